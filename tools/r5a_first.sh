#!/bin/bash
# round 5, first GPU call: the GPU suite, smoke, then same-box A/B of the one-launch datum (QSMC_FOLD_MAX_GRID) on the
# config-5 share and on a 1.25e6-particle precession shard
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r5a
mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1
echo "pytest rc=$?" | tee -a $O/pytest.log
tail -5 $O/pytest.log
timeout 300 python __graft_entry__.py smoke > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $O/smoke.log
for rep in 1 2 3; do
  for cap in 1024 0; do
    QSMC_FOLD_MAX_GRID=$cap timeout 300 python bench.py --only config5_share_tomography --warmup 5 > $O/c5_fold${cap}_$rep.json 2>$O/err.log
    python - <<PY
import json
d=json.load(open("$O/c5_fold${cap}_$rep.json"))["config5_share_tomography"]
print("C5 fold_cap=$cap rep=$rep ms/step %.5f resamples %d upd %.2f us" % (d["ms_per_step"], d["resamples"], d["update_kernel"]["avg_kernel_us"]))
PY
  done
done
for rep in 1 2; do
  for cap in 1024 0; do
    QSMC_FOLD_MAX_GRID=$cap timeout 300 python bench.py --particles 1.25e6 --steps 200 --warmup 20 --no-other-configs --no-cpu-baseline > $O/shard_fold${cap}_$rep.json 2>$O/err.log
    python - <<PY
import json
d=json.load(open("$O/shard_fold${cap}_$rep.json"))
print("shard 1.25e6 fold_cap=$cap rep=$rep ms/step %.5f resamples %d upd %.2f us reps %s" % (d["ms_per_step"], d["config"]["resamples_in_timed_region"], d["roofline"]["avg_kernel_us"], d["repeat_passes_ms_per_step"]))
PY
  done
done
for cap in 1024 0; do
  QSMC_FOLD_MAX_GRID=$cap timeout 300 python bench.py --particles 2.5e6 --steps 200 --warmup 20 --no-other-configs --no-cpu-baseline > $O/shard2_fold${cap}.json 2>$O/err.log
  python - <<PY
import json
d=json.load(open("$O/shard2_fold${cap}.json"))
print("shard 2.5e6 (grid 1221) fold_cap=$cap ms/step %.5f upd %.2f us reps %s" % (d["ms_per_step"], d["roofline"]["avg_kernel_us"], d["repeat_passes_ms_per_step"]))
PY
done
QSMC_FOLD_MAX_GRID=2048 timeout 300 python bench.py --particles 2.5e6 --steps 200 --warmup 20 --no-other-configs --no-cpu-baseline > $O/shard2_fold2048.json 2>$O/err.log
python - <<PY
import json
d=json.load(open("$O/shard2_fold2048.json"))
print("shard 2.5e6 (grid 1221) fold_cap=2048 ms/step %.5f upd %.2f us reps %s" % (d["ms_per_step"], d["roofline"]["avg_kernel_us"], d["repeat_passes_ms_per_step"]))
PY
