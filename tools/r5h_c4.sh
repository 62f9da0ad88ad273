#!/bin/bash
# round 5: config-4 share, bank kernels with the one-sweep prefix (in-tree) against gpurun_ab/lib_prev.so; per-launch list; tests
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r5h
mkdir -p $O
for rep in 1 2; do
  for which in new prev; do
    if [ $which = prev ]; then export QSMC_LIB_PATH=$GRAFT_REPO_ROOT/gpurun_ab/lib_prev.so; else unset QSMC_LIB_PATH; fi
    timeout 300 python bench.py --only config4_share_rb --warmup 5 > $O/c4_${which}_$rep.json 2>$O/err.log
    python - <<PY
import json
d=json.load(open("$O/c4_${which}_$rep.json"))["config4_share_rb"]
print("C4 $which rep=$rep ms/step %.5f resamples %d upd %.1f sampler %.1f mean %s" % (d["ms_per_step"], d["resamples"], d["update_kernel"]["avg_kernel_us"], d["resample_kernel"]["avg_kernel_us"], d["posterior_mean_head"]))
PY
  done
done
unset QSMC_LIB_PATH
bash tools/r4l_c4_launches.sh > $O/c4_launches.txt 2>&1
sed -n 9,40p $O/c4_launches.txt
cd /tmp
QSMC_BENCH_NO_EVENTS=1 rocprofv3 --kernel-trace --stats -d /tmp/pc5 --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --only config5_share_tomography --warmup 5 > /dev/null 2>&1
cp /tmp/pc5/*/*kernel_stats.csv $GRAFT_REPO_ROOT/$O/c5_kernel_stats.csv
QSMC_BENCH_NO_EVENTS=1 rocprofv3 --kernel-trace --stats -d /tmp/pp --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --only other_paths > $GRAFT_REPO_ROOT/$O/paths.json 2>/dev/null
cp /tmp/pp/*/*kernel_stats.csv $GRAFT_REPO_ROOT/$O/paths_kernel_stats.csv
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv
for f in ("gpurun_out/r5h/c5_kernel_stats.csv", "gpurun_out/r5h/paths_kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        if any(s in r["Name"] for s in ("publish", "sum_partials", "sum_columns", "hyp_sums")):
            print(f.split("/")[-1], r["Name"].split("(")[0][:50], r["Calls"], "%.2f us" % (float(r["AverageNs"]) / 1e3))
PY
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_parallel_gloo.py -m gpu -x -q -k "rank_ordered or bank or rb or statistics" 2>&1 | tail -15
