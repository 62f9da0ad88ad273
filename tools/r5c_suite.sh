#!/bin/bash
# round 5: the GPU suite on the current build, then the (f) paths and the config-5 batch_update window (sparse rows)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r5c
mkdir -p $O
timeout 2200 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1
echo "pytest rc=$?" | tee -a $O/pytest.log
tail -6 $O/pytest.log
for rep in 1 2; do
  timeout 600 python bench.py --only other_paths > $O/paths_$rep.json 2>$O/err.log
  python tools/other_paths_print.py < $O/paths_$rep.json 2>/dev/null | head -20 || python - <<PY
import json
d=json.load(open("$O/paths_$rep.json"))["other_paths"]
for k,v in d.items():
    print(k, {kk: v[kk] for kk in v if kk in ("value","ms_per_datum","ms_per_experiment","resamples")}, (v.get("window_kernel") or v.get("kernel") or {}).get("avg_kernel_us"))
PY
done
timeout 600 python tools/tomo_batch_bench.py > $O/tomo_batch.txt 2>&1; cat $O/tomo_batch.txt | tail -12
