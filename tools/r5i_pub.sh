#!/bin/bash
cd /tmp; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r5i; mkdir -p $O
QSMC_BENCH_NO_EVENTS=1 rocprofv3 --kernel-trace --stats -d /tmp/pc5 --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --only config5_share_tomography --warmup 5 > $O/c5.json 2>/dev/null
cp /tmp/pc5/*/*kernel_stats.csv $O/c5_kernel_stats.csv
QSMC_BENCH_NO_EVENTS=1 rocprofv3 --kernel-trace --stats -d /tmp/pp --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --only other_paths > $O/paths.json 2>/dev/null
cp /tmp/pp/*/*kernel_stats.csv $O/paths_kernel_stats.csv
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv, json
for f in ("gpurun_out/r5i/c5_kernel_stats.csv", "gpurun_out/r5i/paths_kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        if any(s in r["Name"] for s in ("publish", "sum_partials", "sum_columns", "hyp_sums")):
            print(f.split("/")[-1], r["Name"].split("(")[0][:50], r["Calls"], "%.2f us" % (float(r["AverageNs"]) / 1e3))
d = json.loads(open("gpurun_out/r5i/c5.json").read().strip().splitlines()[-1])["config5_share_tomography"]
print("C5 ms/step", d["ms_per_step"])
p = json.loads(open("gpurun_out/r5i/paths.json").read().strip().splitlines()[-1])["other_paths"]
print("bayes_risk ms/exp", p["bayes_risk_26_outcomes"]["ms_per_experiment"], p["bayes_risk_26_outcomes"]["kernel"]["avg_kernel_us"])
PY
for i in 1 2 3; do python bench.py --only other_paths 2>/dev/null | python -c "
import sys, json
p = json.loads(sys.stdin.read().strip().splitlines()[-1])['other_paths']
print('paths', p['batch_update_interval_5']['value'], p['batch_update_interval_8']['value'], p['batch_update_tomography_interval_5'].get('value'), p['bayes_risk_26_outcomes']['ms_per_experiment'], p['bayes_risk_26_outcomes']['kernel']['avg_kernel_us'])
"; done
