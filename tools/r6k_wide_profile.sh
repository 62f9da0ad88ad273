#!/bin/bash
# round 6: profiles/r6_h_* -- the wide path (three-qubit tomography): bench entry, kernel stats under rocprofv3, timings
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
P=gpurun_out/profiles_new; mkdir -p $P
T=${T:-r6_h}
python3 bench.py --only widening_tomography_3q --warmup 5 2>$P/${T}_3q_bench.err | tail -1 > $P/${T}_3q_bench_entry.json
rm -rf /tmp/prof3q
QSMC_BENCH_NO_EVENTS=1 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof3q -- python3 /root/repo/bench.py --only widening_tomography_3q --warmup 5 > $P/${T}_3q_under_rocprof.log 2>&1
cp $(ls /tmp/prof3q/*/*kernel_stats.csv | tail -1) $P/${T}_3q_kernel_stats.csv
python3 tools/wide_bench.py 1e6 > $P/${T}_wide_bench.txt 2>&1
python3 - <<PY
import json
d=json.load(open('$P/${T}_3q_bench_entry.json'))
e=d.get('widening_tomography_3q', d)
print({k: e.get(k) for k in ('value','ms_per_step','resamples','particles')})
for k in ('update_kernel','resample_kernel','canonicalize','moments_kernel'):
    v=e.get(k) or {}
    print(k, v.get('kernel'), v.get('avg_kernel_us'), v.get('frac'))
PY
head -12 $P/${T}_3q_kernel_stats.csv | cut -c1-150
tail -6 $P/${T}_wide_bench.txt
