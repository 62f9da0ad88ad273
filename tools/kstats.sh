#!/bin/bash
# Run ON THE GPU BOX: rocprofv3 --kernel-trace --stats of `bench.py "$@"` (no HIP events on the launches), one line per kernel:
# name, calls, average / min / max microseconds.   usage: tools/kstats.sh --only config4_share_rb --warmup 5
cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/ks
QSMC_BENCH_NO_EVENTS=1 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks -- python /root/repo/bench.py "$@" > /tmp/ks.log 2>&1
python - <<PY
import csv, glob
f = sorted(glob.glob("/tmp/ks/*/*kernel_stats.csv"))[-1]
for r in csv.DictReader(open(f)):
    print("%-46s %5s  avg %8.1f  min %8.1f  max %8.1f us" % (r["Name"][:46], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
PY
