#!/bin/bash
cd /tmp; export TMPDIR=/tmp
export QSMC_ABL_LIB=/root/repo/tools/abl_libs/libqsmc_abl6.so
for n in 9400000 10000000 11000000 12580000; do
  QSMC_N=$n rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/abl/n$n -- python /root/repo/tools/abl2.py >/dev/null 2>&1
  python - $n <<'PY'
import csv,sys,glob
n=sys.argv[1]
f=sorted(glob.glob(f'/root/repo/gpurun_out/abl/n{n}/*/*kernel_stats.csv'))[-1]
for r in csv.DictReader(open(f)):
    if 'bucket_sample' in r['Name']: print(n, r['Name'][:30], r['Calls'], round(float(r['AverageNs'])/1e3,1))
PY
done
