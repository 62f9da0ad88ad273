#!/bin/bash
# usage: abl_run.sh <lib tags...>; tag "d" = the product library
cd /tmp; export TMPDIR=/tmp
for n in "$@"; do
  if [ "$n" = "d" ]; then unset QSMC_ABL_LIB; else export QSMC_ABL_LIB=/root/repo/tools/abl_libs/libqsmc_abl$n.so; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/abl/$n -- python /root/repo/tools/abl2.py 2>/dev/null | grep "us per"
  python - $n <<'PY'
import csv,sys,glob
n=sys.argv[1]
f=sorted(glob.glob(f'/root/repo/gpurun_out/abl/{n}/*/*kernel_stats.csv'))[-1]
for r in csv.DictReader(open(f)):
    if float(r['Percentage'])>1.0: print('  ',n, r['Name'][:40], r['Calls'], round(float(r['AverageNs'])/1e3,1))
PY
done
