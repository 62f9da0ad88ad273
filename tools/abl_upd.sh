#!/bin/bash
cd /tmp; export TMPDIR=/tmp
for n in "$@"; do
  export QSMC_ABL_LIB=/root/repo/tools/abl_libs/libqsmc_abl$n.so
  rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/ablu/$n -- python /root/repo/tools/upd_only.py >/dev/null 2>&1
  python - $n <<'PY'
import csv,sys,glob
n=sys.argv[1]
f=sorted(glob.glob(f'/root/repo/gpurun_out/ablu/{n}/*/*kernel_stats.csv'))[-1]
for r in csv.DictReader(open(f)):
    if 'k_update_fused' in r['Name']: print('  ',n, r['Name'][:44], r['Calls'], round(float(r['AverageNs'])/1e3,1))
PY
done
