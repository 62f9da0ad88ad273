#!/bin/bash
# usage: pmc_run.sh <script.py> <kernel-substring> COUNTER...   (one rocprofv3 --pmc pass per counter)
cd /tmp; export TMPDIR=/tmp
script=$1; shift; kern=$1; shift
for c in "$@"; do
  rm -rf /root/repo/gpurun_out/pmc/$c
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d /root/repo/gpurun_out/pmc/$c -- python /root/repo/$script >/dev/null 2>&1
  python - "$c" "$kern" <<'PY'
import csv,sys,glob
c,kern=sys.argv[1],sys.argv[2]
fs=glob.glob(f'/root/repo/gpurun_out/pmc/{c}/*/*counter_collection.csv')
if not fs: print(c,'no output'); sys.exit()
v=[float(r['Counter_Value']) for r in csv.DictReader(open(fs[0])) if kern in r['Kernel_Name'] and r['Counter_Name']==c]
print(c, 'n=%d mean=%.4g'%(len(v), sum(v)/max(len(v),1)))
PY
done
