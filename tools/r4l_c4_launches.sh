cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/r4l; mkdir -p $out
QSMC_BENCH_NO_EVENTS=1 rocprofv3 --kernel-trace --output-format csv -d /tmp/c4t -- python $GRAFT_REPO_ROOT/bench.py --only config4_share_rb --warmup 5 > $out/c4.log 2>&1
python - <<'PY'
import csv, glob
f = glob.glob('/tmp/c4t/*/*kernel_trace.csv')[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
t0 = int(rows[0]['Start_Timestamp'])
for i, r in enumerate(rows):
    n = r['Kernel_Name']
    if any(k in n for k in ('redraw', 'bank', 'sample_ordered', 'chunk_scan')) or ('k_bucket_counts' in n and (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) > 10000):
        print('%9.1f us  %-40s %8.1f us  grid %s' % ((int(r['Start_Timestamp']) - t0) / 1e3, n[:40], (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3, r.get('Grid_Size', r.get('Grid_Size_X', '?'))))
PY
