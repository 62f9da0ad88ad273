#!/bin/bash
# Run ON THE GPU BOX: SQ counters (VALU / LDS / wait) per kernel for one bench.py sub-command, two rocprofv3 --pmc passes
# (no trace domains beside --kernel-trace).   usage: sq_counters.sh <tag> <bench.py args...>   -> gpurun_out/sq/<tag>_sq_counters.json
cd /tmp; export TMPDIR=/tmp
tag=$1; shift
out=$GRAFT_REPO_ROOT/gpurun_out/sq
mkdir -p $out
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU"
P2="SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_WAVES"
i=0
for P in "$P1" "$P2"; do
  i=$((i+1))
  rm -rf /tmp/sq_$tag_$i
  QSMC_BENCH_NO_EVENTS=1 rocprofv3 --pmc $P --kernel-trace --output-format csv -d /tmp/sq_${tag}_$i -- python $GRAFT_REPO_ROOT/bench.py "$@" > $out/${tag}_pass$i.log 2>&1
done
python - "$tag" "$out" "$*" <<'PY'
import csv, glob, json, sys, collections
tag, out, cmd = sys.argv[1], sys.argv[2], sys.argv[3]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for i in (1, 2):
    for f in glob.glob('/tmp/sq_%s_%d/*/*counter_collection.csv' % (tag, i)):
        for r in csv.DictReader(open(f)):
            k = r['Kernel_Name'].split('(')[0][:70]
            agg[k][r['Counter_Name']].append(float(r['Counter_Value']))
res = {}
for k, cs in agg.items():
    n = max(len(v) for v in cs.values())
    if n < 2 and not any(s in k for s in ('sample', 'kick', 'anc16', 'redraw', 'bank', 'hyp', 'multi', 'canon', 'moments')):
        continue
    res[k] = {c: sum(v) / len(v) for c, v in sorted(cs.items())}
    res[k]['launches'] = n
    e = res[k]
    if e.get('SQ_BUSY_CYCLES') and e.get('SQ_ACTIVE_INST_VALU'):
        # quad-cycle counters summed over SEs/XCDs: VALU-busy share of the wave-resident time
        e['valu_active_over_wave_cycles'] = e['SQ_ACTIVE_INST_VALU'] / max(e.get('SQ_WAVE_CYCLES', 1), 1)
json.dump({'command': 'rocprofv3 --pmc <pass 1 | pass 2> --kernel-trace -- python bench.py ' + cmd +
                      '  (per-launch averages; two passes of 8 counters)', 'kernels': res},
          open('%s/%s_sq_counters.json' % (out, tag), 'w'), indent=1, sort_keys=True)
print(tag, 'kernels:', len(res))
PY
