#!/bin/bash
# last session of round 6: the suite + smoke + the driver's command three times + the 200-step line on the final tree
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
T=${T:-r6_e}
P=gpurun_out/profiles_new; rm -rf $P; mkdir -p $P
python -m pytest tests -q -m gpu > $P/${T}_pytest_gpu.log 2>&1; tail -3 $P/${T}_pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" > $P/${T}_smoke.log 2>&1; tail -1 $P/${T}_smoke.log
for i in 1 2 3; do python3 bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | tail -1 > $P/${T}_driver_cmd_$i.json; done
python3 bench.py --gpus 1 --steps 200 --warmup 20 2>$P/${T}_bench.err | tail -1 > $P/${T}_bench_line.json
QSMC_BENCH_SHARE_GPU=1 python3 bench.py --gpus 2 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > $P/${T}_share_gpu_2ranks_line.json
python3 - <<PY
import json
for i in (1,2,3):
    d=json.load(open('$P/${T}_driver_cmd_%d.json'%i))
    h=d['headline_200_steps']; r=h.get('reference_outcome_sequence',{})
    print('driver', d['value'], d['ms_per_step'], d['roofline']['avg_kernel_us'], d['roofline']['frac'], '| 200:', h['value'], h['resamples'], '| G1 200:', r.get('steps_200',{}).get('value'), r.get('steps_200',{}).get('resamples'), 'G1 20:', r.get('steps_20',{}).get('value'), r.get('steps_20',{}).get('resamples'), r.get('error'))
d=json.load(open('$P/${T}_bench_line.json'))
print('200 steps', d['value'], d['ms_per_step'], d['roofline']['frac'])
for k,v in d.get('other_configs',{}).items(): print(k, v.get('value'), v.get('ms_per_step'), v.get('error'))
s=json.load(open('$P/${T}_share_gpu_2ranks_line.json')); print('2 ranks', s['value'], list(s.get('sharded_configs',{}).keys()), s.get('strong_scaling',{}).get('value'))
PY
