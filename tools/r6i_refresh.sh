#!/bin/bash
# round 6: what regenerates profiles/r6_a_* (run on the GPU box through gpurun; results land in gpurun_out/profiles_new)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
rm -rf gpurun_out/profiles_new
T=${T:-r6_a}
P=gpurun_out/profiles_new; mkdir -p $P
python -m pytest tests -q -m gpu > $P/${T}_pytest_gpu.log 2>&1; tail -4 $P/${T}_pytest_gpu.log
bash tools/profile_configs.sh $T pmc > gpurun_out/r6i_profile.log 2>&1
for i in 1 2 3; do python3 bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | tail -1 > $P/${T}_driver_cmd_$i.json; done
# one rank's share of a strong-scaling run: update() and batch_update windows at N = 1.25e6, 20 and 200 data
python3 bench.py --gpus 1 --steps 200 --warmup 5 --only shard_preview 2>/dev/null | tail -1 > $P/${T}_shard_preview_200steps.json
python3 bench.py --gpus 1 --steps 20 --warmup 5 --only shard_preview 2>/dev/null | tail -1 > $P/${T}_shard_preview_20steps.json
# two ranks on this one GPU through the self-launcher (control flow: strong_scaling with batch windows + sharded_configs inside the line)
QSMC_BENCH_SHARE_GPU=1 python3 bench.py --gpus 2 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > $P/${T}_share_gpu_2ranks_line.json
# one rank through the full sharded path with the RCCL pass forced and the transport probe forced (both timings in the line)
QSMC_TRANSPORT_PROBE=force QSMC_BENCH_FORCE_RCCL_PASS=1 MASTER_PORT=29671 python3 bench.py --gpus 1 --force-comm --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs 2>/dev/null | tail -1 > $P/${T}_force_comm_world1_line.json
python3 bench.py --only plugin_paths 2>/dev/null | tail -1 > $P/${T}_plugin_paths.json
python3 tools/sort_time.py > $P/${T}_argsort_time.txt 2>&1
python3 tools/plugin_time.py > $P/${T}_plugin_time.txt 2>&1
bash tools/r4l_c4_launches.sh > $P/${T}_c4_resample_launches.txt 2>&1
python3 tools/tomo_batch_bench.py > $P/${T}_c5_batch_update.txt 2>&1
for i in 1 2 3; do python3 -c "
import json
d=json.loads(open('$P/${T}_driver_cmd_$i.json').read())
print('driver', d['value'], d['ms_per_step'], d.get('repeat_passes_ms_per_step'), (d.get('headline_200_steps') or {}).get('value'))
"; done
python3 - <<PY
import json
d=json.load(open('$P/${T}_bench_line.json'))
print('200 steps', d['value'], d['ms_per_step'], d['config']['resamples_in_timed_region'], 'upd us', d['roofline']['avg_kernel_us'], d['roofline']['frac'])
for k,v in d.get('other_configs',{}).items(): print(k, v.get('value'), v.get('ms_per_step'), v.get('resamples'), v.get('error'))
for k,v in d.get('other_paths',{}).items(): print(k, v.get('value'), v.get('ms_per_datum'), v.get('ms_per_experiment'), (v.get('window_kernel') or v.get('kernel') or {}).get('avg_kernel_us') if isinstance(v, dict) else None, v.get('error') if isinstance(v, dict) else None)
sp=d.get('strong_scaling_shard_preview',{})
print('beyond', d.get('roofline_beyond_l3',{}).get('frac'), 'shard', {k: sp.get(k) for k in ('value','ms_per_step','resamples','error')}, {k: (sp.get(k) or {}).get('vs_update') for k in ('batch_update_interval_5','batch_update_interval_8')})
print('cpu', d.get('cpu_baseline',{}).get('value'), d.get('cpu_baseline',{}).get('cores'))
s=json.load(open('$P/${T}_share_gpu_2ranks_line.json'))
ss=s.get('strong_scaling',{})
print('2 ranks: strong', {k: ss.get(k) for k in ('value','ms_per_step','particles_total','ranks','error')}, {k: (ss.get(k) or {}).get('vs_update') for k in ('batch_update_interval_5','batch_update_interval_8')}, 'sharded', list(s.get('sharded_configs',{}).keys()))
f=json.load(open('$P/${T}_force_comm_world1_line.json'))
print('world1 probe', f['config'].get('transport_probe'), f['config'].get('headline_transport'))
print(open('$P/${T}_argsort_time.txt').read())
print(open('$P/${T}_plugin_time.txt').read()[-400:])
PY
tail -30 gpurun_out/r6i_profile.log
