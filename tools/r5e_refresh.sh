#!/bin/bash
# round 5: what regenerates profiles/r5_a_* (run on the GPU box through gpurun; results land in gpurun_out/profiles_new)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
rm -rf gpurun_out/profiles_new gpurun_out/sq
T=r5_a
bash tools/profile_configs.sh $T pmc > gpurun_out/r5e_profile.log 2>&1
P=gpurun_out/profiles_new
for i in 1 2 3; do python3 bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | tail -1 > $P/${T}_driver_cmd_$i.json; done
# one rank's share of a strong-scaling run (1e7 over 8 GPUs): the headline workload at N = 1.25e6, 200 steps
python3 bench.py --particles 1.25e6 --steps 200 --warmup 20 --no-other-configs --no-cpu-baseline 2>/dev/null | tail -1 > $P/${T}_shard_1p25e6_line.json
# two ranks on this one GPU through the self-launcher (control flow: strong_scaling + sharded_configs inside the line)
QSMC_BENCH_SHARE_GPU=1 python3 bench.py --gpus 2 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > $P/${T}_share_gpu_2ranks_line.json
# SQ counters of the (f) kernels and of config 5's resample kernels
bash tools/sq_counters.sh ${T}_paths --only other_paths > /dev/null 2>&1
bash tools/sq_counters.sh ${T}_c5 --only config5_share_tomography --warmup 5 > /dev/null 2>&1
cp gpurun_out/sq/${T}_paths_sq_counters.json gpurun_out/sq/${T}_c5_sq_counters.json $P/ 2>/dev/null
# read requests of config 5 by size: FETCH_SIZE's gfx950 correction (x 2) assumes 64-byte requests; the d = 16 kick kernel gathers
cd /tmp
for c in TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum; do
  rm -rf /tmp/rq
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/rq -- python $GRAFT_REPO_ROOT/bench.py --only config5_share_tomography --warmup 5 > /dev/null 2>&1
  f=$(ls /tmp/rq/*/*counter_collection.csv 2>/dev/null | tail -1)
  [ -n "$f" ] && cp $f $GRAFT_REPO_ROOT/$P/${T}_c5_pmc_$(echo $c | tr A-Z a-z).csv
done
cd $GRAFT_REPO_ROOT
# per-launch durations of a config-4 resample
bash tools/r4l_c4_launches.sh > $P/${T}_c4_resample_launches.txt 2>&1
python3 tools/tomo_batch_bench.py > $P/${T}_c5_batch_update.txt 2>&1
for i in 1 2 3; do python3 -c "
import json
d=json.loads(open('$P/${T}_driver_cmd_$i.json').read())
print('driver', d['value'], d['ms_per_step'], d.get('repeat_passes_ms_per_step'))
"; done
python3 - <<PY
import json
d=json.load(open('$P/${T}_bench_line.json'))
print('200 steps', d['value'], d['ms_per_step'], d['config']['resamples_in_timed_region'], 'upd us', d['roofline']['avg_kernel_us'], d['roofline']['frac'])
for k,v in d.get('other_configs',{}).items(): print(k, v.get('value'), v.get('ms_per_step'), v.get('resamples'), v.get('error'))
for k,v in d.get('other_paths',{}).items(): print(k, v.get('value'), v.get('ms_per_datum'), v.get('ms_per_experiment'), (v.get('window_kernel') or v.get('kernel') or {}).get('avg_kernel_us'), v.get('error'))
print('beyond', d.get('roofline_beyond_l3',{}).get('frac'), 'shard', {k: d.get('strong_scaling_shard_preview',{}).get(k) for k in ('value','ms_per_step','resamples','error')})
print('cpu', d.get('cpu_baseline',{}).get('value'), d.get('cpu_baseline',{}).get('cores'))
s=json.load(open('$P/${T}_share_gpu_2ranks_line.json'))
print('2 ranks: strong', {k: s.get('strong_scaling',{}).get(k) for k in ('value','ms_per_step','particles_total','ranks','error')}, 'sharded', list(s.get('sharded_configs',{}).keys()))
PY
tail -30 gpurun_out/r5e_profile.log
