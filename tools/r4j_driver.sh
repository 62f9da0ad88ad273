cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4j
for i in 1 2 3; do python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-other-configs > gpurun_out/r4j/drv_$i.json 2>/dev/null; done
QSMC_BENCH_NO_EVENTS=1 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-other-configs --no-cpu-baseline > gpurun_out/r4j/noev.json 2>/dev/null
python3 bench.py --gpus 1 --steps 200 --warmup 20 --no-other-configs --no-cpu-baseline > gpurun_out/r4j/s200.json 2>/dev/null
for f in gpurun_out/r4j/*.json; do python3 -c "
import json
d=json.loads(open('$f').read().strip().splitlines()[-1])
print('$f', d.get('value'), d.get('ms_per_step'), d.get('repeat_passes_ms_per_step'), d.get('cpu_baseline',{}).get('gpu_same_sample'), d.get('config',{}).get('resamples_in_timed_region'), d.get('roofline',{}).get('avg_kernel_us'), d.get('roofline',{}).get('timed_launches'), (d.get('resample_kernel') or {}).get('avg_kernel_us'))
"; done
