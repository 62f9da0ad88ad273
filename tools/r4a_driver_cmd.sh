set -x
mkdir -p gpurun_out/r4a
cd $GRAFT_REPO_ROOT
for i in 1 2 3; do python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r4a/drv_$i.json 2> gpurun_out/r4a/drv_$i.err; done
for sw in QSMC_NO_STEP QSMC_NO_STEP_RESAMPLE QSMC_NO_TILE_SUMS QSMC_BENCH_NO_EVENTS; do
  env $sw=1 python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r4a/sw_$sw.json 2> gpurun_out/r4a/sw_$sw.err
done
python3 bench.py --gpus 1 --steps 20 --warmup 10 > gpurun_out/r4a/w10.json 2> gpurun_out/r4a/w10.err
for f in gpurun_out/r4a/*.json; do echo $f; python3 -c "
import json,sys
d=json.loads(open('$f').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d.get('cpu_baseline',{}).get('gpu_same_sample'))
"; done
