# A/B of the reduction folded into the update kernel (QSMC_FOLD_REDUCE=1): the driver's command, the 200-step run, C4 / C5 shares
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4k
for v in off on; do
  if [ $v = on ]; then export QSMC_FOLD_REDUCE=1; else unset QSMC_FOLD_REDUCE; fi
  for i in 1 2 3; do python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-other-configs --no-cpu-baseline > gpurun_out/r4k/drv_${v}_$i.json 2>/dev/null; done
  python3 bench.py --gpus 1 --steps 200 --warmup 20 --no-other-configs --no-cpu-baseline > gpurun_out/r4k/s200_$v.json 2>/dev/null
  python3 bench.py --only config4_share_rb --warmup 5 > gpurun_out/r4k/c4_$v.json 2>/dev/null
  python3 bench.py --only config5_share_tomography --warmup 5 > gpurun_out/r4k/c5_$v.json 2>/dev/null
  python3 bench.py --only config3_binomial_precession --warmup 5 > gpurun_out/r4k/c3_$v.json 2>/dev/null
done
unset QSMC_FOLD_REDUCE
python3 - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r4k/*.json')):
    d = json.loads(open(f).read().strip().splitlines()[-1])
    if 'value' not in d: d = list(d.values())[0]
    print(f.split('/')[-1], '%.4g' % d['value'], '%.5f' % d['ms_per_step'], d.get('repeat_passes_ms_per_step'), d.get('resamples', d.get('config', {}).get('resamples_in_timed_region')),
          (d.get('roofline') or d.get('update_kernel') or {}).get('avg_kernel_us'))
PY
QSMC_FOLD_REDUCE=1 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -4
