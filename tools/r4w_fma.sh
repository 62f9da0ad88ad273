cd $GRAFT_REPO_ROOT
for lib in base fma; do
  if [ $lib = fma ]; then export QSMC_LIB_PATH=$PWD/gpurun_ab/lib_fma.so; else unset QSMC_LIB_PATH; fi
  for i in 1 2; do
  python3 bench.py --only config5_share_tomography --warmup 5 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
c=d['config5_share_tomography']
print('$lib', c.get('value'), c.get('ms_per_step'), c.get('resamples'), [(k, v.get('avg_kernel_us') if isinstance(v,dict) else v) for k,v in c.items() if 'canon' in k or 'list' in k or 'census' in k])
"
  done
done
