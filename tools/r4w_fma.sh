cd $GRAFT_REPO_ROOT
for lib in base tri; do
  if [ $lib = tri ]; then export QSMC_LIB_PATH=$PWD/gpurun_ab/lib_tri.so; else unset QSMC_LIB_PATH; fi
  for i in 1 2; do
  python3 bench.py --only config5_share_tomography --warmup 5 2>/dev/null | tail -1 | python3 -c "
import json,sys
c=json.loads(sys.stdin.read())['config5_share_tomography']
print('$lib', c.get('value'), c.get('ms_per_step'), [(k, v.get('avg_kernel_us') if isinstance(v,dict) else v) for k,v in c.items() if 'canon' in k], c['resample_kernel'].get('kick_us'))
"
  done
done
export QSMC_LIB_PATH=$PWD/gpurun_ab/lib_tri.so
timeout 1200 python3 -m pytest tests -m gpu -x -q -k "tomo or canon or qutrit or c5 or full_size" 2>&1 | tail -3
