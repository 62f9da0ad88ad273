cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4x
timeout 1500 python3 -m pytest tests -m gpu -x -q > gpurun_out/r4x/gpu_tests.log 2>&1; echo "pytest rc $?"; tail -3 gpurun_out/r4x/gpu_tests.log
for i in 1 2; do python3 bench.py --only config5_share_tomography --warmup 5 2>/dev/null | tail -1 | python3 -c "
import json,sys
c=json.loads(sys.stdin.read())['config5_share_tomography']
print(c.get('value'), c.get('ms_per_step'), [(k, v.get('avg_kernel_us') if isinstance(v,dict) else v) for k,v in c.items() if 'canon' in k], c['resample_kernel'].get('kick_us'))
"; done
