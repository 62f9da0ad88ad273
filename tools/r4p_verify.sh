cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4p
timeout 900 python3 -m pytest tests -m gpu -x -q > gpurun_out/r4p/gpu_tests.log 2>&1; echo "pytest rc $?" >> gpurun_out/r4p/gpu_tests.log
tail -5 gpurun_out/r4p/gpu_tests.log
timeout 120 python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r4p/drv_full.json 2>gpurun_out/r4p/drv_full.err; echo "bench rc $?"
python3 -c "
import json
d=json.loads(open('gpurun_out/r4p/drv_full.json').read().strip().splitlines()[-1])
print(d.get('value'), d.get('ms_per_step'), d.get('roofline'))
print({k:(v.get('value'),v.get('ms_per_step')) for k,v in (d.get('other_configs') or {}).items() if isinstance(v,dict)})
"
