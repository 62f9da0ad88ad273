cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4t
for cfg in default NOPAIR LANES; do
  case $cfg in
    default) envs="";;
    NOPAIR) envs="QSMC_HYP_NO_PAIR=1";;
    LANES) envs="QSMC_HYP_NO_CHAIN=1";;
  esac
  echo "== $cfg"
  env $envs timeout 300 python3 tools/design_bench.py 1e7 gpurun_out/r4t/$cfg.npy 2>&1 | grep -v amdgpu.ids
done
python3 - <<'P'
import numpy as np
ref = np.load('gpurun_out/r4t/LANES.npy')
for c in ('default', 'NOPAIR'):
    a = np.load('gpurun_out/r4t/%s.npy' % c)
    both = np.isfinite(a) & np.isfinite(ref)
    rel = np.abs(a[both] - ref[both]) / (np.abs(ref[both]) + 1e-300)
    print(c, 'finite', int(both.sum()), 'of', a.size, 'max rel', rel.max())
P
timeout 900 python3 -m pytest tests -m gpu -x -q -k "bayes_risk or design or hyp or full_size" 2>&1 | tail -3
bash tools/sq_counters.sh r4t_paths --only other_paths > /dev/null 2>&1
python3 - <<'P'
import json
d=json.load(open('gpurun_out/sq/r4t_paths_sq_counters.json'))
for k,v in d['kernels'].items():
    if 'hyp' in k:
        print(k, {c: (round(x,1) if isinstance(x,float) else x) for c,x in v.items()})
P
