cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4q
for cfg in default CHAIN1 LANES NH7 NH13; do
  case $cfg in
    default) envs="";;
    CHAIN1) envs="QSMC_HYP_CHAIN1=1";;
    LANES) envs="QSMC_HYP_NO_CHAIN=1";;
    NH7) envs="QSMC_HYP_NH=7";;
    NH13) envs="QSMC_HYP_NH=13";;
  esac
  echo "== $cfg"
  env $envs timeout 300 python3 tools/design_bench.py 1e7 gpurun_out/r4q/$cfg.npy 2>&1 | grep -v amdgpu.ids
done
python3 - <<'P'
import numpy as np
ref = np.load('gpurun_out/r4q/LANES.npy')
for c in ('default', 'CHAIN1', 'NH7', 'NH13'):
    a = np.load('gpurun_out/r4q/%s.npy' % c)
    both = np.isfinite(a) & np.isfinite(ref)
    rel = np.abs(a[both] - ref[both]) / (np.abs(ref[both]) + 1e-300)
    print(c, 'finite', int(both.sum()), 'of', a.size, 'nan in new only', int((~np.isfinite(a) & np.isfinite(ref)).sum()), 'max rel', rel.max(), 'max abs', np.abs(a[both]-ref[both]).max())
P
timeout 900 python3 -m pytest tests -m gpu -x -q -k "bayes_risk or design or hyp or full_size" 2>&1 | tail -5
