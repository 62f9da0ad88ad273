cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4y
for cfg in default LANES; do
  case $cfg in
    default) envs="";;
    NOWP) envs="QSMC_HYP_NO_WAVEPAIR=1";;
    LANES) envs="QSMC_HYP_NO_CHAIN=1";;
  esac
  echo "== $cfg"
  env $envs timeout 300 python3 tools/design_bench.py 1e7 gpurun_out/r4y/$cfg.npy 2>&1 | grep -v amdgpu.ids
done
python3 - <<'P'
import numpy as np
ref = np.load('gpurun_out/r4y/LANES.npy')
for c in ("default",):
    a = np.load('gpurun_out/r4y/%s.npy' % c)
    both = np.isfinite(a) & np.isfinite(ref)
    rel = np.abs(a[both] - ref[both]) / (np.abs(ref[both]) + 1e-300)
    print(c, 'finite', int(both.sum()), 'of', a.size, 'max rel', rel.max())
P
timeout 900 python3 -m pytest tests -m gpu -x -q -k "bayes_risk or design or hyp" 2>&1 | tail -3
timeout 600 python3 bench.py --only other_paths 2>/dev/null | python3 tools/other_paths_print.py
