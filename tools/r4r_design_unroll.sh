cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4r
for u in 1 2 3; do
  echo "== U=$u"
  QSMC_LIB_PATH=$PWD/gpurun_ab/lib_u$u.so timeout 300 python3 tools/design_bench.py 1e7 gpurun_out/r4r/u$u.npy 2>&1 | grep -v amdgpu.ids
  echo "== U=$u NH7"
  QSMC_HYP_NH=7 QSMC_LIB_PATH=$PWD/gpurun_ab/lib_u$u.so timeout 300 python3 tools/design_bench.py 1e7 2>&1 | grep "n_meas  25 bayes"
done
QSMC_HYP_NO_CHAIN=1 timeout 300 python3 tools/design_bench.py 1e7 gpurun_out/r4r/LANES.npy > /dev/null 2>&1
python3 - <<'P'
import numpy as np
ref = np.load('gpurun_out/r4r/LANES.npy')
for c in ('u1', 'u2', 'u3'):
    a = np.load('gpurun_out/r4r/%s.npy' % c)
    both = np.isfinite(a) & np.isfinite(ref)
    rel = np.abs(a[both] - ref[both]) / (np.abs(ref[both]) + 1e-300)
    print(c, 'finite', int(both.sum()), 'of', a.size, 'max rel', rel.max())
P
