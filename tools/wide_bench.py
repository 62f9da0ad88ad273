"""Round 6: three-qubit tomography (d = 64) on the wide kernels (csrc/kernels/wide.hpp): per-kernel times (HIP events through
the profiling ring) and ms per datum of an SMCUpdater run with resamples, N = 1e6 by default (N x 64 x 8 B = 512 MB of cloud)."""
import os, sys, time, warnings
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "python-qinfer_amd")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import torch
import qinfer_amd as qi
from qinfer_amd.engine import get_engine
warnings.simplefilter("ignore")
eng = get_engine()
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_000_000
b = qi.tomography.pauli_basis(3)
m = qi.TomographyModel(b)
rs = np.random.RandomState(0)
np.random.seed(0)
t0 = time.perf_counter()
upd = qi.SMCUpdater(m, n, qi.GinibreDistribution(b), device_rng=True, seed=1)
print("prior of %d states: %.1f s (host)" % (n, time.perf_counter() - t0))


def pauli_ep():
    ep = np.zeros((1,), dtype=m.expparams_dtype)
    ep["meas"][0, 0] = np.sqrt(8) / 2
    ep["meas"][0, rs.randint(1, 64)] = np.sqrt(8) / 2
    return ep


def dense_ep():
    v = rs.randn(8) + 1j * rs.randn(8)
    v /= np.linalg.norm(v)
    ep = np.zeros((1,), dtype=m.expparams_dtype)
    ep["meas"][0] = np.real(np.einsum('aij,ij->a', b.data.conj(), np.outer(v, v.conj())))
    return ep


names = {0: "update", 1: "kick", 2: "update_ones", 3: "canon_classify", 4: "canon_list", 5: "moments", 6: "counts",
         7: "counts_skipped", 8: "ancestors", 12: "canon_build", 13: "canon_expand"}
for label, mk in (("sparse (Pauli, nnz = 2)", pauli_ep), ("dense (nnz = 64)", dense_ep)):
    for _ in range(3):
        upd.update(int(rs.randint(2)), mk(), check_for_resample=False)
    eng.set_profiling(1)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20):
        upd.update(int(rs.randint(2)), mk(), check_for_resample=False)
    torch.cuda.synchronize(); wall = (time.perf_counter() - t0) / 20
    ms, tags = eng.profile_read(); eng.set_profiling(0)
    k = float(ms[tags == 0].mean()) * 1e3
    nnz = 2 if "sparse" in label else 64
    print("update %-24s %.1f us/datum, kernel %.1f us = %.2f TB/s of (16 + 8 nnz) B" % (label, wall * 1e6, k, n * (16 + 8 * nnz) / k / 1e6))
# resample pieces
for rep in range(3):
    eng.set_profiling(1 if rep == 2 else 0)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    upd.resample()
    torch.cuda.synchronize(); wall = time.perf_counter() - t0
    upd.update(int(rs.randint(2)), pauli_ep(), check_for_resample=False)
ms, tags = eng.profile_read(); eng.set_profiling(0)
print("resample (moments + sqrt + ancestors + kick + canonicalize): %.0f us wall;" % (wall * 1e6),
      {names.get(int(t), int(t)): round(float(ms[tags == t].mean()) * 1e3, 1) for t in np.unique(tags)})
print("   moments: %.2f TB/s of 520 B; kick: %.2f TB/s of 1028 B (read ancestor + write child + index)" % (
    n * 520 / (float(ms[tags == 5].mean()) * 1e3) / 1e6 if np.any(tags == 5) else -1,
    n * 1028 / (float(ms[tags == 1].mean()) * 1e3) / 1e6 if np.any(tags == 1) else -1))
# a run: 200 random Pauli measurements of a random state
true = np.asarray(qi.GinibreDistribution(b).sample(1))[0]
upd.reset()
eps = [pauli_ep() for _ in range(200)]
outs = [int(rs.random_sample() < np.clip(e["meas"][0] @ true, 0, 1)) for e in eps]
for k in range(20):
    upd.update(outs[k], eps[k])
rc0 = upd.resample_count
torch.cuda.synchronize(); t0 = time.perf_counter()
for k in range(20, 200):
    upd.update(outs[k], eps[k])
torch.cuda.synchronize(); wall = time.perf_counter() - t0
print("run: N = %d, 180 data, %d resamples: %.4f ms per datum = %.3g particle-updates/s; |mean - true| = %.3f" % (
    n, upd.resample_count - rc0, wall / 180 * 1e3, n * 180 / wall, np.linalg.norm(upd.est_mean() - true)))
