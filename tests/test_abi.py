"""CPU checks of the drop-in boundary: libqsmc_hip.so loads and exports exactly the symbols
declared in include/qsmc.h, the ctypes table agrees with the header, and the one host-side routine
(qsmc_sqrtm_psd) matches the oracle / golden vectors.  No GPU compute is invoked."""
import ctypes as C
import os
import re
import subprocess
import warnings

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_functions():
    text = open(os.path.join(ROOT, "include", "qsmc.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(qsmc_[a-z_0-9]+)\s*\(", text)))


@pytest.fixture(scope="module")
def lib():
    import __graft_entry__ as ge
    ge.build_library()
    from qinfer_amd import _native
    return _native.load()


def test_header_symbols_exported(lib):
    from qinfer_amd import _native
    out = subprocess.run(["nm", "-D", "--defined-only", _native.lib_path()], capture_output=True,
                         text=True, check=True).stdout
    exported = sorted(set(re.findall(r" T (qsmc_[a-z_0-9]+)", out)))
    declared = _header_functions()
    assert declared == exported, (set(declared) ^ set(exported))
    assert len(declared) >= 20


def test_ctypes_table_matches_header(lib):
    from qinfer_amd import _native
    assert sorted(_native.SIGNATURES) == _header_functions()
    assert lib.qsmc_abi_version() == 3
    assert lib.qsmc_strerror(0) == b"ok"
    assert lib.qsmc_strerror(-1) == b"invalid argument"


def test_struct_layouts(lib):
    """ctypes mirrors of the header structs have the sizes the C compiler gives them."""
    from qinfer_amd import _native
    src = '#include "qsmc.h"\n#include <stdio.h>\n#include <stddef.h>\nint main(){printf("%zu %zu %zu %zu %zu %zu %zu\\n", ' \
          'sizeof(qsmc_model_t), sizeof(qsmc_expparam_t), sizeof(qsmc_update_stats_t), sizeof(qsmc_step_t), ' \
          'offsetof(qsmc_step_t, lw), offsetof(qsmc_step_t, status), offsetof(qsmc_step_t, moments));return 0;}'
    exe = "/tmp/qsmc_sizeof"
    subprocess.run(["gcc", "-x", "c", "-", "-I", os.path.join(ROOT, "include"), "-o", exe],
                   input=src, text=True, check=True)
    sizes = [int(v) for v in subprocess.run([exe], capture_output=True, text=True, check=True).stdout.split()]
    assert sizes == [C.sizeof(_native.ModelDesc), C.sizeof(_native.ExpParam), C.sizeof(_native.UpdateStats),
                     C.sizeof(_native.Step), _native.Step.lw.offset, _native.Step.status.offset,
                     _native.Step.moments.offset]


def test_invalid_arguments_return_status(lib):
    assert lib.qsmc_destroy(None) == 0
    assert lib.qsmc_sqrtm_psd(None, 2, 1.0, None, None) == -1
    assert lib.qsmc_fill(None, None, 10, 0.0, None) == -1


def test_sqrtm_psd_vs_oracle(lib, golden):
    import np_oracle as orc
    import parity_tols as tol
    from qinfer_amd.utils import sqrtm_psd
    g = golden("g3_moments")
    for tag in g["tags"]:
        cov = g[tag + "_cov"]
        S, err = sqrtm_psd(cov)
        at = tol.atol_sqrtm_psd(cov)
        np.testing.assert_allclose(S, g[tag + "_sqrt"], rtol=0, atol=max(at, 1e-15), err_msg=tag)
        np.testing.assert_allclose(S @ S, np.where(np.isfinite(cov), cov, 0), atol=max(at ** 2, 1e-14))
        assert abs(err - g[tag + "_sqrt_err"]) <= max(at, 1e-14)
    # reference's own property test (tests/test_utils.py:132-152): sqrt(Y) sqrt(Y) ~= Y, incl. singular Y
    rs = np.random.RandomState(3)
    for n in (1, 2, 5, 16):
        X = rs.randn(n, n)
        Y = X @ X.T
        S, _ = sqrtm_psd(Y)
        np.testing.assert_allclose(S @ S, Y, atol=1e-10 * max(1, np.abs(Y).max()))
    Y = np.zeros((3, 3))
    Y[0, 0] = 2.0
    S, err = sqrtm_psd(Y)
    np.testing.assert_allclose(S @ S, Y, atol=1e-14)
    # negative eigenvalues are truncated
    S = sqrtm_psd(np.diag([4.0, -1.0]), est_error=False)
    np.testing.assert_allclose(S, np.diag([2.0, 0.0]), atol=1e-15)


def test_compute_without_gpu_fails_loudly():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import qinfer_amd as qi
    with pytest.raises(qi.NativeLibraryError):
        qi.SMCUpdater(qi.SimplePrecessionModel(), 10, qi.UniformDistribution([0, 1]))
    with pytest.raises(qi.NativeLibraryError):
        qi.SimplePrecessionModel().likelihood(np.array([0]), np.array([[0.5]]), np.array([1.0]))


def _plan(lib, seed, epoch, w, n_total):
    w = np.ascontiguousarray(w, dtype=np.float64)
    out = np.empty(len(w), dtype=np.int64)
    rc = lib.qsmc_shard_plan_totals(C.c_uint64(seed), C.c_uint64(epoch), w.ctypes.data_as(C.POINTER(C.c_double)), len(w),
                                    int(n_total), out.ctypes.data_as(C.POINTER(C.c_int64)))
    assert rc == 0
    return out


@pytest.mark.parametrize("n,p", [(40, 0.3), (25, 0.97), (400, 0.02), (400, 0.2), (3000, 0.5), (100000, 0.07),
                                 (80000000, 0.125), (80000000, 0.5), (12345678, 0.999)])
def test_shard_plan_binomial_law(lib, n, p):
    """qsmc_shard_plan_totals with two shards IS a binomial draw: its law against the exact pmf (scipy), through both
    samplers (inversion below n min(p, 1-p) = 30, BTPE above) -- a chi-square over the bins that hold the mass, plus the
    first two moments; and the obvious invariants (sum, determinism, epoch and seed dependence)."""
    from scipy import stats
    draws = 40000
    x = np.array([_plan(lib, 7, e, [p, 1 - p], n)[0] for e in range(draws)])
    assert x.min() >= 0 and x.max() <= n
    mu, var = n * p, n * p * (1 - p)
    assert abs(x.mean() - mu) < 5 * np.sqrt(var / draws)
    assert abs(x.var() / var - 1) < 5 * np.sqrt(2.0 / draws) + 3.0 / max(var, 1.0) ** 0.5 / np.sqrt(draws)
    # bins of (about) equal probability from the exact quantiles
    qs = np.unique(stats.binom.ppf(np.linspace(0, 1, 41)[1:-1], n, p)).astype(np.int64)
    edges = np.concatenate([[-1], qs, [n]])
    probs = np.diff(stats.binom.cdf(edges, n, p))
    keep = probs > 0
    obs = np.histogram(x, bins=edges.astype(np.float64) + 0.5)[0]
    chi2 = (((obs - draws * probs) ** 2)[keep] / (draws * probs[keep])).sum()
    assert stats.chi2.sf(chi2, keep.sum() - 1) > 1e-5, (chi2, keep.sum())


def test_shard_plan_multinomial(lib):
    w = np.array([3.0, 0.0, 1.0, 2.0, 2.0])
    n = 1000003
    t = _plan(lib, 11, 5, w, n)
    assert t.sum() == n and t[1] == 0 and np.array_equal(t, _plan(lib, 11, 5, w, n))
    assert not np.array_equal(t, _plan(lib, 11, 6, w, n)) and not np.array_equal(t, _plan(lib, 12, 5, w, n))
    ts = np.array([_plan(lib, 3, e, w, 5000) for e in range(4000)])
    p = w / w.sum()
    assert np.all(ts.sum(axis=1) == 5000)
    np.testing.assert_allclose(ts.mean(axis=0), 5000 * p, atol=5 * np.sqrt(5000 * 0.25 / 4000))
    cov = np.cov(ts.T)
    want = 5000 * (np.diag(p) - np.outer(p, p))
    np.testing.assert_allclose(cov, want, atol=0.12 * 5000 * 0.25)
    assert np.array_equal(_plan(lib, 1, 1, [0.0, 2.5], 77), [0, 77])
    bad = np.array([1.0, -1.0])
    out = np.empty(2, dtype=np.int64)
    assert lib.qsmc_shard_plan_totals(C.c_uint64(1), C.c_uint64(1), bad.ctypes.data_as(C.POINTER(C.c_double)), 2, 10,
                                      out.ctypes.data_as(C.POINTER(C.c_int64))) != 0
