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
    assert lib.qsmc_abi_version() == 2
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
