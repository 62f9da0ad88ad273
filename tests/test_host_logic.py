"""CPU tests of the host-side mirror of the reference interface (no GPU compute): class
surface, expparams translation into the C ABI structs, priors, bases, resampler parameters."""
import numpy as np
import pytest

import np_oracle as orc
import qinfer_amd as qi
from qinfer_amd import _native


def test_public_surface_matches_reference_names():
    for name in ["SMCUpdater", "LiuWestResampler", "Resampler", "Model", "Simulatable", "FiniteOutcomeModel",
                 "SimplePrecessionModel", "SimpleInversionModel", "BinomialModel", "DerivedModel",
                 "RandomizedBenchmarkingModel", "TomographyModel", "Distribution", "UniformDistribution",
                 "PostselectedDistribution", "ProductDistribution", "MultivariateNormalDistribution",
                 "ParticleDistribution", "ResamplerError", "ResamplerWarning", "ApproximationWarning"]:
        assert hasattr(qi, name), name
    assert issubclass(qi.ResamplerError, RuntimeError)
    assert issubclass(qi.ResamplerWarning, RuntimeWarning)
    assert issubclass(qi.ApproximationWarning, RuntimeWarning)
    assert issubclass(qi.SMCUpdater, qi.ParticleDistribution)
    import inspect
    sig = inspect.signature(qi.SMCUpdater.__init__)
    ref_args = ["model", "n_particles", "prior", "resample_a", "resampler", "resample_thresh", "debug_resampling",
                "track_resampling_divergence", "zero_weight_policy", "zero_weight_thresh", "canonicalize"]
    assert list(sig.parameters)[1:1 + len(ref_args)] == ref_args
    sig = inspect.signature(qi.LiuWestResampler.__init__)
    assert list(sig.parameters)[1:9] == ["a", "h", "maxiter", "debug", "postselect", "zero_cov_comp",
                                         "default_n_particles", "kernel"]
    for meth in ["update", "batch_update", "hypothetical_update", "resample", "reset", "est_mean",
                 "est_covariance_mtx", "est_meanfn", "est_entropy", "sample"]:
        assert callable(getattr(qi.SMCUpdater, meth))
    for prop in ["resample_count", "just_resampled", "normalization_record", "log_total_likelihood",
                 "min_n_ess", "data_record", "resampling_divergences", "n_particles", "n_ess", "n_rvs",
                 "particle_locations", "particle_weights"]:
        assert isinstance(getattr(qi.SMCUpdater, prop), property), prop


def test_model_metadata():
    m = qi.SimplePrecessionModel()
    assert m.n_modelparams == 1 and m.expparams_dtype == 'float' and m.n_outcomes(None) == 2
    assert m.is_n_outcomes_constant and m.modelparam_names == [r'\omega']
    assert qi.SimpleInversionModel().expparams_dtype == [('t', 'float'), ('w_', 'float')]
    assert m.domain(None).values.tolist() == [0, 1]
    b = qi.BinomialModel(m)
    assert b.expparams_dtype == [('x', 'float'), ('n_meas', 'uint')]
    assert not b.is_n_outcomes_constant and b.underlying_model is m and b.base_model is m
    assert b.model_chain == (m,) and b.decorated_model is m
    ep = np.empty((2,), dtype=b.expparams_dtype)
    ep['x'], ep['n_meas'] = [1.0, 2.0], [25, 7]
    assert b.n_outcomes(ep).tolist() == [26, 8]
    assert [d.max for d in b.domain(ep)] == [25, 7]
    rb = qi.RandomizedBenchmarkingModel()
    assert rb.n_modelparams == 3 and rb.modelparam_names == ['p', 'A', 'B']
    assert rb.expparams_dtype == [('m', 'uint')]
    rbi = qi.RandomizedBenchmarkingModel(interleaved=True)
    assert rbi.n_modelparams == 4 and rbi.expparams_dtype == [('m', 'uint'), ('reference', bool)]
    with pytest.raises(NotImplementedError):
        qi.RandomizedBenchmarkingModel(order=1)
    with pytest.raises(ValueError):
        qi.BinomialModel(b)                      # not a two-outcome model
    tm = qi.TomographyModel(qi.tomography.pauli_basis(2))
    assert tm.n_modelparams == 16 and tm.dim == 4
    assert np.dtype(tm.expparams_dtype)['meas'].shape == (16,)
    assert tm.are_models_valid(np.zeros((5, 16))).all()
    assert np.array_equal(m.update_timestep(np.ones((3, 1)), np.zeros(2)), np.ones((3, 1, 2)))
    assert m.Q.tolist() == [1.0] and m.experiment_cost(np.zeros(3)).tolist() == [1, 1, 1]


def test_expparam_translation():
    m = qi.SimplePrecessionModel()
    eps = m._native_expparams(np.array([0.5, 2.0]))
    assert [e.t for e in eps] == [0.5, 2.0] and all(e.w_ == 0 for e in eps)
    rec = np.array([(3.0, 0.25)], dtype=qi.SimpleInversionModel().expparams_dtype)
    e = qi.SimpleInversionModel()._native_expparams(rec)[0]
    assert (e.t, e.w_) == (3.0, 0.25)
    e = m._native_expparams(np.array([(4.0, 0.0)], dtype=[('t', float), ('w_', float)]))[0]
    assert e.t == 4.0
    b = qi.BinomialModel(m)
    ep = np.empty((1,), dtype=b.expparams_dtype)
    ep['x'], ep['n_meas'] = 1.5, 25
    e = b._native_expparams(ep)[0]
    assert (e.t, e.n_meas) == (1.5, 25)
    assert b._native_desc().kind == _native.MODEL_BINOMIAL_PRECESSION
    rbi = qi.RandomizedBenchmarkingModel(interleaved=True)
    ep = np.empty((2,), dtype=rbi.expparams_dtype)
    ep['m'], ep['reference'] = [5, 9], [True, False]
    es = rbi._native_expparams(ep)
    assert [(e.m, e.reference) for e in es] == [(5, 1), (9, 0)]
    assert rbi._native_desc().d == 4
    tm = qi.TomographyModel(qi.tomography.pauli_basis(1))
    ep = np.zeros((1,), dtype=tm.expparams_dtype)
    ep['meas'][0] = [1, 0, 0, 1]
    e = tm._native_expparams(ep)[0]
    assert list(e.meas)[:4] == [1, 0, 0, 1]
    wide = _native.make_expparam(meas=np.arange(17.0))           # d > QSMC_MAX_D: a host array of its own behind a pointer
    assert wide.meas_wide and list(wide._wide) == list(np.arange(17.0)) and list(wide.meas) == [0.0] * 16
    with pytest.raises(ValueError):
        _native.make_expparam(meas=np.zeros(65))
    big = qi.TomographyModel(qi.tomography.gell_mann_basis(5))   # d = 25: the wide kernels (csrc/kernels/wide.hpp)
    assert big._native is True and big._native_canonicalize_ok()
    ep = np.zeros((1,), dtype=big.expparams_dtype)
    ep['meas'][0, 3] = 0.25
    e = _native.ExpParam()
    assert big._native_fill_expparam(e, ep) and e.meas_wide and e._wide[3] == 0.25
    huge = qi.TomographyModel(qi.tomography.gell_mann_basis(9))  # d = 81 > QSMC_MAX_D_WIDE: plugin path
    assert huge._native is False and not huge._native_canonicalize_ok()


def test_bases_match_oracle():
    np.testing.assert_allclose(qi.tomography.gell_mann_basis(3).data, orc.gell_mann_data(3), atol=1e-15)
    np.testing.assert_allclose(qi.tomography.pauli_basis(2).data, orc.pauli_data(2), atol=1e-15)
    b = qi.tomography.pauli_basis(2)
    gram = np.einsum('aij,bij->ab', b.data.conj(), b.data)
    np.testing.assert_allclose(gram, np.eye(16), atol=1e-14)            # orthonormal
    np.testing.assert_allclose(b.data[0], np.eye(4) / 2, atol=1e-15)    # B_0 = 1 / sqrt(dim)
    rho = np.diag([0.4, 0.3, 0.2, 0.1]).astype(complex)
    x = b.state_to_modelparams(rho)
    np.testing.assert_allclose(b.modelparams_to_state(x), rho, atol=1e-14)
    assert x[0] == pytest.approx(0.5)
    assert len(b) == 16 and b.dim == 4 and "pauli_basis" in repr(b)
    tp = qi.tomography.tensor_product_basis(qi.tomography.gell_mann_basis(2), qi.tomography.gell_mann_basis(3))
    assert tp.data.shape == (36, 6, 6) and tp.dims == [2, 3]


def test_priors():
    np.random.seed(0)
    u = qi.UniformDistribution([[0, 1], [2, 4]])
    s = u.sample(1000)
    assert s.shape == (1000, 2) and s[:, 1].min() >= 2 and s[:, 1].max() <= 4 and u.n_rvs == 2
    np.random.seed(0)
    ref = np.random.random((1000, 2)) * np.array([1, 2]) + np.array([0, 2])
    np.testing.assert_array_equal(s, ref)                   # same stream consumption as the reference
    assert qi.UniformDistribution([0, 2]).n_rvs == 1
    p = qi.ProductDistribution(qi.UniformDistribution([0, 1]), u)
    assert p.n_rvs == 3 and p.sample(7).shape == (7, 3)
    assert qi.ProductDistribution([u, u]).n_rvs == 4
    mvn = qi.MultivariateNormalDistribution(np.array([1.0, -1.0]), np.array([[2.0, 0.3], [0.3, 1.0]]))
    s = mvn.sample(20000)
    np.testing.assert_allclose(s.mean(axis=0), [1, -1], atol=0.05)
    np.testing.assert_allclose(np.cov(s.T), mvn.cov, atol=0.08)

    class Half:
        def are_models_valid(self, mp):
            return mp[:, 0] > 0.5
    ps = qi.PostselectedDistribution(qi.UniformDistribution([0, 1]), Half())
    s = ps.sample(500)
    assert s.min() > 0.5 and ps.n_rvs == 1

    class Never:
        def are_models_valid(self, mp):
            return np.zeros(mp.shape[0], dtype=bool)
    with pytest.raises(RuntimeError):
        qi.PostselectedDistribution(qi.UniformDistribution([0, 1]), Never(), maxiters=3).sample(4)
    g = qi.GinibreDistribution(qi.tomography.pauli_basis(2))
    x = g.sample(20)
    assert x.shape == (20, 16)
    np.testing.assert_allclose(x[:, 0], 0.5, atol=1e-14)    # trace one
    rho = np.tensordot(x, qi.tomography.pauli_basis(2).data, 1)
    assert np.linalg.eigvalsh(rho).min() > -1e-14


def test_liu_west_parameters():
    r = qi.LiuWestResampler()
    assert r.a == 0.98 and r.h == pytest.approx(np.sqrt(1 - 0.98 ** 2))
    r.a = 0.9
    assert r.h == pytest.approx(np.sqrt(1 - 0.81))
    r = qi.LiuWestResampler(a=1.0, h=0.005)
    r.a = 0.5
    assert r.h == 0.005                                     # an explicit h is never overridden
    assert isinstance(r, qi.Resampler)
    with pytest.raises(TypeError):
        qi.Resampler()


def test_pr0_to_likelihood_array_and_domain():
    pr0 = np.array([[0.2, 0.7]])
    L = qi.FiniteOutcomeModel.pr0_to_likelihood_array(np.array([0, 1, 1]), pr0)
    assert L.shape == (3, 1, 2)
    np.testing.assert_allclose(L[0], pr0)
    np.testing.assert_allclose(L[1], 1 - pr0)
    d = qi.IntegerDomain(min=0, max=3)
    assert d.n_members == 4 and d.values.tolist() == [0, 1, 2, 3] and d.in_domain([1, 2]) and not d.in_domain([4])


def test_simple_est_data_tables():
    """data_to_params / load_data_or_txt (simple_est.py:69-118): positional columns of a 2-D array, named
    columns of a record array and a CSV file give the same (outcomes, expparams)."""
    import io
    from qinfer_amd.simple_est import data_to_params, load_data_or_txt
    dtype = [('x', 'float'), ('n_meas', 'uint')]
    table = np.array([[3, 1.5, 10], [7, 2.5, 12], [0, 4.0, 9]], dtype=float)
    cols = {'x': (1, 't'), 'n_meas': (2, 'n_shots')}
    o1, e1 = data_to_params(table, dtype, cols_expparams=cols)
    assert o1.dtype.kind == 'i' and o1.tolist() == [3, 7, 0]
    assert e1['x'].tolist() == [1.5, 2.5, 4.0] and e1['n_meas'].tolist() == [10, 12, 9]
    rec = np.array([(3, 1.5, 10), (7, 2.5, 12), (0, 4.0, 9)], dtype=[('counts', 'uint'), ('t', float), ('n_shots', 'uint')])
    o2, e2 = data_to_params(rec, dtype, cols_expparams=cols)
    assert np.array_equal(o1, o2) and np.array_equal(e1, e2)
    csv = io.StringIO("3,1.5,10\n7,2.5,12\n0,4.0,9\n")
    loaded = load_data_or_txt(csv, [('counts', 'uint'), ('t', float), ('n_shots', 'uint')])
    o3, e3 = data_to_params(loaded, dtype, cols_expparams=cols)
    assert np.array_equal(o1, o3) and np.array_equal(e1, e3)
    assert load_data_or_txt(table, None) is table
    with pytest.raises(TypeError):
        load_data_or_txt(12345, None)
    # scalar expparams dtype: one column
    o4, e4 = data_to_params(table, np.float64, cols_expparams=(1, 't'))
    assert e4.tolist() == [1.5, 2.5, 4.0]


def test_mvee_and_in_ellipsoid(golden):
    """utils.mvee / in_ellipsoid (reference utils.py:314-374) against the reference's own output (G13)."""
    from qinfer_amd import utils as u
    g = golden("g13_regions")
    A, c = u.mvee(g["mvee3_pts"], 1e-5)
    np.testing.assert_allclose(A, g["mvee3_A"], rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(c, g["mvee3_c"], rtol=1e-9, atol=1e-12)
    inside = u.in_ellipsoid(g["mvee3_pts"], np.linalg.inv(A) * (1 + 1e-3), c)       # every point is enclosed
    assert inside.all()
    assert bool(u.in_ellipsoid(c, np.linalg.inv(A), c)) and not bool(u.in_ellipsoid(c + 100.0, np.linalg.inv(A), c))
    assert u.uniquify([3, 1, 3, 2, 1]) == [3, 1, 2]


def test_f64_ptr_paths():
    """_native.f64_ptr: the from_buffer fast path and the data_as fall-back address the same memory."""
    import ctypes as C
    a = np.arange(6, dtype=np.float64)
    p = _native.f64_ptr(a)                                   # writeable: a ctypes array over the buffer
    assert C.addressof(p) == a.ctypes.data and list(p) == a.tolist()
    p[2] = -1.0
    assert a[2] == -1.0
    ro = np.arange(4, dtype=np.float64)
    ro.setflags(write=False)
    q = _native.f64_ptr(ro)                                  # read-only: ndarray.ctypes.data_as
    assert C.cast(q, C.c_void_p).value == ro.ctypes.data and q[3] == 3.0
    with pytest.raises(AssertionError):
        _native.f64_ptr(np.zeros(3, dtype=np.float32))
    with pytest.raises(AssertionError):
        _native.f64_ptr(np.zeros((4, 4))[:, ::2])


def test_cov_from_sums_scalar_path_equals_matrix_path():
    """ParticleDistribution._cov_from_sums: the one-parameter scalar path gives the matrix path's number bit for bit
    and the same warning / assertion behaviour (distributions.py:386-399)."""
    import warnings
    rs = np.random.RandomState(2)
    f = qi.ParticleDistribution._cov_from_sums
    for _ in range(200):
        m = rs.uniform(-3, 3)
        s1, s2 = np.array([m]), np.array([[m * m + rs.uniform(0, 1e-3) * rs.choice([1, 1e-9])]])
        want = s2 - np.outer(s1, s1)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            got = f(s1, s2)
        assert got.shape == (1, 1) and got[0, 0] == want[0, 0]
    with pytest.warns(qi.ApproximationWarning):
        f(np.array([1.0]), np.array([[1.0 - 1e-12]]))        # negative variance from cancellation: warned, returned
    with pytest.raises(AssertionError):
        f(np.array([np.nan]), np.array([[1.0]]))
    with pytest.warns(qi.ApproximationWarning):               # d = 2: the eigenvalue test of the matrix path
        f(np.array([0.0, 0.0]), np.array([[1.0, 2.0], [2.0, 1.0]]))


def test_make_expparam_measurement_vector():
    v = np.linspace(0.0, 1.5, 16)
    e = _native.make_expparam(meas=v)
    assert list(e.meas) == v.tolist()
    e = _native.make_expparam(meas=v[:5])
    assert list(e.meas)[:5] == v[:5].tolist() and all(x == 0.0 for x in list(e.meas)[5:])



def test_device_backed_array_copies_do_not_write_through():
    """A write-through snapshot (distributions.DeviceBackedArray) uploads on in-place writes to itself and to VIEWS of
    itself -- and only those: what fancy / boolean-mask indexing returns owns its memory (as with the reference's
    plain arrays) and must neither upload nor raise 'stale snapshot' when edited after the cloud changed."""
    from qinfer_amd.distributions import DeviceBackedArray

    class Owner:
        _view_version = 0

        def __init__(self):
            self.uploads = []

        def _write_back(self, what, arr):
            self.uploads.append((what, np.array(arr)))
            self._view_version += 1

    own = Owner()
    snap = DeviceBackedArray(np.arange(10.0), own, "weights")
    mask = np.arange(10) % 2 == 0
    sel, fancy, taken = snap[mask], snap[[1, 2, 3]], snap.take([4, 5])
    view = snap[2:5]
    assert not np.shares_memory(sel, snap) and np.shares_memory(view, snap)
    own._view_version += 1                                # an update / resample happened: `snap` is stale now
    sel[0] = -1.0                                         # private copies: no upload, no raise
    fancy += 1.0
    taken.fill(0.0)
    sel.sort()
    assert own.uploads == [] and snap[0] == 0.0
    with pytest.raises(RuntimeError, match="snapshot"):
        view[0] = 7.0                                     # a view of the stale snapshot is stale too
    # a fresh snapshot: every in-place path uploads, through views as well
    own = Owner()
    snap = DeviceBackedArray(np.arange(10.0), own, "locations")
    snap[3] = 30.0
    snap[2:5][0] = 20.0
    snap.T[9] = 90.0
    np.copyto(snap, snap[::-1].copy())
    np.put(snap, [0], [5.0])
    np.add.at(snap, [1, 1], 1.0)
    np.place(snap, snap > 80, [1.0])
    np.putmask(snap, np.asarray(snap) == 1.0, 2.0)
    snap.sort()
    snap.put([0], [-3.0])
    assert len(own.uploads) == 10
    assert own.uploads[-1][1][0] == -3.0 and own.uploads[0][1][3] == 30.0
    out = snap * 2                                        # arithmetic: a plain result
    out[0] = 1e9
    assert len(own.uploads) == 10


def test_native_ok_judges_by_defining_class():
    """abstract_model.native_ok: the library's kernels stand for a model only if every kernel-backed method is DEFINED
    by a class of the library -- judged along the MRO (a property, a functools.partial or a builtin has no __module__ of
    its own), cached per class, and a user module whose name merely starts with 'qinfer_amd.' does not pass."""
    import functools
    import qinfer_amd as qi
    from qinfer_amd import abstract_model as am

    class Plain(qi.SimplePrecessionModel):                # adds nothing kernel-backed: still native
        def extra(self):
            return 1

    class WithPartial(qi.SimplePrecessionModel):          # an override without a __module__ of its own: plugin path
        likelihood = functools.partial(lambda self, *a, **k: None)
    assert getattr(WithPartial.likelihood, "__module__", None) in (None, "functools")

    class Spoof(qi.SimplePrecessionModel):
        def are_models_valid(self, modelparams):
            return np.ones(modelparams.shape[0], dtype=bool)
    Spoof.__module__ = "qinfer_amd.userstuff"
    assert am.native_ok(qi.SimplePrecessionModel()) and am.native_ok(Plain())
    assert not am.native_ok(WithPartial()) and not am.native_ok(Spoof())
    assert am._CLASS_OK[Plain] is True and am._CLASS_OK[Spoof] is False
    # BinomialModel follows the same rule for its underlying model
    assert qi.BinomialModel(Plain())._native and not qi.BinomialModel(Spoof())._native
    assert qi.BinomialModel(qi.RandomizedBenchmarkingModel())._native


def test_gaussian_random_walk_step_laws_seed_for_seed(golden):
    """GaussianRandomWalkModel.update_timestep, all four covariance variants (given diagonal / given dense / learned
    diagonal / learned dense), against ONE seeded call of the reference each (fixture g10_grw_steps, generated by
    oracle/gen_golden.py from derived_models.py:920-963): the same draw shapes in the same order from the legacy
    global stream, so a seeded host-RNG run consumes np.random as QInfer does."""
    g = golden("g10_grw_steps")
    t2 = qi.UnknownT2Model()
    ep = np.empty((3,), dtype=t2.expparams_dtype)
    ep['t'] = g["expparam_t"]
    kws = {"known_diag": dict(fixed_covariance=np.array([4e-4, 9e-4])),
           "known_dense": dict(fixed_covariance=g["cov_dense"], diagonal=False),
           "learned_diag": dict(), "learned_dense": dict(diagonal=False)}
    assert sorted(kws) == sorted(str(t) for t in g["tags"])
    for tag, kw in kws.items():
        m = qi.GaussianRandomWalkModel(t2, scale_mult=lambda e: np.sqrt(e['t']), **kw)
        mp = g[tag + "_in"]
        assert m.n_modelparams == mp.shape[1], tag
        np.random.seed(4242)
        out = m.update_timestep(mp.copy(), ep)
        np.testing.assert_allclose(out, g[tag + "_out"], rtol=0, atol=1e-15, err_msg=tag)


def _pack_lower(A):
    out = np.empty((A.shape[0], 16))
    k = 0
    for r in range(4):
        for c in range(r + 1):
            out[:, k] = A[:, r, c].real
            k += 1
    for r in range(4):
        for c in range(r):
            out[:, k] = A[:, r, c].imag
            k += 1
    return out


def test_psd_projection_without_eigenvectors_host_harness(tmp_path):
    """psd_project4 (qsmc_device.h, round 5: the clamp of tomography/models.py:185-192 as a polynomial in rho over
    eigenvalues from eigenvector-free Jacobi sweeps) and jacobi_clamp (the eigenvector form it replaces on the list pass
    of a 2-qubit canonicalize), both compiled for the HOST from the device header (tests/harness/canon_host.hip), against
    numpy.linalg.eigh: the workload's spectra (Ginibre states plus Hermitian noise) and prescribed spectra with gaps from
    1e-1 down to 1e-15 -- clusters of negative eigenvalues, a negative next to a positive, everything around zero.
    Every result psd_project4 vouches for (verdict 1) is within 1e-13 of the exact projection; the one family it cannot do
    (three eigenvalues clustered across zero) is flagged (verdict 2), never returned wrong; verdict 0 = no negative
    eigenvalue."""
    import os
    import shutil
    import subprocess
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    if not (os.path.exists(hipcc) or shutil.which(hipcc)):
        pytest.skip("no hipcc")
    src = os.path.join(os.path.dirname(os.path.abspath(__file__)), "harness", "canon_host.hip")
    exe = str(tmp_path / "canon_host")
    subprocess.run([hipcc, "--offload-arch=gfx950", "-O2", "-std=c++17", "-ffp-contract=off", src, "-o", exe], check=True)
    rs = np.random.RandomState(3)

    def herm(M):
        return (M + M.conj().transpose(0, 2, 1)) / 2

    def run(A):
        raw = subprocess.run([exe], input=_pack_lower(A).tobytes(), capture_output=True, check=True).stdout
        o = np.frombuffer(raw, dtype=np.float64).reshape(-1, 66)
        return (o[:, 0].astype(int), o[:, 1:17].reshape(-1, 4, 4) + 1j * o[:, 17:33].reshape(-1, 4, 4),
                o[:, 33].astype(int), o[:, 34:50].reshape(-1, 4, 4) + 1j * o[:, 50:66].reshape(-1, 4, 4))

    def check(name, A, lam=None, U=None, expect_flag=None):
        if lam is None:
            lam, U = np.linalg.eigh(A)
        ref = (U * np.maximum(lam, 0)[:, None, :]) @ U.conj().transpose(0, 2, 1)
        v, R, jn, J = run(A)
        scale = np.sqrt((np.abs(A) ** 2).sum(axis=(1, 2)))
        ok = v == 1
        if ok.any():
            err = np.abs(R - ref).max(axis=(1, 2))[ok] / np.maximum(scale[ok], 1e-300)
            assert err.max() < 1e-13, (name, err.max())
        # the eigenvector form on everything with a negative eigenvalue (the fallback of the flagged ones)
        if (jn == 1).any():
            errj = np.abs(J - ref).max(axis=(1, 2))[jn == 1] / np.maximum(scale[jn == 1], 1e-300)
            assert errj.max() < 1e-13, (name, "jacobi_clamp", errj.max())
        neg = lam[:, 0] < -1e-14 * scale
        assert not np.any((v == 0) & neg), name               # a negative eigenvalue is never missed
        if expect_flag is not None:
            assert (v == 2).mean() >= expect_flag, (name, (v == 2).mean())
        return v

    n = 20000
    g = rs.randn(n, 4, 4) + 1j * rs.randn(n, 4, 4)
    rho = g @ g.conj().transpose(0, 2, 1)
    rho /= np.trace(rho, axis1=1, axis2=2).real[:, None, None]
    for noise in (0.05, 0.15, 0.4):
        h = herm(rs.randn(n, 4, 4) + 1j * rs.randn(n, 4, 4))
        v = check("ginibre + %.2f" % noise, rho + noise * h)
        assert (v == 2).sum() == 0                            # the workload never needs the fallback
    n = 4000
    U = np.linalg.qr(rs.randn(n, 4, 4) + 1j * rs.randn(n, 4, 4))[0]
    one, u = np.ones(n), rs.rand
    for gap in (1e-1, 1e-2, 1e-4, 1e-7, 1e-10, 1e-13, 1e-15):
        fams = {
            "two negative close": [-0.1 * one, -0.1 + gap * u(n), 0.4 * one, 0.8 * one],
            "across zero": [-gap * u(n), gap * u(n), 0.3 * one, 0.7 * one],
            "negative, two positive close": [-0.05 * one, 0.2 * one, 0.2 + gap * u(n), 0.65 * one],
            "three negative close": [-0.1 * one, -0.1 + gap * u(n), -0.1 + 2 * gap * u(n), 1.3 * one],
            "three across zero": [-gap * u(n), gap * u(n), 2 * gap * u(n), one],
            "all four around zero": [gap * (u(n) - 0.5) for _ in range(4)],
        }
        for name, cols in fams.items():
            lam = np.sort(np.stack(cols, 1), axis=1)
            A = herm((U * lam[:, None, :]) @ U.conj().transpose(0, 2, 1))
            check("%s, gap %g" % (name, gap), A, lam, U,
                  expect_flag=0.9 if (name == "three across zero" and 1e-13 <= gap <= 1e-3) else None)
