"""Quantum-state tomography model on the SMC path (reference `qinfer/tomography/`):

    TomographyBasis, gell_mann_basis, pauli_basis, tensor_product_basis   bases.py:71-154, 180-374
    TomographyModel (likelihood / canonicalize / renormalize)             models.py:82-226
    GinibreDistribution                                                   distributions.py:168-196

States are real coefficient vectors x in an orthonormal Hermitian operator basis {B_a} whose
element 0 is identity / sqrt(dim); Pr(1 | x; meas) = clip(meas . x, 0, 1).  `canonicalize` clamps
negative eigenvalues of rho(x) = sum_a x_a B_a and renormalises the trace -- on the GPU this is a
per-particle complex-Hermitian Jacobi (csrc/qsmc_device.h `tomo_canon_particle`), replacing the
reference's per-particle Python `np.linalg.eig` loop.

QuTiP is not required: the reference uses it only to convert to/from `Qobj`, and for
`rand_dm_ginibre`, which is restated here directly (X = randn + i randn, rho = X X^+ / tr).
"""
import itertools as it
from functools import reduce

import numpy as np

from . import _native
from .abstract_model import FiniteOutcomeModel, NativeModelMixin
from .distributions import Distribution

__all__ = ["TomographyBasis", "gell_mann_basis", "pauli_basis", "tensor_product_basis",
           "TomographyModel", "GinibreDistribution"]


class TomographyBasis:
    """Orthonormal Hermitian operator basis; `data[a, i, j]` is element (i, j) of B_a."""

    def __init__(self, data, dims, labels=None, superrep=None, name=None):
        self.data = np.asarray(data, dtype=complex)
        self.dims = list(dims)
        self.superrep = superrep
        self._name = name if name is not None else "(unnamed)"
        n = self.dim ** 2
        if isinstance(labels, str):
            self.labels = ["{}_{{{}}}".format(labels, i) for i in range(n)]
        elif labels is None:
            self.labels = ["B_{}".format(i) for i in range(n)]
        else:
            self.labels = list(labels)
        self._flat = self.data.reshape((self.data.shape[0], -1))

    def __repr__(self):
        return "<TomographyBasis {} dims={} at 0x{:0x}>".format(self._name, self.dims, id(self))

    def __len__(self):
        return self.dim ** 2

    @property
    def dim(self):
        return int(np.prod(self.dims))

    @property
    def name(self):
        return self._name

    def flat(self):
        return self._flat

    def state_to_modelparams(self, state):
        """Density matrix (dim x dim array-like) -> real coefficient vector."""
        rho = np.asarray(getattr(state, "full", lambda: state)(), dtype=complex)
        return np.real(np.dot(self._flat.conj(), rho.flatten()))

    def modelparams_to_state(self, modelparams):
        """Coefficient vector(s) -> dim x dim complex array(s)."""
        modelparams = np.asarray(modelparams)
        if modelparams.ndim == 1:
            return np.tensordot(modelparams, self.data, 1)
        return [self.modelparams_to_state(mp) for mp in modelparams]

    def covariance_mtx_to_superop(self, mtx):
        M = self._flat
        return np.dot(np.dot(M.conj().T, mtx), M)


def gell_mann_basis(dim):
    """Generalised Gell-Mann matrices, normalised to tr(B_a B_b) = delta_ab; B_0 = 1/sqrt(dim)."""
    data = np.zeros((dim * dim, dim, dim), dtype=complex)
    data[0] = np.eye(dim) / np.sqrt(dim)
    for r in range(1, dim):                                  # diagonal family
        diag = np.concatenate([np.ones(r), [-r], np.zeros(dim - r - 1)])
        data[r] = np.diag(diag) / np.sqrt(r + r * r)
    n_pairs = dim * (dim - 1) // 2
    for i in range(1, dim):                                  # symmetric / antisymmetric families
        for j in range(i):
            k = (i - 1) * i // 2 + j + dim
            data[k, i, j] = data[k, j, i] = 1 / np.sqrt(2)
            data[k + n_pairs, i, j] = 1j / np.sqrt(2)
            data[k + n_pairs, j, i] = -1j / np.sqrt(2)
    return TomographyBasis(data, [dim], r'\gamma', name='gell_mann_basis')


def tensor_product_basis(*bases):
    """Basis of all tensor products of elements of the factor bases (row-major over factors)."""
    dim = int(np.prod([b.data.shape[1] for b in bases]))
    data = np.zeros((dim * dim, dim, dim), dtype=complex)
    for k, factors in enumerate(it.product(*[b.data for b in bases])):
        data[k] = reduce(np.kron, factors)
    labels = [r"\otimes".join(ls) for ls in it.product(*[b.labels for b in bases])]
    return TomographyBasis(data, sum((b.dims for b in bases), []), labels)


def pauli_basis(nq=1):
    """Normalised n-qubit Pauli basis, ordered (I, X, Y, Z) per qubit."""
    single = TomographyBasis(gell_mann_basis(2).data[[0, 2, 3, 1]], [2],
                             [u'\U0001D7D9', r'\sigma_x', r'\sigma_y', r'\sigma_z'])
    basis = tensor_product_basis(*([single] * nq))
    basis._name = 'pauli_basis'
    return basis


class TomographyModel(NativeModelMixin, FiniteOutcomeModel):
    """Two-outcome tomography: outcome 1 has probability <<meas | rho>> clipped to [0, 1]."""

    def __init__(self, basis, allow_subnormalized=False):
        self._dim = basis.dim
        self._basis = basis
        self._allow_subnormalized = bool(allow_subnormalized)
        self._basis_dev = None
        self._is_pauli = None
        super().__init__()
        # d <= 16: the narrow kernels; 16 < d <= 64 (dim 5 .. 8, three qubits): the wide ones (csrc/kernels/wide.hpp)
        self._native = self.n_modelparams <= _native.QSMC_MAX_D_WIDE

    @property
    def dim(self):
        return self._dim

    @property
    def basis(self):
        return self._basis

    @property
    def n_modelparams(self):
        return self._dim ** 2

    @property
    def modelparam_names(self):
        return [r'\langle\!\langle{} | \rho\rangle\!\rangle'.format(lbl) for lbl in self._basis.labels]

    @property
    def is_n_outcomes_constant(self):
        return True

    @property
    def expparams_dtype(self):
        return [(str('meas'), float, self._dim ** 2)]

    def n_outcomes(self, expparams):
        return 2

    # native hooks
    def _native_desc(self):
        return _native.ModelDesc(_native.MODEL_TOMOGRAPHY, self.n_modelparams, 0.0, 1, 0)

    def _native_fill_expparam(self, ep, expparams):
        d = self.n_modelparams
        if type(expparams) is np.ndarray and expparams.shape == (1,) and d <= _native.QSMC_MAX_D_WIDE:
            # (a NumPy view of the struct's array, made once per struct: 0.5 us per datum against 1.6 us for a list
            #  assigned to a ctypes slice -- this sits inside every update() of a 38 us step)
            view = ep.__dict__.get("_meas_view")
            if view is None:
                if d <= _native.QSMC_MAX_D:
                    view = ep._meas_view = np.ctypeslib.as_array(ep.meas)
                else:                        # wide: the struct points at a host array that lives with it
                    view = ep._meas_view = ep._wide = np.zeros(d, dtype=np.float64)
                    ep.meas_wide = view.ctypes.data
            view[:d] = expparams['meas'][0]
            return True
        return False

    def _native_expparams(self, expparams):
        expparams = np.atleast_1d(expparams)
        meas = np.asarray(expparams['meas'], dtype=np.float64).reshape(-1, self.n_modelparams)
        return [_native.make_expparam(meas=row) for row in meas]

    def _device_basis(self, eng):
        if self._basis_dev is None or self._basis_dev.device != eng.device:
            inter = np.ascontiguousarray(self._basis.data).view(np.float64)     # (d, dim, 2 dim)
            self._basis_dev = eng.to_device(inter.reshape(-1))
        return self._basis_dev

    def _native_canonicalize_ok(self):
        """Device canonicalize kernels exist for dim 2 .. 8 (dim 2, 3, 4: the narrow kernels; dim 5 .. 8, up to three
        qubits: classify + Jacobi list, csrc/kernels/wide.hpp); larger systems take `canonicalize` on the host."""
        return 2 <= self._dim <= 8

    def _pauli(self):
        if self._is_pauli is None:               # is this the reference's Pauli basis, element for element?
            nq = int(round(np.log2(self._dim)))
            self._is_pauli = bool(2 ** nq == self._dim and np.array_equal(self._basis.data, pauli_basis(nq).data))
        return self._is_pauli

    def _native_canonicalize_(self, eng, x):
        """In-place canonicalize of a device SoA cloud."""
        if not self._native_canonicalize_ok():
            raise NotImplementedError("native canonicalize supports dim 2 .. 8")
        eng.tomo_canonicalize(self._device_basis(eng), self._dim, x, self._allow_subnormalized, pauli=self._pauli())

    def _native_canonicalize_fused(self, eng):
        """(kind, basis device tensor or None, allow_subnormalized) if the device-RNG Liu-West resample can fold
        `canonicalize` of its output into its own kernels (qsmc_lw_fuse_canonicalize: 2 qubits), else None.
        kind 1: the reference's Pauli basis (sparse contraction), 2: a dense basis."""
        if self._dim != 4:
            return None
        if self._pauli():
            return (1, None, self._allow_subnormalized)
        return (2, self._device_basis(eng), self._allow_subnormalized)

    # NumPy contract
    def are_models_valid(self, modelparams):
        # tomography/models.py:143-147: deliberately always true
        return np.ones((np.asarray(modelparams).shape[0],), dtype=bool)

    def likelihood(self, outcomes, modelparams, expparams):
        super().likelihood(outcomes, modelparams, expparams)
        if not self._native:
            # d > 64 (four qubits and up): no kernels of the library's own.  Host arrays in, host arrays out -- what
            # `simulate_experiment` and other host callers ask for; an updater keeps its cloud on the GPU through the
            # torch hooks below (`likelihood_device`, `canonicalize_device`: the plugin surface's device form)
            meas = np.asarray(np.atleast_1d(expparams)['meas'], dtype=np.float64).reshape(-1, self.n_modelparams)
            pr1 = np.clip(np.asarray(modelparams, dtype=np.float64) @ meas.T, 0, 1)
            return FiniteOutcomeModel.pr0_to_likelihood_array(outcomes, 1 - pr1)
        return self._native_likelihood(outcomes, modelparams, expparams)

    def __getattr__(self, name):
        # the device hooks exist only for models WITHOUT native kernels (an updater asks with getattr(model, hook, None));
        # a native model must not grow them: its kernels are the path
        if name in ("likelihood_device", "canonicalize_device") and not self.__dict__.get("_native", True):
            return getattr(self, "_torch_" + name)
        raise AttributeError(name)

    def _torch_likelihood_device(self, outcomes, x_dev, expparams):
        """(n_o, n_e, N) device tensor: Pr(outcome | x) = clip(<<meas | x>>, 0, 1) or one minus it (tomography/models.py:211-226)
        by a torch product on the (d, N) cloud."""
        import torch
        self.count_likelihood_calls(len(np.ravel(outcomes)), x_dev.shape[1], np.atleast_1d(expparams).shape[0])
        meas = torch.as_tensor(np.asarray(np.atleast_1d(expparams)['meas'], dtype=np.float64).reshape(-1, self.n_modelparams),
                               device=x_dev.device)
        pr1 = (meas @ x_dev).clamp_(0, 1)                                   # (n_e, N)
        return torch.stack([pr1 if int(o) == 1 else 1 - pr1 for o in np.ravel(outcomes)])

    def _torch_canonicalize_device(self, x_dev):
        """tomography/models.py:149-209 on the device: rho of every particle, batched Hermitian eigendecomposition, negative
        eigenvalues clamped, re-expanded, trace renormalised -- in blocks (a block of 4096 sixteen-by-sixteen matrices)."""
        import torch
        d, n = x_dev.shape
        flat = torch.as_tensor(np.ascontiguousarray(self._basis.flat()), device=x_dev.device)        # (d, dim^2) complex
        out = x_dev.clone()
        for i0 in range(0, n, 4096):
            xb = x_dev[:, i0:i0 + 4096].T.to(torch.complex128)                                       # (m, d)
            rho = (xb @ flat).reshape(-1, self._dim, self._dim)
            rho = 0.5 * (rho + rho.conj().transpose(1, 2))
            lam, v = torch.linalg.eigh(rho)
            fixed = (v * lam.clamp(min=0).to(v.dtype)[:, None, :]) @ v.conj().transpose(1, 2)
            xb_new = (fixed.reshape(fixed.shape[0], -1) @ flat.conj().T).real                        # (m, d)
            neg = (lam < 0).any(dim=1)
            blk = out[:, i0:i0 + 4096]
            blk[:, neg] = xb_new[neg].T
        if not self._allow_subnormalized:
            out = out / (out[0:1] * np.sqrt(self._dim))
        return out

    def canonicalize(self, modelparams):
        """Clamp negative eigenvalues of rho(x), re-expand, renormalise the trace (tomography/models.py:149-209)."""
        modelparams = np.asarray(modelparams, dtype=np.float64)
        if self._native_canonicalize_ok():
            eng = self._engine()
            x = eng.locs_to_soa(modelparams)
            self._native_canonicalize_(eng, x)
            return np.ascontiguousarray(x.cpu().numpy().T)
        # host path for the dimensions without a kernel: batched Hermitian eigendecomposition
        flat = self._basis.flat()                                           # (d, dim * dim)
        rho = (modelparams @ flat).reshape(-1, self._dim, self._dim)
        rho = 0.5 * (rho + rho.conj().transpose(0, 2, 1))
        lam, v = np.linalg.eigh(rho)
        neg = np.any(lam < 0, axis=1)
        out = modelparams.copy()
        if neg.any():
            lam_c = np.where(lam[neg] < 0, 0.0, lam[neg])
            fixed = (v[neg] * lam_c[:, None, :]) @ v[neg].conj().transpose(0, 2, 1)
            out[neg] = np.real(fixed.reshape(fixed.shape[0], -1) @ flat.conj().T)
        return out if self._allow_subnormalized else self.renormalize(out)

    def renormalize(self, modelparams):
        modelparams = np.asarray(modelparams, dtype=np.float64)
        norm = modelparams[:, 0] * np.sqrt(self._dim)
        assert not np.sum(norm == 0)
        return modelparams / norm[:, None]


class GinibreDistribution(Distribution):
    """Ginibre-ensemble prior over density operators of a given rank, as coefficient vectors."""

    def __init__(self, basis, rank=None, device=False):
        self._basis = basis
        self._dim = basis.dim
        self._rank = self._dim if rank is None else int(rank)
        self._device = bool(device)          # extension: `SMCUpdater(device_rng=True)` draws the prior on the GPU (sample_device)

    @property
    def n_rvs(self):
        return self._dim ** 2

    def sample(self, n=1):
        # one legacy-RNG call for all n states, in the order a per-state loop would consume the stream
        # (state i: real parts, then imaginary parts), in blocks so that 1e6 states do not need 1e6 x 16 x 16 temporaries
        flat = self._basis.flat()
        out = np.empty((n, self.n_rvs))
        for i0 in range(0, n, 65536):
            m = min(65536, n - i0)
            z = np.random.randn(m, 2, self._dim, self._rank)
            g = z[:, 0] + 1j * z[:, 1]
            rho = g @ g.conj().transpose(0, 2, 1)
            rho /= np.trace(rho, axis1=1, axis2=2).real[:, None, None]
            out[i0:i0 + m] = np.real(rho.reshape(m, -1) @ flat.conj().T)
        return out

    def sample_device(self, engine, n, seed, epoch, maxiter=None):
        """The same ensemble drawn on the device (`SMCUpdater(device_rng=True)`): torch's own generator keyed by
        (seed, epoch), the states built in blocks by torch products, written straight into the (d, N) cloud.  Prior
        sampling is outside the path (once per `reset`); the host form above is the one the reference's stream
        replays.  Not the same numbers as `sample` -- the same law (invariants tested: trace, positivity, moments)."""
        if not self._device:
            raise NotImplementedError      # (the updater then takes `sample`: the host draw, the default)
        import torch
        dev = engine.device
        gen = torch.Generator(device=dev)
        gen.manual_seed((int(seed) * 0x9E3779B97F4A7C15 + int(epoch)) & (2 ** 63 - 1))
        flat_h = torch.from_numpy(np.ascontiguousarray(self._basis.flat().conj().T)).to(dev)     # (dim^2, d) complex
        x_out = engine.empty(self.n_rvs, n)
        block = 1 << 20
        for i0 in range(0, n, block):
            m = min(block, n - i0)
            z = torch.randn(m, 2, self._dim, self._rank, dtype=torch.float64, device=dev, generator=gen)
            g = torch.complex(z[:, 0], z[:, 1])
            rho = g @ g.conj().transpose(1, 2)
            tr = torch.diagonal(rho, dim1=1, dim2=2).sum(1).real
            rho = rho / tr[:, None, None]
            x_out[:, i0:i0 + m] = (rho.reshape(m, -1) @ flat_h).real.T
        return x_out, 0
