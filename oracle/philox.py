"""NumPy emulation of the device Philox4x32-10 streams (TEST INFRASTRUCTURE).

Mirrors `PhiloxStream` in python-qinfer_amd/csrc/qsmc_device.h bit for bit so the device-RNG
resample / prior kernels can be checked against the oracle on identical random numbers.
"""
import numpy as np

M0, M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
W0, W1 = 0x9E3779B9, 0xBB67AE85
MASK = np.uint64(0xFFFFFFFF)


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    c0, c1, c2, c3 = (np.asarray(c, dtype=np.uint64) & MASK for c in (c0, c1, c2, c3))
    k0, k1 = int(k0) & 0xFFFFFFFF, int(k1) & 0xFFFFFFFF
    for _ in range(10):
        p0, p1 = M0 * c0, M1 * c2
        hi0, lo0 = p0 >> np.uint64(32), p0 & MASK
        hi1, lo1 = p1 >> np.uint64(32), p1 & MASK
        c0, c1, c2, c3 = (hi1 ^ c1 ^ np.uint64(k0)) & MASK, lo1, (hi0 ^ c3 ^ np.uint64(k1)) & MASK, lo0
        k0 = (k0 + W0) & 0xFFFFFFFF
        k1 = (k1 + W1) & 0xFFFFFFFF
    return c0, c1, c2, c3


def u53(a, b):
    return ((a >> np.uint64(5)).astype(np.float64) * 67108864.0 +
            (b >> np.uint64(6)).astype(np.float64)) / 9007199254740992.0


def keys(seed, epoch):
    seed = int(seed) & (2 ** 64 - 1)
    return seed & 0xFFFFFFFF, ((seed >> 32) ^ (int(epoch) >> 16)) & 0xFFFFFFFF


def uniforms(particles, seed, epoch, rnd, slot):
    """(u0, u1) for each particle index at (epoch, round, slot)."""
    particles = np.asarray(particles, dtype=np.uint64)
    k0, k1 = keys(seed, epoch)
    er = np.uint64(((int(epoch) & 0xFFFF) << 16) | int(rnd))
    r = philox4x32_10(particles & MASK, particles >> np.uint64(32), np.full(particles.shape, er),
                      np.full(particles.shape, np.uint64(slot)), k0, k1)
    return u53(r[0], r[1]), u53(r[2], r[3])


def normals(particles, seed, epoch, rnd, slot):
    u0, u1 = uniforms(particles, seed, epoch, rnd, slot)
    r = np.sqrt(-2.0 * np.log(1.0 - u0))
    return r * np.cos(2 * np.pi * u1), r * np.sin(2 * np.pi * u1)


def liu_west_philox(w, x, valid_fn, a, h, seed, epoch, n_out, maxiter=1000, postselect=True,
                    mean=None, cov=None, zero_cov_comp=1e-10):
    """Oracle of qsmc_lw_resample_philox: same draws, same redraw rule (ancestor AND kick)."""
    import np_oracle as orc
    N, d = x.shape
    mean = orc.particle_mean(w, x) if mean is None else mean
    cov = orc.particle_cov(w, x, warn=False) if cov is None else cov
    if np.linalg.norm(cov, 'fro') == 0:
        cov = zero_cov_comp * np.eye(d)
    S = h * orc.sqrtm_psd(cov)[0]
    cdf = np.cumsum(w)
    out = np.empty((n_out, d))
    todo = np.arange(n_out)
    for rnd in range(maxiter):
        if not todo.size:
            break
        u, _ = uniforms(todo, seed, epoch, rnd, 0)
        js = np.minimum(cdf.searchsorted(u, side='right'), N - 1)
        z = np.empty((d, todo.size))
        for q in range(0, d, 2):
            z0, z1 = normals(todo, seed, epoch, rnd, 1 + q // 2)
            z[q] = z0
            if q + 1 < d:
                z[q + 1] = z1
        out[todo] = (a * x[js] + (1 - a) * mean) + (S @ z).T
        ok = valid_fn(out[todo]) if postselect else np.ones(todo.size, dtype=bool)
        todo = todo[~ok]
    return out, todo.size


BUCKET_CHUNK = 4096


def liu_west_philox_bucketed(w, x, valid_fn, a, h, seed, epoch, n_out, maxiter=1000, postselect=True,
                             mean=None, cov=None, zero_cov_comp=1e-10, cdf=None):
    """Oracle of the bucketed device-RNG resampler (k_bucket_count / _plan / _sample): outputs are
    ordered by ancestor CHUNK; counts come from word 0 of each output's Philox block, the position
    inside the chunk from word 1 (independent), retries redraw a global ancestor."""
    import np_oracle as orc
    N, d = x.shape
    mean = orc.particle_mean(w, x) if mean is None else mean
    cov = orc.particle_cov(w, x, warn=False) if cov is None else cov
    if np.linalg.norm(cov, 'fro') == 0:
        cov = zero_cov_comp * np.eye(d)
    S = h * orc.sqrtm_psd(cov)[0]
    cdf = np.cumsum(w) if cdf is None else cdf
    chunks = (N + BUCKET_CHUNK - 1) // BUCKET_CHUNK
    edge_idx = np.minimum((np.arange(chunks) + 1) * BUCKET_CHUNK, N) - 1
    edges = cdf[edge_idx]
    ids = np.arange(n_out)
    u0, _ = uniforms(ids, seed, epoch, 0, 0)
    chunk_of = np.minimum(np.searchsorted(edges, u0, side='right'), chunks - 1)
    counts = np.bincount(chunk_of, minlength=chunks)
    slot_off = np.concatenate([[0], np.cumsum(counts)])
    c_of_slot = np.repeat(np.arange(chunks), counts)                 # chunk of every output slot
    _, u1 = uniforms(ids, seed, epoch, 0, 0)
    lo = np.where(c_of_slot == 0, 0.0, edges[np.maximum(c_of_slot - 1, 0)])
    hi = edges[c_of_slot]
    u = lo + u1 * (hi - lo)
    base = c_of_slot * BUCKET_CHUNK
    end = np.minimum(base + BUCKET_CHUNK, N)
    js = np.minimum(np.maximum(np.searchsorted(cdf, u, side='right'), base), end - 1)
    out = np.empty((n_out, d))

    def kick(todo, rnd, centres):
        z = np.empty((d, todo.size))
        for q in range(0, d, 2):
            z0, z1 = normals(todo, seed, epoch, rnd, 1 + q // 2)
            z[q] = z0
            if q + 1 < d:
                z[q + 1] = z1
        return (a * centres + (1 - a) * mean) + (S @ z).T

    out[:] = kick(ids, 0, x[js])
    ok = valid_fn(out) if postselect else np.ones(n_out, dtype=bool)
    todo = ids[~ok]
    for rnd in range(1, maxiter):
        if not todo.size:
            break
        ur, _ = uniforms(todo, seed, epoch, rnd, 0)
        jr = np.minimum(cdf.searchsorted(ur, side='right'), N - 1)
        out[todo] = kick(todo, rnd, x[jr])
        okr = valid_fn(out[todo]) if postselect else np.ones(todo.size, dtype=bool)
        todo = todo[~okr]
    return out, todo.size, js, counts
