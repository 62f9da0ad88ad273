// kernels/walk_tomo.hpp -- time-step random walk (Model.update_timestep) and tomography canonicalize.
// Part of the single translation unit qsmc_kernels.hip (included there, in this order; not a stand-alone header).
#pragma once

// =============================================================================================
// time-step updates (smc.py:447-449, Model.update_timestep): the cloud takes a random-walk step between
// data.  x[m][i] += scale[m] * z, in place; rows with scale 0 do not move and are not touched.
//   z given (device, [row r of the walking parameters][i]): the host drew the steps (parity mode: the
//     reference's np.random.normal call, or an arbitrary step distribution of a RandomWalkModel);
//   z == nullptr: standard normals from Philox -- pair index P = i >> 1 shares a block across the two
//     particles of a pair for ONE walking parameter r: block (P, epoch, slot r), Box-Muller comp i & 1.
// HBM-bound: reads and writes the walking rows once (16 B per particle per walking parameter).
// =============================================================================================
struct WalkArgs {
    double scale[QSMC_MAX_D_WIDE];      // (wide clouds walk too: up to 64 rows)
    int row[QSMC_MAX_D_WIDE];           // parameter index of walking row r
    int n_rw;
};

__global__ __launch_bounds__(QSMC_BLOCK) void k_random_walk(double *__restrict__ x, int64_t ldx, int64_t n,
                                                            WalkArgs wa, const double *__restrict__ z, int64_t ldz,
                                                            uint32_t k0, uint32_t k1, uint32_t epoch) {
    const int64_t n_pairs = (n + 1) >> 1;
    for (int64_t P = (int64_t)blockIdx.x * QSMC_BLOCK + threadIdx.x; P < n_pairs;
         P += (int64_t)gridDim.x * QSMC_BLOCK) {
        const int64_t i0 = 2 * P, i1 = 2 * P + 1;
        for (int r = 0; r < wa.n_rw; ++r) {
            double z0, z1;
            if (z) {
                z0 = z[r * ldz + i0];
                z1 = i1 < n ? z[r * ldz + i1] : 0.0;
            } else {
                PhiloxStream rng{(uint64_t)P, (epoch << 16), k0, k1};
                rng.normals((uint32_t)r, z0, z1);
            }
            double *row = x + (int64_t)wa.row[r] * ldx;
            row[i0] += wa.scale[r] * z0;
            if (i1 < n) row[i1] += wa.scale[r] * z1;
        }
    }
}

// =============================================================================================
// tomography canonicalize: per-particle dim x dim complex Hermitian Jacobi, clamp, re-expand
// =============================================================================================
template <int DIM, class Basis>
__global__ __launch_bounds__(QSMC_BLOCK) void k_tomo_canon(Basis B, double *__restrict__ x, int64_t ldx, int64_t n,
                                                           int allow_subnormalized) {
    constexpr int D = DIM * DIM;
    for (int64_t i = (int64_t)blockIdx.x * QSMC_BLOCK + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * QSMC_BLOCK) {
        double p[D];
#pragma unroll
        for (int a = 0; a < D; ++a) p[a] = x[a * ldx + i];
        if (tomo_canon_particle<DIM>(B, p, allow_subnormalized != 0)) {
#pragma unroll
            for (int a = 0; a < D; ++a) x[a * ldx + i] = p[a];
        }
    }
}

// Two passes for dim = 4 (2 qubits): the eigendecomposition is ~6000 flops and 240 VGPRs per particle, but a
// particle whose rho is positive definite only needs its trace renormalised -- and about two thirds of a
// freshly resampled cloud are (36 % non-PSD measured after a Liu-West kick).  Deciding per lane inside one
// kernel would not help (a wave is as slow as its slowest lane), so pass 1 classifies with a pivot test
// (tomo_clearly_positive), finishes the clear cases and compacts the others into an index list
// (one atomic per wave); pass 2 runs the Jacobi path on the list only, densely packed.
// The hard particles' indices are collected in LDS (one LDS atomic per wave) and appended to the global list in
// batches, one global atomic per flush: a returning atomic on ONE global word serialises at ~88 per microsecond on this
// part, and one per wave (19531 of them at N = 1.25e6) was 220 of this kernel's 250 us.
constexpr int CLASSIFY_BUF = 2048;
template <int DIM, class Basis>
__global__ __launch_bounds__(QSMC_BLOCK) void k_tomo_classify(Basis B, double *__restrict__ x, int64_t ldx, int64_t n,
                                                              int allow_subnormalized, unsigned int *__restrict__ list,
                                                              unsigned int *__restrict__ count) {
    constexpr int D = DIM * DIM;
    __shared__ unsigned int buf[CLASSIFY_BUF + QSMC_BLOCK];
    __shared__ unsigned int bcount, gbase;
    const int lane = threadIdx.x & (QSMC_WAVE - 1);
    if (threadIdx.x == 0) bcount = 0u;
    __syncthreads();
    auto flush = [&]() {                                           // workgroup-uniform call
        const unsigned int m = bcount;
        if (threadIdx.x == 0) gbase = atomicAdd(count, m);
        __syncthreads();
        for (unsigned int t = threadIdx.x; t < m; t += QSMC_BLOCK) list[gbase + t] = buf[t];
        __syncthreads();
        if (threadIdx.x == 0) bcount = 0u;
        __syncthreads();
    };
    for (int64_t i0 = (int64_t)blockIdx.x * QSMC_BLOCK; i0 < n; i0 += (int64_t)gridDim.x * QSMC_BLOCK) {
        const int64_t i = i0 + threadIdx.x;
        bool hard = false;
        if (i < n) {
            double p[D];
#pragma unroll
            for (int a = 0; a < D; ++a) p[a] = x[a * ldx + i];
            if (tomo_clearly_positive<DIM>(B, p)) {
                if (!allow_subnormalized) {                   // tomography/models.py:194-209
                    const double inv = 1.0 / (p[0] * sqrt((double)DIM));
#pragma unroll
                    for (int a = 0; a < D; ++a) x[a * ldx + i] = p[a] * inv;
                }
            } else {
                hard = true;
            }
        }
        const unsigned long long m = __ballot(hard);
        if (m) {
            unsigned int base = 0;
            if (lane == 0) base = atomicAdd(&bcount, (unsigned int)__popcll(m));
            base = __shfl(base, 0, QSMC_WAVE);
            if (hard) buf[base + __popcll(m & ((1ull << lane) - 1ull))] = (unsigned int)i;
        }
        __syncthreads();
        if (bcount >= CLASSIFY_BUF) flush();                      // (uniform: bcount is read after the barrier)
    }
    if (bcount) flush();
}

template <int DIM, class Basis>
__global__ __launch_bounds__(QSMC_BLOCK) void k_tomo_canon_list(Basis B, double *__restrict__ x, int64_t ldx,
                                                                int allow_subnormalized,
                                                                const unsigned int *__restrict__ list,
                                                                const unsigned int *__restrict__ count) {
    constexpr int D = DIM * DIM;
    const unsigned int m = *count;
    for (unsigned int t = blockIdx.x * QSMC_BLOCK + threadIdx.x; t < m; t += gridDim.x * QSMC_BLOCK) {
        const int64_t i = list[t];
        double p[D];
#pragma unroll
        for (int a = 0; a < D; ++a) p[a] = x[a * ldx + i];
        auto reload = [&](double *pp) {                      // (a listed particle without a negative eigenvalue: rare)
#pragma unroll
            for (int a = 0; a < D; ++a) pp[a] = x[a * ldx + i];
        };
        if (tomo_canon_particle<DIM>(B, p, allow_subnormalized != 0, reload)) {
#pragma unroll
            for (int a = 0; a < D; ++a) x[a * ldx + i] = p[a];
        }
    }
}

// Round 5: the list pass of a 2-qubit canonicalize through psd_project4 (qsmc_device.h) -- the clamped reconstruction as a
// polynomial in rho over eigenvalues from eigenvector-free Jacobi sweeps: ~3400 instead of ~8000 instructions per listed
// particle, and every lane of a wave finishes within a sweep of its neighbours' (the reconstruction no longer depends on
// how many eigenvalues are negative).  A particle psd_project4 flags (three eigenvalues clustered across zero: a nearly
// pure state seen through noise; never on a Ginibre-like cloud) is appended to a second list, which k_tomo_canon_list --
// the eigenvector form, launched right behind this kernel on that list -- works off; it leaves at once on an empty one.
template <class Basis>
__global__ __launch_bounds__(QSMC_BLOCK) void k_tomo_canon_list_fast(Basis B, double *__restrict__ x, int64_t ldx,
                                                                     int allow_subnormalized,
                                                                     const unsigned int *__restrict__ list,
                                                                     unsigned int *__restrict__ count,      // [0]: length of list; [1]: of list2
                                                                     unsigned int *__restrict__ list2) {
    constexpr int D = 16;
    const unsigned int m = count[0];
    for (unsigned int t = blockIdx.x * QSMC_BLOCK + threadIdx.x; t < m; t += gridDim.x * QSMC_BLOCK) {
        const int64_t i = list[t];
        double p[D];
#pragma unroll
        for (int a = 0; a < D; ++a) p[a] = x[a * ldx + i];
        auto reload = [&](double *pp) {                      // (a listed particle without a negative eigenvalue: rare)
#pragma unroll
            for (int a = 0; a < D; ++a) pp[a] = x[a * ldx + i];
        };
        const int verdict = tomo_canon_particle4_fast(B, p, allow_subnormalized != 0, reload);
        if (verdict == 1) {
#pragma unroll
            for (int a = 0; a < D; ++a) x[a * ldx + i] = p[a];
        } else if (verdict == 2) {
            list2[atomicAdd(&count[1], 1u)] = (unsigned int)i;
        }
    }
}
