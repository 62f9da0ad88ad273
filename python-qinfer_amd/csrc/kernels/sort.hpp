// kernels/sort.hpp -- stable LSD radix sort of (float64 key, int64 index) pairs for the posterior read-outs
// (est_credible_region, posterior_marginal: SURVEY 8(f)4; the reference sorts with np.argsort, distributions.py:558-614,
// smc.py:672-716).  Round 6: written here for gfx950 in place of rocPRIM's onesweep, whose per-architecture trampoline
// kernels were 340 of the library's 554 kernel instantiations and most of its build time for a pass that is off the hot path.
//
// Eight passes of eight bits over the order-preserving 64-bit image of the key (sign flip; descending = complemented).
// A pass is four launches on the caller's stream:
//   k_sort_hist      per 4096-element tile, the count of every digit (LDS atomics) -> hist[digit][tile]
//   k_sort_scan      exclusive scan of hist in 4096-entry chunks (in place) + each chunk's total
//   k_sort_scan_top  exclusive scan of the chunk totals (one workgroup)
//   k_sort_scatter   every tile again: wave w of the workgroup owns the w-th 1024 elements of the tile; a lane finds the
//                    lanes of its wave that hold the same digit with eight ballots (no LDS atomics, no sorting network),
//                    its rank among them is a popcount, and the wave's running offset per digit lives in LDS -- so equal
//                    digits leave the tile in input order (stable) and land at hist[digit][tile] + offsets of earlier waves.
// Keys travel as their 64-bit images between passes; pass 0 reads the doubles (index = position), pass 7 writes doubles.
#pragma once

constexpr int SORT_ITEMS = 16;                               // elements per lane and tile
constexpr int SORT_TILE = QSMC_BLOCK * SORT_ITEMS;           // 4096
constexpr int SORT_WAVE_TILE = QSMC_WAVE * SORT_ITEMS;       // 1024: one wave's contiguous share
constexpr int SORT_SCAN_CHUNK = 4096;

__device__ inline unsigned long long sort_image(double v, int desc) {
    unsigned long long b = (unsigned long long)__double_as_longlong(v);
    b = (b >> 63) ? ~b : (b | 0x8000000000000000ull);       // negative: all bits flipped; non-negative: sign bit set
    return desc ? ~b : b;
}

__device__ inline double sort_unimage(unsigned long long b, int desc) {
    if (desc) b = ~b;
    b = (b >> 63) ? (b & 0x7fffffffffffffffull) : ~b;
    return __longlong_as_double((long long)b);
}

template <bool FIRST>
__device__ inline unsigned long long sort_load(const void *kin, long long i, int desc) {
    if (FIRST) return sort_image(static_cast<const double *>(kin)[i], desc);
    return static_cast<const unsigned long long *>(kin)[i];
}

template <bool FIRST>
__global__ __launch_bounds__(QSMC_BLOCK) void k_sort_hist(const void *__restrict__ kin, long long n, int shift, int desc,
                                                          unsigned int *__restrict__ hist, int ntiles) {
    __shared__ unsigned int h[256];
    h[threadIdx.x] = 0u;
    __syncthreads();
    const long long base = (long long)blockIdx.x * SORT_TILE;
#pragma unroll 4
    for (int r = 0; r < SORT_ITEMS; ++r) {
        const long long i = base + (long long)r * QSMC_BLOCK + threadIdx.x;
        if (i < n) atomicAdd(&h[(unsigned)(sort_load<FIRST>(kin, i, desc) >> shift) & 255u], 1u);
    }
    __syncthreads();
    hist[(size_t)threadIdx.x * ntiles + blockIdx.x] = h[threadIdx.x];
}

// exclusive scan of v[0..m) in chunks of 4096 (one workgroup of 1024 threads per chunk, four entries a thread), in place;
// totals[chunk] = the chunk's sum
__global__ __launch_bounds__(1024) void k_sort_scan(unsigned int *__restrict__ v, long long m, unsigned int *__restrict__ totals) {
    __shared__ unsigned int wsum[16];
    const long long i0 = (long long)blockIdx.x * SORT_SCAN_CHUNK + (long long)threadIdx.x * 4;
    unsigned int a[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) a[k] = (i0 + k < m) ? v[i0 + k] : 0u;
    const unsigned int mine = a[0] + a[1] + a[2] + a[3];
    unsigned int inc = mine;
    const int lane = threadIdx.x & (QSMC_WAVE - 1), wv = threadIdx.x / QSMC_WAVE;
    for (int off = 1; off < QSMC_WAVE; off <<= 1) {
        const unsigned int t = __shfl_up(inc, off, QSMC_WAVE);
        if (lane >= off) inc += t;
    }
    if (lane == QSMC_WAVE - 1) wsum[wv] = inc;
    __syncthreads();
    unsigned int before = 0u, total = 0u;
    for (int w = 0; w < 16; ++w) {
        const unsigned int s = wsum[w];
        if (w < wv) before += s;
        total += s;
    }
    unsigned int run = before + inc - mine;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (i0 + k < m) v[i0 + k] = run;
        run += a[k];
    }
    if (threadIdx.x == 0) totals[blockIdx.x] = total;
}

// exclusive scan of the chunk totals, one workgroup (at most a few thousand entries)
__global__ __launch_bounds__(1024) void k_sort_scan_top(unsigned int *__restrict__ totals, int nchunks) {
    __shared__ unsigned int wsum[16];
    __shared__ unsigned int carry_s;
    if (threadIdx.x == 0) carry_s = 0u;
    __syncthreads();
    const int lane = threadIdx.x & (QSMC_WAVE - 1), wv = threadIdx.x / QSMC_WAVE;
    for (int base = 0; base < nchunks; base += 1024) {
        const int i = base + threadIdx.x;
        const unsigned int mine = i < nchunks ? totals[i] : 0u;
        unsigned int inc = mine;
        for (int off = 1; off < QSMC_WAVE; off <<= 1) {
            const unsigned int t = __shfl_up(inc, off, QSMC_WAVE);
            if (lane >= off) inc += t;
        }
        if (lane == QSMC_WAVE - 1) wsum[wv] = inc;
        __syncthreads();
        unsigned int before = 0u, total = 0u;
        for (int w = 0; w < 16; ++w) {
            const unsigned int s = wsum[w];
            if (w < wv) before += s;
            total += s;
        }
        const unsigned int carry = carry_s;
        if (i < nchunks) totals[i] = carry + before + inc - mine;
        __syncthreads();
        if (threadIdx.x == 0) carry_s = carry + total;
        __syncthreads();
    }
}

template <bool FIRST, bool LAST>
__global__ __launch_bounds__(QSMC_BLOCK) void k_sort_scatter(const void *__restrict__ kin, const long long *__restrict__ iin,
                                                             void *__restrict__ kout, long long *__restrict__ iout,
                                                             long long n, int shift, int desc,
                                                             const unsigned int *__restrict__ hist,
                                                             const unsigned int *__restrict__ chunk_off, int ntiles) {
    __shared__ unsigned int cnt[QSMC_WAVES_PER_BLOCK][256];          // per-wave digit counts, then running output offsets
    const int lane = threadIdx.x & (QSMC_WAVE - 1), wv = threadIdx.x / QSMC_WAVE;
    for (int w = 0; w < QSMC_WAVES_PER_BLOCK; ++w) cnt[w][threadIdx.x] = 0u;
    __syncthreads();
    const long long sub = (long long)blockIdx.x * SORT_TILE + (long long)wv * SORT_WAVE_TILE;
    const unsigned long long below = (1ull << lane) - 1ull;
    volatile unsigned int *mine = cnt[wv];
    unsigned long long kb[SORT_ITEMS];
    // phase A: this wave's share into registers; its digit counts (one lane per distinct digit and round adds the group's size)
#pragma unroll
    for (int r = 0; r < SORT_ITEMS; ++r) {
        const long long i = sub + (long long)r * QSMC_WAVE + lane;
        const bool valid = i < n;
        kb[r] = valid ? sort_load<FIRST>(kin, i, desc) : 0ull;
        const unsigned int dg = (unsigned)(kb[r] >> shift) & 255u;
        unsigned long long same = __ballot(valid);
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            const bool bit = (dg >> b) & 1u;
            const unsigned long long bal = __ballot(valid && bit);
            same &= bit ? bal : ~bal;
        }
        if (valid && (same & below) == 0ull) mine[dg] += (unsigned)__popcll(same);
    }
    __syncthreads();
    // phase B: digit t's first output slot for every wave of this tile (earlier tiles and smaller digits come from the scan)
    {
        const size_t e = (size_t)threadIdx.x * ntiles + blockIdx.x;
        unsigned int g = hist[e] + chunk_off[e / SORT_SCAN_CHUNK];
        for (int w = 0; w < QSMC_WAVES_PER_BLOCK; ++w) {
            const unsigned int c = cnt[w][threadIdx.x];
            cnt[w][threadIdx.x] = g;
            g += c;
        }
    }
    __syncthreads();
    // phase C: the same rounds again; rank inside the group of equal digits = lanes below, the wave's running offset in LDS
#pragma unroll
    for (int r = 0; r < SORT_ITEMS; ++r) {
        const long long i = sub + (long long)r * QSMC_WAVE + lane;
        const bool valid = i < n;
        const unsigned int dg = (unsigned)(kb[r] >> shift) & 255u;
        unsigned long long same = __ballot(valid);
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            const bool bit = (dg >> b) & 1u;
            const unsigned long long bal = __ballot(valid && bit);
            same &= bit ? bal : ~bal;
        }
        unsigned int pos = 0u;
        if (valid) pos = mine[dg] + (unsigned)__popcll(same & below);
        __builtin_amdgcn_wave_barrier();                                  // every lane has read before the group's leader adds
        if (valid && (same & below) == 0ull) mine[dg] += (unsigned)__popcll(same);
        __builtin_amdgcn_wave_barrier();
        if (valid) {
            const long long idx = FIRST ? i : iin[i];
            if (LAST) static_cast<double *>(kout)[pos] = sort_unimage(kb[r], desc);
            else static_cast<unsigned long long *>(kout)[pos] = kb[r];
            iout[pos] = idx;
        }
    }
}
