// Pricing of round 6's "small-shard datum from the host side" (measurement tooling, nothing here is shipped).
//
// A datum on a 1.25e6-particle shard is an update kernel of ~611 workgroups (2048 particles each) that leaves one
// partial row per workgroup, a one-workgroup reducing launch that sums the rows in index order and publishes 6 doubles +
// a completion word to pinned host memory, and a host spin on that word.  The question: is it cheaper to let every
// workgroup store its row (+ a sequence word) straight into pinned host memory and have the HOST add the rows in index
// order -- no reducing launch, no device-side arrival ticket?
//
//   a) two launches (what ships): k_upd<0> -> device partials; k_red<<<1, 256>>> -> pinned totals + flag; host spins on flag
//   b) rows to host, fenced:      k_upd<1>: lanes 0..5 store the row to pinned, thread 0 fences at system scope and stores
//                                 the sequence word of its row; the host spins on every row's word and adds in index order
//   c) rows to host, one line:    k_upd<2>: lanes 0..7 of wave 0 store {6 doubles, pad, seq} as ONE 64-byte wave store, no
//                                 fence; the host validates the sequence word inside the line (in-line validation relies on
//                                 a 64-byte store arriving as one PCIe write -- not an architectural guarantee: priced only)
//
// Usage: hostrows [iterations]   (prints us per datum for grids of 153, 306, 611 and 1024 workgroups)
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <immintrin.h>

constexpr int BLOCK = 256, PER_BLOCK = 2048, NS = 6, ROW = 8;      // a row = one 64-byte line: 6 sums, pad, seq

static double now() {
    return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

__device__ inline double wave_sum(double v) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}

// one pass of the update kernel's shape: read x and w, write w', six block sums
template <int MODE>
__global__ __launch_bounds__(BLOCK) void k_upd(const double *__restrict__ x, const double *__restrict__ w,
                                               double *__restrict__ w_out, long n, double t, double *__restrict__ partials,
                                               double *__restrict__ rows_host, unsigned long long seq) {
    __shared__ double red[4][NS];
    double acc[NS] = {0, 0, 0, 0, 0, 0};
    const long base = (long)blockIdx.x * PER_BLOCK;
    for (int u = 0; u < PER_BLOCK / BLOCK; ++u) {
        const long i = base + u * BLOCK + threadIdx.x;
        if (i < n) {
            const double xi = x[i], c = cos(0.5 * t * xi), L = c * c, wn = w[i] * L;
            w_out[i] = wn;
            acc[0] += wn; acc[1] += wn * wn; acc[2] += (wn < 0.0) ? 1.0 : 0.0;
            acc[3] += wn * xi; acc[4] += wn * xi * xi; acc[5] = fmin(acc[5], wn);
        }
    }
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (int k = 0; k < NS; ++k) {
        const double s = wave_sum(acc[k]);
        if (lane == 0) red[wv][k] = s;
    }
    __syncthreads();
    if (wv == 0 && lane < ROW) {
        double v = 0.0;
        if (lane < NS) v = red[0][lane] + red[1][lane] + red[2][lane] + red[3][lane];
        if (MODE == 0) {
            if (lane < NS) partials[(long)blockIdx.x * NS + lane] = v;
        } else if (MODE == 1) {
            if (lane < NS) rows_host[(long)blockIdx.x * ROW + lane] = v;
            __threadfence_system();
            if (lane == 0)
                *reinterpret_cast<volatile unsigned long long *>(&rows_host[(long)blockIdx.x * ROW + 7]) = seq;
        } else {
            if (lane == 7) v = __longlong_as_double((long long)seq);
            rows_host[(long)blockIdx.x * ROW + lane] = v;                     // one 64-byte store of the wave's low 8 lanes
        }
    }
}

__global__ __launch_bounds__(BLOCK) void k_red(const double *__restrict__ partials, int grid, double *__restrict__ out_host,
                                               unsigned long long *flag, unsigned long long seq) {
    __shared__ double red[4][NS];
    double acc[NS] = {0, 0, 0, 0, 0, 0};
    for (int g = threadIdx.x; g < grid; g += BLOCK)
        for (int k = 0; k < NS; ++k) acc[k] += partials[(long)g * NS + k];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (int k = 0; k < NS; ++k) {
        const double s = wave_sum(acc[k]);
        if (lane == 0) red[wv][k] = s;
    }
    __syncthreads();
    if (threadIdx.x < NS) out_host[threadIdx.x] = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence_system();
        *reinterpret_cast<volatile unsigned long long *>(flag) = seq;
    }
}

int main(int argc, char **argv) {
    const int R = argc > 1 ? atoi(argv[1]) : 3000;
    hipStream_t s;
    hipStreamCreate(&s);
    const int grids[4] = {153, 306, 611, 1024};
    const long nmax = 1024L * PER_BLOCK;
    double *x, *w, *w2, *partials;
    hipMalloc(&x, nmax * 8); hipMalloc(&w, nmax * 8); hipMalloc(&w2, nmax * 8); hipMalloc(&partials, 1024 * NS * 8);
    {
        double *hx = (double *)malloc(nmax * 8);
        for (long i = 0; i < nmax; ++i) hx[i] = (double)(i % 9973) / 9973.0;
        hipMemcpy(x, hx, nmax * 8, hipMemcpyHostToDevice);
        for (long i = 0; i < nmax; ++i) hx[i] = 1.0;
        hipMemcpy(w, hx, nmax * 8, hipMemcpyHostToDevice);
        free(hx);
    }
    double *rows, *rows_dev, *tot, *tot_dev;
    unsigned long long *flag, *flag_dev;
    hipHostMalloc(&rows, 1024 * ROW * 8, hipHostMallocMapped); memset(rows, 0, 1024 * ROW * 8);
    hipHostMalloc(&tot, 64, hipHostMallocMapped);
    hipHostMalloc(&flag, 64, hipHostMallocMapped); *flag = 0;
    hipHostGetDevicePointer((void **)&rows_dev, rows, 0);
    hipHostGetDevicePointer((void **)&tot_dev, tot, 0);
    hipHostGetDevicePointer((void **)&flag_dev, flag, 0);
    unsigned long long seq = 0;
    volatile double sink = 0;
    printf("%d iterations per figure; a = update + reducing launch + spin on one word (ships), b = rows to pinned host memory "
           "behind a system fence + host sum, c = rows as one unfenced 64-byte store + host sum\n", R);
    for (int gi = 0; gi < 4; ++gi) {
        const int G = grids[gi];
        const long n = (long)G * PER_BLOCK;
        double res[3] = {0, 0, 0}, host_scan[3] = {0, 0, 0};
        for (int mode = 0; mode < 3; ++mode) {
            for (int pass = 0; pass < 2; ++pass) {            // pass 0: warm-up
                const int iters = pass ? R : 200;
                double t_scan = 0;
                const double t0 = now();
                for (int it = 0; it < iters; ++it) {
                    ++seq;
                    const double t = 1.0 + 1e-3 * (double)(it & 255);
                    const double *wi = (it & 1) ? w2 : w;
                    double *wo = (it & 1) ? w : w2;
                    if (mode == 0) {
                        hipLaunchKernelGGL(k_upd<0>, dim3(G), dim3(BLOCK), 0, s, x, wi, wo, n, t, partials, rows_dev, seq);
                        hipLaunchKernelGGL(k_red, dim3(1), dim3(BLOCK), 0, s, partials, G, tot_dev, flag_dev, seq);
                        while (*(volatile unsigned long long *)flag != seq) _mm_pause();
                        sink = tot[0] + tot[5];
                    } else {
                        if (mode == 1) hipLaunchKernelGGL(k_upd<1>, dim3(G), dim3(BLOCK), 0, s, x, wi, wo, n, t, partials, rows_dev, seq);
                        else hipLaunchKernelGGL(k_upd<2>, dim3(G), dim3(BLOCK), 0, s, x, wi, wo, n, t, partials, rows_dev, seq);
                        // wait for the LAST row first (workgroups retire roughly in index order), then walk the rows in order
                        const volatile unsigned long long *rq = reinterpret_cast<const volatile unsigned long long *>(rows);
                        while (rq[(long)(G - 1) * ROW + 7] != seq) _mm_pause();
                        const double ts = now();
                        double a[NS] = {0, 0, 0, 0, 0, 0};
                        for (int g = 0; g < G; ++g) {
                            while (rq[(long)g * ROW + 7] != seq) _mm_pause();
                            const volatile double *r = rows + (long)g * ROW;
                            for (int k = 0; k < NS; ++k) a[k] += r[k];
                        }
                        t_scan += now() - ts;
                        sink = a[0] + a[5];
                    }
                }
                const double per = (now() - t0) / iters;
                if (pass) { res[mode] = per; host_scan[mode] = t_scan / iters; }
            }
            hipStreamSynchronize(s);
        }
        printf("grid %4d (N = %7ld):  a %.2f us   b %.2f us (host walk after the last row %.2f)   c %.2f us (host walk %.2f)   "
               "a - b = %+.2f   a - c = %+.2f\n", G, n, res[0], res[1], host_scan[1], res[2], host_scan[2], res[0] - res[1],
               res[0] - res[2]);
    }
    (void)sink;
    return 0;
}
