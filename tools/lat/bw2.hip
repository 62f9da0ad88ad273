// Which feature of the update kernel costs time beyond a bare read-read-write stream? (N = 1e7, 240 MB)
#include <hip/hip_runtime.h>
#include <cstdio>
template <int MODE>   // 0 bare; 1 + five running sums, no epilogue; 2 + wave/block reduction epilogue; 3 tiled 4x unroll like k_update_fused
__global__ __launch_bounds__(256) void k(const double *__restrict__ x, const double *__restrict__ w, double *__restrict__ o, long n, double *__restrict__ partials) {
    double s0 = 0, s1 = 0, s2 = 0, s3 = 0, s4 = 0, mn = 1e300;
    if (MODE < 3) {
        for (long i = (blockIdx.x * 256L + threadIdx.x) * 2; i + 1 < n; i += (long)gridDim.x * 512) {
            double2 a = *(const double2 *)(x + i), b = *(const double2 *)(w + i);
            double2 r = make_double2(a.x * b.x, a.y * b.y);
            *(double2 *)(o + i) = r;
            if (MODE >= 1) { s0 += r.x + r.y; s1 += r.x * r.x + r.y * r.y; s2 += (r.x >= 0) + (r.y >= 0); s3 += r.x * a.x + r.y * a.y; s4 += r.x * a.x * a.x + r.y * a.y * a.y; mn = fmin(mn, fmin(r.x, r.y)); }
        }
    } else {
        const long TILE = 2048;
        for (long base = blockIdx.x * TILE; base + TILE <= n; base += (long)gridDim.x * TILE) {
            double2 a[4], b[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) { long i = base + (u * 256L + threadIdx.x) * 2; a[u] = *(const double2 *)(x + i); b[u] = *(const double2 *)(w + i); }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                long i = base + (u * 256L + threadIdx.x) * 2;
                double2 r = make_double2(a[u].x * b[u].x, a[u].y * b[u].y);
                *(double2 *)(o + i) = r;
                s0 += r.x + r.y; s1 += r.x * r.x + r.y * r.y; s2 += (r.x >= 0) + (r.y >= 0); s3 += r.x * a[u].x + r.y * a[u].y; s4 += r.x * a[u].x * a[u].x + r.y * a[u].y * a[u].y; mn = fmin(mn, fmin(r.x, r.y));
            }
        }
    }
    if (MODE == 1) { if (s0 + s1 + s2 + s3 + s4 + mn == 1.2345e-300) partials[0] = s0; }
    if (MODE >= 2) {
        __shared__ double lds[4 * 6];
        double v[6] = {s0, s1, s2, s3, s4, mn};
        for (int k = 0; k < 6; ++k) for (int off = 32; off > 0; off >>= 1) { double t = __shfl_down(v[k], off, 64); v[k] = k < 5 ? v[k] + t : fmin(v[k], t); }
        if ((threadIdx.x & 63) == 0) for (int k = 0; k < 6; ++k) lds[(threadIdx.x >> 6) * 6 + k] = v[k];
        __syncthreads();
        if (threadIdx.x < 6) { double t = lds[threadIdx.x]; for (int wv = 1; wv < 4; ++wv) t += lds[wv * 6 + threadIdx.x]; partials[blockIdx.x * 6 + threadIdx.x] = t; }
    }
}
template <int MODE> void run(const char *name, int grid, const double *x, const double *w1, double *w2, long n, double *partials) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1); float ms;
    for (int rep = 0; rep < 4; ++rep) hipLaunchKernelGGL(k<MODE>, grid, 256, 0, 0, x, rep & 1 ? w2 : w1, (double *)(rep & 1 ? w1 : w2), n, partials);
    float tot = 0;
    for (int rep = 0; rep < 20; ++rep) {   // ping-pong the weight buffers like the updater; time each launch on its own
        hipEventRecord(e0); hipLaunchKernelGGL(k<MODE>, grid, 256, 0, 0, x, rep & 1 ? w2 : w1, (double *)(rep & 1 ? w1 : w2), n, partials); hipEventRecord(e1);
        hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1); tot += ms;
    }
    printf("%-44s grid %5d: %.1f us\n", name, grid, tot / 20 * 1e3);
}
int main() {
    const long n = 10000000; double *x, *w1, *w2, *p;
    hipMalloc(&x, n * 8); hipMalloc(&w1, n * 8); hipMalloc(&w2, n * 8); hipMalloc(&p, 65536 * 8);
    hipMemset(x, 0, n * 8); hipMemset(w1, 0, n * 8); hipMemset(w2, 0, n * 8);
    for (int grid : {2048, 4883}) {
        run<0>("bare read-read-write", grid, x, w1, w2, n, p);
        run<1>("+ five running sums", grid, x, w1, w2, n, p);
        run<2>("+ wave/block reduction epilogue", grid, x, w1, w2, n, p);
        run<3>("tiled, 4 x double2 loads hoisted + sums + epi", grid, x, w1, w2, n, p);
    }
    return 0;
}
