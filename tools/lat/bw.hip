// HBM ceiling for the update kernel's access pattern (read x, read w, write w'; 24 B/particle), N = 1e7
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(256) void k_rrw(const double2 *__restrict__ x, const double2 *__restrict__ w, double2 *__restrict__ o, long n2) {
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n2; i += (long)gridDim.x * 256) {
        double2 a = x[i], b = w[i];
        o[i] = make_double2(a.x * b.x, a.y * b.y);
    }
}
__global__ __launch_bounds__(256) void k_rw(const double2 *__restrict__ x, double2 *__restrict__ o, long n2) {
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n2; i += (long)gridDim.x * 256) {
        double2 a = x[i];
        o[i] = make_double2(a.x * 1.5, a.y * 1.5);
    }
}
__global__ __launch_bounds__(256) void k_r(const double2 *__restrict__ x, double *__restrict__ o, long n2) {
    double s = 0;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n2; i += (long)gridDim.x * 256) { double2 a = x[i]; s += a.x + a.y; }
    if (s == 1.2345) o[0] = s;
}
int main() {
    const long n = 10000000, n2 = n / 2;
    double2 *x, *w, *o; hipMalloc(&x, n * 8); hipMalloc(&w, n * 8); hipMalloc(&o, n * 8);
    hipMemset(x, 0, n * 8); hipMemset(w, 0, n * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int grid : {1024, 2048, 4096, 8192, 19532}) {
        float ms;
        for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(k_rrw, grid, 256, 0, 0, x, w, o, n2);
        hipEventRecord(e0); for (int rep = 0; rep < 20; ++rep) hipLaunchKernelGGL(k_rrw, grid, 256, 0, 0, x, w, o, n2); hipEventRecord(e1);
        hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
        double t_rrw = ms / 20 * 1e-3;
        hipEventRecord(e0); for (int rep = 0; rep < 20; ++rep) hipLaunchKernelGGL(k_rw, grid, 256, 0, 0, x, o, n2); hipEventRecord(e1);
        hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
        double t_rw = ms / 20 * 1e-3;
        hipEventRecord(e0); for (int rep = 0; rep < 20; ++rep) hipLaunchKernelGGL(k_r, grid, 256, 0, 0, x, (double *)o, n2); hipEventRecord(e1);
        hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
        double t_r = ms / 20 * 1e-3;
        printf("grid %5d: read+read+write %.1f us = %.0f GB/s | read+write %.1f us = %.0f GB/s | read %.1f us = %.0f GB/s (back-to-back launches, incl. ~2 us gaps)\n",
               grid, t_rrw * 1e6, 24.0 * n / t_rrw / 1e9, t_rw * 1e6, 16.0 * n / t_rw / 1e9, t_r * 1e6, 8.0 * n / t_r / 1e9);
    }
    return 0;
}
