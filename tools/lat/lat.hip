// Launch/sync latency microbenchmark (measurement tooling).
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <atomic>
__global__ void k_empty(int *p) { if (p && threadIdx.x == 9999) *p = 1; }
__global__ void k_flag(volatile unsigned long long *flag, unsigned long long v) {
    if (threadIdx.x == 0 && blockIdx.x == 0) { __threadfence_system(); *flag = v; }
}
__global__ void k_work(double *x, long n, volatile unsigned long long *flag, unsigned long long v) {
    long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
    for (; i < n; i += (long)gridDim.x * blockDim.x) x[i] = x[i] * 1.0000001 + 1e-9;
}
static double now() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
    hipStream_t s; hipStreamCreate(&s);
    int *d; hipMalloc(&d, 4);
    unsigned long long *flag; hipHostMalloc(&flag, 64, hipHostMallocMapped); *flag = 0;
    unsigned long long *dflag; hipHostGetDevicePointer((void **)&dflag, flag, 0);
    double *x; long n = 10000000; hipMalloc(&x, n * 8); hipMemset(x, 0, n * 8);
    const int R = 2000;
    for (int i = 0; i < 100; ++i) { hipLaunchKernelGGL(k_empty, 1, 64, 0, s, d); hipStreamSynchronize(s); }
    double t = now();
    for (int i = 0; i < R; ++i) { hipLaunchKernelGGL(k_empty, 1, 64, 0, s, d); hipStreamSynchronize(s); }
    printf("a) empty kernel + streamSync           %.2f us\n", (now() - t) / R);
    t = now();
    for (int i = 0; i < R; ++i) { hipLaunchKernelGGL(k_empty, 1, 64, 0, s, d); hipLaunchKernelGGL(k_empty, 1, 64, 0, s, d); hipStreamSynchronize(s); }
    printf("b) two dependent kernels + streamSync  %.2f us\n", (now() - t) / R);
    t = now();
    for (int i = 0; i < R; ++i) {
        hipLaunchKernelGGL(k_flag, 1, 64, 0, s, dflag, (unsigned long long)(i + 1));
        while (*(volatile unsigned long long *)flag != (unsigned long long)(i + 1)) {}
    }
    printf("c) flag kernel + host spin on mapped   %.2f us\n", (now() - t) / R);
    hipStreamSynchronize(s);
    t = now();
    for (int i = 0; i < R; ++i) {
        hipLaunchKernelGGL(k_empty, 1, 64, 0, s, d);
        hipLaunchKernelGGL(k_flag, 1, 64, 0, s, dflag, (unsigned long long)(R + i + 1));
        while (*(volatile unsigned long long *)flag != (unsigned long long)(R + i + 1)) {}
    }
    printf("d) empty + flag kernel + host spin     %.2f us\n", (now() - t) / R);
    hipStreamSynchronize(s);
    hipEvent_t ev; hipEventCreateWithFlags(&ev, hipEventDisableTiming);
    t = now();
    for (int i = 0; i < R; ++i) { hipLaunchKernelGGL(k_empty, 1, 64, 0, s, d); hipEventRecord(ev, s); while (hipEventQuery(ev) != hipSuccess) {} }
    printf("e) empty kernel + eventRecord + query spin %.2f us\n", (now() - t) / R);
    // with real work: 80 MB rmw kernel (~30 us) then flag
    for (int i = 0; i < 20; ++i) { hipLaunchKernelGGL(k_work, 2048, 256, 0, s, x, n, dflag, 0ull); } hipStreamSynchronize(s);
    t = now();
    for (int i = 0; i < 500; ++i) { hipLaunchKernelGGL(k_work, 2048, 256, 0, s, x, n, dflag, 0ull); hipLaunchKernelGGL(k_empty, 1, 64, 0, s, d); hipStreamSynchronize(s); }
    double t_sync = (now() - t) / 500;
    t = now();
    for (int i = 0; i < 500; ++i) { hipLaunchKernelGGL(k_work, 2048, 256, 0, s, x, n, dflag, 0ull); }
    hipStreamSynchronize(s);
    double t_back = (now() - t) / 500;
    t = now();
    for (int i = 0; i < 500; ++i) {
        hipLaunchKernelGGL(k_work, 2048, 256, 0, s, x, n, dflag, 0ull);
        hipLaunchKernelGGL(k_flag, 1, 64, 0, s, dflag, (unsigned long long)(3 * R + i + 1));
        while (*(volatile unsigned long long *)flag != (unsigned long long)(3 * R + i + 1)) {}
    }
    double t_spin = (now() - t) / 500;
    printf("f) work kernel back-to-back %.2f us; work+empty+streamSync %.2f us; work+flag+spin %.2f us\n", t_back, t_sync, t_spin);
    return 0;
}
