// Round 6 (last session): price a PRE-QUEUED datum.  update()'s contract puts one host round trip between data; of its
// ~11 us at N = 1e7, ~5 are "launch to start" (hipLaunchKernel -> doorbell -> command processor -> dispatch) and 2-3 the
// launch calls themselves.  Here the next datum's kernels are queued WHILE the current one runs, behind a one-workgroup
// gate kernel that spins (bounded) on a word in pinned host memory; when the datum arrives the host writes its
// parameter + the word, the gate copies the parameter to device memory and exits, and the work kernel -- already in the
// queue, its parameter read through a pointer -- starts at a kernel boundary instead of after a launch.
//   baseline : [think 2 us] launch work(kernarg) ; launch flag ; spin on the completion word
//   gated    : [think 2 us] write param + go ; queue gate', work', flag' for the NEXT datum ; spin on the completion word
// Measurement tooling; nothing here is part of the library.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
typedef unsigned long long u64;
static double now() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static void think(double us) { double t = now(); while (now() - t < us) {} }

__global__ void k_work_arg(double *x, long n, double a) {
    long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
    for (; i < n; i += (long)gridDim.x * blockDim.x) x[i] = x[i] * a + 1e-9;
}
__global__ void k_work_ptr(double *x, long n, const double *params, const u64 *cancelled) {
    if (*cancelled) return;
    const double a = params[0];
    long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
    for (; i < n; i += (long)gridDim.x * blockDim.x) x[i] = x[i] * a + 1e-9;
}
__global__ void k_flag(volatile u64 *flag, u64 v) {
    if (threadIdx.x == 0) { __threadfence_system(); *flag = v; }
}
// host block: [0] go word (datum number), [1] parameter (as bits)
__global__ void k_gate(const u64 *host_blk, u64 want, double *dparams, u64 *cancelled, long long max_ticks) {
    if (threadIdx.x != 0) return;
    const long long t0 = wall_clock64();
    u64 go;
    for (;;) {
        go = __hip_atomic_load(host_blk, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM);
        if (go >= want) break;
        if (wall_clock64() - t0 > max_ticks) break;
        __builtin_amdgcn_s_sleep(1);
    }
    if (go >= want) {
        const u64 bits = __hip_atomic_load(host_blk + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        dparams[0] = __longlong_as_double((long long)bits);
        *cancelled = 0;
    } else {
        *cancelled = 1;                       // timed out: the queued work leaves at once, the host launches the plain form
    }
}

int main(int argc, char **argv) {
    hipStream_t s; hipStreamCreate(&s);
    u64 *flag; hipHostMalloc(&flag, 64, hipHostMallocMapped); *flag = 0;
    u64 *dflag; hipHostGetDevicePointer((void **)&dflag, flag, 0);
    u64 *blk; hipHostMalloc(&blk, 64, hipHostMallocMapped); blk[0] = 0; blk[1] = 0;
    u64 *dblk; hipHostGetDevicePointer((void **)&dblk, blk, 0);
    double *dparams; hipMalloc(&dparams, 64);
    u64 *dcancel; hipMalloc(&dcancel, 8); hipMemset(dcancel, 0, 8);
    const long sizes[3] = {10000000, 1250000, 100000};
    double *x; hipMalloc(&x, sizes[0] * 8); hipMemset(x, 0, sizes[0] * 8);
    const long long max_ticks = 100ll * 1000 * 100;      // wall_clock64 ticks at 100 MHz: 100 ms bound
    u64 seq = 0, datum = 0;
    for (int si = 0; si < 3; ++si) {
        const long n = sizes[si];
        const int grid = (int)((n + 4095) / 4096), R = 1500;
        for (int rep = 0; rep < 2; ++rep) {
            // ---- baseline
            for (int i = 0; i < 50; ++i) { hipLaunchKernelGGL(k_work_arg, grid, 256, 0, s, x, n, 1.0000001); } hipStreamSynchronize(s);
            double t = now();
            for (int i = 0; i < R; ++i) {
                think(2.0);
                hipLaunchKernelGGL(k_work_arg, grid, 256, 0, s, x, n, 1.0000001);
                hipLaunchKernelGGL(k_flag, 1, 64, 0, s, dflag, ++seq);
                while (*(volatile u64 *)flag != seq) {}
            }
            const double base = (now() - t) / R;
            // ---- kernel alone, back to back
            t = now();
            for (int i = 0; i < 200; ++i) hipLaunchKernelGGL(k_work_arg, grid, 256, 0, s, x, n, 1.0000001);
            hipStreamSynchronize(s);
            const double alone = (now() - t) / 200;
            // ---- gated: prime the first datum's chain
            hipLaunchKernelGGL(k_gate, 1, 64, 0, s, dblk, datum + 1, dparams, dcancel, max_ticks);
            hipLaunchKernelGGL(k_work_ptr, grid, 256, 0, s, x, n, dparams, dcancel);
            hipLaunchKernelGGL(k_flag, 1, 64, 0, s, dflag, seq + 1);
            think(50.0);
            t = now();
            for (int i = 0; i < R; ++i) {
                think(2.0);
                const double a = 1.0000001;
                blk[1] = *(const u64 *)&a;
                __atomic_store_n(&blk[0], ++datum, __ATOMIC_RELEASE);          // the datum arrives
                ++seq;
                // the NEXT datum's chain goes into the queue while this one runs
                hipLaunchKernelGGL(k_gate, 1, 64, 0, s, dblk, datum + 1, dparams, dcancel, max_ticks);
                hipLaunchKernelGGL(k_work_ptr, grid, 256, 0, s, x, n, dparams, dcancel);
                hipLaunchKernelGGL(k_flag, 1, 64, 0, s, dflag, seq + 1);
                while (*(volatile u64 *)flag != seq) {}
            }
            const double gated = (now() - t) / R;
            // drain the primed chain (let it through as one more datum)
            __atomic_store_n(&blk[0], ++datum, __ATOMIC_RELEASE); ++seq;
            while (*(volatile u64 *)flag != seq) {}
            hipStreamSynchronize(s);
            printf("n = %8ld  grid %5d : kernel alone %6.2f us | baseline datum %6.2f us | pre-queued behind a gate %6.2f us | gain %5.2f us\n",
                   n, grid, alone, base, gated, base - gated);
            fflush(stdout);
        }
    }
    // a gate nobody opens: how long a cancelled chain takes to leave (timeout 20 us)
    {
        const long n = sizes[0]; const int grid = (int)((n + 4095) / 4096);
        double t = now();
        hipLaunchKernelGGL(k_gate, 1, 64, 0, s, dblk, datum + 1000, dparams, dcancel, 2000ll);
        hipLaunchKernelGGL(k_work_ptr, grid, 256, 0, s, x, n, dparams, dcancel);
        hipLaunchKernelGGL(k_flag, 1, 64, 0, s, dflag, ++seq);
        while (*(volatile u64 *)flag != seq) {}
        printf("a gate left closed (20 us bound) + its cancelled chain: %.2f us\n", now() - t);
    }
    return 0;
}
