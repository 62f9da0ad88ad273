// One datum as three stream launches against the same three kernels as an instantiated hipGraph whose kernel-node
// parameters are rewritten before every launch (measurement tooling).  The shape of a datum: a work kernel
// over n doubles whose scalar argument changes every time (the experiment), a one-workgroup kernel that publishes a
// completion word in pinned memory (the reducing launch), a third that leaves at once (the count kernel at its gate);
// the host spins on the word and only then issues the next datum (update()'s contract).
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
__global__ void k_work(double *x, long n, double a) {
    long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
    for (; i < n; i += (long)gridDim.x * blockDim.x) x[i] = x[i] * a + 1e-9;
}
__global__ void k_flag(volatile unsigned long long *flag, unsigned long long v) {
    if (threadIdx.x == 0 && blockIdx.x == 0) { __threadfence_system(); *flag = v; }
}
__global__ void k_gate(const int *gate, int *out) { if (*gate && threadIdx.x == 0) *out = 1; }
static double now() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main() {
    hipStream_t s; CK(hipStreamCreate(&s));
    unsigned long long *flag; CK(hipHostMalloc(&flag, 64, hipHostMallocMapped)); *flag = 0;
    unsigned long long *dflag; CK(hipHostGetDevicePointer((void **)&dflag, flag, 0));
    int *gate; CK(hipMalloc(&gate, 8)); CK(hipMemset(gate, 0, 8));
    const long sizes[3] = {1250000, 10000000, 100000};
    for (int si = 0; si < 3; ++si) {
        long n = sizes[si];
        double *x; CK(hipMalloc(&x, n * 8)); CK(hipMemset(x, 0, n * 8));
        const int R = 2000;
        unsigned long long tick = *flag;
        const int grid = (int)((n + 2047) / 2048 < 4096 ? (n + 2047) / 2048 : 4096);
        // (a) three launches per datum
        auto datum_launches = [&](double a) {
            ++tick;
            hipLaunchKernelGGL(k_work, grid, 256, 0, s, x, n, a);
            hipLaunchKernelGGL(k_flag, 1, 64, 0, s, dflag, tick);
            hipLaunchKernelGGL(k_gate, 16, 64, 0, s, gate, gate + 1);
            while (*(volatile unsigned long long *)flag != tick) {}
        };
        for (int i = 0; i < 200; ++i) datum_launches(1.0 + 1e-9 * i);
        double t = now();
        for (int i = 0; i < R; ++i) datum_launches(1.0 + 1e-9 * i);
        const double t_l = (now() - t) / R;
        CK(hipStreamSynchronize(s));
        // (b) the same as a graph: capture once, set the parameters of the two nodes that change, launch
        hipGraph_t g; hipGraphExec_t ge;
        double a0 = 1.0; unsigned long long v0 = 0;
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
        hipLaunchKernelGGL(k_work, grid, 256, 0, s, x, n, a0);
        hipLaunchKernelGGL(k_flag, 1, 64, 0, s, dflag, v0);
        hipLaunchKernelGGL(k_gate, 16, 64, 0, s, gate, gate + 1);
        CK(hipStreamEndCapture(s, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        size_t nn = 0; CK(hipGraphGetNodes(g, nullptr, &nn));
        hipGraphNode_t nodes[8]; CK(hipGraphGetNodes(g, nodes, &nn));
        hipGraphNode_t n_work = nullptr, n_flag = nullptr;
        for (size_t i = 0; i < nn; ++i) {
            hipKernelNodeParams p; CK(hipGraphKernelNodeGetParams(nodes[i], &p));
            if (p.func == (void *)k_work) n_work = nodes[i];
            if (p.func == (void *)k_flag) n_flag = nodes[i];
        }
        if (!n_work || !n_flag) { printf("nodes not found\n"); return 1; }
        double a; unsigned long long v;
        void *wargs[3] = {&x, &n, &a};
        void *fargs[2] = {&dflag, &v};
        hipKernelNodeParams pw = {}, pf = {};
        pw.func = (void *)k_work; pw.gridDim = dim3(grid); pw.blockDim = dim3(256); pw.kernelParams = wargs;
        pf.func = (void *)k_flag; pf.gridDim = dim3(1); pf.blockDim = dim3(64); pf.kernelParams = fargs;
        auto datum_graph = [&](double aa) {
            ++tick; a = aa; v = tick;
            CK(hipGraphExecKernelNodeSetParams(ge, n_work, &pw));
            CK(hipGraphExecKernelNodeSetParams(ge, n_flag, &pf));
            CK(hipGraphLaunch(ge, s));
            while (*(volatile unsigned long long *)flag != tick) {}
        };
        for (int i = 0; i < 200; ++i) datum_graph(1.0 + 1e-9 * i);
        t = now();
        for (int i = 0; i < R; ++i) datum_graph(1.0 + 1e-9 * i);
        const double t_g = (now() - t) / R;
        CK(hipStreamSynchronize(s));
        // (c) the graph's host cost alone: set + launch, no wait (queue allowed to run ahead), then one sync
        t = now();
        for (int i = 0; i < R; ++i) {
            ++tick; a = 1.0; v = tick;
            CK(hipGraphExecKernelNodeSetParams(ge, n_work, &pw));
            CK(hipGraphExecKernelNodeSetParams(ge, n_flag, &pf));
            CK(hipGraphLaunch(ge, s));
        }
        const double t_gh = (now() - t) / R;
        CK(hipStreamSynchronize(s));
        t = now();
        for (int i = 0; i < R; ++i) {
            ++tick;
            hipLaunchKernelGGL(k_work, grid, 256, 0, s, x, n, 1.0);
            hipLaunchKernelGGL(k_flag, 1, 64, 0, s, dflag, tick);
            hipLaunchKernelGGL(k_gate, 16, 64, 0, s, gate, gate + 1);
        }
        const double t_lh = (now() - t) / R;
        CK(hipStreamSynchronize(s));
        printf("n = %8ld: datum with host wait: 3 launches %.2f us, graph (2 SetParams + launch) %.2f us | issue rate without "
               "waiting: launches %.2f us, graph %.2f us per datum\n", n, t_l, t_g, t_lh, t_gh);
        CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g)); CK(hipFree(x));
    }
    return 0;
}
