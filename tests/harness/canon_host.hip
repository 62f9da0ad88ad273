// Host harness for the per-particle canonicalize arithmetic of qsmc_device.h (the functions are __host__ __device__):
// reads n Hermitian 4 x 4 matrices (lower triangles as 16 doubles: Ar[r][c], Ai[r][c] for r >= c, row by row) from stdin
// as raw float64, writes for each the verdict of psd_project4 and its R (32 doubles, full matrix re / im), then the R of
// jacobi_clamp (the eigenvector form) -- tests/test_host_logic.py compares both with numpy.linalg.eigh.
#include <cstdio>
#include <vector>
#include "../../python-qinfer_amd/csrc/qsmc_device.h"
using namespace qsmc;
int main() {
    std::vector<double> in;
    double buf[4096];
    size_t got;
    while ((got = fread(buf, sizeof(double), 4096, stdin)) > 0) in.insert(in.end(), buf, buf + got);
    const size_t n = in.size() / 16;
    for (size_t i = 0; i < n; ++i) {
        double Ar[4][4] = {}, Ai[4][4] = {}, Rr[4][4] = {}, Ri[4][4] = {};
        const double *p = &in[16 * i];
        int k = 0;
        for (int r = 0; r < 4; ++r)
            for (int c = 0; c <= r; ++c) { Ar[r][c] = p[k++]; }
        for (int r = 0; r < 4; ++r)
            for (int c = 0; c < r; ++c) { Ai[r][c] = p[k++]; }
        const int v = psd_project4(Ar, Ai, Rr, Ri);          // (lower triangle; mirrored for the comparison)
        for (int r = 0; r < 4; ++r)
            for (int c = r + 1; c < 4; ++c) { Rr[r][c] = Rr[c][r]; Ri[r][c] = -Ri[c][r]; }
        double out[1 + 32 + 1 + 32];
        out[0] = (double)v;
        for (int r = 0; r < 4; ++r) for (int c = 0; c < 4; ++c) { out[1 + r * 4 + c] = Rr[r][c]; out[17 + r * 4 + c] = Ri[r][c]; }
        double Jr[4][4] = {}, Ji[4][4] = {};
        const bool neg = jacobi_clamp<4>(Ar, Ai, Jr, Ji);          // (destroys Ar / Ai: after psd_project4, which copies)
        out[33] = neg ? 1.0 : 0.0;
        for (int r = 0; r < 4; ++r) for (int c = 0; c < 4; ++c) { out[34 + r * 4 + c] = Jr[r][c]; out[50 + r * 4 + c] = Ji[r][c]; }
        fwrite(out, sizeof(double), 66, stdout);
    }
    return 0;
}
