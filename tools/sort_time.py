"""Round 6: qsmc_argsort (kernels/sort.hpp, hand-written LSD radix sort) timed beside torch.sort on the same keys."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'python-qinfer_amd'))
import torch
from qinfer_amd.engine import get_engine
eng = get_engine()
for n in (1_000_000, 10_000_000, 100_000_000):
    k = torch.rand(n, dtype=torch.float64, device="cuda")
    for _ in range(2):
        eng.argsort(k)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5):
        s, i = eng.argsort(k)
    torch.cuda.synchronize(); t_q = (time.perf_counter() - t0) / 5
    for _ in range(2):
        torch.sort(k, stable=True)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5):
        s2, i2 = torch.sort(k, stable=True)
    torch.cuda.synchronize(); t_t = (time.perf_counter() - t0) / 5
    assert torch.equal(s, s2) and torch.equal(i, i2)
    print("n = %.0e  qsmc_argsort %.2f ms (%.1f GB/s of 16 n bytes in + out)   torch.sort(stable) %.2f ms" % (n, t_q * 1e3, 32 * n / t_q / 1e9, t_t * 1e3))
    del k, s, i, s2, i2
    torch.cuda.empty_cache()
