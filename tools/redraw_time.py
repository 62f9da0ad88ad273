"""Resample time when the redraw path is triggered: a precession cloud hugging omega = 0, so that a chosen
fraction of the kicked children is invalid (omega < 0) and must redraw a global ancestor.
Run on the GPU box: python tools/redraw_time.py      (QSMC_LIB_PATH=<other build> for an A/B)."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "python-qinfer_amd"))
import numpy as np
import torch
import qinfer_amd as qi
from qinfer_amd.engine import get_engine

eng = get_engine()
n = 10_000_000
desc = qi.SimplePrecessionModel()._native_desc()
rs = np.random.RandomState(0)
w = eng.to_device(rs.random_sample(n))
norm = float(w.sum().item())
for centre in (0.5, 0.03, 0.01, 0.0):          # sd of the kick = 0.01
    x = eng.to_device(np.abs(centre + 0.002 * rs.randn(1, n)))
    mean, S = np.array([centre]), np.array([[0.01]])
    ts = []
    for rep in range(12):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        a.record()
        out, failed = eng.lw_resample_philox(desc, True, x, w, norm, 0.98, mean, S, n, 7, rep + 1, 100, sync=False)
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) * 1e3)
    frac = float((out < 0).double().mean().item())
    print("centre %.2f: resample median %.1f us  (invalid left %.1e)" % (centre, np.median(ts[2:]), frac))
