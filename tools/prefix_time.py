"""Time of the weight-only resample prefix (chunk sums -> offsets -> chunk counts -> plan) alone.
Run on the GPU box: python tools/prefix_time.py [N]   (QSMC_COUNT_BY_DRAWS=1 for the histogram variant)."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "python-qinfer_amd"))
import numpy as np
import torch
from qinfer_amd.engine import get_engine

eng = get_engine()
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10_000_000
rs = np.random.RandomState(0)
w = eng.to_device(rs.random_sample(n) ** 2)
norm = float(w.sum().item())
for skew in ("flat", "peaked"):
    if skew == "peaked":
        ww = rs.random_sample(n) ** 2 * np.exp(-0.5 * ((np.arange(n) / n - 0.3) / 0.02) ** 2)
        w = eng.to_device(ww)
        norm = float(w.sum().item())
    ts = []
    for rep in range(30):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        a.record()
        eng.lw_resample_prepare(w, n, norm, n, 11, rep + 1)
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) * 1e3)
    print(skew, "prefix: median %.1f us  min %.1f us" % (np.median(ts[5:]), min(ts[5:])))
