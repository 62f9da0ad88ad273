"""Per-kernel HBM bytes per launch from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs).
Counters are in KB; on gfx950 FETCH_SIZE reports half of a coalesced streaming read (MI355X_MICROARCH.md,
HBM section), hence bytes = (2 * FETCH + WRITE) * 1024."""
import collections
import csv
import json
import re
import sys


def per_kernel(path, counter):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            acc[re.sub(r"\(.*", "", r["Kernel_Name"])].append(float(r["Counter_Value"]))
    return {k: (len(v), sum(v) / len(v)) for k, v in acc.items()}


fetch = per_kernel(sys.argv[1], "FETCH_SIZE")
write = per_kernel(sys.argv[2], "WRITE_SIZE")
allk = {}
for k in sorted(fetch):
    n, f = fetch[k]
    w = write.get(k, (0, 0.0))[1]
    allk[k] = {"calls": n, "FETCH_SIZE_KB_avg": f, "WRITE_SIZE_KB_avg": w, "hbm_bytes_corrected": (2 * f + w) * 1024}
upd = next(v for k, v in allk.items() if "k_update_fused<1, 2, false" in k)
ones = next((v for k, v in allk.items() if "k_update_fused<1, 2, true" in k), None)
print(json.dumps({
    "update_kernel_bytes_per_launch": upd["hbm_bytes_corrected"],
    "kernel": "k_update_fused<PRECESSION,2,false,false> (24 B/particle variant)",
    "n_particles": 10000000,
    "method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in SEPARATE passes (+ --kernel-trace only) over "
              "`bench.py --steps 40 --warmup 5 --no-cpu-baseline` (tools/refresh_profiles.sh); per-launch averages; "
              "bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024: counters are in KB and on gfx950 FETCH_SIZE reports exactly "
              "half of a coalesced streaming read (MI355X_MICROARCH.md, HBM section) -- confirmed here by k_chunk_sums "
              "whose read is known (80 MB)",
    "FETCH_SIZE_KB_avg": upd["FETCH_SIZE_KB_avg"], "WRITE_SIZE_KB_avg": upd["WRITE_SIZE_KB_avg"],
    "algorithmic_bytes_per_launch": 240000000,
    "ones_variant_bytes_per_launch": None if ones is None else ones["hbm_bytes_corrected"],
    "all_kernels": allk}, indent=1))
