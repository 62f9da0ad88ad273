"""Per-kernel HBM bytes per launch from the rocprofv3 --pmc passes tools/profile_configs.sh leaves behind
(<dir>/<tag>_{c2,c3,c4,c5}_pmc_{fetch,write}.csv: FETCH_SIZE and WRITE_SIZE in SEPARATE runs, kernel-trace only).
Counters are in KB; on gfx950 FETCH_SIZE reports half of a coalesced streaming read (MI355X_MICROARCH.md, HBM
section), hence bytes = (2 * FETCH + WRITE) * 1024 -- checked here on kernels whose reads are known.

    python tools/make_traffic_json.py <dir> <tag> > profiles/hbm_traffic.json
"""
import collections
import csv
import json
import os
import re
import sys


def per_kernel(path, counter):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            acc[re.sub(r"\(.*", "", r["Kernel_Name"]).strip()].append(float(r["Counter_Value"]))
    return {k: (len(v), sum(v) / len(v)) for k, v in acc.items()}


def config_table(d, tag, cfg):
    f = os.path.join(d, "%s_%s_pmc_fetch.csv" % (tag, cfg))
    w = os.path.join(d, "%s_%s_pmc_write.csv" % (tag, cfg))
    if not (os.path.exists(f) and os.path.exists(w)):
        return None
    fetch, write = per_kernel(f, "FETCH_SIZE"), per_kernel(w, "WRITE_SIZE")
    out = {}
    for k in sorted(fetch):
        n, fv = fetch[k]
        wv = write.get(k, (0, 0.0))[1]
        out[k] = {"calls": n, "FETCH_SIZE_KB_avg": fv, "WRITE_SIZE_KB_avg": wv, "hbm_bytes_corrected": (2 * fv + wv) * 1024}
    return out


def pick(table, needle):
    for k, v in table.items():
        if needle in k:
            return v
    return None


d, tag = sys.argv[1], sys.argv[2]
tables = {c: config_table(d, tag, c) for c in ("c2", "c3", "c4", "c5")}
c2 = tables["c2"] or {}
upd = pick(c2, "k_update_fused<1, 2, false")
ones = pick(c2, "k_update_fused<1, 2, true")
samp = pick(c2, "k_bucket_sample<1")
res = {
    "update_kernel_bytes_per_launch": upd and upd["hbm_bytes_corrected"],
    "sample_kernel_bytes_per_launch": samp and samp["hbm_bytes_corrected"],
    "kernel": "k_update_fused<PRECESSION,2,false,false> (24 B/particle variant)",
    "n_particles": 10000000,
    "algorithmic_bytes_per_launch": 240000000,
    "ones_variant_bytes_per_launch": ones and ones["hbm_bytes_corrected"],
    "method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in SEPARATE passes (+ --kernel-trace only) over the bench "
              "commands of tools/profile_configs.sh (headline: `bench.py --steps 200 --warmup 20 --no-cpu-baseline "
              "--no-other-configs`; configs 3-5: `bench.py --only <config>`); per-launch averages; bytes = (2*FETCH_SIZE "
              "+ WRITE_SIZE)*1024: counters are in KB and on gfx950 FETCH_SIZE reports exactly half of a coalesced "
              "streaming read (MI355X_MICROARCH.md, HBM section)",
    "tag": tag,
    "configs": {c: t for c, t in tables.items() if t},
}
print(json.dumps(res, indent=1))
