import json,sys
for ln in sys.stdin:
    if ln.startswith("{"):
        d=json.loads(ln)["other_paths"]
        for k,v in d.items():
            kk = v.get("window_kernel") or v.get("kernel") or {}
            print(k, "value", v.get("value") or v.get("hypothetical_likelihoods_per_s"), "ms", v.get("ms_per_datum") or v.get("ms_per_experiment"), "kernel_us", kk.get("avg_kernel_us"), "frac", kk.get("frac"))
