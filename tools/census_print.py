import json,sys
for ln in sys.stdin:
    if ln.startswith("{"):
        d=json.loads(ln)
        print("ms/step", d["ms_per_step"], "resamples", d["config"]["resamples_in_timed_region"])
        for k,v in d.get("kernel_census",{}).get("kernels",{}).items():
            print("   ", k, v)
