#!/bin/bash
# round 6: the item-aware slot mapping of k_bucket_kick16 -- suite, config 5's line, kernel stats and the read requests of the
# kick kernel (TCC_EA0_RDREQ, 64-byte requests) -- and the plugin surface's three ways (run on the GPU box through gpurun)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r6d; mkdir -p $O
python -m pytest tests -q -m gpu > $O/pytest_all.log 2>&1; tail -6 $O/pytest_all.log
python bench.py --only plugin_paths 2>/dev/null | tail -1 > $O/plugin_paths.json
for i in 1 2; do python bench.py --only config5_share_tomography --warmup 5 2>/dev/null | tail -1 > $O/c5_line_$i.json; done
bash tools/kstats.sh --only config5_share_tomography --warmup 5 > $O/c5_kernel_stats.txt 2>&1
cd /tmp
for c in TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum; do
  rm -rf /tmp/rq
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/rq -- python $GRAFT_REPO_ROOT/bench.py --only config5_share_tomography --warmup 5 > /dev/null 2>&1
  f=$(ls /tmp/rq/*/*counter_collection.csv 2>/dev/null | tail -1)
  [ -n "$f" ] && cp $f $GRAFT_REPO_ROOT/$O/c5_pmc_$(echo $c | tr A-Z a-z).csv
done
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv, json, collections
O = "gpurun_out/r6d"
for c in ("tcc_ea0_rdreq_sum", "tcc_ea0_rdreq_32b_sum"):
    try:
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open("%s/c5_pmc_%s.csv" % (O, c))):
            acc[r["Kernel_Name"][:40]].append(float(r["Counter_Value"]))
        for k, v in sorted(acc.items()):
            if "kick16" in k or "anc16" in k or "canon" in k:
                print(c, k, "n=%d mean=%.4g  (x 64 B = %.1f MB)" % (len(v), sum(v) / len(v), sum(v) / len(v) * 64 / 1e6))
    except Exception as e:
        print(c, "error", e)
for i in (1, 2):
    d = json.load(open("%s/c5_line_%d.json" % (O, i)))["config5_share_tomography"]
    print("c5", d["value"], d["ms_per_step"], d["resamples"], {k: d[k].get("avg_kernel_us") for k in d if isinstance(d[k], dict) and "avg_kernel_us" in d[k]}, d.get("resample_kernel", {}).get("kick_us"))
p = json.load(open(O + "/plugin_paths.json"))["plugin_device_hook"]
print({k: (v.get("ms_per_datum"), v.get("resamples"), v.get("vs_native")) if isinstance(v, dict) else v for k, v in p.items() if k != "note"})
PY
cat $O/c5_kernel_stats.txt | head -30
