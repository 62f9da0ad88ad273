#!/bin/bash
# round 6: the plugin surface's tests + four ways (native / hip / torch / numpy)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r6h; mkdir -p $O
python -m pytest tests/test_plugin_device.py tests/test_parallel_gloo.py -q -m gpu -k "plugin" > $O/pytest_plugin.log 2>&1; tail -30 $O/pytest_plugin.log
python tools/plugin_time.py 2>&1 | tail -2
python bench.py --only plugin_paths 2>$O/plugin_paths.err | tail -1 > $O/plugin_paths.json
python - <<'PY'
import json
p = json.load(open("gpurun_out/r6h/plugin_paths.json"))["plugin_device_hook"]
print({k: (round(v["ms_per_datum"], 4), v["resamples"], v.get("vs_native")) if isinstance(v, dict) and "ms_per_datum" in v else v for k, v in p.items() if k not in ("note", "workload")})
PY
tail -3 $O/plugin_paths.err
