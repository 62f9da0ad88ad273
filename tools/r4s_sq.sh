cd $GRAFT_REPO_ROOT
bash tools/sq_counters.sh r4z_paths --only other_paths > /dev/null 2>&1
python3 - <<'P'
import json
d=json.load(open('gpurun_out/sq/r4z_paths_sq_counters.json'))
for k,v in d['kernels'].items():
    if 'hyp' in k or 'multi' in k:
        print(k, {c: (round(x,1) if isinstance(x,float) else x) for c,x in v.items()})
P
