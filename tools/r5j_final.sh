#!/bin/bash
# round 5, last refresh: the GPU suite + smoke(), then profiles/${T}_* (kernel stats and gaps per config without the PMC
# passes, the default bench line, three runs of the driver's command, config 4's resample launches)
cd $GRAFT_REPO_ROOT
T=${T:-r5_b}
export TMPDIR=/tmp
bash tools/r5f_suite.sh
rm -rf gpurun_out/profiles_new
bash tools/profile_configs.sh ${T} > gpurun_out/r5j_profile.log 2>&1
P=gpurun_out/profiles_new
for i in 1 2 3; do python3 bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | tail -1 > $P/${T}_driver_cmd_$i.json; done
bash tools/r4l_c4_launches.sh > $P/${T}_c4_resample_launches.txt 2>&1
for i in 1 2 3; do python3 -c "
import json
d = json.load(open('$P/${T}_driver_cmd_$i.json'))
print('driver cmd', d['value'], d['ms_per_step'], d['roofline']['frac'])
"; done
python3 -c "
import json
d = json.load(open('$P/${T}_bench_line.json'))
print('200 steps', d['value'], d['ms_per_step'])
for k, v in d['other_configs'].items(): print(k, v.get('value'), v.get('ms_per_step'))
print({k: (v.get('value') if isinstance(v, dict) else v) for k, v in d.get('other_paths', {}).items()})
"
