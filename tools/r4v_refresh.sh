cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/profiles_new gpurun_out/sq
bash tools/profile_configs.sh r4_c pmc > gpurun_out/r4v_profile.log 2>&1
for i in 1 2 3; do python3 bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | tail -1 > gpurun_out/profiles_new/r4_c_driver_cmd_$i.json; done
bash tools/sq_counters.sh r4_c_paths --only other_paths > /dev/null 2>&1
cp gpurun_out/sq/r4_c_paths_sq_counters.json gpurun_out/profiles_new/
for i in 1 2 3; do python3 -c "
import json
d=json.loads(open('gpurun_out/profiles_new/r4_c_driver_cmd_$i.json').read())
print('driver', d['value'], d['ms_per_step'], d.get('repeat_passes_ms_per_step'))
"; done
tail -40 gpurun_out/r4v_profile.log
