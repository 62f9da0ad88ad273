# Does something OUTSIDE the process (the driver samples GPU utilisation while bench.py runs: BENCH_r03 "gpu_busy") put
# milliseconds into a 3.5 ms timed region?  The driver's command, 4x quiet, 4x with rocm-smi polling beside it.
mkdir -p gpurun_out/r4b
cd $GRAFT_REPO_ROOT
for i in 1 2 3 4; do python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-other-configs --no-cpu-baseline > gpurun_out/r4b/quiet_$i.json 2>/dev/null; done
( while true; do rocm-smi --showuse --showmemuse --json > /dev/null 2>&1; sleep 0.2; done ) &
POLL=$!
for i in 1 2 3 4; do python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-other-configs --no-cpu-baseline > gpurun_out/r4b/smi_$i.json 2>/dev/null; done
kill $POLL
( while true; do cat /sys/class/drm/card*/device/gpu_busy_percent > /dev/null 2>&1; sleep 0.05; done ) &
POLL=$!
for i in 1 2 3 4; do python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-other-configs --no-cpu-baseline > gpurun_out/r4b/sysfs_$i.json 2>/dev/null; done
kill $POLL
for f in gpurun_out/r4b/*.json; do python3 -c "
import json
d=json.loads(open('$f').read().strip().splitlines()[-1])
print('$f', d['value'], d['ms_per_step'])
"; done
