import sys, os, numpy as np, warnings, time, cProfile, pstats
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'python-qinfer_amd'))
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", RANK="0", WORLD_SIZE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
import torch, torch.distributed as dist
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
import qinfer_amd as qi
from qinfer_amd.parallel import ParticleShardGroup
comm = ParticleShardGroup()
warnings.simplefilter('ignore')
for thresh, tag in ((0.0, 'update only'), (1.0, 'update+resample')):
    upd=qi.SMCUpdater(qi.SimplePrecessionModel(), 40960, qi.UniformDistribution([0,1]), device_rng=True, resample_thresh=thresh, comm=comm)
    t=np.array([0.3 if thresh else 0.01])
    for _ in range(100): upd.update(0,t)
    torch.cuda.synchronize(); t0=time.perf_counter()
    for _ in range(2000): upd.update(0,t)
    torch.cuda.synchronize(); print(tag, 'per step us', (time.perf_counter()-t0)/2000*1e6)
    pr=cProfile.Profile(); pr.enable()
    for _ in range(2000): upd.update(0,t)
    pr.disable(); pstats.Stats(pr).sort_stats('tottime').print_stats(14)
comm.close(); dist.destroy_process_group()
