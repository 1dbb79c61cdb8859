import sys, os, cProfile, pstats
sys.path.insert(0, "/root/repo")
os.environ.setdefault("GPU_MAX_HW_QUEUES", "4")
import torch
from avlen_amd.harness import Workload
from avlen_amd import rollout_storage as RS
wl = Workload(64, 150, spectrogram=(257, 101, 2), precision="bf16", pretraining=True)
wl.cycle()
torch.cuda.synchronize()
pr = cProfile.Profile()
orig = RS.RolloutStorage.insert
def wrapped(self, *a, **k):
    pr.enable()
    try:
        return orig(self, *a, **k)
    finally:
        pr.disable()
RS.RolloutStorage.insert = wrapped
for _ in range(150):
    wl.rollout_step()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(18)
