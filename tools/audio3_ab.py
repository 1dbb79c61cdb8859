"""Lab: the AudioCNN's convs as one LDS-resident launch (csrc/audio3.hip) against cast + one launch per conv (avlen_set_audio3(0)):
free-running rollout step period, alternating in one process (a fresh workload per setting: the choice is captured in its graphs).
Usage: python tools/audio3_ab.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "4")
import torch
from avlen_amd.harness import Workload
from avlen_amd import _lib as L

res = {0: [], 1: []}
for rep in range(6):
    v = rep & 1
    L.lib.avlen_set_audio3(v)
    wl = Workload(64, 150, spectrogram=(257, 101, 2), precision="bf16x3", pretraining=True)
    wl.cycle()
    for _ in range(20):
        wl.rollout_step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(wl.T - 20):
        wl.rollout_step()
    torch.cuda.synchronize()
    per = (time.perf_counter() - t0) / (wl.T - 20) * 1e6
    res[v].append(per)
    print("fused=%d: step %.1f us" % (v, per), flush=True)
    del wl
    torch.cuda.empty_cache()
L.lib.avlen_set_audio3(1)
for v in (0, 1):
    r = sorted(res[v])
    print("fused=%d: median %.1f us, min %.1f us" % (v, r[len(r) // 2], r[0]))
