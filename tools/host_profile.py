"""Lab: where the HOST time of a rollout step goes (cProfile over the harness' default flow).
Usage: python tools/host_profile.py [envs] [steps]"""
import cProfile, io, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "4")
import torch
from avlen_amd.harness import Workload

N = int(sys.argv[1]) if len(sys.argv) > 1 else 64
STEPS = int(sys.argv[2]) if len(sys.argv) > 2 else 100
wl = Workload(N, 150, spectrogram=(257, 101, 2), precision="bf16x3", pretraining=True)
wl.cycle()
for _ in range(10):
    wl.rollout_step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(STEPS):
    wl.rollout_step()
torch.cuda.synchronize()
print("plain: %.1f us per step" % ((time.perf_counter() - t0) / STEPS * 1e6))
wl.update()
for _ in range(10):
    wl.rollout_step()
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
t0 = time.perf_counter()
for _ in range(STEPS):
    wl.rollout_step()
torch.cuda.synchronize()
dt = time.perf_counter() - t0
pr.disable()
print("profiled: %.1f us per step" % (dt / STEPS * 1e6))
for key in ("tottime", "cumulative"):
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats(key).print_stats(45)
    print(s.getvalue())
