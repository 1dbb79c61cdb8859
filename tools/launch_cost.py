"""Lab: host cost of the pieces of the recorded step (hipGraphLaunch per captured graph, staging copy, events) with the GPU idle,
and node counts.  Usage: python tools/launch_cost.py"""
import os, sys, time, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "4")
import torch
from avlen_amd.harness import Workload
from avlen_amd import _lib as L

wl = Workload(64, 150, spectrogram=(257, 101, 2), precision="bf16x3", pretraining=True)
wl.cycle()
for _ in range(5):
    wl.rollout_step()
torch.cuda.synchronize()
plan = next(iter(wl.seq._plans.values()))
names = {1: "GRAPH", 2: "RECORD", 3: "WAIT", 4: "MULTICOPY"}
bt, bb = next(iter(plan.b_variants.values()))
for label, cmds in (("phase A (early)", plan.a_early), ("phase B: text tower", bt), ("phase B: pi_l's dialog half", bb)):
    print(label)
    for i in range(len(cmds)):
        one = (L.Cmd * 1)(cmds[i])
        ts = []
        for _ in range(30):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            L.lib.avlen_cmds_run(one, 1)
            ts.append((time.perf_counter() - t0) * 1e6)
        ts.sort()
        print(f"   {names[cmds[i].op]:9s} n={cmds[i].n:2d}  median {ts[15]:6.1f} us  min {ts[0]:6.1f} us")
    torch.cuda.synchronize()
# the encoder half (towers + audio) of the leader graph
g = plan.gq
for label, ex in (("leader graph1 (towers + fc + audio)", g.exec1), ("leader graph2 (pi_q rest)", g.exec2), ("pi_g graph", plan.gg.exec1),
                  ("pi_l graph1", plan.gl.exec1), ("pi_l graph2", plan.gl.exec2), ("text graph", plan.gt.exec1)):
    one = (L.Cmd * 1)()
    one[0].op, one[0].a, one[0].b = L.CMD_GRAPH, ex, plan.main
    ts = []
    for _ in range(30):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        L.lib.avlen_cmds_run(one, 1)
        ts.append((time.perf_counter() - t0) * 1e6)
    ts.sort()
    print(f"{label:40s} launch: median {ts[15]:6.1f} us  min {ts[0]:6.1f} us")
torch.cuda.synchronize()
# draws
B = 64
t0 = time.perf_counter()
for _ in range(1000):
    wl.pi_q._noise_dev("option", B, 2, None).exponential_(1)
print("exponential_ into mapped (64 x 2): %.2f us" % ((time.perf_counter() - t0) / 1000 * 1e6))
t0 = time.perf_counter()
for _ in range(1000):
    torch.get_rng_state()
print("get_rng_state: %.2f us" % ((time.perf_counter() - t0) / 1000 * 1e6))
