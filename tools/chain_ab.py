"""Lab: A/B of avlen_set_chain_one_xcd (the fused chains' working blocks on one XCD) -- GPU time of the captured pieces that hold a
chain, and the free-running step period.  Usage: python tools/chain_ab.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "4")
import torch
from avlen_amd.harness import Workload
from avlen_amd import _lib as L

for on in (1, 0, 1, 0):
    L.lib.avlen_set_chain_one_xcd(on)
    wl = Workload(64, 150, spectrogram=(257, 101, 2), precision="bf16x3", pretraining=True)
    wl.cycle()
    for _ in range(20):
        wl.rollout_step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(100):
        wl.rollout_step()
    torch.cuda.synchronize()
    per = (time.perf_counter() - t0) / 100 * 1e6
    plan = next(iter(wl.seq._plans.values()))
    out = []
    for name, g in (("pi_q rest", plan.gq.graph2), ("pi_g", plan.gg.graph), ("pi_l half 2", plan.gl.graph2)):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(30):
            g.replay()
        e1.record()
        torch.cuda.synchronize()
        out.append("%s %.1f us" % (name, e0.elapsed_time(e1) / 30 * 1e3))
    print("one_xcd=%d: step %.1f us | %s" % (on, per, " | ".join(out)), flush=True)
    del wl
    torch.cuda.empty_cache()
L.lib.avlen_set_chain_one_xcd(1)
