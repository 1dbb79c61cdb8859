"""Rollout-step time of mixed-precision configurations (graphs on, no sharing / launch-ahead): which modules' compensated path costs what."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from avlen_amd.harness import Workload
N, T = 64, 12
kw = dict(spectrogram=(257, 101, 2), pretraining=True, seed=0, use_graphs=True, share_encoders=False, launch_ahead=False)
ALL = ("towers", "audio", "smt", "clip", "dialog")
for name, fast in (("all x3", ()), ("towers bf16", ("towers",)), ("clip bf16", ("clip",)), ("towers+clip bf16", ("towers", "clip")),
                   ("towers+clip+audio bf16", ("towers", "clip", "audio")), ("only smt x3", ("towers", "clip", "audio", "dialog")),
                   ("only dialog x3", ("towers", "clip", "audio", "smt")), ("all bf16", ALL)):
    wl = Workload(N, T, precision="bf16x3", **kw)
    for pol in (wl.pi_q, wl.pi_g, wl.pi_l):
        pol.module_precision = {m: "bf16" for m in fast}
    for _ in range(3):
        wl.rollout_step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(8):
        wl.rollout_step()
    torch.cuda.synchronize()
    print("%-24s %.3f ms/step" % (name, (time.perf_counter() - t0) / 8 * 1e3), flush=True)
    del wl
    torch.cuda.empty_cache()
