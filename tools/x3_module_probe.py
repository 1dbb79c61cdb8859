"""Which modules need compensated arithmetic?  Base mode bf16x3 everywhere; one module at a time is switched to the bf16 fast
path and the three policies are compared with the fp32 parity mode on the fp32 workload's state."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from avlen_amd.harness import Workload

N = int(sys.argv[1]) if len(sys.argv) > 1 else 64
T = int(sys.argv[2]) if len(sys.argv) > 2 else 6
kw = dict(spectrogram=(257, 101, 2), pretraining=True, seed=0, use_graphs=False, share_encoders=False, launch_ahead=False)
wl32 = Workload(N, T, precision="fp32", **kw)
wlx = Workload(N, T, precision="bf16x3", **kw)
# a few optimiser steps so that the heads are not at their 0.01-gain initialisation
for _ in range(2):
    wl32.cycle()
wlx.pi_q.load_state_dict(wl32.pi_q.state_dict())
states = []
cases = [None, "towers", "audio", "smt", "clip", "dialog", "all"]
res = {c: {} for c in cases}
for t in range(T):
    o32 = None
    for c in cases:
        mp = {} if c is None else ({m: "bf16" for m in ("towers", "audio", "smt", "clip", "dialog")} if c == "all" else {c: "bf16"})
        for pol in (wlx.pi_q, wlx.pi_g, wlx.pi_l):
            pol.module_precision = mp
        rng = torch.get_rng_state()
        ox = {k: v.clone() for k, v in wlx.policies_on(wl32, t).items()}
        torch.set_rng_state(rng)
        if o32 is None:
            rng2 = torch.get_rng_state()
            o32 = {k: v.clone() for k, v in wl32.policies_on(wl32, t).items()}
            torch.set_rng_state(rng2)
        for k in ("q_value", "g_value", "l_value", "q_prob", "g_prob", "l_prob"):
            res[c][k] = max(res[c].get(k, 0.0), float((ox[k] - o32[k]).abs().max()))
    wl32.rollout_step()
for c in cases:
    print("%-8s" % (c or "x3"), {k: "%.2e" % v for k, v in res[c].items()}, flush=True)
