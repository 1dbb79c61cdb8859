"""Determinism / agreement probe of the text tower's two column splits on the benched token mix."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from avlen_amd import _lib as L
from avlen_amd.harness import Workload
wl = Workload(64, 4, precision="bf16x3", use_graphs=False, share_encoders=False, launch_ahead=False, with_goal_policy=False)
pol = wl.pi_l
for n in (64, 33, 16, 5):
    tok = wl.dialog[1][:n].contiguous()
    outs = {}
    for lim in (0, 256):
        L.lib.avlen_set_clip_tower_split4_wgs(lim)
        runs = [pol.net.encode_text(pol, tok).clone() for _ in range(6)]
        torch.cuda.synchronize()
        same = all(torch.equal(runs[0], r) for r in runs[1:])
        outs[lim] = runs[0]
        print(f"n={n} split4_limit={lim}: run-to-run identical {same}, finite {bool(torch.isfinite(runs[0]).all())}")
    print(f"   |4-way - 2-way| max {float((outs[0] - outs[256]).abs().max()):.3e}")
L.lib.avlen_set_clip_tower_split4_wgs(-1)
