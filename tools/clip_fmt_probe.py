"""CLIP text tower: fp16 / bf16 fast paths against the fp32 path (embedding after dialog_layer), and their times."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from avlen_amd import policy as P
from avlen_amd.harness import Workload
from avlen_amd.spaces import savi_observation_space, ActionSpace, SMT_KW
wl = Workload(64, 2, precision="fp32", use_graphs=False, share_encoders=False, launch_ahead=False, with_goal_policy=False)
tok = wl.dialog[0]
ref = None
for mode in ("fp32", "bf16x3", "bf16"):
    torch.manual_seed(0)
    pol = P.AudioNavDialogPolicy(savi_observation_space((257, 101, 2)), ActionSpace(4), pretraining=False, num_steps=3,
                                 precision=mode, **SMT_KW).to("cuda")
    pol.load_state_dict(wl.pi_l.state_dict())
    f = lambda: pol.net._dialog_embed(pol, pol.net.encode_text(pol, tok))
    out = f().clone()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        f()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 10
    if ref is None:
        ref = out
    print("%-7s clip=%s  max|d| %.3e  rms %.3e  (ref rms %.3e)  %.3f ms" % (
        mode, pol.module_precision.get("clip", mode), float((out - ref).abs().max()), float((out - ref).pow(2).mean().sqrt()),
        float(ref.pow(2).mean().sqrt()), dt * 1e3), flush=True)
