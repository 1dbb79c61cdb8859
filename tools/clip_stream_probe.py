"""CLIP text tower: the one-launch sequence-stationary tower (csrc/clip_tower.hip) against the launch-per-GEMM fast path and the
fp32 path (raw encode_text output and the embedding after dialog_layer), with times.  AVLEN_CLIP_STREAM=0 builds no stream."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from avlen_amd import policy as P
from avlen_amd.harness import Workload
from avlen_amd.spaces import savi_observation_space, ActionSpace, SMT_KW
wl = Workload(64, 2, precision="fp32", use_graphs=False, share_encoders=False, launch_ahead=False, with_goal_policy=False)
tok = wl.dialog[0]
ref = None
for mode, stream in (("fp32", "0"), ("bf16x3", "0"), ("bf16x3", "1"), ("bf16", "0"), ("bf16", "1")):
    from avlen_amd import config as CFG
    CFG.CLIP_STREAM = stream != "0"
    torch.manual_seed(0)
    pol = P.AudioNavDialogPolicy(savi_observation_space((257, 101, 2)), ActionSpace(4), pretraining=False, num_steps=3,
                                 precision=mode, **SMT_KW).to("cuda")
    pol.load_state_dict(wl.pi_l.state_dict())
    f = lambda: pol.net.encode_text(pol, tok)
    out = f().clone()
    emb = pol.net._dialog_embed(pol, out).clone()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        f()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 10
    if ref is None:
        ref, ref_e = out, emb
    print("%-7s stream=%s  encode_text max|d| %.3e rms %.3e (ref rms %.3e, finite %s) | dialog embedding max|d| %.3e | %.3f ms" % (
        mode, stream, float((out - ref).abs().max()), float((out - ref).pow(2).mean().sqrt()), float(ref.pow(2).mean().sqrt()),
        bool(torch.isfinite(out).all()), float((emb - ref_e).abs().max()), dt * 1e3), flush=True)
