"""Towers: compensated bf16 (tower_x3.hip) and the bf16 fused kernels against the fp32 path; times at the rollout batch."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from avlen_amd import policy as P
from avlen_amd.harness import Workload
from avlen_amd.spaces import savi_observation_space, ActionSpace, SMT_KW
N = int(sys.argv[1]) if len(sys.argv) > 1 else 64
wl = Workload(N, 2, precision="fp32", use_graphs=False, share_encoders=False, launch_ahead=False, with_goal_policy=False, with_dialog_policy=False)
obs = {k: v[0] for k, v in wl.rollouts.observations.items()}
prev = wl.rollouts.prev_actions[0]
ref = None
for mode in ("fp32", "bf16x3", "bf16"):
    torch.manual_seed(0)
    pol = P.AudioNavOptionPolicy(savi_observation_space((257, 101, 2)), ActionSpace(4), pretraining=True, query_count_emb_size=32,
                                 precision=mode, **SMT_KW).to("cuda")
    pol.load_state_dict(wl.pi_q.state_dict())
    f = lambda: pol.net.features(pol, obs, prev, extra=wl.query_state[0])[0]
    out = f().clone()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        f()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 10
    if ref is None:
        ref = out
    d = (out - ref)
    print("%-7s visual max|d| %.3e rms %.3e (ref rms %.3e) | audio max|d| %.3e | finite %s | %.3f ms" % (
        mode, float(d[:, :128].abs().max()), float(d[:, :128].pow(2).mean().sqrt()), float(ref[:, :128].pow(2).mean().sqrt()),
        float(d[:, 144:272].abs().max()), bool(torch.isfinite(out).all()), dt * 1e3), flush=True)
