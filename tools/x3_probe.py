"""bf16x3 (compensated bf16) against the fp32 parity mode: per-step max |value| / |prob| differences of the three policies on the
fp32 workload's state, and the cycle time of both modes.  usage: python tools/x3_probe.py [N] [T] [mode]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from avlen_amd.harness import Workload

N = int(sys.argv[1]) if len(sys.argv) > 1 else 64
T = int(sys.argv[2]) if len(sys.argv) > 2 else 10
mode = sys.argv[3] if len(sys.argv) > 3 else "bf16x3"
kw = dict(spectrogram=(257, 101, 2), pretraining=True, seed=0, use_graphs=True, share_encoders=True, launch_ahead=(mode != "fp32"))
wl32 = Workload(N, T, precision="fp32", **dict(kw, launch_ahead=False))
wlx = Workload(N, T, precision=mode, **kw)
mv = {}
flips = 0
torch.manual_seed(4242)
for t in range(T):
    rng = torch.get_rng_state()
    ox = {k: v.clone() for k, v in wlx.policies_on(wl32, t).items()}
    torch.set_rng_state(rng)
    o32 = wl32.rollout_step(return_outs=True)
    for k in ("q_value", "g_value", "l_value", "q_prob", "g_prob", "l_prob", "row_q", "row_g", "row_l", "row_d"):
        mv[k] = max(mv.get(k, 0.0), float((ox[k] - o32[k]).abs().max()))
    for k in "qgl":
        flips += int((ox["a_" + k] != o32["a_" + k]).sum())
print(mode, "vs fp32:", {k: "%.2e" % v for k, v in mv.items()}, "flips", flips, flush=True)
out32 = wl32.update()
for name, wl in (("fp32", wl32), (mode, wlx)):
    wl.cycle()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(T):
        wl.rollout_step()
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    out = wl.update()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("%s: rollout %.3f ms/step, update %.1f ms (T=%d N=%d) -> %.0f env-steps/s at T=150: %.0f" % (
        name, (t1 - t0) / T * 1e3, (t2 - t1) * 1e3, T, N, N * T / (t2 - t0),
        N * 150 / ((t1 - t0) / T * 150 + (t2 - t1) * 150 / T)), [round(float(x), 5) for x in out], flush=True)
