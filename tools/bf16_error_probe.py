"""Where does the bf16 mode's value error come from?  Same weights, same observations: towers / audio CNN / state encoder, one at a time."""
import sys, os
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from avlen_amd.harness import Workload

N = int(os.environ.get("N", 64))
kw = dict(spectrogram=(257, 101, 2), pretraining=True, em_capacity=150, seed=0, use_graphs=False, share_encoders=False, launch_ahead=False)
w16 = Workload(N, 2, precision="bf16", **kw)
w32 = Workload(N, 2, precision="fp32", **kw)
for name in ("pi_q", "pi_g"):
    p16, p32 = getattr(w16, name), getattr(w32, name)
    v = w32._step_views(0)
    ro = w32.rollouts
    extra = v["qs"] if name == "pi_q" else None
    f16, g16 = p16.net.features(p16, v["obs"], v["prev"], extra=extra)
    f32, g32 = p32.net.features(p32, v["obs"], v["prev"], extra=extra)
    torch.cuda.synchronize()
    sl = {"rgb": slice(0, 64), "depth": slice(64, 128), "action": slice(128, 144), "audio": slice(144, 272)}
    for k, s in sl.items():
        d = (f16[:, s] - f32[:, s]).abs()
        print(f"{name} {k:7s}: max|err| {float(d.max()):.4g}  rms err {float(d.pow(2).mean().sqrt()):.4g}  rms value {float(f32[:, s].pow(2).mean().sqrt()):.4g}")
    mem = ro.external_memory_option[:, 0] if name == "pi_q" else ro.external_memory_goal[:, 0]
    # state encoder alone: bf16 SMT on the fp32 features vs fp32 SMT on the fp32 features
    x16, _ = p16.net.smt(p16, f32, g32, mem, v["em_masks"])
    x32, _ = p32.net.smt(p32, f32, g32, mem, v["em_masks"])
    xx, _ = p16.net.smt(p16, f16, g16, mem, v["em_masks"], save_key="b")
    torch.cuda.synchronize()
    which = "option" if name == "pi_q" else "goal"
    h16a, h32, h16 = p16._heads_first(which, x16), p32._heads_first(which, x32), p16._heads_first(which, xx)
    torch.cuda.synchronize()
    print(f"{name} state encoder alone (fp32 feats): max|x err| {float((x16 - x32).abs().max()):.4g}; value err {float((h16a['value'] - h32['value']).abs().max()):.4g}; "
          f"prob err {float((h16a['probs'] - h32['probs']).abs().max()):.4g}")
    print(f"{name} end to end: value err {float((h16['value'] - h32['value']).abs().max()):.4g} (|value| max {float(h32['value'].abs().max()):.3g}); prob err {float((h16['probs'] - h32['probs']).abs().max()):.4g}")
    # fp32 SMT on bf16 features: the towers' share
    x3, _ = p32.net.smt(p32, f16, g16, mem, v["em_masks"], save_key="c")
    h3 = p32._heads_first(which, x3)
    torch.cuda.synchronize()
    print(f"{name} towers' share (fp32 SMT on bf16 feats): value err {float((h3['value'] - h32['value']).abs().max()):.4g}")
