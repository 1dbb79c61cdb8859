"""GPU probe: where the bf16 MFMA attention backward differs from the fp32 kernels."""
import sys, os, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from avlen_amd import _lib as L
torch.manual_seed(6)
B, H, S, D = 3, 8, 301, 32
d = H * D
q, k, v, do = (torch.randn(B, S, d, device="cuda") for _ in range(4))
mask = (torch.rand(B, S) < 0.6).float(); mask[:, -1] = 1.0; mask[0, :-1] = 0.0
md = mask.cuda()
scale = 1.0 / math.sqrt(D)
o = torch.empty(B, S, d, device="cuda"); lse = torch.empty(B, H, S, device="cuda")
P = L.ptr
L.call("avlen_attention_fwd", P(q), d, P(k), d, P(v), d, P(o), d, P(md), P(lse), B, H, S, S, D, 0, scale, L.stream())
outs = {}
for name in ("avlen_attention_bwd", "avlen_attention_bwd_bf16"):
    dq, dk, dv = (torch.zeros(B, S, d, device="cuda") for _ in range(3)); delta = torch.empty(B, H, S, device="cuda")
    L.call(name, P(q), d, P(k), d, P(v), d, P(o), d, P(do), d, P(md), P(lse), P(delta), P(dq), d, P(dk), d, P(dv), d, B, H, S, S, D, 0,
           scale, L.stream())
    torch.cuda.synchronize()
    outs[name] = (dq, dk, dv)
for i, nm in enumerate(("dq", "dk", "dv")):
    a, b = outs["avlen_attention_bwd"][i], outs["avlen_attention_bwd_bf16"][i]
    e = (a - b).abs()
    print(nm, "max|ref|", float(a.abs().max()), "per-sample max err", [round(float(e[s].max()), 4) for s in range(B)],
          "per-sample max|ref|", [round(float(a[s].abs().max()), 3) for s in range(B)])
    s_, r_, c_ = [int(x) for x in torch.nonzero(e == e.max())[0]]
    print("   worst at sample", s_, "row", r_, "col", c_, "ref", float(a[s_, r_, c_]), "got", float(b[s_, r_, c_]), "mask[row]", float(mask[s_, r_]))
    # error by key-valid / masked rows
    for s in range(B):
        mv = mask[s].bool().cuda()
        print("   sample", s, "rows valid: max err", round(float(e[s][mv].max()), 4), " rows masked:", round(float(e[s][~mv].max()) if (~mv).any() else 0.0, 4))
