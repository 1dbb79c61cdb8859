"""GPU probe: time of the attention backward kernels at the 2nd-stage minibatch shape (2400 samples x 8 heads x 301 tokens)."""
import sys, os, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from avlen_amd import _lib as L
B, H, S, D = 2400, 8, 301, 32
d = H * D
q, k, v, do = (torch.randn(B, S, d, device="cuda") for _ in range(4))
md = (torch.rand(B, S, device="cuda") < 0.6).float(); md[:, -1] = 1
scale = 1 / math.sqrt(D)
o = torch.empty(B, S, d, device="cuda"); lse = torch.empty(B, H, S, device="cuda")
P = L.ptr
L.call("avlen_attention_fwd", P(q), d, P(k), d, P(v), d, P(o), d, P(md), P(lse), B, H, S, S, D, 0, scale, L.stream())
dq, dk, dv = (torch.zeros(B, S, d, device="cuda") for _ in range(3)); delta = torch.empty(B, H, S, device="cuda")
for name in ("avlen_attention_bwd", "avlen_attention_bwd_bf16"):
    f = lambda: L.call(name, P(q), d, P(k), d, P(v), d, P(o), d, P(do), d, P(md), P(lse), P(delta), P(dq), d, P(dk), d, P(dv), d, B, H, S, S,
                       D, 0, scale, L.stream())
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); f(); f(); f(); e1.record(); torch.cuda.synchronize()
    print(name, round(e0.elapsed_time(e1) / 3, 2), "ms")
