"""Direct conv (16ch @ 64x64, 384 images) with and without the fused GroupNorm statistics / fused input norm."""
import sys, os, math
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from avlen_amd import _lib as L
from avlen_amd.engine import P
from roofline_probe import measure

def make(stats_on, C=16, W=64, B=384):
    x16 = torch.randn(B, W, W, C, device="cuda").bfloat16()
    w = torch.randn(C, C, 3, 3, device="cuda") / math.sqrt(C * 9)
    wp16 = torch.empty(C, 3, 3, C, device="cuda", dtype=torch.bfloat16)
    L.call("avlen_pack_conv_weight_bf16", P(w), P(wp16), C, C, 3, 3, C, L.stream())
    y = torch.empty(B, W, W, C, device="cuda", dtype=torch.bfloat16); stats = torch.zeros(B, 2, C, device="cuda")
    fn = lambda: L.call("avlen_conv_direct_bf16", P(x16), P(wp16), P(y), P(stats) if stats_on else None, B, W, C, C, 3, L.stream())
    return fn, (x16, wp16, y, stats)

for C, W in ((16, 64), (32, 32)):
    for on in (True, False):
        t = measure(lambda: make(on, C, W))
        byt = 384 * W * W * C * 2 * 2
        print(f"dconv C={C} W={W} stats={on}: {t*1e6:.1f} us  {byt/t/1e9:.0f} GB/s")
