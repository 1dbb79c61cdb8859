"""Runs ONLY the dominant kernel of the rollout (g2_kernel<128,16,4,1,2>: the 3x3 16->16 implicit-GEMM conv of the
ResNet towers' layer1 at 64x64, six towers x 64 envs = 384 images per launch, fused GroupNorm statistics) so that
rocprofv3 --pmc passes can attribute HBM traffic to it.  Also prints its event-timed duration and algorithmic bytes."""
import math, sys, os, json
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from avlen_amd import _lib as L
from avlen_amd.engine import P

B, H, C = 384, 64, 16


def make():
    x16 = torch.randn(B, H, H, C, device="cuda").bfloat16()
    w = torch.randn(C, C, 3, 3, device="cuda") / math.sqrt(C * 9)
    wp16 = torch.empty(C, 3, 3, C, device="cuda", dtype=torch.bfloat16)
    L.call("avlen_pack_conv_weight_bf16", P(w), P(wp16), C, C, 3, 3, C, L.stream())
    y = torch.empty(B, H, H, C, device="cuda")
    stats = torch.zeros(B, 2, C, device="cuda")
    nb = L.lib.avlen_gemm_bf16_workspace_bytes(1, 1)
    ws = torch.empty(nb, dtype=torch.uint8, device="cuda")
    keep = (x16, wp16, y, stats, ws)
    fn = lambda: L.call("avlen_conv2d_nhwc_bf16", P(x16), P(wp16), None, None, P(y), None, P(stats), B, H, H, C, C, 3, 3, 1, 1, 0,
                        P(ws), nb, L.stream())
    return fn, keep


def algorithmic_bytes():
    # read the bf16 activation once, write the fp32 raw output once (weights 4.6 KB, statistics 48 KB: negligible)
    return B * H * H * C * 2 + B * H * H * C * 4


def measure(iters=40):
    fn, keep = make()
    for _ in range(5):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 1e3 / iters


if __name__ == "__main__":
    sec = measure()
    print(json.dumps({"kernel": "g2_kernel<128,16,4,1,2>", "us_per_launch": sec * 1e6, "algorithmic_bytes": algorithmic_bytes(),
                      "GBps": algorithmic_bytes() / sec / 1e9}))
