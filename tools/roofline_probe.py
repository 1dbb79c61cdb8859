"""Runs the two kernels bench.py reports a roofline for, in isolation, so that rocprofv3 --pmc passes can attribute HBM
traffic to them:
  * `gemm`  : g2_kernel<64,128,2,4,2,512> (8-wave ping-pong tile, 2 LDS stages) -- the dominant kernel of the rollout by GPU
              time (profiles/r01_rocprof_summary.md: 630 of 2400 us per step), on its heaviest call site: the CLIP text MLP
              up-projection c_fc on the ragged batch (M = 2464 live rows = half of 64 x 77, N = 2048, K = 512), bf16 operands
              via global_load_lds, bias + QuickGELU epilogue, bf16 output.  MFMA-bound class.
  * `dconv` : dconv3x3_kernel<16,16,64,3> -- the layer-1 3x3 convolution of the six ResNet towers (384 images of 64x64x16 per
              launch, bf16 in / bf16 out, fused GroupNorm statistics).  HBM-bound class.
Prints event-timed durations and algorithmic FLOPs / bytes per launch as JSON."""
import math, sys, os, json
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from avlen_amd import _lib as L
from avlen_amd.engine import P

GEMM_KERNEL_NAME = "g2_kernel<64,128,2,4,2,512>"
GEMM = dict(M=2464, N=2048, K=512, epilogue="bf16", act=2)
CONV = dict(B=384, W=64, C=16)


FMT = 0          # 16-bit operand format of the probed GEMM: 0 = bf16, 1 = fp16 (the CLIP text tower of the bf16x3 mode)


def make_gemm(M=None, N=None, K=None, epilogue=None, act=None):
    """epilogue: "residual32" = fp32 residual in/out (out_proj / c_proj), "bf16" = bias (+ activation), 16-bit out only (in_proj / c_fc)."""
    M, N, K = M or GEMM["M"], N or GEMM["N"], K or GEMM["K"]
    epilogue = epilogue or GEMM["epilogue"]
    act = GEMM["act"] if act is None else act
    dt = torch.float16 if FMT == 1 else torch.bfloat16
    A = torch.randn(M, K, device="cuda").to(dt); Wt = (torch.randn(N, K, device="cuda") / math.sqrt(K)).to(dt)
    b = torch.randn(N, device="cuda"); X = torch.randn(M, N, device="cuda")
    Y16 = torch.empty(M, N, device="cuda", dtype=dt)
    nb = L.lib.avlen_gemm_bf16_workspace_bytes(M, N); ws = torch.empty(nb, dtype=torch.uint8, device="cuda")
    if epilogue == "residual32":
        fn = lambda: L.call("avlen_gemm_h16", P(A), K, P(Wt), K, P(X), N, None, 0, P(b), P(X), N, M, N, K, 0, FMT, None, 0, L.stream())
    else:
        fn = lambda: L.call("avlen_gemm_h16", P(A), K, P(Wt), K, None, 0, P(Y16), N, P(b), None, 0, M, N, K, act, FMT, None, 0, L.stream())
    return fn, (A, Wt, b, X, Y16, ws)


def make_conv():
    B, W, C = CONV["B"], CONV["W"], CONV["C"]
    x16 = torch.randn(B, W, W, C, device="cuda").bfloat16()
    w = torch.randn(C, C, 3, 3, device="cuda") / math.sqrt(C * 9)
    wp16 = torch.empty(C, 3, 3, C, device="cuda", dtype=torch.bfloat16)
    L.call("avlen_pack_conv_weight_bf16", P(w), P(wp16), C, C, 3, 3, C, L.stream())
    y = torch.empty(B, W, W, C, device="cuda", dtype=torch.bfloat16); stats = torch.zeros(B, 2, C, device="cuda")
    fn = lambda: L.call("avlen_conv_direct_bf16", P(x16), P(wp16), P(y), P(stats), B, W, C, C, 3, L.stream())
    return fn, (x16, wp16, y, stats)


def gemm_work():
    M, N, K = GEMM["M"], GEMM["N"], GEMM["K"]
    return {"flops": 2.0 * M * N * K, "bytes": M * K * 2 + N * K * 2 + M * N * 2}         # A + W (bf16) in, bf16 out


def conv_work():
    B, W, C = CONV["B"], CONV["W"], CONV["C"]
    return {"flops": 2.0 * B * W * W * C * C * 9, "bytes": B * W * W * C * 2 * 2}          # bf16 activation in + out, once each


# the other GEMMs of one CLIP text block on the same ragged batch (same kernel family): qkv, out_proj (NS = 2), c_proj (NS = 4)
CLIP_SITES = {"in_proj": (2464, 1536, 512, "bf16", 0), "out_proj": (2464, 512, 512, "residual32", 0),
              "c_proj": (2464, 512, 2048, "residual32", 0)}


def clip_call_sites():
    out = {}
    for name, (M, N, K, ep, act) in CLIP_SITES.items():
        s = measure(lambda: make_gemm(M, N, K, ep, act))
        out[name] = {"M": M, "N": N, "K": K, "us_per_launch": round(s * 1e6, 2), "TFLOPs": round(2.0 * M * N * K / s / 1e12, 1)}
    return out


# per-conv table of one visual tower at the rollout batch (6 towers x 64 images = 384 images per launch):
# (name, H_in, Cin, Cout, K, stride, launches per tower); M = B*OH*OW rows, N = Cout, K = k*k*Cin
TOWER_CONVS = [("conv1 7x7 s2 (Cin 3->8 padded)", 64, 8, 16, 7, 1, 1), ("layer1 3x3", 64, 16, 16, 3, 1, 4),
               ("layer2.0.conv1 3x3 s2", 64, 16, 32, 3, 2, 1), ("layer2 3x3", 32, 32, 32, 3, 1, 3),
               ("layer2.0.down 1x1 s2", 64, 16, 32, 1, 2, 1),
               ("layer3.0.conv1 3x3 s2", 32, 32, 64, 3, 2, 1), ("layer3 3x3", 16, 64, 64, 3, 1, 3),
               ("layer4.0.conv1 3x3 s2", 16, 64, 128, 3, 2, 1), ("layer4 3x3", 8, 128, 128, 3, 1, 3)]


def make_tower_conv(B, H, Cin, Cout, K, stride):
    x16 = torch.randn(B, H, H, Cin, device="cuda").bfloat16()
    w = torch.randn(Cout, Cin, K, K, device="cuda") / math.sqrt(Cin * K * K)
    wp16 = torch.empty(Cout, K, K, Cin, device="cuda", dtype=torch.bfloat16)
    L.call("avlen_pack_conv_weight_bf16", P(w), P(wp16), Cout, Cin, K, K, Cin, L.stream())
    OH = (H + 2 * (K // 2) - K) // stride + 1
    y = torch.empty(B, OH, OH, Cout, device="cuda", dtype=torch.bfloat16)
    stats = torch.zeros(B, 2, Cout, device="cuda")
    direct = stride == 1 and (Cin, Cout, H, K) in ((16, 16, 64, 3), (32, 32, 32, 3), (8, 16, 64, 7))
    if direct:
        fn = lambda: L.call("avlen_conv_direct_bf16", P(x16), P(wp16), P(y), P(stats), B, H, Cin, Cout, K, L.stream())
        ws = None
    else:
        nb = L.lib.avlen_gemm_bf16_workspace_bytes(B * OH * OH, Cout)
        ws = torch.empty(max(nb, 16), dtype=torch.uint8, device="cuda")
        fn = lambda: L.call("avlen_conv2d_nhwc_bf16", P(x16), P(wp16), None, None, None, P(y), P(stats), B, H, H, Cin, Cout, K, K,
                            stride, K // 2, 0, P(ws), nb, L.stream())
    return fn, (x16, wp16, y, stats, ws), OH, ("dconv3x3_kernel" if direct else "g2_kernel<CONV>")


def tower_conv_table(B=384):
    """Every conv shape of a visual tower as a standalone launch at the rollout batch: M, N, K, us, TFLOP/s, fraction of the bf16
    MFMA peak, and the HBM rate of its algorithmic bytes (bf16 activation in + out once).  Layers 3-4 run FUSED in
    tower_tail_kernel in the product (one launch, activations in LDS); their standalone rows are the unfused comparison."""
    rows = []
    for name, H, Cin, Cout, K, stride, count in TOWER_CONVS:
        holder = {}
        def mk():
            fn, keep, OH, kern = make_tower_conv(B, H, Cin, Cout, K, stride)
            holder.update(OH=OH, kern=kern)
            return fn, keep
        s = measure(mk, iters=20)
        OH = holder["OH"]
        M, N, Kd = B * OH * OH, Cout, K * K * Cin
        fl = 2.0 * M * N * Kd
        by = B * H * H * Cin * 2 + M * N * 2
        rows.append({"conv": name, "kernel": holder["kern"], "launches_per_tower": count, "M": M, "N": N, "K": Kd,
                     "us": round(s * 1e6, 2), "TFLOPs": round(fl / s / 1e12, 1), "frac_mfma": round(fl / s / 2.5e15, 4),
                     "GBps": round(by / s / 1e9, 1), "frac_hbm": round(by / s / 8e12, 4)})
    return rows


def measure(make, iters=40):
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


def product_kernels():
    """The two big kernels of a bf16x3 rollout step AS THE PRODUCT LAUNCHES THEM (so that the --pmc passes see them under their
    own names): the tower group (tower_x3_kernel: 6 towers x 64 images, persistent work queue) and the CLIP text tower
    (clip_tower_kernel: 64 dialogs).  Returns their event-timed durations."""
    from avlen_amd.harness import Workload
    wl = Workload(64, 2, precision="bf16x3", use_graphs=True)
    grp = wl.pi_q._enc_group
    obs = {k: v[0] for k, v in wl.rollouts.observations.items()}
    pol, toks = wl.pi_l, wl.dialog[0]
    st = torch.cuda.current_stream()
    tw = lambda: grp.run_all(wl.pi_q, obs["rgb"], obs["depth"])
    tx = lambda: pol.net.encode_text(pol, toks)
    out = {}
    for name, fn in (("tower_x3_us", tw), ("clip_tower_us", tx)):
        for _ in range(3):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            fn()
        e1.record()
        torch.cuda.synchronize()
        out[name] = e0.elapsed_time(e1) / 10 * 1e3
    return out


if __name__ == "__main__":
    sg = measure(make_gemm)
    res = {"gemm_us": sg * 1e6, "gemm_TFLOPs": gemm_work()["flops"] / sg / 1e12}
    res.update(product_kernels())
    print(json.dumps(res))
