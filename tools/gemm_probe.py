"""GPU probe: correctness + timing of the fp32-staged (v1) and bf16/glds (v2) implicit-GEMM kernels on the
shapes of the hot path.  Usage: python tools/gemm_probe.py"""
import math, sys, os, time
import torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from avlen_amd import _lib as L
from avlen_amd.engine import P


def timeit(fn, iters=30):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3      # us


def gemm_case(M, N, K, act=0):
    torch.manual_seed(0)
    A = torch.randn(M, K, device="cuda"); W = torch.randn(N, K, device="cuda") / math.sqrt(K); b = torch.randn(N, device="cuda")
    A16, W16 = A.bfloat16().contiguous(), W.bfloat16().contiguous()
    C1 = torch.empty(M, N, device="cuda"); C2 = torch.empty(M, N, device="cuda"); C2h = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    nb1 = L.lib.avlen_gemm_workspace_bytes(M, N, K, 64); ws1 = torch.empty(nb1, dtype=torch.uint8, device="cuda")
    nb2 = L.lib.avlen_gemm_bf16_workspace_bytes(M, N); ws2 = torch.empty(nb2, dtype=torch.uint8, device="cuda")
    sk = L.lib.avlen_gemm_pick_splitk(M, N, K)
    st = L.stream()
    f1 = lambda: L.call("avlen_gemm", P(A), K, 0, P(W), K, 0, P(C1), N, P(b), None, 0, M, N, K, act, 1, sk, 0.0, P(ws1), nb1, st)
    f2 = lambda: L.call("avlen_gemm_bf16", P(A16), K, P(W16), K, P(C2), N, P(C2h), N, P(b), None, 0, M, N, K, act, P(ws2), nb2, st)
    f3 = lambda: L.call("avlen_gemm_bf16", P(A16), K, P(W16), K, None, N, P(C2h), N, P(b), None, 0, M, N, K, act, P(ws2), nb2, st)
    t1, t2, t3 = timeit(f1), timeit(f2), timeit(f3)
    ref = A16.float() @ W16.float().t() + b
    if act == 2:
        ref = ref * torch.sigmoid(1.702 * ref)
    e1 = float((C1 - ref).abs().max() / ref.abs().max()); e2 = float((C2 - ref).abs().max() / ref.abs().max())
    e3 = float((C2h.float() - ref).abs().max() / ref.abs().max())
    fl = 2.0 * M * N * K
    tr = timeit(lambda: torch.matmul(A16, W16.t()))
    print(f"gemm M={M:6d} N={N:5d} K={K:5d}: v1 {t1:8.1f}us {fl/t1/1e6:7.1f}TF err {e1:.1e} | v2 {t2:8.1f}us {fl/t2/1e6:7.1f}TF err {e2:.1e} "
          f"| v2(bf16 out only) {t3:8.1f}us {fl/t3/1e6:7.1f}TF err {e3:.1e} | torch.matmul bf16 {tr:8.1f}us {fl/tr/1e6:7.1f}TF", flush=True)


def conv_case(B, H, Cin, Cout, k, s, p):
    torch.manual_seed(1)
    x = torch.randn(B, H, H, Cin, device="cuda"); w = torch.randn(Cout, Cin, k, k, device="cuda") / math.sqrt(Cin * k * k)
    Cp = max(8, Cin)
    x16 = torch.zeros(B, H, H, Cp, device="cuda", dtype=torch.bfloat16); x16[..., :Cin] = x.bfloat16()
    wp = torch.empty(Cout, k, k, Cin, device="cuda"); wp16 = torch.empty(Cout, k, k, Cp, device="cuda", dtype=torch.bfloat16)
    st = L.stream()
    L.call("avlen_pack_conv_weight", P(w), P(wp), Cout, Cin, k, k, st)
    L.call("avlen_pack_conv_weight_bf16", P(w), P(wp16), Cout, Cin, k, k, Cp, st)
    OH = (H + 2 * p - k) // s + 1
    y1 = torch.empty(B, OH, OH, Cout, device="cuda"); y2 = torch.empty(B, OH, OH, Cout, device="cuda")
    nb2 = L.lib.avlen_gemm_bf16_workspace_bytes(B * OH * OH, Cout); ws2 = torch.empty(nb2, dtype=torch.uint8, device="cuda")
    f1 = lambda: L.call("avlen_conv2d_nhwc", P(x), P(wp), None, None, P(y1), B, H, H, Cin, Cout, k, k, s, p, 0, 1, st)
    f2 = lambda: L.call("avlen_conv2d_nhwc_bf16", P(x16), P(wp16), None, None, P(y2), None, None, B, H, H, Cp, Cout, k, k, s, p, 0, P(ws2), nb2, st)
    t1, t2 = timeit(f1), timeit(f2)
    ref = F.conv2d(x.bfloat16().float().permute(0, 3, 1, 2), w.bfloat16().float(), None, stride=s, padding=p).permute(0, 2, 3, 1)
    e1 = float((y1 - ref).abs().max() / ref.abs().max()); e2 = float((y2 - ref).abs().max() / ref.abs().max())
    fl = 2.0 * B * OH * OH * Cout * Cin * k * k
    print(f"conv B={B:4d} H={H:3d} {Cin:3d}->{Cout:3d} k{k} s{s}: v1 {t1:8.1f}us {fl/t1/1e6:7.1f}TF err {e1:.1e} | v2 {t2:8.1f}us {fl/t2/1e6:7.1f}TF err {e2:.1e}", flush=True)


if __name__ == "__main__":
    print(L.lib.avlen_build_info())
    for (M, N, K, act) in [(4928, 2048, 512, 2), (4928, 512, 2048, 0), (4928, 1536, 512, 0), (4928, 512, 512, 0),
                           (19264, 256, 320, 1), (19264, 768, 256, 0), (64, 256, 256, 0), (64, 64, 8192, 0), (4800, 64, 8192, 0),
                           (8192, 8192, 8192, 0)]:
        gemm_case(M, N, K, act)
    for cfg in [(64, 64, 3, 16, 7, 1, 3), (64, 64, 16, 16, 3, 1, 1), (64, 64, 16, 32, 3, 2, 1), (64, 32, 32, 32, 3, 1, 1),
                (64, 32, 32, 64, 3, 2, 1), (64, 16, 64, 64, 3, 1, 1), (64, 16, 64, 128, 3, 2, 1), (64, 8, 128, 128, 3, 1, 1),
                (64, 32, 16, 32, 1, 2, 0), (4800, 64, 16, 16, 3, 1, 1), (4800, 8, 128, 128, 3, 1, 1)]:
        conv_case(*cfg)
