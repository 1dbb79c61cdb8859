"""GPU-side (graph-replayed, no host launch overhead) timing of small launches."""
import math, sys, os
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from avlen_amd import _lib as L
from avlen_amd.engine import P


def graph_time(fn, n=50, reps=5):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n):
            fn()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps / n * 1e3


def gemm16(M, N, K, act=0):
    A = torch.randn(M, K, device="cuda").bfloat16(); W = (torch.randn(N, K, device="cuda") / math.sqrt(K)).bfloat16()
    b = torch.randn(N, device="cuda"); C = torch.empty(M, N, device="cuda")
    nb = L.lib.avlen_gemm_bf16_workspace_bytes(M, N); ws = torch.empty(nb, dtype=torch.uint8, device="cuda")
    return lambda: L.call("avlen_gemm_bf16", P(A), K, P(W), K, P(C), N, None, 0, P(b), None, 0, M, N, K, act, P(ws), nb, L.stream())


def ln(rows, d):
    x = torch.randn(rows, d, device="cuda"); g = torch.ones(d, device="cuda"); b = torch.zeros(d, device="cuda"); y = torch.empty_like(x)
    return lambda: L.call("avlen_layernorm_fwd", P(x), None, P(g), P(b), P(y), None, None, rows, d, 1e-5, L.stream())


def attn(B, H, S, D, causal):
    d = H * D
    q = torch.randn(B, S, 3 * d, device="cuda"); o = torch.empty(B, S, d, device="cuda")
    return lambda: L.call("avlen_attention_fwd", P(q), 3 * d, E(q, d), 3 * d, E(q, 2 * d), 3 * d, P(o), d, None, None, B, H, S, S, D,
                          causal, 0.125, L.stream())


def E(t, off):
    import ctypes
    return ctypes.c_void_p(t.data_ptr() + 4 * off)


if __name__ == "__main__":
    for (M, N, K) in [(64, 256, 64), (64, 256, 256), (64, 256, 512), (64, 64, 8192), (64, 768, 256), (256, 256, 256), (4928, 512, 512),
                      (4928, 512, 2048), (4928, 2048, 512), (4928, 1536, 512), (19264, 256, 320)]:
        print(f"gemm_bf16 M={M} N={N} K={K}: {graph_time(gemm16(M, N, K)):7.2f} us", flush=True)
    print(f"layernorm 64x256: {graph_time(ln(64, 256)):6.2f} us; 4928x512: {graph_time(ln(4928, 512)):6.2f} us")
    print(f"attention CLIP (64,8,77,64 causal): {graph_time(attn(64, 8, 77, 64, 1)):6.2f} us")
    print(f"attention SMT unmasked (64,8,301,32): {graph_time(attn(64, 8, 301, 32, 0)):6.2f} us")
