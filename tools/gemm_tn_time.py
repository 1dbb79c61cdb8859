"""Times the row-contracted weight-gradient product (avlen_gemm_tn_bf16) at the 2nd-stage update's shapes next to the route it
replaces (two transposing casts + the row-times-row GEMM): python tools/gemm_tn_time.py"""
import sys, torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from avlen_amd import _lib as L

M = 722400
for N1, N2 in ((256, 256), (768, 256), (512, 256), (256, 272)):
    A = torch.randn(M, N1, device="cuda").bfloat16(); B = torch.randn(M, N2, device="cuda").bfloat16()
    C = torch.zeros(N1, N2, device="cuda")
    nb = L.lib.avlen_gemm_tn_bf16_workspace_bytes(M, N1, N2)
    ws = torch.empty(nb, dtype=torch.uint8, device="cuda")
    def run():
        L.call("avlen_gemm_tn_bf16", L.ptr(A), N1, L.ptr(B), N2, M, N1, N2, L.ptr(C), N2, 0.0, L.ptr(ws), nb, L.stream())
    for _ in range(3): run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): run()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 20
    byt = 2.0 * M * (N1 + N2)
    print("dW %4d x %4d over %d rows: %7.1f us  %6.1f TFLOP/s  %5.2f TB/s (operands read once)" %
          (N1, N2, M, us, 2.0 * M * N1 * N2 / us * 1e-6, byt / us * 1e-6), flush=True)
    ref = (A[:20000].float().t() @ B[:20000].float())
    C2 = torch.zeros(N1, N2, device="cuda")
    L.call("avlen_gemm_tn_bf16", L.ptr(A), N1, L.ptr(B), N2, 20000, N1, N2, L.ptr(C2), N2, 0.0, L.ptr(ws), nb, L.stream())
    torch.cuda.synchronize()
    print("   max |d| vs torch over 20000 rows: %.3e (scale %.1f)" % (float((C2 - ref).abs().max()), float(ref.abs().max())))
