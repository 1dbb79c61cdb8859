"""HIP primitive kernels (through the C ABI) vs plain PyTorch fp32 on the CPU.  Needs an MI355X."""
import math
import ctypes as C
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def L():
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    from avlen_amd import _lib
    return _lib


_KEEP = []


def dev(t):
    """Device copy that stays alive for the whole test (a temporary could be freed and its block re-used by the
    next allocation before the asynchronous kernel has run)."""
    d = t.cuda().contiguous()
    _KEEP.append(d)
    if len(_KEEP) > 64:
        torch.cuda.synchronize()
        del _KEEP[:32]
    return d


def rel_err(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-12))


TOL = {0: 2e-5, 1: 2e-2}          # fp32 MFMA (exact fma chain) / bf16 operands


def gemm(L, A, Bw, M, N, K, lda, ldb, ldc, transA=0, transB=0, bias=None, res=None, act=0, prec=0, splitk=1, beta=0.0,
         C0=None):
    out = C0.clone() if C0 is not None else torch.zeros(M, ldc, device="cuda")
    nb = L.lib.avlen_gemm_workspace_bytes(M, N, K, max(splitk, 1))
    ws = torch.empty(nb, dtype=torch.uint8, device="cuda")
    L.call("avlen_gemm", L.ptr(A), lda, transA, L.ptr(Bw), ldb, transB, L.ptr(out), ldc, L.ptr(bias), L.ptr(res),
           ldc if res is not None else 0, M, N, K, act, prec, splitk, beta, L.ptr(ws), nb, L.stream())
    torch.cuda.synchronize()
    return out


@pytest.mark.parametrize("prec", [0, 1])
@pytest.mark.parametrize("M,N,K,splitk", [(64, 64, 8192, 16), (300, 256, 288, 1), (5000, 768, 256, 1), (77, 16, 147, 1),
                                          (130, 21, 5, 1), (129, 130, 67, 2), (4928, 2048, 512, 1)])
def test_gemm_nt(L, M, N, K, splitk, prec):
    torch.manual_seed(0)
    A, W = torch.randn(M, K), torch.randn(N, K) / math.sqrt(K)
    b, r = torch.randn(N), torch.randn(M, N)
    ref = torch.relu(A @ W.t() + b) + r
    out = gemm(L, dev(A), dev(W), M, N, K, K, K, N, bias=dev(b), res=dev(r), act=1, prec=prec, splitk=splitk)
    assert rel_err(out, ref) < TOL[prec]


@pytest.mark.parametrize("prec", [0, 1])
def test_gemm_quickgelu(L, prec):
    torch.manual_seed(1)
    M, N, K = 200, 96, 64
    A, W = torch.randn(M, K), torch.randn(N, K) / 8
    h = A @ W.t()
    ref = h * torch.sigmoid(1.702 * h)
    out = gemm(L, dev(A), dev(W), M, N, K, K, K, N, act=2, prec=prec)
    assert rel_err(out, ref) < TOL[prec]


@pytest.mark.parametrize("prec", [0, 1])
def test_gemm_backward_products(L, prec):
    """dX = dY W (transB) and dW += dY^T X (transA, transB, split over rows, beta=1)."""
    torch.manual_seed(2)
    R, N, K = 3000, 256, 320
    dY, W, X = torch.randn(R, N), torch.randn(N, K) / 16, torch.randn(R, K)
    dX = gemm(L, dev(dY), dev(W), R, K, N, N, K, K, transB=1, prec=prec)
    assert rel_err(dX, dY @ W) < TOL[prec]
    G0 = torch.randn(N, K)
    dW = gemm(L, dev(dY), dev(X), N, K, R, N, K, K, transA=1, transB=1, prec=prec, splitk=8, beta=1.0, C0=dev(G0))
    assert rel_err(dW, G0 + dY.t() @ X) < TOL[prec]
    # odd sizes / unaligned leading dims (scalar load path)
    R, N, K = 333, 21, 309
    dY, X = torch.randn(R, N), torch.randn(R, K)
    dW = gemm(L, dev(dY), dev(X), N, K, R, N, K, K, transA=1, transB=1, prec=prec, splitk=3, beta=1.0,
              C0=torch.zeros(N, K, device="cuda"))
    assert rel_err(dW, dY.t() @ X) < TOL[prec]


CONVS = [  # B, H, W, Cin, Cout, k, stride, pad, bias, act
    (2, 64, 64, 3, 16, 7, 1, 3, False, 0), (2, 64, 64, 1, 16, 7, 1, 3, False, 0),
    (2, 64, 64, 16, 16, 3, 1, 1, False, 0), (2, 64, 64, 16, 32, 3, 2, 1, False, 0),
    (2, 64, 64, 16, 32, 1, 2, 0, False, 0), (3, 8, 8, 128, 128, 3, 1, 1, False, 0),
    (2, 65, 26, 2, 32, 5, 2, 0, True, 1), (2, 257, 101, 2, 32, 8, 4, 0, True, 1),
    (2, 31, 11, 32, 64, 3, 2, 0, True, 1), (2, 128, 128, 4, 32, 8, 4, 0, True, 1),
]


@pytest.mark.parametrize("prec", [0, 1])
@pytest.mark.parametrize("cfg", CONVS)
def test_conv2d_nhwc(L, cfg, prec):
    B, H, W, Cin, Cout, k, s, p, has_bias, act = cfg
    torch.manual_seed(3)
    x = torch.randn(B, H, W, Cin)
    w = torch.randn(Cout, Cin, k, k) / math.sqrt(Cin * k * k)
    b = torch.randn(Cout) if has_bias else None
    ref = F.conv2d(x.permute(0, 3, 1, 2), w, b, stride=s, padding=p)
    if act:
        ref = torch.relu(ref)
    ref = ref.permute(0, 2, 3, 1).contiguous()
    wd, wp = dev(w), torch.empty(Cout, k, k, Cin, device="cuda")
    L.call("avlen_pack_conv_weight", L.ptr(wd), L.ptr(wp), Cout, Cin, k, k, L.stream())
    assert torch.equal(wp.cpu(), w.permute(0, 2, 3, 1).contiguous())
    y = torch.empty(ref.shape, device="cuda")
    L.call("avlen_conv2d_nhwc", L.ptr(dev(x)), L.ptr(wp), L.ptr(dev(b)) if has_bias else None, None, L.ptr(y), B, H, W,
           Cin, Cout, k, k, s, p, act, prec, L.stream())
    torch.cuda.synchronize()
    assert rel_err(y, ref) < TOL[prec]


@pytest.mark.parametrize("B,HW,C,relu,res", [(3, 4096, 16, 1, False), (2, 1024, 32, 1, True), (5, 256, 64, 0, False),
                                             (64, 64, 128, 1, True)])
def test_groupnorm_nhwc(L, B, HW, C, relu, res):
    torch.manual_seed(4)
    x = torch.randn(B, HW, C) * 2 + 0.5
    g, b = torch.rand(C) + 0.5, torch.randn(C)
    r = torch.randn(B, HW, C) if res else None
    ref = F.group_norm(x.permute(0, 2, 1), 16, g, b, 1e-5).permute(0, 2, 1)
    if res:
        ref = ref + r
    if relu:
        ref = torch.relu(ref)
    y = torch.empty(B, HW, C, device="cuda")
    L.call("avlen_groupnorm_nhwc", L.ptr(dev(x)), L.ptr(dev(g)), L.ptr(dev(b)), L.ptr(dev(r)) if res else None, L.ptr(y),
           B, HW, C, 16, relu, 1e-5, L.stream())
    torch.cuda.synchronize()
    assert rel_err(y, ref) < 1e-5


@pytest.mark.parametrize("rows,d", [(301 * 3, 256), (77 * 2, 512), (5, 256)])
def test_layernorm_fwd_bwd(L, rows, d):
    torch.manual_seed(5)
    x = (torch.randn(rows, d) * 1.5 + 0.3).requires_grad_(True)
    r = torch.randn(rows, d)
    g, b = (torch.rand(d) + 0.5).requires_grad_(True), torch.randn(d).requires_grad_(True)
    y_ref = F.layer_norm(x + r, (d,), g, b, 1e-5)
    dy = torch.randn(rows, d)
    y_ref.backward(dy)
    y, mean, rstd = torch.empty(rows, d, device="cuda"), torch.empty(rows, device="cuda"), torch.empty(rows, device="cuda")
    L.call("avlen_layernorm_fwd", L.ptr(dev(x.detach())), L.ptr(dev(r)), L.ptr(dev(g.detach())), L.ptr(dev(b.detach())),
           L.ptr(y), L.ptr(mean), L.ptr(rstd), rows, d, 1e-5, L.stream())
    torch.cuda.synchronize()
    assert rel_err(y, y_ref.detach()) < 1e-5
    dx, dg, db = torch.empty(rows, d, device="cuda"), torch.zeros(d, device="cuda"), torch.zeros(d, device="cuda")
    xs = dev((x + r).detach())
    L.call("avlen_layernorm_bwd", L.ptr(dev(dy)), L.ptr(xs), L.ptr(dev(g.detach())), L.ptr(mean), L.ptr(rstd), L.ptr(dx),
           L.ptr(dg), L.ptr(db), rows, d, L.stream())
    torch.cuda.synchronize()
    assert rel_err(dx, x.grad) < 2e-5 and rel_err(dg, g.grad) < 2e-5 and rel_err(db, b.grad) < 2e-5


def _attn_ref(q, k, v, mask, causal, scale):
    s = (q @ k.transpose(-1, -2)) * scale                  # (B,H,Sq,Sk)
    if mask is not None:
        s = s.masked_fill(mask[:, None, None, :] == 0, float("-inf"))
    if causal:
        Sq, Sk = s.shape[-2:]
        s = s.masked_fill(torch.ones(Sq, Sk, dtype=torch.bool).triu(1), float("-inf"))
    return torch.softmax(s, -1) @ v


@pytest.mark.parametrize("B,H,Sq,Sk,D,causal,masked", [(3, 8, 301, 301, 32, 0, True), (2, 8, 77, 77, 64, 1, False),
                                                       (4, 8, 1, 301, 32, 0, True), (2, 8, 4, 4, 32, 0, True),
                                                       (2, 8, 130, 130, 32, 0, False)])
def test_attention_fwd_bwd(L, B, H, Sq, Sk, D, causal, masked):
    torch.manual_seed(6)
    d = H * D
    q = torch.randn(B, Sq, d).requires_grad_(True)
    k = torch.randn(B, Sk, d).requires_grad_(True)
    v = torch.randn(B, Sk, d).requires_grad_(True)
    mask = None
    if masked:
        mask = (torch.rand(B, Sk) < 0.6).float()
        mask[:, -1] = 1.0
        mask[0, :-1] = 0.0                                   # a sample whose memory is empty
    scale = 1.0 / math.sqrt(D)
    sp = lambda t, S: t.view(B, S, H, D).permute(0, 2, 1, 3)
    o_ref = _attn_ref(sp(q, Sq), sp(k, Sk), sp(v, Sk), mask, causal, scale).permute(0, 2, 1, 3).reshape(B, Sq, d)
    do = torch.randn(B, Sq, d)
    o_ref.backward(do)
    qd, kd, vd = dev(q.detach()), dev(k.detach()), dev(v.detach())
    o = torch.empty(B, Sq, d, device="cuda")
    lse = torch.empty(B, H, Sq, device="cuda")
    md = dev(mask) if masked else None
    L.call("avlen_attention_fwd", L.ptr(qd), d, L.ptr(kd), d, L.ptr(vd), d, L.ptr(o), d, L.ptr(md), L.ptr(lse), B, H, Sq,
           Sk, D, causal, scale, L.stream())
    torch.cuda.synchronize()
    assert rel_err(o, o_ref.detach()) < 2e-5
    dq, dk, dv = (torch.empty(B, S, d, device="cuda") for S in (Sq, Sk, Sk))
    delta = torch.empty(B, H, Sq, device="cuda")
    L.call("avlen_attention_bwd", L.ptr(qd), d, L.ptr(kd), d, L.ptr(vd), d, L.ptr(o), d, L.ptr(dev(do)), d, L.ptr(md),
           L.ptr(lse), L.ptr(delta), L.ptr(dq), d, L.ptr(dk), d, L.ptr(dv), d, B, H, Sq, Sk, D, causal, scale, L.stream())
    torch.cuda.synchronize()
    assert rel_err(dq, q.grad) < 5e-5 and rel_err(dk, k.grad) < 5e-5 and rel_err(dv, v.grad) < 5e-5
    if Sq == Sk and D == 32 and not causal:
        # the matrix-core backward (bf16 operands, fp32 accumulate; the 2nd-stage training path): bf16 rounding of q, k, v, dO,
        # P and dS -> 2e-2 of each gradient's largest element
        dq2, dk2, dv2 = (torch.full((B, S, d), float("nan"), device="cuda") for S in (Sq, Sk, Sk))
        delta2 = torch.empty(B, H, Sq, device="cuda")
        L.call("avlen_attention_bwd_bf16", L.ptr(qd), d, L.ptr(kd), d, L.ptr(vd), d, L.ptr(o), d, L.ptr(dev(do)), d, L.ptr(md),
               L.ptr(lse), L.ptr(delta2), L.ptr(dq2), d, L.ptr(dk2), d, L.ptr(dv2), d, B, H, Sq, Sk, D, causal, scale, L.stream())
        torch.cuda.synchronize()
        assert rel_err(delta2, delta) < 1e-5
        e = (rel_err(dq2, q.grad), rel_err(dk2, k.grad), rel_err(dv2, v.grad))
        assert max(e) < 2e-2, e


def test_preprocess_and_pack(L):
    torch.manual_seed(7)
    B = 2
    rgb = torch.randint(0, 256, (B, 128, 128, 3)).float()
    ref = F.interpolate((rgb.permute(0, 3, 1, 2) / 255.0), size=(64, 64), mode="area").permute(0, 2, 3, 1)
    y = torch.empty(B, 64, 64, 3, device="cuda")
    L.call("avlen_preprocess_image", L.ptr(dev(rgb)), 0, L.ptr(y), B, 128, 3, 255.0, L.stream())
    torch.cuda.synchronize()
    assert float((y.cpu() - ref).abs().max()) < 1e-6
    dep = torch.rand(B, 128, 128, 1)
    ref = F.interpolate(dep.permute(0, 3, 1, 2), size=(64, 64), mode="area").permute(0, 2, 3, 1)
    y = torch.empty(B, 64, 64, 1, device="cuda")
    L.call("avlen_preprocess_image", L.ptr(dev(dep)), 0, L.ptr(y), B, 128, 1, 1.0, L.stream())
    torch.cuda.synchronize()
    assert float((y.cpu() - ref).abs().max()) < 1e-6
    w = torch.randn(64, 128 * 64)
    wp = torch.empty(64, 64 * 128, device="cuda")
    L.call("avlen_pack_fc_after_flatten", L.ptr(dev(w)), L.ptr(wp), 64, 128, 64, L.stream())
    torch.cuda.synchronize()
    assert torch.equal(wp.cpu(), w.view(64, 128, 64).permute(0, 2, 1).reshape(64, -1))


def test_gae_adam_extmem(L):
    import fixtures as fx
    import restate as R
    from conftest import golden
    T, N = 150, 4
    r, v = fx.sym("gae.r", (T, N, 1)), fx.sym("gae.v", (T + 1, N, 1))
    m = torch.from_numpy((fx.unit("gae.m", (T + 1) * N) >= 1 / 15).astype("float32")).view(T + 1, N, 1)
    vd, ret, adv = dev(v), torch.zeros(T + 1, N, 1, device="cuda"), torch.zeros(T, N, 1, device="cuda")
    L.call("avlen_gae_scan", L.ptr(dev(r)), L.ptr(vd), L.ptr(dev(m)), L.ptr(dev(fx.sym("gae.nv", (N, 1)))), L.ptr(ret),
           L.ptr(adv), T, N, 0.99, 0.95, L.stream())
    torch.cuda.synchronize()
    np.testing.assert_allclose(ret.cpu().numpy()[:T], golden("gae")["returns"][:T], rtol=1e-5, atol=1e-6)
    # external-memory ring vs the reference's ExternalMemory.insert trace
    g = golden("extmem")
    mem, masks = torch.zeros(8, 3, 5, device="cuda"), torch.zeros(3, 8, device="cuda")
    idx = 0
    for t in range(20):
        nd = torch.from_numpy((fx.unit(f"em.nd{t}", 3) >= 0.12).astype("float32")).view(3, 1)
        snap = torch.empty(3, 8, device="cuda")
        L.call("avlen_extmem_insert", L.ptr(mem), L.ptr(masks), L.ptr(dev(fx.sym(f"em.f{t}", (3, 5)))), 5, L.ptr(dev(nd)),
               L.ptr(snap), idx, 8, 4, 3, 5, L.stream())
        idx = (idx + 1) % 8
        torch.cuda.synchronize()
        assert np.array_equal(snap.cpu().numpy(), g["masks"][t])
    assert np.array_equal(mem.cpu().numpy(), g["memory"])
    # clip-norm + Adam, two steps, vs the oracle's restatement of torch.optim.Adam
    n = 5000
    p0, g1, g2 = fx.sym("ad.p", (n,)), fx.sym("ad.g1", (n,), 0.05), fx.sym("ad.g2", (n,), 0.05)
    p, mm, vv = p0.clone(), torch.zeros(n), torch.zeros(n)
    pd, md_, vd_ = dev(p0), torch.zeros(n, device="cuda"), torch.zeros(n, device="cuda")
    for step, gg in enumerate((g1, g2), 1):
        (gc,), _ = R.clip_grad_norm([gg], 0.2)
        R.adam_step(p, gc, mm, vv, step, 2.5e-4, 1e-5)
        ns = torch.zeros(1, dtype=torch.float64, device="cuda")
        gd = dev(gg)
        L.call("avlen_grad_sumsq", L.ptr(gd), n, L.ptr(ns), L.stream())
        L.call("avlen_adam_step", L.ptr(pd), L.ptr(gd), L.ptr(md_), L.ptr(vd_), n, 2.5e-4, 0.9, 0.999, 1e-5, step, 0.2,
               L.ptr(ns), L.stream())
    torch.cuda.synchronize()
    assert rel_err(pd, p) < 1e-6 and rel_err(md_, mm) < 1e-5


# ------------------------------------------------------------------ bf16 fast path (igemm2.hip)
@pytest.mark.parametrize("M,N,K,act", [(4928, 2048, 512, 2), (4928, 512, 2048, 0), (300, 256, 320, 1), (64, 64, 8192, 0),
                                       (130, 21, 72, 0), (19264, 768, 256, 0), (257, 130, 200, 0),
                                       (16384, 4096, 128, 0), (2464, 512, 2048, 0), (8192, 1024, 192, 1)])
def test_gemm_bf16_fast_path(L, M, N, K, act):
    torch.manual_seed(10)
    A, W = torch.randn(M, K), torch.randn(N, K) / math.sqrt(K)
    b, r = torch.randn(N), torch.randn(M, N)
    A16, W16 = dev(A.bfloat16()), dev(W.bfloat16())
    ref = A16.float().cpu() @ W16.float().cpu().t() + b
    ref = torch.relu(ref) if act == 1 else (ref * torch.sigmoid(1.702 * ref) if act == 2 else ref)
    ref = ref + r
    C32 = torch.empty(M, N, device="cuda"); C16 = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    nb = L.lib.avlen_gemm_bf16_workspace_bytes(M, N); ws = torch.empty(nb, dtype=torch.uint8, device="cuda")
    L.call("avlen_gemm_bf16", L.ptr(A16), K, L.ptr(W16), K, L.ptr(C32), N, L.ptr(C16), N, L.ptr(dev(b)), L.ptr(dev(r)), N, M,
           N, K, act, L.ptr(ws), nb, L.stream())
    torch.cuda.synchronize()
    assert rel_err(C32, ref) < 1e-5                 # same bf16 operands, fp32 accumulate
    assert rel_err(C16.float(), ref) < 1e-2


@pytest.mark.parametrize("cfg", [(2, 64, 3, 16, 7, 1, 3), (2, 64, 16, 16, 3, 1, 1), (3, 64, 16, 32, 3, 2, 1),
                                 (2, 32, 16, 32, 1, 2, 0), (64, 8, 128, 128, 3, 1, 1), (5, 16, 64, 128, 3, 2, 1)])
def test_conv_bf16_with_fused_groupnorm_stats(L, cfg):
    """bf16 implicit-GEMM conv (zero-padded channels) + GroupNorm statistics from its epilogue + bf16 GN apply,
    against conv2d + group_norm in fp32 on the bf16-rounded operands."""
    B, H, Cin, Cout, k, s, p = cfg
    torch.manual_seed(11)
    x = torch.randn(B, H, H, Cin); w = torch.randn(Cout, Cin, k, k) / math.sqrt(Cin * k * k)
    g, be = torch.rand(Cout) + 0.5, torch.randn(Cout)
    Cp = max(8, Cin)
    x16 = torch.zeros(B, H, H, Cp, dtype=torch.bfloat16); x16[..., :Cin] = x.bfloat16()
    OH = (H + 2 * p - k) // s + 1
    res = torch.randn(B, OH, OH, Cout).bfloat16()
    raw_ref = F.conv2d(x.bfloat16().float().permute(0, 3, 1, 2), w.bfloat16().float(), None, stride=s, padding=p)
    y_ref = torch.relu(F.group_norm(raw_ref, 16, g, be, 1e-5).permute(0, 2, 3, 1) + res.float())
    wp16 = torch.empty(Cout, k, k, Cp, device="cuda", dtype=torch.bfloat16)
    L.call("avlen_pack_conv_weight_bf16", L.ptr(dev(w)), L.ptr(wp16), Cout, Cin, k, k, Cp, L.stream())
    raw = torch.empty(B, OH, OH, Cout, device="cuda"); stats = torch.zeros(B, 2, Cout, device="cuda")
    nb = L.lib.avlen_gemm_bf16_workspace_bytes(B * OH * OH, Cout); ws = torch.empty(nb, dtype=torch.uint8, device="cuda")
    L.call("avlen_conv2d_nhwc_bf16", L.ptr(dev(x16)), L.ptr(wp16), None, None, L.ptr(raw), None, L.ptr(stats), B, H, H, Cp,
           Cout, k, k, s, p, 0, L.ptr(ws), nb, L.stream())
    torch.cuda.synchronize()
    assert rel_err(raw, raw_ref.permute(0, 2, 3, 1)) < 1e-5
    sums = raw_ref.sum(dim=(2, 3)); sq = (raw_ref ** 2).sum(dim=(2, 3))
    assert rel_err(stats[:, 0], sums) < 1e-4 and rel_err(stats[:, 1], sq) < 1e-4
    # bf16 GN-apply through the module-level tower is covered by the policy tests; here check the stats only


@pytest.mark.parametrize("cfg", [(3, 64, 16, 16, 3), (2, 32, 32, 32, 3), (3, 64, 3, 16, 7), (2, 64, 1, 16, 7)])
def test_direct_conv_bf16(L, cfg):
    """Direct (LDS-halo) convolution of the small-channel tower stages + fused GroupNorm statistics vs conv2d."""
    B, W, Cin, Cout, k = cfg
    torch.manual_seed(12)
    x = torch.randn(B, W, W, Cin); w = torch.randn(Cout, Cin, k, k) / math.sqrt(Cin * k * k)
    Cp = max(8, Cin)
    x16 = torch.zeros(B, W, W, Cp, dtype=torch.bfloat16); x16[..., :Cin] = x.bfloat16()
    ref = F.conv2d(x.bfloat16().float().permute(0, 3, 1, 2), w.bfloat16().float(), None, stride=1, padding=k // 2)
    wp16 = torch.empty(Cout, k, k, Cp, device="cuda", dtype=torch.bfloat16)
    L.call("avlen_pack_conv_weight_bf16", L.ptr(dev(w)), L.ptr(wp16), Cout, Cin, k, k, Cp, L.stream())
    y = torch.empty(B, W, W, Cout, device="cuda", dtype=torch.bfloat16); stats = torch.zeros(B, 2, Cout, device="cuda")
    L.call("avlen_conv_direct_bf16", L.ptr(dev(x16)), L.ptr(wp16), L.ptr(y), L.ptr(stats), B, W, Cp, Cout, k, L.stream())
    torch.cuda.synchronize()
    assert rel_err(y.float(), ref.permute(0, 2, 3, 1)) < 8e-3          # bf16 output rounding
    assert rel_err(stats[:, 0], ref.sum(dim=(2, 3))) < 1e-4 and rel_err(stats[:, 1], (ref ** 2).sum(dim=(2, 3))) < 1e-4


def test_multi_copy(L):
    """Batched device-to-device copies (storage inserts): mixed sizes / alignments / dtypes, > 32 pairs."""
    torch.manual_seed(13)
    srcs, dsts = [], []
    for i, n in enumerate([1, 3, 4, 64, 1000, 4099, 12 * 128 * 128 * 3] + [17 + 5 * j for j in range(40)]):
        if i % 3 == 0:
            s_ = torch.randint(0, 1 << 40, (n,), dtype=torch.int64)
        elif i % 3 == 1:
            s_ = torch.randn(n)
        else:
            s_ = torch.randint(0, 255, (n,), dtype=torch.uint8)
        srcs.append(dev(s_)); dsts.append(torch.zeros_like(srcs[-1]))
    base = torch.zeros(101, device="cuda")               # a misaligned float view
    srcs.append(dev(torch.randn(100))); dsts.append(base[1:])
    L.multi_copy(list(zip(dsts, srcs)))
    torch.cuda.synchronize()
    for d, s_ in zip(dsts, srcs):
        assert torch.equal(d, s_)


@pytest.mark.parametrize("M,N1,N2,lda,ldb,beta", [(1000, 256, 256, 256, 256, 0.0), (37, 256, 64, 256, 64, 0.0), (5, 21, 9, 24, 16, 1.0),
                                                  (70001, 768, 268, 768, 272, 1.0), (4096, 512, 256, 520, 256, 0.0),
                                                  (300 * 77, 300, 520, 304, 520, 0.5)])
def test_gemm_tn_rows_contracted(L, M, N1, N2, lda, ldb, beta):
    """avlen_gemm_tn_bf16: C = beta C + A^T B over the rows of two row-major bf16 operands (the weight gradient of the 2nd-stage
    update) against the fp64 product of the same bf16 values; twice the same bits (fixed-order partial sums)."""
    torch.manual_seed(3)
    A = torch.zeros(M, lda); A[:, :N1] = torch.randn(M, N1)
    B = torch.zeros(M, ldb); B[:, :N2] = torch.randn(M, N2)
    A16, B16 = dev(A.bfloat16()), dev(B.bfloat16())
    C0 = torch.randn(N1, N2)
    ref = beta * C0.double() + A16.cpu().double()[:, :N1].t() @ B16.cpu().double()[:, :N2]
    nb = L.lib.avlen_gemm_tn_bf16_workspace_bytes(M, N1, N2)
    ws = torch.empty(nb, dtype=torch.uint8, device="cuda")
    outs = []
    for _ in range(2):
        out = dev(C0.clone())
        L.call("avlen_gemm_tn_bf16", L.ptr(A16), lda, L.ptr(B16), ldb, M, N1, N2, L.ptr(out), N2, beta, L.ptr(ws), nb, L.stream())
        torch.cuda.synchronize()
        outs.append(out.cpu())
    assert torch.equal(outs[0], outs[1])
    scale = float(ref.abs().max())
    assert float((outs[0].double() - ref).abs().max()) < 2e-5 * scale + 1e-4 * math.sqrt(M) * 1e-2, (float((outs[0].double() - ref).abs().max()), scale)


@pytest.mark.parametrize("B,S,planes", [(5, 301, 2), (3, 17, 1), (9, 320, 2)])
def test_single_query_cross_attention_in_memory_space(L, B, S, planes):
    """csrc/cross1.hip (the decoder's cross attention of the 2nd-stage training path): expand -> fwd -> reduce and its backward
    (dw, expand, bwd, reduce, dw) against torch autograd (fp64) of the textbook formulation -- K | V = mem W_kv^T + b_kv, one query
    per (sample, head), key-padding mask -- on the same memory rows (the bf16 hi (+ lo) planes, added back)."""
    torch.manual_seed(7)
    d, H, D = 256, 8, 32
    scale = 1.0 / math.sqrt(D)
    mem = torch.randn(B * S, d)
    hi = mem.bfloat16()
    lo = (mem - hi.float()).bfloat16()
    memv = (hi.float() + (lo.float() if planes == 2 else 0)).double().view(B, S, d).requires_grad_(True)
    q = torch.randn(B, d, dtype=torch.float64, requires_grad=True)
    Wk = (torch.randn(d, d, dtype=torch.float64) / 16).requires_grad_(True)
    Wv = (torch.randn(d, d, dtype=torch.float64) / 16).requires_grad_(True)
    bk = torch.randn(d, dtype=torch.float64, requires_grad=True)
    bv = torch.randn(d, dtype=torch.float64, requires_grad=True)
    mask = (torch.rand(B, S) > 0.3).float()
    mask[:, -1] = 1.0                                        # the current token is always valid
    K = memv @ Wk.t() + bk
    V = memv @ Wv.t() + bv
    sc = scale * torch.einsum("bhd,bshd->bhs", q.view(B, H, D), K.view(B, S, H, D))
    sc = sc.masked_fill(mask.double()[:, None, :] == 0, float("-inf"))
    p = torch.softmax(sc, -1)
    out = torch.einsum("bhs,bshd->bhd", p, V.view(B, S, H, D)).reshape(B, d)
    dout = torch.randn(B, d, dtype=torch.float64)
    out.backward(dout)
    planes16 = dev(torch.cat([hi, lo], 0) if planes == 2 else hi)
    lo_off = B * S * d if planes == 2 else 0
    f = lambda t: dev(t.detach().float())
    qd, Wkd, Wvd, bvd, md, dod = f(q), f(Wk), f(Wv), f(bv), dev(mask), f(dout)
    z = lambda *s: torch.zeros(*s, device="cuda")
    A, P, Mo, o, DM, dA, dMEM, dq, dWk, dWv = z(B, H, d), z(B, H, S), z(B, H, d), z(B, d), z(B, H, d), z(B, H, d), z(B * S, d), z(B, d), z(d, d), z(d, d)
    st = L.stream()
    L.call("avlen_cross1_expand", L.ptr(qd), d, L.ptr(Wkd), d, L.ptr(A), B, st)
    L.call("avlen_cross1_fwd", L.ptr(A), L.ptr(planes16), lo_off, L.ptr(md), L.ptr(P), L.ptr(Mo), B, S, scale, st)
    L.call("avlen_cross1_reduce", L.ptr(Mo), L.ptr(Wvd), d, L.ptr(bvd), L.ptr(o), d, B, st)
    L.call("avlen_cross1_dw", L.ptr(dod), d, L.ptr(Mo), L.ptr(dWv), d, B, st)
    L.call("avlen_cross1_expand", L.ptr(dod), d, L.ptr(Wvd), d, L.ptr(DM), B, st)
    L.call("avlen_cross1_bwd", L.ptr(P), L.ptr(DM), L.ptr(A), L.ptr(planes16), lo_off, L.ptr(dA), L.ptr(dMEM), B, S, scale, st)
    L.call("avlen_cross1_reduce", L.ptr(dA), L.ptr(Wkd), d, None, L.ptr(dq), d, B, st)
    L.call("avlen_cross1_dw", L.ptr(qd), d, L.ptr(dA), L.ptr(dWk), d, B, st)
    torch.cuda.synchronize()
    assert rel_err(P, p.detach()) < 2e-5 and rel_err(o, out.detach()) < 2e-5
    assert rel_err(dq, q.grad) < 5e-5 and rel_err(dWk, Wk.grad) < 5e-5 and rel_err(dWv, Wv.grad) < 5e-5
    assert rel_err(dMEM, memv.grad.reshape(B * S, d)) < 5e-5
    assert float(bk.grad.abs().max()) < 1e-9 * max(1.0, float(Wk.grad.abs().max()))   # the shift invariance the kernel relies on (db_k = 0)
    assert rel_err(dod.sum(0), bv.grad) < 1e-5                                        # db_v = column sums of dout
