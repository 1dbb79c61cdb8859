"""The fused visual-tower kernels (tower_head: preprocessing + stem + layers 1-2; tower_tail: layers 3-4; grouped fc) through the C
ABI (`avlen_resnet18_group_fwd[_indexed]`, bf16) against the oracle's fp32 restatement of SMTCNN (smt_cnn.py:78-115) on the fixture
weights: every sensor size the preprocessing handles (64: identity, 128: the 2x2 fast path, 256: 4x4 generic path), uint8 and fp32
rgb, determinism (the kernels reduce their GroupNorm statistics in a fixed order), and the row-indexed minibatch form."""
import ctypes as C
import pytest
import torch

import fixtures as fx
import restate as R
from conftest import param_specs
from avlen_amd import policy as P, _lib as L, engine as E
from avlen_amd.spaces import savi_observation_space, ActionSpace, SMT_KW

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def towers():
    specs = param_specs()
    pol = P.AudioNavOptionPolicy(savi_observation_space((65, 26, 2)), ActionSpace(4), pretraining=True, precision="bf16",
                                 use_category_input=False, query_count_emb_size=32, **SMT_KW)
    sd = fx.state_dict_for({k: tuple(v) for k, v in specs["option"].items()})
    pol.load_state_dict(sd, strict=False)
    pol.cuda()
    return pol, sd


def run_group(pol, rgb, depth, index=None, rows=None):
    eng = pol._engine()
    B = rows if rows is not None else rgb.shape[0]
    out = torch.empty(B, 128, device="cuda")
    nets = (C.POINTER(L.ResNet18) * 2)(C.pointer(eng["rgb"]), C.pointer(eng["depth"]))
    imgs = (C.c_void_p * 2)(rgb.data_ptr(), depth.data_ptr())
    outs = (C.c_void_p * 2)(out.data_ptr(), out.data_ptr() + 4 * 64)
    chans, divs = (C.c_int * 2)(rgb.shape[3], depth.shape[3]), (C.c_float * 2)(255.0, 1.0)
    u8 = (C.c_int * 2)(int(rgb.dtype == torch.uint8), 0)
    nb = L.lib.avlen_resnet18_group_workspace_bytes(2, B)
    ws = torch.empty(nb, dtype=torch.uint8, device="cuda")
    if index is None:
        L.call("avlen_resnet18_group_fwd", nets, imgs, u8, chans, divs, outs, 128, 2, B, rgb.shape[1], E.P(ws), nb, L.stream())
    else:
        L.call("avlen_resnet18_group_fwd_indexed", nets, imgs, u8, chans, divs, outs, 128, 2, B, rgb.shape[1], E.P(index), E.P(ws), nb,
               L.stream())
    torch.cuda.synchronize()
    return out


@pytest.mark.parametrize("S", [64, 128, 256])
@pytest.mark.parametrize("u8", [True, False])
def test_fused_towers_match_oracle(towers, S, u8):
    pol, sd = towers
    g = torch.Generator().manual_seed(S + int(u8))
    B = 5
    # images with spatial structure (a coarse random field, upsampled, + pixel noise): white noise alone loses its contrast in
    # the k x k block mean and the per-channel GroupNorm then amplifies the bf16 rounding of a nearly constant image
    field = lambda c: torch.nn.functional.interpolate(torch.rand(B, c, 8, 8, generator=g), size=(S, S), mode="bilinear", align_corners=False)
    noisy = lambda c: (0.8 * field(c) + 0.2 * torch.rand(B, c, S, S, generator=g)).permute(0, 2, 3, 1).contiguous()
    rgb8 = (noisy(3) * 255).round().clamp(0, 255).to(torch.uint8)
    depth = noisy(1)
    ref = R.smt_cnn(sd, "net.visual_encoder", {"rgb": rgb8.float(), "depth": depth})
    out = run_group(pol, rgb8.cuda() if u8 else rgb8.float().cuda(), depth.cuda())
    d = out.cpu() - ref
    err, rms = float(d.abs().max()) / float(ref.abs().max()), float(d.pow(2).mean().sqrt() / ref.pow(2).mean().sqrt())
    # bf16 operands through 20 convs + 20 GroupNorms: 2.7 % rms measured on the benched batch
    assert rms < 4e-2 and err < 6e-2, (S, u8, rms, err)
    again = run_group(pol, rgb8.cuda() if u8 else rgb8.float().cuda(), depth.cuda())
    assert torch.equal(out, again)           # fixed-order statistics: run-to-run identical


def test_indexed_rows_equal_gathered_rows(towers):
    """The PPO minibatch form: rows `index` of a larger observation buffer read in place == the same rows gathered first."""
    pol, _ = towers
    g = torch.Generator().manual_seed(5)
    rgb = torch.randint(0, 256, (12, 128, 128, 3), generator=g, dtype=torch.uint8).cuda()
    depth = torch.rand(12, 128, 128, 1, generator=g).cuda()
    idx = torch.tensor([7, 0, 11, 3, 3, 9], dtype=torch.int32, device="cuda")
    a = run_group(pol, rgb, depth, index=idx, rows=6)
    b = run_group(pol, rgb[idx.long()].contiguous(), depth[idx.long()].contiguous())
    assert torch.equal(a, b)


def test_fragment_order_weight_copy():
    """avlen_pack_conv_weight_frag: w16 [cout][K] -> [cout/16][K/32][lane = 16 q + r][8] with lane's chunk = w16[16 t + r][32 i + 8 q ..]."""
    cout, K = 64, 576
    w = torch.randn(cout, K, device="cuda").bfloat16()
    f = torch.empty_like(w)
    L.call("avlen_pack_conv_weight_frag", E.P(w), E.P(f), cout, K, L.stream())
    torch.cuda.synchronize()
    ref = w.view(cout // 16, 16, K // 32, 4, 8).permute(0, 2, 3, 1, 4).contiguous()       # [t][i][q][r][8]
    assert torch.equal(f.view(-1), ref.view(-1))
    with pytest.raises(L.AvlenHipError):
        L.call("avlen_pack_conv_weight_frag", E.P(w), E.P(f), 60, K, L.stream())            # cout % 16 != 0


# ---- the benched mode: compensated bf16 towers (tower_x3.hip: one persistent launch, stem + layer 1 / layers 2-4 work items) ----
@pytest.fixture(scope="module")
def towers_x3():
    specs = param_specs()
    pol = P.AudioNavOptionPolicy(savi_observation_space((65, 26, 2)), ActionSpace(4), pretraining=True, precision="bf16x3",
                                 use_category_input=False, query_count_emb_size=32, **SMT_KW)
    sd = fx.state_dict_for({k: tuple(v) for k, v in specs["option"].items()})
    pol.load_state_dict(sd, strict=False)
    pol.cuda()
    return pol, sd


def run_group_x3(pol, rgb, depth, index=None, rows=None):
    eng = pol._engine()
    B = rows if rows is not None else rgb.shape[0]
    out = torch.full((B, 128), float("nan"), device="cuda")
    nets = (C.POINTER(L.ResNet18) * 2)(C.pointer(eng["rgb"]), C.pointer(eng["depth"]))
    imgs = (C.c_void_p * 2)(rgb.data_ptr(), depth.data_ptr())
    outs = (C.c_void_p * 2)(out.data_ptr(), out.data_ptr() + 4 * 64)
    chans, divs = (C.c_int * 2)(rgb.shape[3], depth.shape[3]), (C.c_float * 2)(255.0, 1.0)
    u8 = (C.c_int * 2)(int(rgb.dtype == torch.uint8), 0)
    nb = L.lib.avlen_resnet18_group_x3_workspace_bytes(2, B)
    ws = torch.empty(nb, dtype=torch.uint8, device="cuda")
    L.call("avlen_resnet18_group_fwd_x3", nets, imgs, u8, chans, divs, outs, 128, 2, B, rgb.shape[1],
           E.P(index) if index is not None else None, E.P(ws), nb, L.stream())
    torch.cuda.synchronize()
    return out


@pytest.mark.parametrize("S", [64, 128, 256])
@pytest.mark.parametrize("u8", [True, False])
def test_x3_towers_match_oracle_on_white_noise(towers_x3, S, u8):
    """`avlen_resnet18_group_fwd_x3` against the oracle's fp32 SMTCNN (smt_cnn.py:78-115) on WHITE-NOISE images -- what bench.py
    feeds (uniform uint8 rgb, U[0,1) depth): the hardest input for the per-channel GroupNorm after the block mean.  Tolerance: 1e-3
    of the feature scale (the north-star's), plain and row-indexed forms, bit-reproducible."""
    pol, sd = towers_x3
    g = torch.Generator().manual_seed(100 + S + int(u8))
    B = 7
    rgb8 = torch.randint(0, 256, (B, S, S, 3), generator=g, dtype=torch.uint8)
    depth = torch.rand(B, S, S, 1, generator=g)
    ref = R.smt_cnn(sd, "net.visual_encoder", {"rgb": rgb8.float(), "depth": depth})
    rgb = rgb8.cuda() if u8 else rgb8.float().cuda()
    out = run_group_x3(pol, rgb, depth.cuda())
    assert torch.isfinite(out).all()
    scale = float(ref.abs().max())
    err = float((out.cpu() - ref).abs().max())
    print(f"x3 towers S={S} u8={u8}: max |d| {err:.3e} on scale {scale:.3f}")
    assert err <= 1e-3 * max(1.0, scale), (S, u8, err, scale)
    assert torch.equal(out, run_group_x3(pol, rgb, depth.cuda()))                       # fixed-order statistics
    # the PPO minibatch form: rows `idx` of the buffers read in place == the same rows gathered first
    idx = torch.tensor([5, 0, 6, 2, 2], dtype=torch.int32, device="cuda")
    a = run_group_x3(pol, rgb, depth.cuda(), index=idx, rows=5)
    assert torch.equal(a, out[idx.long()])
