"""The one-launch CLIP text tower (csrc/clip_tower.hip; AVLEN_CLIP_STREAM=0 selects the launch-per-GEMM tower: sequence-stationary
workgroups, per-wave weight streams, two column halves per dialog, two row halves with a K / V hand-off for dialogs of 5 row tiles) against the launch-per-GEMM tower and the fp32
path: same 16-bit formats, so both fast towers must sit at the same distance from fp32; dialog lengths cover every tile count
(1 .. 5 tiles, i.e. every kernel instance incl. the split one) and the extremes (EOT at position 1 and at 76)."""
import os

import pytest
import torch

from avlen_amd import policy as P
from avlen_amd.spaces import savi_observation_space, ActionSpace, SMT_KW

pytestmark = pytest.mark.gpu


def _tokens(n, gen):
    toks = torch.zeros(n, 77, dtype=torch.long)
    lens = [1, 76, 15, 16, 17, 31, 32, 33, 47, 48, 49, 63, 64, 65, 72] + [int(x) for x in torch.randint(2, 76, (n - 15,), generator=gen)]
    for b, ln in enumerate(lens[:n]):
        toks[b, :ln] = torch.randint(1, 49406, (ln,), generator=gen)
        toks[b, 0] = 49406
        toks[b, ln] = 49407
    return toks.cuda()


def _policy(mode, stream, sd=None):
    from avlen_amd import config as CFG
    keep, CFG.CLIP_STREAM = CFG.CLIP_STREAM, bool(stream)      # (AVLEN_CLIP_STREAM: read when the engine builds the tower's views)
    try:
        torch.manual_seed(3)
        pol = P.AudioNavDialogPolicy(savi_observation_space((65, 26, 2)), ActionSpace(4), pretraining=False, num_steps=3,
                                     precision=mode, **SMT_KW).to("cuda")
        if sd is not None:
            pol.load_state_dict(sd)
        pol._engine()
    finally:
        CFG.CLIP_STREAM = keep
    return pol


@pytest.mark.parametrize("mode", ["bf16x3", "bf16"])
def test_stream_tower_matches_the_gemm_chain_tower(mode):
    gen = torch.Generator().manual_seed(11)
    tok = _tokens(24, gen)
    ref_pol = _policy("fp32", False)
    sd = ref_pol.state_dict()
    ref = ref_pol.net.encode_text(ref_pol, tok).clone()
    chain_pol, stream_pol = _policy(mode, False, sd), _policy(mode, True, sd)
    assert not chain_pol._engine()["clip"].wstream and stream_pol._engine()["clip"].wstream
    chain = chain_pol.net.encode_text(chain_pol, tok).clone()
    out = stream_pol.net.encode_text(stream_pol, tok).clone()
    again = stream_pol.net.encode_text(stream_pol, tok).clone()
    torch.cuda.synchronize()
    assert torch.isfinite(out).all()
    assert torch.equal(out, again)                                  # fixed-order reductions: bit-reproducible
    e_chain, e_stream = float((chain - ref).abs().max()), float((out - ref).abs().max())
    scale = float(ref.abs().max())
    print(f"{mode}: |chain - fp32| {e_chain:.3e}  |stream - fp32| {e_stream:.3e}  |stream - chain| {float((out - chain).abs().max()):.3e}  (max |ref| {scale:.3f})")
    tol = (1e-2 if mode == "bf16x3" else 6e-2) * max(1.0, scale)     # fp16 / bf16 operands through 12 blocks
    assert e_stream < tol and e_stream < 2.0 * e_chain + 1e-3


@pytest.mark.parametrize("mode", ["bf16x3", "bf16"])
def test_folded_dialog_layer_matches_the_two_step_form_and_follows_weight_updates(mode):
    """The rollout's text graph folds text_projection into dialog_layer (policy.py:847-849: dialog_layer(encode_text(x)), nothing
    in between): folded vs two-step on the same policy, against the fp32 policy, and again after dialog_layer moved (an optimiser
    step marks the parameters changed: the folded weight is derived data and must follow)."""
    gen = torch.Generator().manual_seed(5)
    tok = _tokens(20, gen)
    ref_pol = _policy("fp32", False)
    pol = _policy(mode, True, ref_pol.state_dict())
    assert "dialog_fold" in pol._engine()
    for rnd in range(2):
        ref = ref_pol.net._dialog_embed(ref_pol, ref_pol.net.encode_text(ref_pol, tok)).clone()
        two = pol.net._dialog_embed(pol, pol.net.encode_text(pol, tok)).clone()
        fold = pol.net._text_to_dialog(pol, tok).clone()
        torch.cuda.synchronize()
        scale = max(1.0, float(ref.abs().max()))
        e_two, e_fold = float((two - ref).abs().max()), float((fold - ref).abs().max())
        print(f"{mode} round {rnd}: |two-step - fp32| {e_two:.3e}  |folded - fp32| {e_fold:.3e}  (max |ref| {scale:.3f})")
        assert e_fold < (1e-2 if mode == "bf16x3" else 6e-2) * scale and e_fold < 2.0 * e_two + 2e-3
        with torch.no_grad():                              # "training": dialog_layer moves in both policies
            for p_ in (ref_pol, pol):
                p_.net.dialog_layer.weight.mul_(0.5).add_(0.01)
                p_.net.dialog_layer.bias.add_(0.1)
                p_.mark_params_changed()


@pytest.mark.parametrize("mode", ["bf16x3", "bf16"])
@pytest.mark.parametrize("B", [600, 1100])
def test_encode_text_above_one_pass(mode, B):
    """PPO.update_dialog evaluates the frozen tower on T * N rows (ppo.py:99-154; policy.py:847-849): batches above the one-launch
    tower's 512 dialogs run as consecutive passes -- the same rows as separate <= 512-row calls, bit for bit."""
    gen = torch.Generator().manual_seed(21)
    tok = _tokens(B, gen)
    pol = _policy(mode, True)
    out = pol.net.encode_text(pol, tok).clone()
    parts = torch.cat([pol.net.encode_text(pol, tok[i:i + 512].contiguous()).clone() for i in range(0, B, 512)])
    torch.cuda.synchronize()
    assert out.shape == (B, 512) and torch.isfinite(out).all()
    assert torch.equal(out, parts)
    ref_pol = _policy("fp32", False, pol.state_dict())
    ref = ref_pol.net.encode_text(ref_pol, tok[-40:].contiguous()).clone()        # the last rows belong to the last pass
    torch.cuda.synchronize()
    err = float((out[-40:] - ref).abs().max())
    assert err < (1e-2 if mode == "bf16x3" else 6e-2) * max(1.0, float(ref.abs().max())), err


@pytest.mark.parametrize("mode", ["bf16x3", "bf16"])
def test_embedding_does_not_depend_on_the_batch_it_is_packed_in(mode):
    """The work list packs whole dialogs into groups of <= 4 row tiles ([4] | [3+1] | [2+2] | [2+1+1] | [1+1+1+1] | 5-tile halves);
    a tile attends to its own dialog's keys only, the column split is fixed: a dialog alone in a call, in a reversed batch, or
    among other tile classes must produce the same bits (the per-row memo recomputes single rows and relies on it)."""
    gen = torch.Generator().manual_seed(12)
    tok = _tokens(24, gen)
    pol = _policy(mode, True)
    # ln_final(tower output), without the text projection: that GEMM's summation order legitimately depends on the row count
    enc = lambda t: pol.net.encode_text(pol, t, project=False)
    full = enc(tok).clone()
    rev = enc(tok.flip(0).contiguous()).clone().flip(0)
    assert torch.equal(full, rev)
    for b in (0, 1, 2, 5, 8, 11, 13, 14, 23):                       # 1 .. 5 tiles, alone
        one = enc(tok[b:b + 1].contiguous()).clone()
        assert torch.equal(one[0], full[b]), b
    ones = [b for b in range(24) if int((tok[b] == 49407).nonzero()[0]) < 16]                 # only 1-tile dialogs: [1+1+1+1] groups
    sub = enc(tok[ones].contiguous()).clone()
    assert torch.equal(sub, full[ones])
    for pick in ([4, 0, 2], [4, 5, 6, 0, 2], [6, 7, 0]):              # [2+1+1] | [3+1] [2+2] [1] | [3+1] [3]
        assert torch.equal(enc(tok[pick].contiguous()).clone(), full[pick]), pick


def test_benched_fp16_split_tower_against_the_oracle():
    """The tower the bench runs (precision="bf16x3": fp16 operands, packed groups, 4-way column split, one launch) directly against
    the ORACLE's CLIP text transformer (oracle/restate.py clip_encode_text: the published definition, equal to Hugging Face's
    CLIPTextModelWithProjection to 2e-5, tests/test_clip_independent.py) on every tile class (1 .. 5 row tiles, EOT at 1 and 76):
    <= 5e-3 of the embedding scale (measured 1.0e-3: 3.7e-3 on a scale of 3.6).  And what that error does to pi_l: with the
    reference-golden weights of `policy_dlg` -- whose heads are He-scaled, logits O(1), not the reference's gain-0.01 initialisation
    -- the action logits computed from the HIP tower's embedding and from the oracle's embedding differ by <= 1e-3 of the logit
    scale (measured 1.9e-3 on logits up to 2.35, i.e. 8e-4)."""
    import numpy as np
    import fixtures as fx
    import restate as R
    from conftest import param_specs
    gen = torch.Generator().manual_seed(11)
    tok = _tokens(24, gen)
    pol = _policy("bf16x3", True)
    assert pol._engine()["clip"].wstream and pol.prec_of("clip") == P.L.PREC_FP16
    sd = {k: v.detach().cpu().clone() for k, v in pol.state_dict().items() if k.startswith("net.clip.")}
    with torch.no_grad():
        ref = R.clip_encode_text(sd, "net.clip", tok.cpu())
    out = pol.net.encode_text(pol, tok).clone()
    noproj = pol.net.encode_text(pol, tok, project=False).clone()
    torch.cuda.synchronize()
    scale = max(1.0, float(ref.abs().max()))
    err = float((out.cpu() - ref).abs().max())
    print(f"fp16 4-way tower vs oracle: max |diff| {err:.3e} on scale {scale:.3f} ({err / scale:.3e} of scale); rows {tuple(out.shape)}")
    assert torch.isfinite(out).all() and torch.isfinite(noproj).all()
    assert err < 5e-3 * scale, err
    # --- induced error on pi_l's logits, reference-golden weights (tests/golden/policy_dlg) ---
    specs = param_specs()
    B, M, tag = 3, 3, "dlg"
    torch.manual_seed(3)                                         # the CLIP tower keeps its seeded initialisation (no reference fixture holds it)
    pl = P.AudioNavDialogPolicy(savi_observation_space((65, 26, 2)), ActionSpace(4), pretraining=False, num_steps=3,
                                precision="bf16x3", **SMT_KW)
    fsd = fx.state_dict_for({k: tuple(v) for k, v in specs["dialog"].items()})
    assert not pl.load_state_dict(fsd, strict=False).unexpected_keys
    pl.cuda()
    obs = {k: v.cuda() for k, v in fx.observations(tag, B).items()}
    mem, memd = fx.memory(tag, M, B, 276, 272).cuda(), fx.sym(tag + ".memd", (M, B, 256)).cuda()
    mk = fx.mask_patterns(tag, B, M).cuda()
    pa, act = fx.ints(tag + ".pa", (B, 1), 4).cuda(), fx.ints(tag + ".a", (B, 1), 4).cuda()
    toks = fx.dialog_tokens(tag, B).cuda()
    astep = fx.ints(tag + ".as", (B,), 3).float().cuda()
    h0, ones = torch.zeros(1, B, 512, device="cuda"), torch.ones(B, 1, device="cuda")
    hip = pl.evaluate_actions_dialog(obs, h0, pa, ones, act, mem, memd, mk, toks, astep)[6].clone()
    csd = {k: v.detach().cpu() for k, v in pl.state_dict().items() if k.startswith("net.clip.")}
    with torch.no_grad():
        e_ref = R.clip_encode_text(csd, "net.clip", toks.cpu()).cuda()
    pl.net.text_encoder_override = lambda t: e_ref
    orc = pl.evaluate_actions_dialog(obs, h0, pa, ones, act, mem, memd, mk, toks, astep)[6].clone()
    torch.cuda.synchronize()
    dl = float((hip - orc).abs().max())
    print(f"pi_l logits, HIP fp16 tower vs oracle embedding (fixture weights): max |diff| {dl:.3e} (logit scale {float(orc.abs().max()):.3f})")
    assert dl < 1e-3 * max(1.0, float(orc.abs().max())), dl
