"""Pin the oracle (oracle/restate.py, oracle/flow.py) against golden vectors produced by the
REFERENCE's own modules (oracle/make_goldens.py).  CPU only."""
import json
import os
import numpy as np
import pytest
import torch

import fixtures as fx
import restate as R
import flow
import cycle as cyc
from conftest import golden, GOLDEN

torch.set_num_threads(8)
TOL = dict(rtol=1e-3, atol=2e-4)   # fp32 re-association noise through the 20-conv towers


def sd_for(specs, kind):
    sd = fx.state_dict_for({k: tuple(v) for k, v in specs[kind].items()})
    return sd


def close(a, b, **kw):
    tol = dict(TOL); tol.update(kw)
    np.testing.assert_allclose(a.detach().numpy() if torch.is_tensor(a) else a, b, **tol)


def test_encoders(specs):
    sd = sd_for(specs, "option")
    obs = fx.observations("enc", 2)
    g = golden("enc_visual")
    close(R.smt_cnn(sd, "net.visual_encoder", obs), g["out"])
    close(R.audio_cnn(sd, "net.goal_encoder", obs["spectrogram"]), golden("enc_audio_65")["out"])
    sd2 = sd_for(specs, "option_257")
    obs2 = fx.observations("enc", 2, (257, 101))
    close(R.audio_cnn(sd2, "net.goal_encoder", obs2["spectrogram"]), golden("enc_audio_257")["out"])


@pytest.mark.parametrize("pre", [True, False])
@pytest.mark.parametrize("M", [4, 300])
def test_option_policy(specs, pre, M):
    B = 3
    sd = sd_for(specs, "option")
    tag = f"opt_p{int(pre)}_m{M}"
    g = golden("policy_" + tag)
    obs = fx.observations(tag, B)
    mem, mk = fx.memory(tag, M, B, 308, 272), fx.mask_patterns(tag, B, M)
    qs, lqi = fx.sym(tag + ".qs", (B, 32)), fx.sym(tag + ".lqi", (B, 32))
    pa, act = fx.ints(tag + ".pa", (B, 1), 4), fx.ints(tag + ".a", (B, 1), 2)
    feats, row = R.option_net(sd, obs, pa, mem, mk, qs, lqi, pretraining=pre)
    h = R.heads(sd, "option", feats, action=act)
    close(h["value"], g["value"]); close(h["unct"], g["unct"]); close(h["log_prob"], g["log_prob"])
    close(h["entropy"], g["entropy"]); close(h["probs"], g["probs"]); close(row, g["row"])
    close(h["value"], g["get_value"])
    torch.manual_seed(1234)
    hs = R.heads(sd, "option", feats)
    assert np.array_equal(hs["action"].numpy(), g["sampled"])       # bit-exact sampling
    close(hs["log_prob"], g["sampled_log_prob"])
    assert np.array_equal(R.heads(sd, "option", feats, deterministic=True)["action"].numpy(), g["mode"])


def test_pretraining_collapses_to_current_token(specs):
    """D8: with pretraining=True the memory contents/masks cannot influence the output."""
    sd = sd_for(specs, "option")
    B = 3
    obs = fx.observations("col", B)
    qs, lqi, pa = fx.sym("col.qs", (B, 32)), fx.sym("col.l", (B, 32)), fx.ints("col.pa", (B, 1), 4)
    a, _ = R.option_net(sd, obs, pa, fx.memory("c1", 7, B, 308, 272), fx.mask_patterns("c1", B, 7), qs, lqi, True)
    b, _ = R.option_net(sd, obs, pa, fx.memory("c2", 1, B, 308, 272), torch.zeros(B, 1), qs, lqi, True)
    close(a, b.detach().numpy(), rtol=1e-4, atol=1e-5)


def test_option_distractor(specs):
    B, M = 3, 6
    sd = sd_for(specs, "option_distractor")
    tag = "opt_dis"
    g = golden("policy_" + tag)
    obs = fx.observations(tag, B)
    mem, mk = fx.memory(tag, M, B, 329, 293), fx.mask_patterns(tag, B, M)
    qs, lqi = fx.sym(tag + ".qs", (B, 32)), fx.sym(tag + ".lqi", (B, 32))
    pa, act = fx.ints(tag + ".pa", (B, 1), 4), fx.ints(tag + ".a", (B, 1), 2)
    feats, row = R.option_net(sd, obs, pa, mem, mk, qs, lqi, pretraining=False, use_category_input=True)
    h = R.heads(sd, "option", feats, action=act)
    close(h["value"], g["value"]); close(h["probs"], g["probs"]); close(row, g["row"])
    close(h["log_prob"], g["log_prob"]); close(h["unct"], g["unct"])


@pytest.mark.parametrize("M", [4, 300])
def test_goal_policy(specs, M):
    B = 3
    sd = sd_for(specs, "goal")
    tag = f"goal_m{M}"
    g = golden("policy_" + tag)
    obs = fx.observations(tag, B)
    mem, mk = fx.memory(tag, M, B, 276, 272), fx.mask_patterns(tag, B, M)
    pa, act = fx.ints(tag + ".pa", (B, 1), 4), fx.ints(tag + ".a", (B, 1), 4)
    feats, row = R.smt_net(sd, obs, pa, mem, mk)
    h = R.heads(sd, "goal", feats, action=act)
    close(h["value"], g["value"]); close(h["log_prob"], g["log_prob"]); close(h["entropy"], g["entropy"])
    close(h["probs"], g["probs"]); close(row, g["row"])
    torch.manual_seed(77)
    assert np.array_equal(R.heads(sd, "goal", feats)["action"].numpy(), g["sampled"])


@pytest.mark.parametrize("with_dialog", [True, False])
def test_dialog_policy(specs, with_dialog):
    B, M = 3, 3
    sd = sd_for(specs, "dialog")
    tag = "dlg"
    g = golden("policy_dlg" if with_dialog else "policy_dlg_nodialog")
    obs = fx.observations(tag, B)
    mem, memd = fx.memory(tag, M, B, 276, 272), fx.sym(tag + ".memd", (M, B, 256))
    mk = fx.mask_patterns(tag, B, M)
    pa, act = fx.ints(tag + ".pa", (B, 1), 4), fx.ints(tag + ".a", (B, 1), 4)
    toks = fx.dialog_tokens(tag, B) if with_dialog else None
    astep = fx.ints(tag + ".as", (B,), 3).float()
    xd, row = R.dialog_net(sd, obs, pa, mem, memd, mk, toks, astep, clip_fn=fx.stub_text_embedding)
    h = R.heads(sd, "vln", xd, action=act)
    close(xd, g["xd"]); close(row, g["row"]); close(h["logits"], g["logits"]); close(h["value"], g["value"])
    close(h["log_prob"], g["log_prob"]); close(h["entropy"], g["entropy"])
    torch.manual_seed(5)
    assert np.array_equal(R.heads(sd, "vln", xd)["action"].numpy(), g["sampled"])


def test_baseline_policy(specs):
    sd = sd_for(specs, "baseline")
    g = golden("policy_base")
    N, T = 3, 5
    obs = fx.observations("base", N)
    h0 = fx.sym("base.h0", (1, N, 512), 0.5)
    m1 = torch.tensor([[1.0], [0.0], [1.0]])
    x, h1 = R.baseline_net(sd, obs, h0, m1)
    hd = R.heads(sd, "goal", x, action=torch.zeros(N, 1, dtype=torch.long))
    close(hd["value"], g["value"]); close(hd["probs"], g["probs"]); close(h1, g["hidden"])
    obs_seq = fx.observations("base.seq", T * N)
    ms = torch.from_numpy((fx.unit("base.m", T * N) >= 0.3).astype("float32")).view(T * N, 1)
    act = fx.ints("base.a", (T * N, 1), 4)
    x2, h2 = R.baseline_net(sd, obs_seq, h0, ms)
    hd2 = R.heads(sd, "goal", x2, action=act)
    close(hd2["value"], g["seq_value"]); close(hd2["log_prob"], g["seq_log_prob"])
    close(hd2["entropy"], g["seq_entropy"]); close(h2, g["seq_hidden"])


def test_gae():
    T, N = 150, 4
    r, v = fx.sym("gae.r", (T, N, 1)), fx.sym("gae.v", (T + 1, N, 1))
    m = torch.from_numpy((fx.unit("gae.m", (T + 1) * N) >= 1 / 15).astype("float32")).view(T + 1, N, 1)
    ret, _ = R.gae_returns(r, v, m, fx.sym("gae.nv", (N, 1)), 0.99, 0.95)
    close(ret[:T], golden("gae")["returns"][:T], rtol=1e-5, atol=1e-6)
    ret, _ = R.gae_returns(r, v, m, fx.sym("gae.nv", (N, 1)), 0.99, 0.95, steps=97)
    close(ret[:97], golden("gae_short")["returns"][:97], rtol=1e-5, atol=1e-6)


def test_extmem_ring():
    g = golden("extmem")
    em = R.ExtMemoryRing(3, 8, 4, 5)
    for t in range(20):
        nd = torch.from_numpy((fx.unit(f"em.nd{t}", 3) >= 0.12).astype("float32")).view(3, 1)
        em.insert(fx.sym(f"em.f{t}", (3, 5)), nd)
        assert np.array_equal(em.masks.numpy(), g["masks"][t])
    assert np.array_equal(em.memory.numpy(), g["memory"]) and em.idx == int(g["idx"])


def test_host_rng_equivalence():
    """Categorical.sample on the CPU generator == exponential race; randperm follows (App. B)."""
    g = golden("rng")
    p = torch.softmax(fx.sym("rng.p", (16, 4), 2.0), 1)
    torch.manual_seed(31337)
    s1 = R.sample_host(p)[:, 0]
    s2 = R.sample_host(p[:, :2] / p[:, :2].sum(1, keepdim=True))[:, 0]
    perm = torch.randperm(8)
    assert np.array_equal(s1.numpy(), g["s1"]) and np.array_equal(s2.numpy(), g["s2"])
    assert np.array_equal(perm.numpy(), g["perm"])


@pytest.mark.parametrize("pre", [True, False])
def test_full_cycle(specs, pre):
    """act_option x T -> insert -> get_value -> GAE -> PPO.update (2 epochs x 2 minibatches) ->
    6-tuple and post-step parameters, against the reference's PPO/RolloutStorage/Policy."""
    g = golden(f"cycle_p{int(pre)}")
    keys = json.load(open(os.path.join(GOLDEN, f"cycle_p{int(pre)}_keys.json")))
    T, N, EMS, EMC = 6, 4, 12, 6
    sd = sd_for(specs, "option")
    agent = flow.OptionAgent(sd, pretraining=pre)
    st = flow.Storage(T, N, cyc.first_obs(N), EMS, EMC)
    torch.manual_seed(2024)
    for t in range(T):
        si = cyc.step_inputs(t, N)
        so = {k: v[st.step] for k, v in st.obs.items()}
        h, row = agent.act(so, st.prev_actions[st.step], st.em_option.memory, st.em_masks[st.step],
                           si["query_state"], si["last_query_info"])
        close(h["value"], g["value"][t]); close(h["probs"], g["probs"][t])
        assert np.array_equal(h["action"].numpy(), g["action_option"][t])
        st.insert(si["next_obs"], si["actions"], h["action"], h["log_prob"], h["value"], si["rewards"],
                  si["not_done"], si["not_done"], row[:, :276], row, row[:, :276], torch.zeros(N, 256),
                  torch.zeros(N, 77, dtype=torch.long), si["rl_masks"], si["ucnt_gt"], si["query_state"],
                  si["last_query_info"], si["agent_step"])
    out = agent.update(st)
    assert np.array_equal(st.em_masks.numpy(), g["em_masks"])      # golden is taken after after_update()
    close(st.returns[:T], g["returns"][:T])
    np.testing.assert_allclose(np.array(out), g["update"], rtol=2e-4, atol=2e-5)
    psum = np.array([float(sd[k].double().sum()) for k in keys])
    pabs = np.array([float(sd[k].double().abs().sum()) for k in keys])
    np.testing.assert_allclose(pabs, g["param_abs"], rtol=1e-5)
    np.testing.assert_allclose(psum, g["param_sum"], rtol=1e-3, atol=2e-3)
    close(sd["net.smt_state_encoder.fusion_encoder.2.weight"][:4, :8], g["fusion2_w"], rtol=1e-4, atol=1e-6)
    close(sd["critic_option.fc.weight"], g["critic_w"], rtol=1e-4, atol=1e-6)
    check_step(sd, sd_for(specs, "option"), keys, g, 2e-3)


def check_step(post, pre, keys, g, tol):
    """The parameter step of EVERY tensor against the reference's (goldens `delta_l2`, `delta_chk`: fixtures.delta_stats): its
    norm within `tol` (relative) and its signed weighted checksum within tol * norm; tensors the reference left untouched
    (encoders, unused heads) must not have moved at all."""
    l2, chk = fx.delta_stats(post, pre, keys)
    worst = 0.0
    for k, a, c, ga, gc in zip(keys, l2, chk, g["delta_l2"], g["delta_chk"]):
        if ga == 0.0:
            assert a == 0.0, (k, a)
            continue
        e = max(abs(a - ga), abs(c - gc)) / ga
        worst = max(worst, e)
        assert e < tol, (k, a, ga, c, gc)
    return worst


def test_param_counts(specs):
    assert specs["option__nparams"] == 4035805 and specs["goal__nparams"] == 4028063
    assert specs["baseline__nparams"] == 4995983
