"""Generate tests/golden/*.npz from the REFERENCE's own modules (build container only).

    python oracle/make_goldens.py

Imports the unmodified reference through oracle/ref_harness.py, loads the closed-form fixture
weights (oracle/fixtures.py) into the reference's modules, runs them on the fixture inputs and
stores ONLY outputs (small arrays).  The GPU box never sees the reference: tests regenerate
weights/inputs from the same closed forms and compare against these files.
"""
import json
import os
import sys
import warnings
import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import fixtures as fx          # noqa: E402
import ref_harness as rh       # noqa: E402
import cycle as cyc            # noqa: E402

warnings.filterwarnings("ignore")
OUT = os.environ.get("AVLEN_GOLDEN_OUT", os.path.join(os.path.dirname(HERE), "tests", "golden"))
os.makedirs(OUT, exist_ok=True)
torch.set_num_threads(8)

SMT_KW = dict(hidden_size=256, nhead=8, num_encoder_layers=1, num_decoder_layers=1, dropout=0.0,
              activation="relu", use_pretrained=False, pretrained_path="", use_belief_encoding=False,
              use_belief_as_goal=True, use_label_belief=True, use_location_belief=True,
              normalize_category_distribution=False)


def save(name, **arrs):
    np.savez_compressed(os.path.join(OUT, name + ".npz"),
                        **{k: (v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v))
                           for k, v in arrs.items()})
    print("wrote", name, {k: tuple(np.shape(v)) for k, v in arrs.items()})


def load_fixture_weights(module, tag=""):
    spec = {k: tuple(v.shape) for k, v in module.state_dict().items()}
    sd = fx.state_dict_for(spec, tag)
    missing = module.load_state_dict(sd, strict=False)
    assert not missing.unexpected_keys
    return spec


def build(ns, kind, spectrogram=(65, 26, 2), pretraining=False, distractor=False):
    osp, asp = rh.observation_space(spectrogram), rh.ActionSpace(4)
    P = ns.policy
    if kind == "option":
        pol = P.AudioNavOptionPolicy(osp, asp, pretraining=pretraining, use_category_input=distractor,
                                     query_count_emb_size=32, **SMT_KW)
    elif kind == "goal":
        pol = P.AudioNavSMTPolicy(osp, asp, pretraining=False, use_category_input=distractor, **SMT_KW)
    elif kind == "dialog":
        pol = P.AudioNavDialogPolicy(osp, asp, pretraining=False, use_category_input=distractor,
                                     num_steps=3, **SMT_KW)
    elif kind == "baseline":
        pol = P.AudioNavBaselinePolicy(osp, asp, "spectrogram", hidden_size=512)
    spec = load_fixture_weights(pol)
    return pol, spec


def main():
    ns = rh.load()
    specs = {}

    # ---- G1: parameter name/shape contract + parameter counts ------------------------
    for kind, kw in [("option", {}), ("goal", {}), ("dialog", {}), ("baseline", {}),
                     ("option_257", dict(spectrogram=(257, 101, 2))),
                     ("option_distractor", dict(distractor=True))]:
        pol, spec = build(ns, kind.split("_")[0], **kw)
        specs[kind] = {k: list(v) for k, v in spec.items()}
        specs[kind + "__nparams"] = int(sum(p.numel() for p in pol.parameters()))
    with open(os.path.join(OUT, "param_specs.json"), "w") as f:
        json.dump(specs, f, indent=0, sort_keys=True)
    print("param counts", {k: v for k, v in specs.items() if k.endswith("nparams")})

    # ---- G2: encoders ------------------------------------------------------------------
    B = 2
    pol, _ = build(ns, "option")
    obs = fx.observations("enc", B)
    with torch.no_grad():
        save("enc_visual", out=pol.net.visual_encoder(obs),
             rgb=pol.net.visual_encoder.rgb_encoder(
                 pol.net.visual_encoder.obs_transform(obs["rgb"].permute(0, 3, 1, 2) / 255.0)))
        save("enc_audio_65", out=pol.net.goal_encoder(obs))
    pol257, _ = build(ns, "option", spectrogram=(257, 101, 2))
    obs257 = fx.observations("enc", B, (257, 101))
    with torch.no_grad():
        save("enc_audio_257", out=pol257.net.goal_encoder(obs257))

    # ---- G3: pi_q forward, pretraining on/off, M in {4,300} ----------------------------
    B = 3
    for pre in (True, False):
        pol, _ = build(ns, "option", pretraining=pre)
        for M in (4, 300):
            tag = f"opt_p{int(pre)}_m{M}"
            obs = fx.observations(tag, B)
            mem = fx.memory(tag, M, B, 308, 272)
            mk = fx.mask_patterns(tag, B, M)
            qs, lqi = fx.sym(tag + ".qs", (B, 32)), fx.sym(tag + ".lqi", (B, 32))
            pa, act = fx.ints(tag + ".pa", (B, 1), 4), fx.ints(tag + ".a", (B, 1), 2)
            h0 = torch.zeros(1, B, 512)
            with torch.no_grad():
                v, u, lp, ent, _, row, probs = pol.evaluate_actions_option(
                    obs, h0, pa, torch.ones(B, 1), act, mem, mk, qs, lqi)
                torch.manual_seed(1234)
                v2, u2, a2, lp2, _, row2, probs2 = pol.act_option(obs, h0, pa, torch.ones(B, 1), mem, mk, qs, lqi)
                det = pol.act_option(obs, h0, pa, torch.ones(B, 1), mem, mk, qs, lqi, deterministic=True)[2]
                gv = pol.get_value_option(obs, h0, pa, torch.ones(B, 1), mem, mk, qs, lqi)
            save("policy_" + tag, value=v, unct=u, log_prob=lp, entropy=ent, row=row, probs=probs,
                 sampled=a2, sampled_log_prob=lp2, mode=det, get_value=gv)

    # distractor variant (use_category_input: feature dims 297 / 329)
    pol, _ = build(ns, "option", pretraining=False, distractor=True)
    tag, M = "opt_dis", 6
    obs = fx.observations(tag, B)
    mem = fx.memory(tag, M, B, 329, 293)
    mk = fx.mask_patterns(tag, B, M)
    qs, lqi = fx.sym(tag + ".qs", (B, 32)), fx.sym(tag + ".lqi", (B, 32))
    pa, act = fx.ints(tag + ".pa", (B, 1), 4), fx.ints(tag + ".a", (B, 1), 2)
    with torch.no_grad():
        v, u, lp, ent, _, row, probs = pol.evaluate_actions_option(
            obs, torch.zeros(1, B, 512), pa, torch.ones(B, 1), act, mem, mk, qs, lqi)
    save("policy_" + tag, value=v, unct=u, log_prob=lp, entropy=ent, row=row, probs=probs)

    # ---- G4: pi_g ----------------------------------------------------------------------
    pol, _ = build(ns, "goal")
    for M in (4, 300):
        tag = f"goal_m{M}"
        obs = fx.observations(tag, B)
        mem = fx.memory(tag, M, B, 276, 272)
        mk = fx.mask_patterns(tag, B, M)
        pa, act = fx.ints(tag + ".pa", (B, 1), 4), fx.ints(tag + ".a", (B, 1), 4)
        with torch.no_grad():
            v, lp, ent, _, row = pol.evaluate_actions(obs, torch.zeros(1, B, 512), pa, torch.ones(B, 1), act, mem, mk)
            torch.manual_seed(77)
            v2, a2, lp2, _, row2, probs = pol.act(obs, torch.zeros(1, B, 512), pa, torch.ones(B, 1), mem, mk)
        save("policy_" + tag, value=v, log_prob=lp, entropy=ent, row=row, probs=probs, sampled=a2)

    # ---- G5: pi_l with the stub text embedding (CLIP itself is unpinned) ---------------
    pol, _ = build(ns, "dialog")
    tag, M = "dlg", 3
    obs = fx.observations(tag, B)
    mem = fx.memory(tag, M, B, 276, 272)
    memd = fx.sym(tag + ".memd", (M, B, 256))
    mk = fx.mask_patterns(tag, B, M)
    pa, act = fx.ints(tag + ".pa", (B, 1), 4), fx.ints(tag + ".a", (B, 1), 4)
    toks = fx.dialog_tokens(tag, B)
    astep = fx.ints(tag + ".as", (B,), 3).float()
    with torch.no_grad():
        for wd, nm in ((False, "policy_dlg"), (True, "policy_dlg_nodialog")):
            _, lp, ent, _, row, xd, logits = pol.evaluate_actions_dialog(
                obs, torch.zeros(1, B, 512), pa, torch.ones(B, 1), act, mem, memd, mk, toks, astep, without_dialog=wd)
            torch.manual_seed(5)
            v, a2, lp2, _, _, _, probs = pol.act_dialog(obs, torch.zeros(1, B, 512), pa, torch.ones(B, 1), mem, memd,
                                                       mk, toks, astep, without_dialog=wd)
            save(nm, value=v, log_prob=lp, entropy=ent, row=row, xd=xd, logits=logits, probs=probs, sampled=a2)

    # ---- G6: GRU baseline: one step + a (T,N) sequence ---------------------------------
    pol, _ = build(ns, "baseline")
    tag = "base"
    N, T = 3, 5
    obs = fx.observations(tag, N)
    h0 = fx.sym(tag + ".h0", (1, N, 512), 0.5)
    m1 = torch.tensor([[1.0], [0.0], [1.0]])
    with torch.no_grad():
        torch.manual_seed(9)
        v, a, lp, h1, _, probs = pol.act(obs, h0, None, m1, None, None)
        obs_seq = fx.observations(tag + ".seq", T * N)
        ms = torch.from_numpy((fx.unit(tag + ".m", T * N) >= 0.3).astype("float32")).view(T * N, 1)
        act = fx.ints(tag + ".a", (T * N, 1), 4)
        v2, lp2, ent2, h2, _ = pol.evaluate_actions(obs_seq, h0, None, ms, act, None, None)
    save("policy_base", value=v, probs=probs, hidden=h1, sampled=a, seq_value=v2, seq_log_prob=lp2,
         seq_entropy=ent2, seq_hidden=h2)

    # ---- G7: GAE ------------------------------------------------------------------------
    T, N = 150, 4
    osp = rh.observation_space()
    st = ns.RolloutStorage(T, N, rh.ObsSpace({"pose": ns.Box(shape=(4,))}), rh.ActionSpace(4), 512, False,
                           2, 1, 2, 1, 1, 1, 4, 4, 4, 4, num_recurrent_layers=-1)
    st.rewards.copy_(fx.sym("gae.r", (T, N, 1)))
    st.value_preds.copy_(fx.sym("gae.v", (T + 1, N, 1)))
    st.masks.copy_(torch.from_numpy((fx.unit("gae.m", (T + 1) * N) >= 1 / 15).astype("float32")).view(T + 1, N, 1))
    st.step = T
    st.compute_returns(fx.sym("gae.nv", (N, 1)), True, 0.99, 0.95)
    save("gae", returns=st.returns)
    st.step = 97                                   # preempted rollout (variable length)
    st.returns.zero_()
    st.compute_returns(fx.sym("gae.nv", (N, 1)), True, 0.99, 0.95)
    save("gae_short", returns=st.returns)

    # ---- G8: external-memory ring -------------------------------------------------------
    em = ns.ExternalMemory(3, 8, 4, 5, num_copies=4, num_steps=3)
    hist = []
    for t in range(20):
        nd = torch.from_numpy((fx.unit(f"em.nd{t}", 3) >= 0.12).astype("float32")).view(3, 1)
        em.insert(fx.sym(f"em.f{t}", (3, 5)), nd)
        hist.append(em.masks.clone())
    save("extmem", masks=torch.stack(hist), memory=em.memory[:, 0], idx=em.idx)

    # ---- G9: one full rollout -> update cycle through the reference's PPO ---------------
    for pre in (True, False):
        T, N, EMS, EMC = 6, 4, 12, 6
        pol, _ = build(ns, "option", pretraining=pre)
        sd_pre = {k: v.detach().clone() for k, v in pol.state_dict().items()}
        agent = ns.PPO(pol, 0.2, 2, 2, 0.5, 0.05, lr=2.5e-4, eps=1e-5, max_grad_norm=0.2,
                       use_normalized_advantage=False)
        st = ns.RolloutStorage(T, N, rh.observation_space(), rh.ActionSpace(4), 512, True, EMS, EMC, EMS, EMC,
                               3, 3, 276, 276, 308, 256, num_recurrent_layers=-1, max_dialog_len=77,
                               use_state_memory=True)
        o0 = cyc.first_obs(N)
        for k in st.observations:
            st.observations[k][0].copy_(o0[k])
        torch.manual_seed(2024)
        rec = {k: [] for k in ("value", "action_option", "log_prob", "probs")}
        for t in range(T):
            si = cyc.step_inputs(t, N)
            st.query_state[st.step].copy_(si["query_state"])
            st.last_query_info[st.step].copy_(si["last_query_info"])
            so = {k: v[st.step] for k, v in st.observations.items()}
            with torch.no_grad():
                v, u, ao, lp, h, row, probs = pol.act_option(
                    so, st.recurrent_hidden_states[st.step], st.prev_actions[st.step], st.masks[st.step],
                    st.em_option.memory[:, st.step].contiguous(), st.em_masks[st.step],
                    st.query_state[st.step], st.last_query_info[st.step])
            for k, x in zip(rec, (v, ao, lp, probs)):
                rec[k].append(x.clone())
            st.insert(si["next_obs"], h, si["actions"], ao, lp, v, si["rewards"], si["not_done"], si["not_done"],
                      row[:, :276], row, row[:, :276], torch.zeros(N, 256), torch.zeros(N, 77, dtype=torch.long),
                      torch.zeros(N), torch.ones(N, dtype=torch.long), si["rl_masks"], si["ucnt_gt"],
                      torch.zeros(N, 4), si["query_state"], si["last_query_info"], si["agent_step"])
        with torch.no_grad():
            lo = {k: v[-1] for k, v in st.observations.items()}
            nv = pol.get_value_option(lo, st.recurrent_hidden_states[st.step], st.prev_actions[st.step],
                                      st.masks[st.step], st.em_option.memory[:, st.step].contiguous(),
                                      st.em_masks[st.step], st.query_state[st.step - 1],
                                      st.last_query_info[st.step - 1])
        st.compute_returns(nv, True, 0.99, 0.95)
        returns = st.returns.clone()
        out = agent.update(st)
        st.after_update()
        sd = pol.state_dict()
        keys = sorted(k for k in sd if sd[k].dtype == torch.float32)
        d_l2, d_chk = fx.delta_stats(sd, sd_pre, keys)        # the step of EVERY tensor: norm + signed weighted checksum
        save(f"cycle_p{int(pre)}", next_value=nv, returns=returns, update=np.array(out, dtype=np.float64),
             em_masks=st.em_masks, **{k: torch.stack(v) for k, v in rec.items()}, delta_l2=d_l2, delta_chk=d_chk,
             param_sum=np.array([float(sd[k].double().sum()) for k in keys]),
             param_abs=np.array([float(sd[k].double().abs().sum()) for k in keys]),
             fusion2_w=sd["net.smt_state_encoder.fusion_encoder.2.weight"][:4, :8],
             critic_w=sd["critic_option.fc.weight"])
        with open(os.path.join(OUT, f"cycle_p{int(pre)}_keys.json"), "w") as f:
            json.dump(keys, f)

    # ---- G10: host RNG equivalence (Categorical.sample vs exponential race) -------------
    p = torch.softmax(fx.sym("rng.p", (16, 4), 2.0), 1)
    torch.manual_seed(31337)
    s1 = torch.distributions.Categorical(probs=p).sample()
    s2 = torch.distributions.Categorical(probs=p[:, :2] / p[:, :2].sum(1, keepdim=True)).sample()
    perm = torch.randperm(8)
    save("rng", s1=s1, s2=s2, perm=perm)

    # ---- G11: initialiser parity (same seed, same construction order) -------------------
    torch.manual_seed(0)
    pol = ns.policy.AudioNavOptionPolicy(rh.observation_space(), rh.ActionSpace(4), pretraining=True,
                                         use_category_input=False, query_count_emb_size=32, **SMT_KW)
    sd = pol.state_dict()
    keys = sorted(sd)
    save("init_option_seed0", sums=np.array([float(sd[k].double().sum()) for k in keys]),
         abss=np.array([float(sd[k].double().abs().sum()) for k in keys]))


if __name__ == "__main__":
    main()
