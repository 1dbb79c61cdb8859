"""Golden vectors for the GRU-baseline training cycle (BASELINE configs[1]) from the REFERENCE's own classes (build container
only):   python oracle/make_goldens_gru.py

The savi `AudioNavBaselinePolicy` (ss_baselines/savi/ppo/policy.py:299-320, 379-498) is trained by the reference's av_nav
`PPO` (ss_baselines/av_nav/ppo/ppo.py:16-165) over the plain `RolloutStorage` (ss_baselines/common/rollout_storage.py:16-235),
all imported unmodified.  The only glue is a signature adapter: av_nav's PPO calls `evaluate_actions(obs, h, prev, masks, actions)`
and reads a 4-tuple, the savi policy takes two more (unused) memory arguments and returns a 5-tuple.
Stores outputs only: per-step values / probabilities / sampled actions, returns, the update's 3-tuple, post-step parameter sums.
"""
import importlib.util
import json
import os
import sys
import numpy as np
import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import fixtures as fx          # noqa: E402
import ref_harness as rh       # noqa: E402
import cycle as cyc            # noqa: E402
from make_goldens import save, build, OUT      # noqa: E402

torch.set_num_threads(8)


def _load_file(name, rel):
    spec = importlib.util.spec_from_file_location(name, os.path.join(rh.REF, rel))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


class Adapter(nn.Module):
    def __init__(self, pol):
        super().__init__()
        self.pol = pol

    def evaluate_actions(self, obs, h, prev, masks, actions):
        v, lp, ent, h2, _ = self.pol.evaluate_actions(obs, h, prev, masks, actions, None, None)
        return v, lp, ent, h2


GRU_CFG = dict(clip_param=0.2, ppo_epoch=4, num_mini_batch=2, value_loss_coef=0.5, entropy_coef=0.01, lr=7e-4, eps=1e-5,
               max_grad_norm=0.5, use_normalized_advantage=False)       # savi/config/default.py RL.PPO defaults


def main():
    ns = rh.load()
    ppo_mod = _load_file("ref_av_nav_ppo", "ss_baselines/av_nav/ppo/ppo.py")
    st_mod = _load_file("ref_common_rollout_storage", "ss_baselines/common/rollout_storage.py")
    for tag, spectro, use_gae in (("gru_cycle", (65, 26, 2), True), ("gru_cycle_257_nogae", (257, 101, 2), False)):
        T, N = 5, 4
        pol, spec = build(ns, "baseline", spectrogram=spectro)
        agent = ppo_mod.PPO(Adapter(pol), **GRU_CFG)
        osp = rh.observation_space(spectro)
        st = st_mod.RolloutStorage(T, N, osp, rh.ActionSpace(4), 512, num_recurrent_layers=1)
        o0 = cyc.first_obs(N, spectro[:2], tag="gru")
        for k in st.observations:
            st.observations[k][0].copy_(o0[k])
        st.recurrent_hidden_states[0].copy_(fx.sym("gru.h0", (1, N, 512), 0.5))
        torch.manual_seed(777)
        rec = {k: [] for k in ("value", "action", "log_prob", "probs", "hidden")}
        for t in range(T):
            si = cyc.step_inputs(t, N, spectro[:2], tag="gru")
            so = {k: v[st.step] for k, v in st.observations.items()}
            with torch.no_grad():
                v, a, lp, h, _, probs = pol.act(so, st.recurrent_hidden_states[st.step], st.prev_actions[st.step],
                                                st.masks[st.step], None, None)
            for k, x in zip(rec, (v, a, lp, probs, h)):
                rec[k].append(x.clone())
            st.insert(si["next_obs"], h, a, lp, v, si["rewards"], si["not_done"])
        with torch.no_grad():
            lo = {k: v[-1] for k, v in st.observations.items()}
            nv = pol.get_value(lo, st.recurrent_hidden_states[-1], st.prev_actions[-1], st.masks[-1], None, None)
        st.compute_returns(nv, use_gae, 0.99, 0.95)
        returns = st.returns.clone()
        out = agent.update(st)
        st.after_update()
        sd = pol.state_dict()
        keys = sorted(k for k in sd if sd[k].dtype == torch.float32)
        save(tag, next_value=nv, returns=returns, update=np.array(out, dtype=np.float64),
             **{k: torch.stack(v) for k, v in rec.items()},
             param_sum=np.array([float(sd[k].double().sum()) for k in keys]),
             param_abs=np.array([float(sd[k].double().abs().sum()) for k in keys]),
             conv0_w=sd["net.visual_encoder.cnn.0.weight"][:2, :, :3, :3],
             afc_w=sd["net.audio_encoder.cnn.6.weight"][:3, :16],
             whh=sd["net.state_encoder.rnn.weight_hh_l0"][:4, :8], critic_w=sd["critic_goal.fc.weight"][:, :16])
        with open(os.path.join(OUT, tag + "_keys.json"), "w") as f:
            json.dump({"keys": keys, "spec": {k: list(v) for k, v in spec.items()}}, f)


if __name__ == "__main__":
    main()
