"""Golden for the distractor variant (BASELINE configs[4]: use_category_input=True, memory dims 297 / 329) through one full
rollout -> GAE -> PPO.update cycle of the REFERENCE's own PPO / RolloutStorage / AudioNavOptionPolicy (build container only):
    python oracle/make_goldens_dis.py
Same protocol as G9 of make_goldens.py (cycle_p0 / cycle_p1)."""
import json
import os
import sys
import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import fixtures as fx          # noqa: E402
import ref_harness as rh       # noqa: E402
import cycle as cyc            # noqa: E402
from make_goldens import save, build, OUT      # noqa: E402

torch.set_num_threads(8)


def main():
    ns = rh.load()
    pre = True
    T, N, EMS, EMC = 6, 4, 12, 6
    pol, spec = build(ns, "option", pretraining=pre, distractor=True)
    sd_pre = {k: v.detach().clone() for k, v in pol.state_dict().items()}
    agent = ns.PPO(pol, 0.2, 2, 2, 0.5, 0.05, lr=2.5e-4, eps=1e-5, max_grad_norm=0.2, use_normalized_advantage=False)
    st = ns.RolloutStorage(T, N, rh.observation_space(), rh.ActionSpace(4), 512, True, EMS, EMC, EMS, EMC, 3, 3, 297, 276, 329,
                           256, num_recurrent_layers=-1, max_dialog_len=77, use_state_memory=True)
    o0 = cyc.first_obs(N, tag="dis")
    for k in st.observations:
        st.observations[k][0].copy_(o0[k])
    torch.manual_seed(2025)
    rec = {k: [] for k in ("value", "action_option", "log_prob", "probs")}
    for t in range(T):
        si = cyc.step_inputs(t, N, tag="dis")
        st.query_state[st.step].copy_(si["query_state"])
        st.last_query_info[st.step].copy_(si["last_query_info"])
        so = {k: v[st.step] for k, v in st.observations.items()}
        with torch.no_grad():
            v, u, ao, lp, h, row, probs = pol.act_option(
                so, st.recurrent_hidden_states[st.step], st.prev_actions[st.step], st.masks[st.step],
                st.em_option.memory[:, st.step].contiguous(), st.em_masks[st.step], st.query_state[st.step],
                st.last_query_info[st.step])
        for k, x in zip(rec, (v, ao, lp, probs)):
            rec[k].append(x.clone())
        st.insert(si["next_obs"], h, si["actions"], ao, lp, v, si["rewards"], si["not_done"], si["not_done"],
                  row[:, :297], row, row[:, :276], torch.zeros(N, 256), torch.zeros(N, 77, dtype=torch.long),
                  torch.zeros(N), torch.ones(N, dtype=torch.long), si["rl_masks"], si["ucnt_gt"],
                  torch.zeros(N, 4), si["query_state"], si["last_query_info"], si["agent_step"])
    with torch.no_grad():
        lo = {k: v[-1] for k, v in st.observations.items()}
        nv = pol.get_value_option(lo, st.recurrent_hidden_states[st.step], st.prev_actions[st.step], st.masks[st.step],
                                  st.em_option.memory[:, st.step].contiguous(), st.em_masks[st.step],
                                  st.query_state[st.step - 1], st.last_query_info[st.step - 1])
    st.compute_returns(nv, True, 0.99, 0.95)
    returns = st.returns.clone()
    out = agent.update(st)
    st.after_update()
    sd = pol.state_dict()
    keys = sorted(k for k in sd if sd[k].dtype == torch.float32)
    d_l2, d_chk = fx.delta_stats(sd, sd_pre, keys)
    save("cycle_dis", next_value=nv, returns=returns, update=np.array(out, dtype=np.float64), em_masks=st.em_masks,
         **{k: torch.stack(v) for k, v in rec.items()}, delta_l2=d_l2, delta_chk=d_chk,
         param_abs=np.array([float(sd[k].double().abs().sum()) for k in keys]),
         fusion0_w=sd["net.smt_state_encoder.fusion_encoder.0.weight"][:4, 270:300])
    with open(os.path.join(OUT, "cycle_dis_keys.json"), "w") as f:
        json.dump(keys, f)


if __name__ == "__main__":
    main()
