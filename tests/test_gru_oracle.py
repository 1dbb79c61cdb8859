"""Oracle restatement of the GRU-baseline training cycle (oracle/flow.py: PlainStorage + BaselineAgent) against goldens produced by
the reference's own av_nav PPO / common RolloutStorage / savi AudioNavBaselinePolicy (oracle/make_goldens_gru.py)."""
import json
import os
import numpy as np
import pytest
import torch

import fixtures as fx
import cycle as cyc
import flow
import restate as R
from conftest import golden, GOLDEN


def run_oracle_cycle(tag, spectro, use_gae):
    meta = json.load(open(os.path.join(GOLDEN, tag + "_keys.json")))
    sd = fx.state_dict_for({k: tuple(v) for k, v in meta["spec"].items()})
    T, N = 5, 4
    agent = flow.BaselineAgent(sd)
    st = flow.PlainStorage(T, N, cyc.first_obs(N, spectro, tag="gru"))
    st.hidden[0].copy_(fx.sym("gru.h0", (1, N, 512), 0.5))
    torch.manual_seed(777)
    rec = {k: [] for k in ("value", "action", "probs", "hidden")}
    for t in range(T):
        si = cyc.step_inputs(t, N, spectro, tag="gru")
        so = {k: v[st.step] for k, v in st.obs.items()}
        h, hid = agent.act(so, st.hidden[st.step], st.masks[st.step])
        for k, x in zip(rec, (h["value"], h["action"], h["probs"], hid)):
            rec[k].append(x.clone())
        st.insert(si["next_obs"], hid, h["action"], h["log_prob"], h["value"], si["rewards"], si["not_done"])
    nv = agent.value({k: v[-1] for k, v in st.obs.items()}, st.hidden[-1], st.masks[-1])
    st.compute_returns(nv, use_gae, 0.99, 0.95)
    returns = st.returns.clone()
    out = agent.update(st)
    return rec, nv, returns, out, sd, meta["keys"]


@pytest.mark.parametrize("tag,spectro,use_gae", [("gru_cycle", (65, 26), True), ("gru_cycle_257_nogae", (257, 101), False)])
def test_gru_cycle_oracle_matches_reference(tag, spectro, use_gae):
    g = golden(tag)
    rec, nv, returns, out, sd, keys = run_oracle_cycle(tag, spectro, use_gae)
    tol = dict(rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(torch.stack(rec["value"]).numpy(), g["value"], **tol)
    np.testing.assert_allclose(torch.stack(rec["probs"]).numpy(), g["probs"], **tol)
    np.testing.assert_allclose(torch.stack(rec["hidden"]).numpy(), g["hidden"], **tol)
    assert np.array_equal(torch.stack(rec["action"]).numpy(), g["action"])
    np.testing.assert_allclose(nv.numpy(), g["next_value"], **tol)
    np.testing.assert_allclose(returns.numpy(), g["returns"], **tol)
    np.testing.assert_allclose(np.array(out), g["update"], rtol=2e-3, atol=1e-5)   # 8 Adam steps at lr 7e-4: near-zero gradients step by +-lr
    pabs = np.array([float(sd[k].detach().double().abs().sum()) for k in keys])
    np.testing.assert_allclose(pabs, g["param_abs"], rtol=2e-4)     # thread-count dependent summation order feeding Adam
    np.testing.assert_allclose(sd["net.visual_encoder.cnn.0.weight"][:2, :, :3, :3].detach().numpy(), g["conv0_w"], rtol=1e-2, atol=2e-5)
    np.testing.assert_allclose(sd["net.state_encoder.rnn.weight_hh_l0"][:4, :8].detach().numpy(), g["whh"], rtol=1e-2, atol=2e-5)
