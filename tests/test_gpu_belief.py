"""avlen_amd.belief_predictor.BeliefPredictor (HIP path through the C ABI) against
 (1) goldens produced by the REFERENCE's BeliefPredictor (location half: custom_resnet18 + odometry filter), and
 (2) the CPU oracle on the same inputs (both halves; the torchvision classifier is parity-unpinned).
Tolerances: fp32 mode 2e-3 on the network outputs / filtered beliefs; bf16 mode stated per assertion."""
import json
import os
import types
import numpy as np
import pytest
import torch

import fixtures as fx
import restate as R
from conftest import golden, GOLDEN
from avlen_amd.belief_predictor import BeliefPredictor

pytestmark = pytest.mark.gpu


def cfg(label=False, location=True, current_pred_only=False, w=0.5):
    return types.SimpleNamespace(use_label_belief=label, use_location_belief=location, online_training=True,
                                 current_pred_only=current_pred_only, weighting_factor=w)


def cu(o):
    return {k: v.cuda() for k, v in o.items()}


def predictor_sd(name):
    specs = json.load(open(os.path.join(GOLDEN, "belief_specs.json")))
    return fx.state_dict_for({k: tuple(v) for k, v in specs[name].items()}, name + ".")


def classifier_sd(bp, tag="belief_lab."):
    spec = {k: tuple(v.shape) for k, v in bp.state_dict().items() if k.startswith("classifier.") and v.dtype == torch.float32}
    sd = fx.state_dict_for(spec, tag)
    for k in sd:
        if k.endswith("running_var"):
            sd[k] = sd[k].abs() + 0.5
    return sd


@pytest.mark.parametrize("name,distractor", [("belief_loc", False), ("belief_loc_distractor", True)])
def test_location_half_matches_reference_golden(name, distractor):
    g = golden(name)
    sd = predictor_sd(name)
    bp = BeliefPredictor(cfg(), "cuda", None, None, 512, num_env=3, has_distractor_sound=distractor, load_pretrained=False)
    assert int(g["nparams"]) == sum(p.numel() for p in bp.parameters())
    assert not bp.load_state_dict(sd, strict=False).unexpected_keys
    bp = bp.cuda()
    for t, (obs, dones) in enumerate(fx.belief_scenario(name, 3)):
        o = cu(obs)
        live = (obs["spectrogram"].flatten(1).sum(1) != 0).numpy()
        pg = bp.cnn_forward(o).cpu().numpy()
        np.testing.assert_allclose(pg[live], g["pointgoals"][t][live], rtol=2e-3, atol=2e-3)
        bp.update(o, dones)
        np.testing.assert_allclose(o["location_belief"].cpu().numpy(), g["location_belief"][t], rtol=2e-3, atol=4e-3)
    last = np.stack([np.full(2, np.nan) if v is None else v for v in bp.last_pointgoal])
    np.testing.assert_allclose(last, g["last_pointgoal"], rtol=2e-3, atol=4e-3, equal_nan=True)


def test_filter_alone_is_float32_exact_vs_oracle():
    """Same network outputs in -> the device filter reproduces the numpy loop to float32 rounding of sin/cos/atan2."""
    from avlen_amd import _lib as L
    from avlen_amd.engine import P
    N = 5
    filt = R.BeliefFilter(N, 0.3, False)
    z = lambda *s, dt=torch.float32: torch.zeros(*s, dtype=dt, device="cuda")
    last_pg, has_pg, last_lab, has_lab, ssum = z(N, 2), z(N, dt=torch.int32), z(N, 21), z(N, dt=torch.int32), z(N)
    g = torch.Generator().manual_seed(5)
    for t in range(12):
        obs = fx.observations(f"bf.{t}", N, (65, 26), step=t)
        for i in range(N):
            if torch.rand((), generator=g) < 0.35:
                obs["spectrogram"][i] = 0.0
        dones = None if t % 5 == 0 else [bool(torch.rand((), generator=g) < 0.25) for _ in range(N)]
        pg = fx.sym(f"bf.pg{t}", (N, 2), 4.0)
        lab = fx.sym(f"bf.lab{t}", (N, 21), 3.0)
        filt.update(obs, dones, pointgoals=pg, labels=lab)
        o = cu(obs)
        loc, cb = z(N, 2), z(N, 21)
        d = torch.tensor(dones, dtype=torch.uint8, device="cuda") if dones is not None else None
        pg_d, lab_d = pg.cuda(), lab.cuda()                # keep the device copies alive across the launch
        L.call("avlen_belief_update", P(pg_d), 2, P(lab_d), 21, P(o["pose"]), 4, P(o["spectrogram"]),
               o["spectrogram"][0].numel(), P(d) if d is not None else None, P(last_pg), P(has_pg), P(last_lab), P(has_lab),
               P(loc), P(cb), P(ssum), N, 21, 0.3, 0, L.stream())
        np.testing.assert_allclose(loc.cpu().numpy(), obs["location_belief"].numpy(), rtol=2e-5, atol=2e-5)
        np.testing.assert_allclose(cb.cpu().numpy(), obs["category_belief"].numpy(), rtol=1e-6, atol=1e-7)
    assert [bool(h) for h in has_pg.cpu()] == [v is not None for v in filt.last_pointgoal]
    assert [bool(h) for h in has_lab.cpu()] == [v is not None for v in filt.last_label]


@pytest.mark.parametrize("precision,tol", [("fp32", 2e-3), ("bf16", 6e-2)])
def test_both_halves_vs_oracle(precision, tol):
    N = 4
    bp = BeliefPredictor(cfg(label=True), "cuda", None, None, 512, num_env=N, precision=precision, load_pretrained=False)
    sd = predictor_sd("belief_loc")
    sd.update(classifier_sd(bp))
    assert not bp.load_state_dict(sd, strict=False).unexpected_keys
    bp = bp.cuda()
    filt = R.BeliefFilter(N, 0.5, False)
    for t, (obs, dones) in enumerate(fx.belief_scenario("belief_both", N, T=5)):
        o = cu(obs)
        bp.update(o, dones)
        with torch.no_grad():
            pg = R.belief_cnn_forward(sd, obs, False)
            lab = R.tv_resnet18(sd, "classifier", obs["spectrogram"].permute(0, 3, 1, 2))
        filt.update(obs, dones, pointgoals=pg, labels=lab)
        scale = max(1.0, float(lab.abs().max()))
        np.testing.assert_allclose(o["category_belief"].cpu().numpy(), obs["category_belief"].numpy(), rtol=tol, atol=tol * scale)
        np.testing.assert_allclose(o["location_belief"].cpu().numpy(), obs["location_belief"].numpy(), rtol=tol,
                                   atol=2 * tol * max(1.0, float(pg.abs().max())))


def test_current_pred_only_and_rollout_batch():
    """config.current_pred_only=True (belief_predictor.py:161,191) and a rollout-sized batch (64 envs)."""
    N = 64
    bp = BeliefPredictor(cfg(label=True, current_pred_only=True), "cuda", None, None, 512, num_env=N, load_pretrained=False)
    sd = predictor_sd("belief_loc")
    sd.update(classifier_sd(bp))
    bp.load_state_dict(sd, strict=False)
    bp = bp.cuda()
    for t in range(2):
        obs = fx.observations(f"b64.{t}", N, (65, 26), step=t)
        o = cu(obs)
        bp.update(o, None)
        with torch.no_grad():
            pg = R.belief_cnn_forward(sd, obs, False)
        exp = torch.stack([-pg[:, 1], pg[:, 0]], 1)               # no averaging: the current prediction, base frame
        np.testing.assert_allclose(o["location_belief"].cpu().numpy(), exp.numpy(), rtol=2e-3, atol=2e-3)
        assert torch.isfinite(o["category_belief"]).all()


def test_no_cpu_fallback():
    bp = BeliefPredictor(cfg(), "cpu", None, None, 512, num_env=2, load_pretrained=False)
    obs = fx.observations("cpu", 2)
    with pytest.raises(AssertionError, match="no CPU fallback"):
        bp.update(obs, None)


def test_graph_replay_matches_eager():
    """use_graphs=True: captured update == launch-by-launch update, over steps with dones=None / lists / silent envs."""
    N = 3
    bps = []
    for ug in (False, True):
        bp = BeliefPredictor(cfg(label=True), "cuda", None, None, 512, num_env=N, precision="bf16", load_pretrained=False,
                             use_graphs=ug)
        sd = predictor_sd("belief_loc")
        sd.update(classifier_sd(bp))
        bp.load_state_dict(sd, strict=False)
        bps.append(bp.cuda())
    for t, (obs, dones) in enumerate(fx.belief_scenario("belief_graph", N)):
        o1, o2 = cu(obs), cu(obs)
        bps[0].update(o1, dones)
        bps[1].update(o2, dones)
        for k in ("location_belief", "category_belief"):
            assert torch.equal(o1[k], o2[k]), (t, k)
