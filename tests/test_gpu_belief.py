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


@pytest.mark.parametrize("precision,tol", [("fp32", 2e-3), ("bf16", 6e-2), ("bf16x3", 8e-3)])     # bf16x3 policies: fp16 inference here
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


def _train_batch(tag, R, distractor):
    spec = torch.log1p(3.0 * fx.uni(tag + ".spec", (R, 65, 26, 2), 0.0, 2.0))
    spec[1] = 0.0
    spec[R - 2] = 0.0
    obs = {"spectrogram": spec, "pointgoal_with_gps_compass": fx.ints(tag + ".pg", (R, 2), 9).float() - 4.0}
    if distractor:
        cat = torch.zeros(R, 21)
        cat[torch.arange(R), fx.ints(tag + ".cat", (R,), 21)] = 1.0
        obs["category"] = cat
    return obs


@pytest.mark.parametrize("name,distractor", [("belief_train", False), ("belief_train_distractor", True)])
def test_online_regression_matches_reference(name, distractor):
    """`train_belief_predictor`'s minibatch body (ppo_trainer.py:996-1022): GroupNorm ResNet-18 forward + backward on the HIP
    path, masked MSE, Adam -- three optimiser steps vs the reference's BeliefPredictor + torch Adam (oracle/make_goldens_belief_train.py):
    per-step loss and predictions, accuracy counters, every parameter tensor after the steps."""
    g = golden(name)
    sd = predictor_sd("belief_loc_distractor" if distractor else "belief_loc")
    bp = BeliefPredictor(cfg(), "cuda", None, None, 512, num_env=2, has_distractor_sound=distractor, load_pretrained=False)
    bp.load_state_dict(sd, strict=False)
    bp = bp.cuda()
    bp.optimizer = torch.optim.Adam(bp.predictor.parameters(), lr=1e-3)
    acc = torch.zeros(3, device="cuda")
    R = 6
    prev = 0.0
    for step in range(3):
        preds = bp.regression_step(cu(_train_batch(f"{name}.{step % 2}", R, distractor)), acc)
        torch.cuda.synchronize()
        live = [0, 2, 3, 5]
        # step 0 is the plain forward (2e-3); later steps run on weights that took Adam steps of +-lr wherever the gradient is
        # near zero (sign-sensitive): 1 % of the prediction scale
        err = np.abs(preds.cpu().numpy()[live] - g["preds"][step][live]).max() / np.abs(g["preds"][step][live]).max()
        a = float(acc[0])
        lerr = abs((a - prev) - g["losses"][step]) / g["losses"][step]
        print(f"step {step}: prediction err {err:.3g} of scale, loss err {lerr:.3g}")
        assert err < (2e-3 if step == 0 else 1e-2) and lerr < (2e-3 if step == 0 else 2e-2)
        prev = a
    assert float(acc[1]) == float(g["correct"]) and float(acc[2]) == float(g["nsample"])
    new = {k[len("predictor."):]: v.detach().cpu() for k, v in bp.state_dict().items() if k.startswith("predictor.")}
    keys = sorted(new)
    pabs = np.array([float(new[k].double().abs().sum()) for k in keys])
    np.testing.assert_allclose(pabs, g["param_abs"], rtol=1e-2)          # small bias tensors: three sign-sensitive steps of lr each
    old = {k[len("predictor."):]: v for k, v in sd.items() if k.startswith("predictor.")}
    # the parameter step itself: direction of the three-step delta vs the reference's (element-wise an Adam step is +-lr where the
    # gradient is near zero, so single elements may flip; the tensors as a whole must agree)
    for key, gold, sl in (("conv1.weight", g["conv1_w"], np.s_[:2, :2]), ("fc.weight", g["fc_w"], np.s_[:, :16]),
                          ("layer4.1.conv2.weight", g["l4_w"], np.s_[:2, :4, 1, 1]),
                          ("layer2.0.downsample.1.weight", g["bn_g"], np.s_[:])):
        d_ours = (new[key].numpy()[sl] - old[key].numpy()[sl]).ravel().astype(np.float64)
        d_ref = (gold - old[key].numpy()[sl]).ravel().astype(np.float64)
        cos = float(d_ours @ d_ref / (np.linalg.norm(d_ours) * np.linalg.norm(d_ref) + 1e-30))
        assert cos > 0.9, (key, cos)
    # inference after training uses the stepped weights (packed copies refreshed)
    o = cu(_train_batch(f"{name}.0", R, distractor))
    again = bp.cnn_forward(o).cpu().numpy()
    assert np.abs(again - g["preds"][0]).max() > 1e-3


def test_train_belief_predictor_over_the_rollout_storage():
    """The trainer-side loop (ppo_trainer.py:959-1030): 5 epochs x 1 minibatch over every stored step of the RolloutStorage."""
    from avlen_amd.belief_predictor import train_belief_predictor
    from avlen_amd.rollout_storage import RolloutStorage
    from avlen_amd.spaces import ActionSpace

    class Box:
        def __init__(self, shape):
            self.shape = shape

    class Space:
        spaces = {"spectrogram": Box((65, 26, 2)), "pointgoal_with_gps_compass": Box((2,)), "pose": Box((4,))}
    T, N = 4, 3
    st = RolloutStorage(T, N, Space(), ActionSpace(4), 8, True, 6, 3, 6, 3, 3, 3, 5, 5, 5, 5, num_recurrent_layers=1,
                        max_dialog_len=7, use_state_memory=True, device="cuda")
    st.observations["spectrogram"].copy_(torch.log1p(3.0 * fx.uni("tb.spec", (T + 1, N, 65, 26, 2), 0.0, 2.0)))
    st.observations["pointgoal_with_gps_compass"].copy_(fx.ints("tb.pg", (T + 1, N, 2), 7).float() - 3.0)
    st.step = T
    bp = BeliefPredictor(cfg(), "cuda", None, None, 512, num_env=N, load_pretrained=False)
    bp.load_state_dict(predictor_sd("belief_loc"), strict=False)
    bp = bp.cuda()
    torch.manual_seed(0)
    l1, a1 = train_belief_predictor(bp, st)
    l2, a2 = train_belief_predictor(bp, st)
    assert np.isfinite(l1) and np.isfinite(l2) and l2 < l1 and 0.0 <= a1 <= 1.0       # the regression makes progress


@pytest.mark.parametrize("precision,distractor", [("fp32", False), ("fp32", True), ("bf16", False)])
def test_predictor_gradients_match_oracle_autograd(precision, distractor):
    """Every parameter gradient of the masked-MSE regression loss through the GroupNorm ResNet-18 (7x7 stem, 16 3x3 convs, three
    strided 1x1 downsamples, 20 GroupNorms, residual adds, fc): HIP backward (im2col / col2im GEMMs, GroupNorm backward with the
    ReLU masks folded in) vs torch autograd on the oracle restatement."""
    name = "belief_loc_distractor" if distractor else "belief_loc"
    sd = predictor_sd(name)
    bp = BeliefPredictor(cfg(), "cuda", None, None, 512, num_env=2, has_distractor_sound=distractor, load_pretrained=False,
                         precision=precision)
    bp.load_state_dict(sd, strict=False)
    bp = bp.cuda()
    R_ = 5
    obs = _train_batch("bgrad", R_, distractor)
    osd = {k: v.clone().requires_grad_(True) for k, v in sd.items() if k.startswith("predictor.")}
    preds = R.belief_cnn_forward(osd, obs, has_distractor_sound=distractor)
    masks = (obs["spectrogram"].reshape(R_, -1).sum(1, keepdim=True) != 0).float()
    gts = obs["pointgoal_with_gps_compass"]
    tg = torch.stack([gts[:, 1], -gts[:, 0]], 1)
    loss = torch.nn.functional.mse_loss(masks * preds, masks * tg)
    loss.backward()
    acc = torch.zeros(3, device="cuda")
    ours = bp.regression_step(cu(obs), acc, apply=False)
    torch.cuda.synchronize()
    fp = precision == "fp32"
    np.testing.assert_allclose(float(acc[0]), float(loss), rtol=1e-3 if fp else 5e-2)
    flat, worst, dot, n1, n2 = bp._flat, 0.0, 0.0, 0.0, 0.0
    for k, v in osd.items():
        n = k[len("predictor."):]
        mine = flat.grad_view(n, v.shape).cpu().double()
        ref = v.grad.double()
        err = float((mine - ref).norm() / (ref.norm() + 1e-30))
        worst = max(worst, err)
        dot, n1, n2 = dot + float((mine * ref).sum()), n1 + float((mine * mine).sum()), n2 + float((ref * ref).sum())
        if fp:
            assert err < 2e-3, (k, err)
    cos = dot / ((n1 * n2) ** 0.5 + 1e-30)
    print(f"{precision} distractor={distractor}: max relative L2 gradient error over {len(osd)} tensors: {worst:.3g}; cosine of the "
          f"whole gradient {cos:.4f}")
    # bf16 operands through 20 convolutions and 20 GroupNorms of a random-weight network: single small tensors (cancelling sums
    # such as a GroupNorm bias) deviate by tens of percent; the gradient as a whole must still point the same way
    assert cos > (0.999999 if fp else 0.97), cos


def test_asynchronous_belief_update_equals_the_synchronous_one():
    """`BeliefPredictor.update_async` + `Policy.late_inputs` (the two belief networks write into the storage slot on their own stream
    beside the next step's visual towers; the rest of the forward waits for their event) against the reference's order (beliefs
    written before the observation is stored, belief_predictor.py:139-206 / ppo_trainer.py:890-894): the same rollout, bit for bit."""
    import os
    from avlen_amd.harness import Workload
    from avlen_amd import _lib as L
    snaps = []
    try:
        for mode in (True, False):
            torch.manual_seed(5)
            wl = Workload(6, 7, spectrogram=(65, 26, 2), precision="bf16x3", belief_predictor=True, em_capacity=5, belief_async=mode)
            assert wl._belief_async == mode
            for _ in range(7):
                wl.rollout_step()
            torch.cuda.synchronize()                         # (no optimiser step: its loss sums are not bit-reproducible run to run)
            wl.rollouts.after_update()
            for _ in range(2):
                wl.rollout_step()
            torch.cuda.synchronize()
            ro = wl.rollouts
            snaps.append(({k: v.clone() for k, v in ro.observations.items() if k.endswith("belief")},
                          ro.value_preds.clone(), ro.actions.clone(), ro.em.memory.clone(), ro.em_option.memory.clone()))
            del wl
    finally:
        L.lib.avlen_set_tower_x3_reserved_cus(0)
    a, b = snaps
    for k in a[0]:
        assert float(a[0][k].abs().sum()) > 0 and torch.equal(a[0][k], b[0][k]), k
    for x, y in zip(a[1:], b[1:]):
        assert torch.equal(x, y)
