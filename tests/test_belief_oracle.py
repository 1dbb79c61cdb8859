"""BeliefPredictor (belief_predictor.py:56-206): the oracle's restatement against goldens produced by the REFERENCE's own
class (oracle/make_goldens_belief.py).  CPU only.  The label classifier (torchvision resnet18) has no golden: parity
unpinned; its oracle restatement is only checked for shape/finite output here."""
import json
import os
import numpy as np
import pytest
import torch

import fixtures as fx
import restate as R
from conftest import golden, GOLDEN

torch.set_num_threads(8)


def belief_sd(name):
    specs = json.load(open(os.path.join(GOLDEN, "belief_specs.json")))
    return fx.state_dict_for({k: tuple(v) for k, v in specs[name].items()}, name + ".")


@pytest.mark.parametrize("name,distractor", [("belief_loc", False), ("belief_loc_distractor", True)])
def test_location_belief_matches_reference(name, distractor):
    g = golden(name)
    sd = belief_sd(name)
    filt = R.BeliefFilter(3, 0.5, False)
    for t, (obs, dones) in enumerate(fx.belief_scenario(name, 3)):
        with torch.no_grad():
            pg = R.belief_cnn_forward(sd, obs, distractor)
        live = (obs["spectrogram"].flatten(1).sum(1) != 0).numpy()
        np.testing.assert_allclose(pg.numpy()[live], g["pointgoals"][t][live], rtol=1e-3, atol=2e-4)
        # an all-zero spectrogram drives every GroupNorm to var ~ 0 (rsqrt(eps) amplifies rounding noise); that output is
        # never used by the filter (belief_predictor.py:157-172), so it only has to be close
        np.testing.assert_allclose(pg.numpy()[~live], g["pointgoals"][t][~live], rtol=2e-2, atol=5e-3)
        filt.update(obs, dones, pointgoals=torch.from_numpy(g["pointgoals"][t]))      # same network outputs -> same floats
        np.testing.assert_allclose(obs["location_belief"].numpy(), g["location_belief"][t], rtol=1e-6, atol=1e-6)
    last = np.stack([np.full(2, np.nan) if v is None else v for v in filt.last_pointgoal])
    np.testing.assert_allclose(last, g["last_pointgoal"], rtol=1e-6, atol=1e-6, equal_nan=True)


def test_scenario_covers_every_branch():
    """silent-without-estimate, silent-with-estimate, reset-while-sounding, reset-while-silent, dones=None"""
    steps = fx.belief_scenario("belief_loc", 3)
    seen = set()
    have = [False] * 3
    for obs, dones in steps:
        if dones is None:
            seen.add("dones_none")
        for i in range(3):
            d = dones is not None and dones[i]
            if d:
                have[i] = False
            s = obs["spectrogram"][i].sum().item() != 0
            seen.add(("sounding" if s else "silent", "has" if have[i] else "none", "done" if d else "live"))
            if s:
                have[i] = True
    for need in [("silent", "none", "live"), ("silent", "has", "live"), ("sounding", "none", "done"),
                 ("silent", "none", "done"), ("sounding", "has", "live"), "dones_none"]:
        assert need in seen, need


def test_label_filter_and_classifier_restatement():
    spec = {}
    from collections import OrderedDict
    # torchvision resnet18 key/shape table (conv1 2->64, fc 512->21), built without torchvision
    def bn(p, c):
        spec.update({p + ".weight": (c,), p + ".bias": (c,), p + ".running_mean": (c,), p + ".running_var": (c,)})
    spec["classifier.conv1.weight"] = (64, 2, 7, 7); bn("classifier.bn1", 64)
    cin = 64
    for li, c in ((1, 64), (2, 128), (3, 256), (4, 512)):
        for bi in (0, 1):
            q = f"classifier.layer{li}.{bi}"
            spec[q + ".conv1.weight"] = (c, cin if bi == 0 else c, 3, 3); bn(q + ".bn1", c)
            spec[q + ".conv2.weight"] = (c, c, 3, 3); bn(q + ".bn2", c)
            if bi == 0 and li > 1:
                spec[q + ".downsample.0.weight"] = (c, cin, 1, 1); bn(q + ".downsample.1", c)
        cin = c
    spec["classifier.fc.weight"] = (21, 512); spec["classifier.fc.bias"] = (21,)
    sd = fx.state_dict_for(spec, "belief_lab.")
    for k in sd:
        if k.endswith("running_var"):
            sd[k] = sd[k].abs() + 0.5
    filt = R.BeliefFilter(3, 0.5, False)
    prev = None
    for t, (obs, dones) in enumerate(fx.belief_scenario("belief_lab", 3, T=4)):
        with torch.no_grad():
            lab = R.tv_resnet18(sd, "classifier", obs["spectrogram"].permute(0, 3, 1, 2))
        assert lab.shape == (3, 21) and torch.isfinite(lab).all()
        filt.update(obs, dones, labels=lab)
        cb = obs["category_belief"].clone()
        if t == 0:
            np.testing.assert_allclose(cb[1].numpy(), np.full(21, 1 / 21), rtol=1e-6)      # silent, no estimate yet
            np.testing.assert_allclose(cb[0].numpy(), lab[0].numpy(), rtol=1e-6)
        if t == 1:
            np.testing.assert_allclose(cb[0].numpy(), 0.5 * lab[0].numpy() + 0.5 * prev[0].numpy(), rtol=1e-5, atol=1e-6)
        prev = cb
