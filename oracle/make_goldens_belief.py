"""Generate tests/golden/belief_*.npz from the REFERENCE's BeliefPredictor (build container only).

    python oracle/make_goldens_belief.py

The location half (predictor = custom_resnet18 + the odometry filter) runs the reference's own code.  The label half needs
torchvision.models.resnet18, which this image does not have and which is not stubbed: the classifier stays unpinned
(SURVEY 8c); only its filter branch structure is shared with the pinned location half.
"""
import json
import os
import sys
import types
import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import fixtures as fx          # noqa: E402
import ref_harness as rh       # noqa: E402
from make_goldens import save, load_fixture_weights, OUT   # noqa: E402


def main():
    ns = rh.load()
    specs = {}
    for name, distractor in (("belief_loc", False), ("belief_loc_distractor", True)):
        cfg = types.SimpleNamespace(use_label_belief=False, use_location_belief=True, online_training=True,
                                    current_pred_only=False, weighting_factor=0.5)
        N = 3
        bp = ns.BeliefPredictor(cfg, "cpu", None, None, 512, num_env=N, has_distractor_sound=distractor)
        specs[name] = {k: list(v) for k, v in load_fixture_weights(bp, name + ".").items()}
        pgs, locs = [], []
        for obs, dones in fx.belief_scenario(name, N):
            with torch.no_grad():
                pgs.append(bp.cnn_forward(obs).clone())
            bp.update(obs, dones)
            locs.append(obs["location_belief"].clone())
        last = np.stack([np.full(2, np.nan) if v is None else np.asarray(v, dtype=np.float64) for v in bp.last_pointgoal])
        save(name, pointgoals=torch.stack(pgs), location_belief=torch.stack(locs), last_pointgoal=last,
             nparams=np.array(sum(p.numel() for p in bp.parameters())))
    with open(os.path.join(OUT, "belief_specs.json"), "w") as f:
        json.dump(specs, f, indent=0, sort_keys=True)


if __name__ == "__main__":
    torch.set_num_threads(8)
    main()
