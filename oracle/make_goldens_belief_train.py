"""Goldens for the location predictor's online regression (`train_belief_predictor`, ppo_trainer.py:959-1030; called after every
PPO update, ddppo_trainer.py:977-978) from the REFERENCE's own BeliefPredictor + torch.optim.Adam (build container only):
    python oracle/make_goldens_belief_train.py
The trainer method itself cannot be imported (it lives in the simulator-bound trainer class); its minibatch body is replayed
statement by statement around the reference module: zero_grad, cnn_forward, the silent-row mask, the (y, -x) goal transform,
MSELoss on the masked values, backward, Adam step, the rounded-prediction accuracy.  Stores outputs only."""
import os
import sys
import types
import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import fixtures as fx          # noqa: E402
import ref_harness as rh       # noqa: E402
from make_goldens import save, load_fixture_weights   # noqa: E402


def batch(tag, R, distractor):
    spec = torch.log1p(3.0 * fx.uni(tag + ".spec", (R, 65, 26, 2), 0.0, 2.0))
    spec[1] = 0.0
    spec[R - 2] = 0.0                                    # silent rows: masked out of the loss
    obs = {"spectrogram": spec, "pointgoal_with_gps_compass": fx.ints(tag + ".pg", (R, 2), 9).float() - 4.0}
    if distractor:
        cat = torch.zeros(R, 21)
        cat[torch.arange(R), fx.ints(tag + ".cat", (R,), 21)] = 1.0
        obs["category"] = cat
    return obs


def main():
    ns = rh.load()
    for name, distractor in (("belief_train", False), ("belief_train_distractor", True)):
        cfg = types.SimpleNamespace(use_label_belief=False, use_location_belief=True, online_training=True,
                                    current_pred_only=False, weighting_factor=0.5)
        bp = ns.BeliefPredictor(cfg, "cpu", None, None, 512, num_env=2, has_distractor_sound=distractor)
        load_fixture_weights(bp, ("belief_loc_distractor." if distractor else "belief_loc."))
        bp.optimizer = torch.optim.Adam(bp.predictor.parameters(), lr=1e-3)            # ddppo_trainer.py:123-129
        R = 6
        losses, correct, nsample, preds_all = [], 0.0, 0.0, []
        for step in range(3):
            obs_batch = batch(f"{name}.{step % 2}", R, distractor)
            bp.optimizer.zero_grad()
            preds = bp.cnn_forward(obs_batch)
            masks = (torch.sum(torch.reshape(obs_batch["spectrogram"], (R, -1)), dim=1, keepdim=True) != 0).float()
            gts = obs_batch["pointgoal_with_gps_compass"]
            tg = torch.stack([gts[:, 1], -gts[:, 0]], dim=1)
            loss = bp.regressor_criterion(masks.expand_as(preds) * preds, masks.expand_as(tg) * tg)
            loss.backward()
            bp.optimizer.step()
            losses.append(loss.item())
            rp = torch.round(preds)
            close = torch.bitwise_and(torch.isclose(rp[:, 0], tg[:, 0]), torch.isclose(rp[:, 1], tg[:, 1]))
            correct += float(torch.sum(torch.bitwise_and(close, masks.bool().squeeze(1))))
            nsample += float(torch.sum(masks))
            preds_all.append(preds.detach().clone())
        sd = bp.predictor.state_dict()
        keys = sorted(sd)
        save(name, losses=np.array(losses), correct=correct, nsample=nsample, preds=torch.stack(preds_all),
             param_abs=np.array([float(sd[k].double().abs().sum()) for k in keys]),
             conv1_w=sd["conv1.weight"][:2, :2], fc_w=sd["fc.weight"][:, :16], l4_w=sd["layer4.1.conv2.weight"][:2, :4, 1, 1],
             bn_g=sd["layer2.0.downsample.1.weight"])


if __name__ == "__main__":
    torch.set_num_threads(8)
    main()
