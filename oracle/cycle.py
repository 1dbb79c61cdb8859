"""Synthetic driver inputs for one rollout->update cycle (TEST INFRASTRUCTURE).

Both the golden generator (reference classes) and the oracle / product tests draw the
per-step "environment" quantities from here so they see identical streams.
"""
import torch
import fixtures as fx
import restate as R

PE = R.sinusoid_table(1000, 32)        # ddppo_trainer.py:506-512 (query-count / last-query encodings)


def step_inputs(t, N, spectrogram=(65, 26), tag="cyc"):
    """Everything the simulator + trainer bookkeeping would have produced for step t."""
    u = fx.unit(f"{tag}.done{t}", N)
    not_done = torch.from_numpy((u >= 0.15).astype("float32")).view(N, 1)
    rl = torch.from_numpy((fx.unit(f"{tag}.rl{t}", N) < 0.7).astype("int64"))
    rl[0] = 1
    return {
        "next_obs": fx.observations(f"{tag}.obs{t + 1}", N, spectrogram, step=t + 1),
        "actions": fx.ints(f"{tag}.act{t}", (N, 1), 4),
        "rewards": fx.sym(f"{tag}.rew{t}", (N, 1), 1.0),
        "not_done": not_done,
        "rl_masks": rl,
        "ucnt_gt": fx.ints(f"{tag}.ug{t}", (N,), 2),
        "query_state": PE[fx.ints(f"{tag}.qc{t}", (N,), 4)],
        "last_query_info": PE[fx.ints(f"{tag}.lq{t}", (N,), 150)],
        "agent_step": fx.ints(f"{tag}.as{t}", (N,), 3).float(),
    }


def first_obs(N, spectrogram=(65, 26), tag="cyc"):
    return fx.observations(f"{tag}.obs0", N, spectrogram, step=0)
