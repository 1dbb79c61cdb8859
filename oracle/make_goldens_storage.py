"""Goldens for the replay / dialog pre-training side of the storage and the eval-time `pop_at`, from the REFERENCE's own
RolloutStorage / ExternalMemory (rollout_storage.py:300-371, 414-588, 943-956; build container only):
    python oracle/make_goldens_storage.py
Stores outputs only (the 17-tuple of dialog_batching after two insert_replay calls, mask traces, non-GAE returns)."""
import os
import sys
import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import fixtures as fx          # noqa: E402
import ref_harness as rh       # noqa: E402
from make_goldens import save  # noqa: E402


def replay_episode(e, T, dg, dd):
    """One stored dialog episode of environment `e` (what ppo_trainer.py:913-951 hands to insert_replay)."""
    t = f"rep{e}"
    return dict(
        obs={"pose": fx.sym(t + ".pose", (T, 4), 3.0), "spectrogram": fx.uni(t + ".spec", (T, 5, 3, 2))},
        h=fx.sym(t + ".h", (T, 1, 8), 0.5), actions=fx.ints(t + ".a", (T, 1), 4), actions_option=fx.ints(t + ".ao", (T, 1), 2),
        logp=fx.sym(t + ".lp", (T, 1)), values=fx.sym(t + ".v", (T, 1)), rewards=fx.sym(t + ".r", (T, 1)),
        masks=torch.from_numpy((fx.unit(t + ".m", T) >= 0.3).astype("float32")).view(T, 1),
        masks_vln=torch.from_numpy((fx.unit(t + ".mv", T) >= 0.3).astype("float32")).view(T, 1),
        em=fx.sym(t + ".em", (T, dg)), emd=fx.sym(t + ".emd", (T, dd)), dialog=fx.ints(t + ".d", (T, 7), 100),
        o_action=fx.ints(t + ".oa", (T,), 4).float(), o_mask=fx.ints(t + ".om", (T,), 2), prob=fx.uni(t + ".p", (T, 4)),
        qs=fx.sym(t + ".qs", (T, 32)), astep=fx.ints(t + ".as", (T,), 3).float())


def main():
    ns = rh.load()
    T, N, dg, dd = 3, 2, 6, 5
    osp = rh.ObsSpace({"pose": ns.Box(shape=(4,)), "spectrogram": ns.Box(shape=(5, 3, 2))})
    st = ns.RolloutStorage(T, N, osp, rh.ActionSpace(4), 8, True, 3, 3, 3, 3, 3, 3, dg, dg, dg, dd, num_recurrent_layers=1,
                           max_dialog_len=7, use_state_memory=True)
    for e in range(N):
        ep = replay_episode(e, T, dg, dd)
        st.insert_replay(ep["obs"], ep["h"], ep["actions"], ep["actions_option"], ep["logp"], ep["values"], ep["rewards"],
                         ep["masks"], ep["masks_vln"], ep["em"], ep["emd"], ep["dialog"], ep["o_action"], ep["o_mask"], ep["prob"],
                         ep["qs"], ep["astep"])
    assert st.env_id == N and st.step == T
    st.compute_returns(fx.sym("rep.nv", (N, 1)), False, 0.99, 0.95)
    out = st.dialog_batching()
    names = ["obs", "h", "actions", "prev_actions", "value_preds", "returns", "masks", "log_probs", "em", "em_vln", "em_dialog",
             "em_masks", "em_vln_masks", "all_dialog", "agent_step", "num_steps", "num_envs"]
    arrs = {}
    for n, v in zip(names, out):
        if n == "obs":
            for k, x in v.items():
                arrs["obs_" + k] = x
        else:
            arrs[n] = v if torch.is_tensor(v) else np.asarray(v)
    arrs.update(o_masks=st.o_masks, o_actions=st.o_actions, action_probs=st.action_probs, returns_full=st.returns)
    save("storage_replay", **arrs)

    em = ns.ExternalMemory(3, 6, 3, 4, num_copies=2, num_steps=2)
    for t in range(5):
        nd = torch.from_numpy((fx.unit(f"pop.nd{t}", 3) >= 0.2).astype("float32")).view(3, 1)
        em.insert(fx.sym(f"pop.f{t}", (3, 4)), nd)
    em.pop_at(1)
    save("extmem_pop", masks=em.masks, memory=em.memory[:, 0], idx=em.idx)


if __name__ == "__main__":
    main()
