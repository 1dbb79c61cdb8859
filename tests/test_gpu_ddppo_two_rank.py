"""A REAL two-rank DDPPO.update on one GPU, after habitat-lab-dialog/test/test_ddppo_reduce.py:26-126 (two spawned workers,
TCP rendezvous on 127.0.0.1, backend gloo -- the reference's own backend, savi_interactive_1st_stage.yaml:85 -- random rollouts,
one update, gradients / parameters compared across ranks) and ss_baselines/savi/ddppo/algo/ddppo.py:61-96:

* per-rank seeds give different initial weights and different observations; `init_distributed` makes the replicas one model;
* rank 1's rollout is PRE-EMPTED (ddppo_trainer.py:952-961: a straggler abandons its rollout once most ranks are done), so its
  storage reaches compute_returns / the minibatch gather with `rollouts.step < T` (rollout_storage.py:398, 654);
* every optimiser step all-reduces the flat gradient: the reduced gradient is the mean of the two local gradients, bit-equal on
  both ranks, each local gradient equals the one a single-process run of that rank's share computes, and the parameters after the
  update (2 epochs x 2 minibatches) are bit-equal on both ranks;
* the pre-empted rank's returns equal the oracle's GAE over `step` steps.
Plus: compute_returns of a storage pre-empted at step 97 of 150 against the reference-generated `gae_short` golden."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

T_FULL, T_PRE, N = 6, 4, 4
KW = dict(spectrogram=(65, 26, 2), precision="fp32", pretraining=False, em_capacity=3, use_graphs=False, share_encoders=False,
          launch_ahead=False, with_goal_policy=False, with_dialog_policy=False)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run_share(rank, weight_seed, record):
    """One rank's share: rollout (pre-empted on rank 1), then update() with the local / reduced gradients of every optimiser step
    recorded.  Same code in the workers (process group up) and in the single-process re-run (no group)."""
    from avlen_amd.harness import Workload
    wl = Workload(N, T_FULL, seed=10 + rank, weight_seed=weight_seed, **KW)
    flat = wl.pi_q._engine()["flat"]
    record["w_after_init"] = flat.flat[:flat.n_trained].clone()
    torch.manual_seed(50 + rank)                         # the action sampling of this rank's environments
    for _ in range(T_FULL if rank == 0 else T_PRE):      # rank 1 is pre-empted: rollouts.step = 4 < T = 6
        wl.rollout_step()
    ro = wl.rollouts
    assert ro.step == (T_FULL if rank == 0 else T_PRE)
    steps = ro.step
    saved = dict(rewards=ro.rewards.clone(), masks=ro.masks.clone())
    local, reduced = [], []
    inner = wl.agent.reduce_gradients

    def hooked(fl):
        local.append(fl.grad.clone())
        inner(fl)
        reduced.append(fl.grad.clone())
    wl.agent.reduce_gradients = hooked
    torch.manual_seed(1234)                              # the epoch permutations (same environment count on both ranks)
    out = wl.update()
    torch.cuda.synchronize()
    assert len(local) == 4                               # 2 epochs x 2 minibatches
    record.update(out=np.array(out), local=torch.stack(local).cpu().numpy(), reduced=torch.stack(reduced).cpu().numpy(),
                  params=flat.flat[:flat.n_trained].cpu().numpy(), returns=ro.returns.cpu().numpy(),
                  value_preds=ro.value_preds.cpu().numpy(), rewards=saved["rewards"].cpu().numpy(),
                  masks=saved["masks"].cpu().numpy(), steps=steps, w_after_init=record["w_after_init"].cpu().numpy())
    return record


def _worker(rank, world, port, out_dir):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    try:
        rec = _run_share(rank, weight_seed=100 + rank, record={})      # per-rank seeds (ddppo_trainer.py:540-548)
        np.savez(os.path.join(out_dir, f"r{rank}.npz"), **rec)
    finally:
        dist.destroy_process_group()


def test_two_rank_update_with_a_preempted_rollout(tmp_path):
    import restate as R
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r = [dict(np.load(tmp_path / f"r{i}.npz")) for i in range(2)]
    # one model after init_distributed (rank 0's), and still one model after the update: bit for bit
    assert np.array_equal(r[0]["w_after_init"], r[1]["w_after_init"])
    assert np.array_equal(r[0]["params"], r[1]["params"])
    assert not np.array_equal(r[0]["params"], r[0]["w_after_init"])
    assert int(r[0]["steps"]) == T_FULL and int(r[1]["steps"]) == T_PRE
    # every optimiser step: reduced == mean of the local gradients, identical on both ranks
    assert np.array_equal(r[0]["reduced"], r[1]["reduced"])
    assert not np.allclose(r[0]["local"], r[1]["local"])
    np.testing.assert_allclose(r[0]["reduced"], 0.5 * (r[0]["local"] + r[1]["local"]), rtol=1e-6, atol=1e-9)
    # a single-process run of each rank's share on rank 0's weights computes the same first local gradient (later steps
    # follow the AVERAGED gradient, which a single share cannot reproduce)
    for rank in range(2):
        solo = _run_share(rank, weight_seed=100, record={})
        assert np.array_equal(solo["w_after_init"], r[0]["w_after_init"])
        np.testing.assert_allclose(solo["local"][0], r[rank]["local"][0], rtol=1e-5, atol=1e-8)
        np.testing.assert_allclose(solo["returns"][:int(solo["steps"])], r[rank]["returns"][:int(solo["steps"])], rtol=1e-6, atol=1e-7)
    # the pre-empted rank's returns: GAE over `step` = 4 steps (rollout_storage.py:394-405 with self.step), next value =
    # value_preds[step] as compute_returns stored it
    x = r[1]
    s = int(x["steps"])
    t = lambda a: torch.from_numpy(a)
    ret, _ = R.gae_returns(t(x["rewards"]), t(x["value_preds"]), t(x["masks"]), t(x["value_preds"][s]), 0.99, 0.95, steps=s)
    np.testing.assert_allclose(x["returns"][:s], ret[:s].numpy(), rtol=1e-5, atol=1e-6)
    assert np.all(np.isfinite(r[0]["out"])) and np.all(np.isfinite(r[1]["out"]))


def test_preempted_storage_returns_match_gae_short_golden():
    """RolloutStorage.compute_returns on the HIP path with rollouts.step = 97 < T = 150 against the reference's own GAE over a
    pre-empted rollout (golden `gae_short`, oracle/make_goldens.py)."""
    import fixtures as fx
    from conftest import golden
    from avlen_amd.rollout_storage import RolloutStorage
    from avlen_amd.spaces import savi_observation_space, ActionSpace
    T, Nn, S = 150, 4, 97
    ro = RolloutStorage(T, Nn, savi_observation_space((65, 26, 2)), ActionSpace(4), 512, True, 8, 4, 8, 4, 3, 3, 276, 276, 308, 256,
                        num_recurrent_layers=-1, max_dialog_len=77, use_state_memory=True, device=torch.device("cuda"))
    ro.rewards.copy_(fx.sym("gae.r", (T, Nn, 1)))
    ro.value_preds.copy_(fx.sym("gae.v", (T + 1, Nn, 1)))
    ro.masks.copy_(torch.from_numpy((fx.unit("gae.m", (T + 1) * Nn) >= 1 / 15).astype("float32")).view(T + 1, Nn, 1))
    ro.step = S
    ro.compute_returns(fx.sym("gae.nv", (Nn, 1)).cuda(), True, 0.99, 0.95)
    torch.cuda.synchronize()
    np.testing.assert_allclose(ro.returns.cpu().numpy()[:S], golden("gae_short")["returns"][:S], rtol=1e-5, atol=1e-6)
    # the minibatch of a pre-empted storage holds step * n_mb rows (rollout_storage.py:654)
    b = ro.gather_minibatch(torch.tensor([0, 2], device="cuda"))
    assert b["T"] == S and b["returns"].shape[0] == S * 2
    np.testing.assert_allclose(b["returns"].view(S, 2, 1).cpu().numpy(), ro.returns[:S][:, [0, 2]].cpu().numpy())
