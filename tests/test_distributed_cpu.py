"""N>1 path on CPU (gloo, world_size 2), after habitat-lab-dialog/test/test_ddppo_reduce.py:26-126: per-rank seeds give different
initial weights, `init_distributed` makes them one model (DDP's constructor broadcast, ddppo.py:61-84); the flat gradient
layout (trained parameters first, parameters the loss never reaches outside the reduced range -- the reference's
`find_unused_params=True` case) is averaged by ONE all-reduce and is identical on every rank afterwards; and the
distributed advantage statistics (ddppo.py:22-59)."""
import os
import socket
import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from avlen_amd import policy as P
    from avlen_amd.engine import FlatParams
    from avlen_amd.ppo import DDPPO, distributed_mean_and_var
    from avlen_amd.spaces import savi_observation_space, ActionSpace, SMT_KW
    torch.manual_seed(100 + rank)                       # ddppo_trainer.py:540-548: SEED + rank * NUM_PROCESSES
    pol = P.AudioNavOptionPolicy(savi_observation_space(), ActionSpace(4), pretraining=True, use_category_input=False,
                                 query_count_emb_size=32, **SMT_KW)
    probe = "net.smt_state_encoder.fusion_encoder.0.weight"
    before = pol.state_dict()[probe].clone()
    agent = DDPPO(pol, 0.2, 2, 2, 0.5, 0.05, lr=2.5e-4, eps=1e-5, max_grad_norm=0.2, use_normalized_advantage=True)
    agent.init_distributed(find_unused_params=True)
    assert agent._distributed
    # one model on every rank after init
    sums = torch.tensor([float(p.double().abs().sum()) for p in pol.parameters()], dtype=torch.float64)
    gathered = [torch.zeros_like(sums) for _ in range(world)]
    dist.all_gather(gathered, sums)
    same_after_init = all(bool(torch.equal(g, gathered[0])) for g in gathered)
    changed = bool((pol.state_dict()[probe] != before).any())           # rank != 0 received rank 0's weights
    # the flat layout the HIP update reduces: trained range first
    flat = FlatParams(pol, pol.TRAINED_PREFIXES)
    trained = set(flat.trained_names)
    names = [n for n, _ in pol.named_parameters()]
    layout_ok = all((flat.offsets[n][0] < flat.n_trained) == (n in trained) for n in names)
    unused = [n for n in names if n not in trained]
    assert "net.policy_selector.weight" in unused and "action_distribution_goal.linear.weight" in unused
    assert "net.visual_encoder.rgb_encoder.conv1.weight" in unused           # policy.py:1035-1036: encoders get no gradient
    torch.manual_seed(7 + rank)
    g_local = torch.randn(flat.n_trained)
    flat.grad.copy_(g_local)
    agent.reduce_gradients(flat)
    named = dict(pol.named_parameters())
    views_ok = all(named[n].grad is not None and named[n].grad.data_ptr() == flat.grad_view(n, named[n].shape).data_ptr()
                   for n in trained) and all(named[n].grad is None for n in unused)
    vals = torch.randn(50, 3, 1) + rank

    class Ro:
        returns = torch.cat([vals, torch.zeros(1, 3, 1)])
        value_preds = torch.zeros(51, 3, 1)
    adv = agent._get_advantages_distributed(Ro)
    mean, var = distributed_mean_and_var(vals)
    np.savez(os.path.join(out_dir, f"r{rank}.npz"), g=g_local.numpy(), red=flat.grad.numpy(), vals=vals.numpy(),
             adv=adv.numpy(), mean=mean.numpy(), var=var.numpy(), same=same_after_init, changed=changed, layout=layout_ok,
             views=views_ok, n_trained=flat.n_trained, n_unused=len(unused))
    dist.destroy_process_group()


def test_two_rank_broadcast_gradient_average_and_advantage_stats(tmp_path):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r = [dict(np.load(tmp_path / f"r{i}.npz")) for i in range(2)]
    assert all(bool(x["same"]) for x in r) and not bool(r[0]["changed"]) and bool(r[1]["changed"])
    assert all(bool(x["layout"]) and bool(x["views"]) for x in r)
    assert int(r[0]["n_trained"]) >= 1_204_320 + 1_285 and int(r[0]["n_unused"]) == 145     # 145 of 191 tensors get no gradient
    avg = (r[0]["g"] + r[1]["g"]) / 2
    for i in range(2):
        np.testing.assert_allclose(r[i]["red"], avg, rtol=1e-6, atol=1e-7)       # identical on every rank
    assert np.array_equal(r[0]["red"], r[1]["red"])
    allv = np.concatenate([r[0]["vals"], r[1]["vals"]])
    np.testing.assert_allclose(r[0]["mean"], allv.mean(), rtol=1e-5)
    np.testing.assert_allclose(r[0]["var"], allv.var(), rtol=1e-4)
    np.testing.assert_allclose(r[1]["adv"], (r[1]["vals"] - allv.mean()) / (np.sqrt(allv.var()) + 1e-5), rtol=1e-4,
                               atol=1e-5)
