"""N>1 path on CPU (gloo, world_size 2): the gradient all-reduce/average of the flat buffer and the distributed
advantage statistics (reference pattern: habitat-lab-dialog/test/test_ddppo_reduce.py:26-126)."""
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


class _Flat:
    def __init__(self, g):
        self.grad = g


def _worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from avlen_amd.ppo import DecentralizedDistributedMixin, distributed_mean_and_var

    class Agent(DecentralizedDistributedMixin):
        use_normalized_advantage = True
    ag = Agent()
    ag.init_distributed(find_unused_params=True)
    assert ag._distributed
    torch.manual_seed(100 + rank)
    g = torch.randn(1000)
    flat = _Flat(g.clone())
    ag.reduce_gradients(flat)
    vals = torch.randn(50, 3, 1) + rank

    class Ro:
        returns = torch.cat([vals, torch.zeros(1, 3, 1)])
        value_preds = torch.zeros(51, 3, 1)
    adv = ag._get_advantages_distributed(Ro)
    mean, var = distributed_mean_and_var(vals)
    np.savez(os.path.join(out_dir, f"r{rank}.npz"), g=g.numpy(), red=flat.grad.numpy(), vals=vals.numpy(),
             adv=adv.numpy(), mean=mean.numpy(), var=var.numpy())
    dist.destroy_process_group()


def test_two_rank_gradient_average_and_advantage_stats(tmp_path):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r = [dict(np.load(tmp_path / f"r{i}.npz")) for i in range(2)]
    avg = (r[0]["g"] + r[1]["g"]) / 2
    for i in range(2):
        np.testing.assert_allclose(r[i]["red"], avg, rtol=1e-6, atol=1e-7)       # identical on every rank
    allv = np.concatenate([r[0]["vals"], r[1]["vals"]])
    np.testing.assert_allclose(r[0]["mean"], allv.mean(), rtol=1e-5)
    np.testing.assert_allclose(r[0]["var"], allv.var(), rtol=1e-4)
    np.testing.assert_allclose(r[1]["adv"], (r[1]["vals"] - allv.mean()) / (np.sqrt(allv.var()) + 1e-5), rtol=1e-4,
                               atol=1e-5)
