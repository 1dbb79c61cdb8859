"""Drop-in mirror of ss_baselines/savi/ppo/ppo.py (PPO :28-303) and
ss_baselines/savi/ddppo/algo/ddppo.py (DecentralizedDistributedMixin / DDPPO :49-100).

`PPO.update(rollouts)` keeps the reference's contract (same minibatch order from the host RNG, same loss,
clip-norm 0.2, Adam, same 6-tuple) but every minibatch step is a fixed sequence of HIP launches:
gather -> encoders -> SMT forward (activations kept in the workspace) -> fused PPO-loss + heads backward ->
SMT backward -> [RCCL all-reduce of the flat gradient] -> grad-norm + clipped Adam on the flat buffers.
No autograd graph is built: the gradient reaches exactly the parameters it reaches in the reference
(SMT state encoder + option heads; policy.py:1035-1036 detaches the encoders).
"""
import ctypes as C
import torch
import torch.nn as nn
import torch.distributed as distrib

from . import _lib as L
from . import engine as E

EPS_PPO = 1e-5


class PPO(nn.Module):
    def __init__(self, actor_critic, clip_param, ppo_epoch, num_mini_batch, value_loss_coef, entropy_coef, lr=None,
                 eps=None, max_grad_norm=None, use_clipped_value_loss=True, use_normalized_advantage=True,
                 unct_coef=0.5):
        super().__init__()
        assert use_clipped_value_loss, "the HIP loss kernel implements the clipped value loss (reference default)"
        self.actor_critic = actor_critic
        self.clip_param, self.ppo_epoch, self.num_mini_batch = clip_param, ppo_epoch, num_mini_batch
        self.value_loss_coef, self.entropy_coef, self.unct_coef = value_loss_coef, entropy_coef, unct_coef
        self.max_grad_norm, self.use_clipped_value_loss = max_grad_norm, use_clipped_value_loss
        self.use_normalized_advantage = use_normalized_advantage
        # param_groups holder for lr schedulers (trainer: LambdaLR(self.agent.optimizer)); .step() is never used
        self.optimizer = torch.optim.Adam(actor_critic.parameters(), lr=lr, eps=eps)
        self.dialog_optimizer = torch.optim.Adam(actor_critic.parameters(), lr=.00001, eps=eps)
        self.device = next(actor_critic.parameters()).device
        self._adam = None
        self._distributed = False

    def forward(self, *x):
        raise NotImplementedError

    # ------------------------------------------------------------------------------------------------
    def get_advantages(self, rollouts):
        adv = rollouts.returns[:-1] - rollouts.value_preds[:-1]
        if not self.use_normalized_advantage:
            return adv
        return (adv - adv.mean()) / (adv.std() + EPS_PPO)

    def _adam_state(self, flat):
        if self._adam is None or self._adam["m"].numel() != flat.n_trained or self._adam["m"].device != flat.flat.device:
            dev = flat.flat.device
            self._adam = {"m": torch.zeros(flat.n_trained, device=dev), "v": torch.zeros(flat.n_trained, device=dev),
                          "step": 0, "norm_sq": torch.zeros(1, dtype=torch.float64, device=dev)}
        return self._adam

    def _grad_views(self, eng):
        if "smt_grad" not in eng:
            pol = self.actor_critic
            flat, p2n = eng["flat"], eng["ptr2name"]
            eng["smt_grad"] = E.grad_struct_like(eng["smt"], p2n, flat)
            eng["heads_option_grad"] = E.grad_struct_like(pol._heads("option"), p2n, flat)
        return eng["smt_grad"], eng["heads_option_grad"]

    def before_backward(self, loss):
        pass

    def after_backward(self, loss):
        pass

    def reduce_gradients(self, flat):
        """Hook for the distributed mixin (K24)."""
        pass

    def _minibatch_step(self, rollouts, b, loss_row):
        pol = self.actor_critic
        net = pol.net
        eng = pol._engine()
        flat = eng["flat"]
        smt_g, heads_g = self._grad_views(eng)
        st = L.stream()
        flat.grad.zero_()
        x_att, _, _ = net.run(pol, b["obs"], None, b["prev_actions"], b["masks"], rollouts.em_option.memory,
                              b["em_masks"], b["query_state"], b["last_query_info"], mem_index=b["mem_index"],
                              save_key="smt_train", save=True)
        feats, goal, (ws, nb, B, M, F, cto) = net._last
        R, d = x_att.shape
        dev = x_att.device
        norm = torch.empty(2, device=dev)
        d_feats = torch.empty(R, d, device=dev)
        L.call("avlen_rl_mask_norm", E.P(b["rl_masks"]), R, E.P(norm), st)
        heads = pol._heads("option")
        L.call("avlen_ppo_loss_heads_bwd", C.byref(heads), C.byref(heads_g), E.P(x_att), d, pol.dim_actions_option,
               E.P(b["actions_option"]), E.P(b["old_log_probs"]), E.P(b["adv"]), E.P(b["rl_masks"]),
               E.P(b["value_preds"]), E.P(b["returns"]), E.P(b["ucnt_gt"]), E.P(norm), float(self.clip_param),
               float(self.value_loss_coef), float(self.entropy_coef), float(self.unct_coef), E.P(loss_row),
               E.P(d_feats), R, st)
        L.call("avlen_smt_bwd", C.byref(eng["smt"]), C.byref(smt_g), E.P(goal), E.P(d_feats), B, M, F, net._x_dims - 4,
               cto, pol.prec, E.P(ws), nb, st)
        self.reduce_gradients(flat)
        ad = self._adam_state(flat)
        ad["step"] += 1
        ad["norm_sq"].zero_()
        lr = self.optimizer.param_groups[0]["lr"]
        eps = self.optimizer.param_groups[0]["eps"]
        L.call("avlen_grad_sumsq", E.P(flat.grad), flat.n_trained, E.P(ad["norm_sq"]), st)
        L.call("avlen_adam_step", E.P(flat.flat), E.P(flat.grad), E.P(ad["m"]), E.P(ad["v"]), flat.n_trained, float(lr),
               0.9, 0.999, float(eps), ad["step"], float(self.max_grad_norm), E.P(ad["norm_sq"]), st)
        flat.refresh16(trained_only=True)                 # bf16 shadows of the updated weights (rollout fast path)

    def update(self, rollouts):
        advantages = self.get_advantages(rollouts).contiguous()
        N = rollouts.rewards.size(1)
        assert N >= self.num_mini_batch, (
            "Trainer requires the number of processes ({}) to be greater than or equal to the number of "
            "trainer mini batches ({}).".format(N, self.num_mini_batch))
        per = N // self.num_mini_batch
        n_updates = 0
        log = torch.zeros(self.ppo_epoch * ((N + per - 1) // per), 6, device=rollouts.rewards.device)
        for _ in range(self.ppo_epoch):
            perm = torch.randperm(N)                                   # host RNG, same draw as the reference
            for start in range(0, N, per):
                env = perm[start:start + per].to(log.device)
                b = rollouts.gather_minibatch(env, advantages, in_place=self.actor_critic.precision == "bf16")
                self._minibatch_step(rollouts, b, log[n_updates])
                n_updates += 1
        s = log[:n_updates].double().sum(0).cpu()                      # one sync per update()
        num_updates = self.ppo_epoch * self.num_mini_batch
        return (float(s[0]) / num_updates, float(s[1]) / num_updates, float(s[2]) / num_updates, float(s[3]),
                float(s[4]), float(s[5]) / num_updates)

    def update_dialog(self, rollouts):
        raise NotImplementedError("dialog pre-training (ppo.py:99-154) is outside the accelerated path (SURVEY §8)")

    def before_step(self):
        pass

    def after_step(self):
        pass


def distributed_mean_and_var(values):
    """ddppo.py:22-46."""
    assert distrib.is_initialized(), "Distributed must be initialized"
    world_size = distrib.get_world_size()
    mean = values.mean()
    distrib.all_reduce(mean)
    mean /= world_size
    sq_diff = (values - mean).pow(2).mean()
    distrib.all_reduce(sq_diff)
    var = sq_diff / world_size
    return mean, var


class DecentralizedDistributedMixin:
    """ddppo.py:49-96.  The reference borrows DDP's bucketed reducer; here the trained gradient is ONE
    contiguous fp32 range (4.8 MB for pi_q), reduced by a single RCCL all-reduce per optimiser step and
    averaged.  Parameters the loss never reaches hold no gradient on any rank (no find_unused bookkeeping)."""

    def _get_advantages_distributed(self, rollouts):
        adv = rollouts.returns[:-1] - rollouts.value_preds[:-1]
        if not self.use_normalized_advantage:
            return adv
        mean, var = distributed_mean_and_var(adv)
        return (adv - mean) / (var.sqrt() + EPS_PPO)

    def init_distributed(self, find_unused_params=True):
        """ddppo.py:61-84.  The reference wraps the policy in DistributedDataParallel here, whose constructor broadcasts rank 0's
        parameters and buffers: replicas seeded per rank (ddppo_trainer.py:540-548) start from ONE model.  Same here, on the
        flat buffer."""
        # a 1-rank group still goes through the collective (as DDP's reducer does): the RCCL path is the same code at any size
        self._distributed = distrib.is_available() and distrib.is_initialized()
        self.get_advantages = self._get_advantages_distributed
        if self._distributed:
            self.broadcast_parameters()

    def broadcast_parameters(self, src=0):
        pol = self.actor_critic
        if next(pol.parameters()).is_cuda:
            flat = pol._engine()["flat"]
            distrib.broadcast(flat.flat, src=src)          # every parameter is a view into this buffer
            for b in pol.buffers():
                distrib.broadcast(b, src=src)
            pol.mark_params_changed()                      # packed conv weights / bf16 shadows are derived data
            pol._engine()
        else:                                              # host-side rehearsal (gloo tests): no engine on a CPU
            for t in list(pol.parameters()) + list(pol.buffers()):
                distrib.broadcast(t.data, src=src)

    def reduce_gradients(self, flat):
        if getattr(self, "_distributed", False):
            distrib.all_reduce(flat.grad)
            flat.grad.mul_(1.0 / distrib.get_world_size())


class DDPPO(DecentralizedDistributedMixin, PPO):
    pass
