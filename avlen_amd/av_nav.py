"""Drop-in mirrors of the two classes that train the GRU baseline policy in the reference (BASELINE configs[1]):

* `RolloutStorage` -- ss_baselines/common/rollout_storage.py:16-235 (the plain recurrent storage: no external memories);
* `PPO`            -- ss_baselines/av_nav/ppo/ppo.py:16-165 (`update` -> 3-tuple; action loss is a plain mean, no uncertainty head).

The savi trainers cannot drive `policy_type: 'rnn'`; the av_nav call pattern is `act -> insert -> get_value ->
compute_returns -> update`, with the savi `AudioNavBaselinePolicy` (category input, 6-tuple `act`).  `PPO.update` keeps the reference's
contract (minibatch permutation from the host generator, same loss, clip-norm, Adam over every parameter that receives a gradient)
and runs each minibatch as a fixed sequence of HIP launches:
gather -> avlen_baseline_train_fwd (CNNs + masked GRU, activations kept) -> avlen_ppo_loss_heads_bwd -> avlen_baseline_train_bwd
(BPTT + conv/Linear backward) -> [RCCL all-reduce] -> grad-norm + clipped Adam on the flat buffers -> packed-weight refresh.
"""
from collections import defaultdict
import ctypes as C
import torch
import torch.nn as nn

from . import _lib as L
from . import engine as E
from .engine import P

EPS_PPO = 1e-5


class RolloutStorage:
    def __init__(self, num_steps, num_envs, observation_space, action_space, recurrent_hidden_state_size,
                 num_recurrent_layers=1, device="cuda", uint8_sensors=("rgb",)):
        T, N, dev = num_steps, num_envs, torch.device(device)
        z = lambda *s, **k: torch.zeros(*s, device=dev, **k)
        # RGB stays uint8 (see avlen_amd/rollout_storage.py): a quarter of the bytes to store and to read back
        self.observations = {k: z(T + 1, N, *sp.shape, dtype=torch.uint8 if k in uint8_sensors else torch.float32)
                             for k, sp in observation_space.spaces.items()}
        self.recurrent_hidden_states = z(T + 1, num_recurrent_layers, N, recurrent_hidden_state_size)
        self.rewards, self.value_preds, self.returns = z(T, N, 1), z(T + 1, N, 1), z(T + 1, N, 1)
        self.action_log_probs = z(T, N, 1)
        discrete = action_space.__class__.__name__ == "ActionSpace"
        ashape = 1 if discrete else action_space.shape[0]
        adt = torch.long if discrete else torch.float32
        self.actions, self.prev_actions = z(T, N, ashape, dtype=adt), z(T + 1, N, ashape, dtype=adt)
        self.masks = torch.ones(T + 1, N, 1, device=dev)
        self.num_steps, self.num_envs, self.device, self.step = T, N, dev, 0

    def to(self, device):
        dev = torch.device(device)
        for k, v in list(self.__dict__.items()):
            if torch.is_tensor(v):
                setattr(self, k, v.to(dev))
        self.observations = {k: v.to(dev) for k, v in self.observations.items()}
        self.device = dev

    def insert(self, observations, recurrent_hidden_states, actions, action_log_probs, value_preds, rewards, masks):
        s = self.step
        pairs = [(self.observations[k][s + 1], observations[k]) for k in observations]
        pairs += [(self.recurrent_hidden_states[s + 1], recurrent_hidden_states), (self.actions[s], actions),
                  (self.prev_actions[s + 1], actions), (self.action_log_probs[s], action_log_probs),
                  (self.value_preds[s], value_preds), (self.rewards[s], rewards), (self.masks[s + 1], masks)]
        L.multi_copy(pairs)                                   # one launch for the step's storage writes
        self.step = (s + 1) % self.num_steps

    def after_update(self):
        pairs = [(v[0], v[-1]) for v in self.observations.values()]
        pairs += [(self.recurrent_hidden_states[0], self.recurrent_hidden_states[-1]), (self.masks[0], self.masks[-1]),
                  (self.prev_actions[0], self.prev_actions[-1])]
        L.multi_copy(pairs)

    def compute_returns(self, next_value, use_gae, gamma, tau):
        nv = next_value.detach().float().contiguous()
        T = self.rewards.size(0)
        if use_gae:
            L.call("avlen_gae_scan", P(self.rewards), P(self.value_preds), P(self.masks), P(nv), P(self.returns), None, T,
                   self.num_envs, float(gamma), float(tau), L.stream())
        else:
            L.call("avlen_discounted_returns", P(self.rewards), P(self.masks), P(nv), P(self.returns), T, self.num_envs,
                   float(gamma), L.stream())

    def _gather(self, src, env, T):
        n_mb = env.numel()
        D = 1
        for d in src.shape[2:]:
            D *= d
        dst = torch.empty((T * n_mb,) + tuple(src.shape[2:]), dtype=src.dtype, device=src.device)
        es = src.element_size()
        if es == 1:                                        # uint8 frames move as 4-byte words
            assert D % 4 == 0
            D, es = D // 4, 4
        L.call("avlen_minibatch_gather", P(src), P(dst), P(env), T, self.num_envs, n_mb, D, es, L.stream())
        return dst

    def recurrent_generator(self, advantages, num_mini_batch):
        """common/rollout_storage.py:137-231: the 9-tuple, rows flattened T-major."""
        N = self.rewards.size(1)
        assert N >= num_mini_batch, (
            "Trainer requires the number of processes ({}) to be greater than or equal to the number of "
            "trainer mini batches ({}).".format(N, num_mini_batch))
        per = N // num_mini_batch
        perm = torch.randperm(N)
        T = self.num_steps
        for start in range(0, N, per):
            env = perm[start:start + per].to(self.device)
            g = lambda x: self._gather(x, env, T)
            obs = defaultdict(list)
            for k, v in self.observations.items():
                obs[k] = g(v)
            yield (obs, self.recurrent_hidden_states[0][:, env].contiguous(), g(self.actions), g(self.prev_actions),
                   g(self.value_preds), g(self.returns), g(self.masks), g(self.action_log_probs), g(advantages))


class PPO(nn.Module):
    def __init__(self, actor_critic, clip_param, ppo_epoch, num_mini_batch, value_loss_coef, entropy_coef, lr=None, eps=None,
                 max_grad_norm=None, use_clipped_value_loss=True, use_normalized_advantage=True):
        super().__init__()
        assert use_clipped_value_loss, "the HIP loss kernel implements the clipped value loss (reference default)"
        self.actor_critic = actor_critic
        self.clip_param, self.ppo_epoch, self.num_mini_batch = clip_param, ppo_epoch, num_mini_batch
        self.value_loss_coef, self.entropy_coef = value_loss_coef, entropy_coef
        self.max_grad_norm, self.use_clipped_value_loss = max_grad_norm, use_clipped_value_loss
        # param_groups holder for lr schedulers; .step() is never used (the HIP Adam steps the flat buffer)
        self.optimizer = torch.optim.Adam(actor_critic.parameters(), lr=lr, eps=eps)
        self.device = next(actor_critic.parameters()).device
        self.use_normalized_advantage = use_normalized_advantage
        from .engine import FlatAdam                     # optimizer.state = views into the flat moments (checkpoint round trip)
        self._adam = FlatAdam(self.optimizer, actor_critic, with_norm=True)
        self._distributed = False

    def forward(self, *x):
        raise NotImplementedError

    def get_advantages(self, rollouts):
        adv = rollouts.returns[:-1] - rollouts.value_preds[:-1]
        if not self.use_normalized_advantage:
            return adv
        return (adv - adv.mean()) / (adv.std() + EPS_PPO)

    def reduce_gradients(self, flat):
        pass

    def before_backward(self, loss):
        pass

    def after_backward(self, loss):
        pass

    def before_step(self):
        pass

    def after_step(self):
        pass

    def _adam_state(self, flat):
        return self._adam.state(flat)

    def _minibatch_step(self, sample, loss_row):
        flat = self._forward_backward(sample, loss_row)
        self._optimizer_step(flat)

    def _forward_backward(self, sample, loss_row):
        """Loss and gradient of one minibatch: fills the flat gradient buffer (zeroed first), returns the FlatParams."""
        obs, h0, actions, _prev, value_preds, returns, masks, old_lp, adv = sample
        pol = self.actor_critic
        eng = pol._engine()
        flat = eng["flat"]
        st = L.stream()
        flat.grad.zero_()
        out, ws, dims = pol.net.train_forward(pol, obs, h0, masks)
        R, d = out.shape
        g = pol.grad_views(eng)
        ones, norm = pol._ones(R), torch.empty(2, device=out.device)
        d_out = torch.empty(R, d, device=out.device)
        L.call("avlen_rl_mask_norm", P(ones), R, P(norm), st)
        heads = pol._heads("goal")
        # av_nav loss = the savi loss with rl_masks == 1 and no uncertainty head (ppo.py:97-129 vs savi ppo.py:219-262)
        L.call("avlen_ppo_loss_heads_bwd", C.byref(heads), C.byref(g["heads"]), P(out), d, pol.dim_actions, P(actions), P(old_lp),
               P(adv), P(ones), P(value_preds), P(returns), None, P(norm), float(self.clip_param), float(self.value_loss_coef),
               float(self.entropy_coef), 0.0, P(loss_row), P(d_out), R, st)
        pol.net.train_backward(pol, g, obs, masks, d_out, ws, dims)
        return flat

    def _optimizer_step(self, flat):
        pol, st = self.actor_critic, L.stream()
        self.reduce_gradients(flat)
        ad = self._adam_state(flat)
        step = ad.advance()
        ad.norm_sq.zero_()
        lr, eps = self.optimizer.param_groups[0]["lr"], self.optimizer.param_groups[0]["eps"]
        L.call("avlen_grad_sumsq", P(flat.grad), flat.n_trained, P(ad.norm_sq), st)
        L.call("avlen_adam_step", P(flat.flat), P(flat.grad), P(ad.m), P(ad.v), flat.n_trained, float(lr), 0.9, 0.999,
               float(eps), step, float(self.max_grad_norm), P(ad.norm_sq), st)
        pol.mark_params_changed()                        # conv / fc weights moved: packed copies + bf16 shadows are stale
        pol._engine()

    def update(self, rollouts):
        advantages = self.get_advantages(rollouts).contiguous()
        n = 0
        log = torch.zeros(self.ppo_epoch * (self.num_mini_batch + 1), 6, device=advantages.device)
        for _ in range(self.ppo_epoch):
            for sample in rollouts.recurrent_generator(advantages, self.num_mini_batch):
                self._minibatch_step(sample, log[n])
                n += 1
        s = log[:n].double().sum(0).cpu()                # one sync per update()
        num_updates = self.ppo_epoch * self.num_mini_batch
        return float(s[0]) / num_updates, float(s[1]) / num_updates, float(s[2]) / num_updates


class DDPPO(PPO):
    """Data-parallel variant (ddppo.py:49-96 semantics): parameters broadcast at init, one RCCL all-reduce of the flat gradient
    per optimiser step."""

    def init_distributed(self, find_unused_params=True):
        import torch.distributed as distrib
        self._distributed = distrib.is_available() and distrib.is_initialized()
        if self._distributed:
            pol = self.actor_critic
            flat = pol._engine()["flat"]
            distrib.broadcast(flat.flat, src=0)
            pol.mark_params_changed()
            pol._engine()

    def reduce_gradients(self, flat):
        if self._distributed:
            import torch.distributed as distrib
            distrib.all_reduce(flat.grad)
            flat.grad.mul_(1.0 / distrib.get_world_size())
