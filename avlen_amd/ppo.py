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
from . import config as CFG
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
        # the moments live in flat buffers stepped by ONE HIP launch; `optimizer.state` holds views into them (E.FlatAdam), so the
        # trainer's `optimizer.state_dict()` / `load_state_dict()` checkpoint round trip carries them (ddppo_trainer.py:812-817, 857-862)
        self._adam = E.FlatAdam(self.optimizer, actor_critic, with_norm=True)
        self._adam_dialog = E.FlatAdam(self.dialog_optimizer, actor_critic)
        self._distributed = False
        # Opt-in, NOT the reference's work count: pi_q's update re-runs the frozen encoders on the stored observations exactly as
        # ppo.py:207-262 does unless this is set -- then the visual / audio feature columns are read back from the rows the rollout
        # wrote into the option memory ring (policy.py:1062-1065 `x_for_memory`; policy.py:1035-1036 detaches them: same values,
        # no gradient either way).  bench.py reports it as `update_feature_reuse`, never as the headline.
        self.feature_reuse = False

    @staticmethod
    def set_x3_mixed_backward_rows(rows):
        """precision="bf16x3" only: the number of SMT token rows per minibatch (B x (M + 1)) from which `update` runs the BACKWARD's
        products and attention on plain bf16 operands (the forward -- logits, ratio, losses -- stays compensated).  Default 65536:
        every 2nd-stage minibatch of the AVLEN configs (1200 samples x 151 .. 301 rows) takes it, the 1st stage never does; 0
        switches it off (compensated backward everywhere, 1.7x the update time at 32 envs); a negative value restores the default.
        Held to the oracle by tests/test_gpu_policy_parity.py::test_gradients_bf16x3_vs_oracle_autograd_per_class[mixed] (per tensor
        class) and tests/test_gpu_harness_parity.py::test_benched_harness_mixed_backward_update_matches_oracle (the parameter step).
        Process-wide (a library setting)."""
        L.lib.avlen_set_x3_mixed_backward_rows(int(rows))

    def forward(self, *x):
        raise NotImplementedError

    # ------------------------------------------------------------------------------------------------
    def get_advantages(self, rollouts):
        adv = rollouts.returns[:-1] - rollouts.value_preds[:-1]
        if not self.use_normalized_advantage:
            return adv
        return (adv - adv.mean()) / (adv.std() + EPS_PPO)

    def _adam_state(self, flat):
        return self._adam.state(flat)

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
                              save_key="smt_train", save=True, stored=b.get("stored"))
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
               cto, pol.prec, None, 0, E.P(ws), nb, st)
        self.reduce_gradients(flat)
        ad = self._adam_state(flat)
        step = ad.advance()
        ad.norm_sq.zero_()
        lr = self.optimizer.param_groups[0]["lr"]
        eps = self.optimizer.param_groups[0]["eps"]
        L.call("avlen_grad_sumsq", E.P(flat.grad), flat.n_trained, E.P(ad.norm_sq), st)
        L.call("avlen_adam_step", E.P(flat.flat), E.P(flat.grad), E.P(ad.m), E.P(ad.v), flat.n_trained, float(lr),
               0.9, 0.999, float(eps), step, float(self.max_grad_norm), E.P(ad.norm_sq), st)
        flat.refresh16(trained_only=True)                 # bf16 shadows of the updated weights (rollout fast path)
        eng["packed"].refresh_pads()                      # ... and the padded shadow of a fusion input that is not 8-aligned

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
                b = rollouts.gather_minibatch(env, advantages, in_place=self.actor_critic.precision in ("bf16", "bf16x3"),
                                              feature_rows=self.feature_reuse)
                self._minibatch_step(rollouts, b, log[n_updates])
                n_updates += 1
        s = log[:n_updates].double().sum(0).cpu()                      # one sync per update()
        num_updates = self.ppo_epoch * self.num_mini_batch
        return (float(s[0]) / num_updates, float(s[1]) / num_updates, float(s[2]) / num_updates, float(s[3]),
                float(s[4]), float(s[5]) / num_updates)

    DIALOG_CLASS_WEIGHTS = (0.0, 0.33, 0.33, 0.33)       # weight_type = 'balanced' (ppo.py:69-76)

    def update_dialog(self, rollouts):
        """ppo.py:99-154 (dialog pre-training of pi_l): ONE optimiser step of `dialog_optimizer` on the class-weighted cross
        entropy between the vln action logits of the rows with o_masks != 0 and the oracle actions, over every stored step of
        every environment.  The gradient reaches the dialog encoder, dialog_layer, the SMT state encoder and -- unlike pi_q's
        update -- the visual towers, the AudioCNN and the action encoder (policy.py:807-808 takes the features with grad); the
        CLIP tower is frozen.  No gradient clipping (the reference calls optimizer.step() directly).  Returns the loss as a
        0-dim device tensor."""
        flat, loss = self._dialog_forward_backward(rollouts)
        pol = self.actor_critic
        dev, st = flat.flat.device, L.stream()
        self.reduce_gradients(flat)
        ad = self._adam_dialog.state(flat)
        step = ad.advance()
        pg = self.dialog_optimizer.param_groups[0]
        L.call("avlen_adam_step", E.P(flat.flat), E.P(flat.grad), E.P(ad.m), E.P(ad.v), flat.n_trained, float(pg["lr"]), 0.9,
               0.999, float(pg["eps"]), step, 0.0, None, st)
        pol.mark_params_changed()                 # conv / fc weights moved: packed copies and bf16 shadows follow
        pol._engine()
        return loss

    def _dialog_forward_backward(self, rollouts):
        """Loss and gradient of update_dialog's batch: fills the flat gradient buffer; -> (FlatParams, loss tensor)."""
        pol, net = self.actor_critic, self.actor_critic.net
        T, N = rollouts.step, rollouts.num_envs
        R = T * N
        (obs, _h, _actions, prev_actions, _, _, _masks, _, _, _em_vln, _em_dlg, _em_masks, em_vln_masks, all_dialog, agent_step,
         _, _) = rollouts.dialog_batching(memories=False)
        eng = pol._engine()
        flat = eng["flat"]
        g = pol.grad_views(eng)
        dev, st = flat.flat.device, L.stream()
        flat.grad.zero_()
        # the ring is stored once: row t*N + n reads column n (no (em_size, T*N, dim) copy for the SMT encoder)
        mem_index = torch.arange(N, device=dev, dtype=torch.int32).repeat(T).contiguous()
        emd = rollouts.em_vln_dialog
        memd = emd.memory.unsqueeze(1).expand(-1, T, -1, -1).reshape(emd.total_size, R, emd.dim).contiguous()
        out, sv = net.train_forward(pol, obs, prev_actions, rollouts.em_vln.memory, mem_index, memd, em_vln_masks, all_dialog,
                                    agent_step)
        d = out.shape[1]
        o_actions = rollouts.o_actions[:T].reshape(-1).contiguous()
        o_masks = rollouts.o_masks[:T].reshape(-1).contiguous()
        wcls = torch.tensor(self.DIALOG_CLASS_WEIGHTS, device=dev)
        scratch = torch.zeros(2, device=dev)                      # [norm, loss]
        d_out = torch.empty(R, d, device=dev)
        heads = pol._heads("vln")
        L.call("avlen_dialog_loss_heads_bwd", C.byref(heads), C.byref(g["heads"]), E.P(out), d, pol.dim_actions, E.P(o_actions),
               E.P(o_masks), E.P(wcls), E.P(scratch), E.P(scratch, 1), E.P(d_out), R, st)
        net.train_backward(pol, g, sv, d_out)
        return flat, scratch[1]

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
        self._comm = None
        if self._distributed:
            self.broadcast_parameters()
            if self.native_allreduce and next(self.actor_critic.parameters()).is_cuda and distrib.get_backend() == "nccl":
                self._comm = self._native_comm()

    native_allreduce = CFG.NATIVE_ALLREDUCE      # the C ABI's grad_allreduce instead of torch.distributed.all_reduce (config.py)

    def _native_comm(self):
        """An RCCL communicator of the library's own over the ranks of the default process group: rank 0 draws the unique id
        (avlen_comm_unique_id), torch.distributed carries it to the others, every rank joins (avlen_comm_init_rank)."""
        dev = next(self.actor_critic.parameters()).device
        buf = (C.c_ubyte * 128)()
        if distrib.get_rank() == 0:
            L.call("avlen_comm_unique_id", buf, 128)
        t = torch.tensor(list(buf), dtype=torch.uint8, device=dev)
        distrib.broadcast(t, src=0)
        buf = (C.c_ubyte * 128)(*t.cpu().tolist())
        comm = C.c_void_p()
        L.call("avlen_comm_init_rank", C.byref(comm), distrib.get_world_size(), buf, distrib.get_rank())
        return comm

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
            if getattr(self, "_comm", None) is not None:       # one in-place ncclAllReduce(avg) on the backward's stream
                L.call("avlen_grad_allreduce", E.P(flat.grad), flat.n_trained, L.PREC_FP32, self._comm, L.stream())
                return
            distrib.all_reduce(flat.grad)
            flat.grad.mul_(1.0 / distrib.get_world_size())


class DDPPO(DecentralizedDistributedMixin, PPO):
    pass


CFG.add_ranges(PPO, ('update', 'update_dialog'), "PPO.")
