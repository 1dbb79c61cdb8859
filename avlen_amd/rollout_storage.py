"""Device-resident mirror of ss_baselines/savi/models/rollout_storage.py (RolloutStorage :16-905,
ExternalMemory :907-960).

Same constructor, `insert` (22 arguments), `compute_returns`, `after_update`, `recurrent_generator`
(22-tuple) and `external_memory_*` accessors as the reference, with two MI355X-first changes that do
not alter any value the trainer can observe:

* the reference keeps T+1 IDENTICAL copies of every external memory (rollout_storage.py:924,933 --
  3.6 GB per memory at N=64); here there is ONE copy (total, N, dim) plus the per-step masks: a slot that
  is visible at step t is never overwritten before the rollout ends (total = capacity + T);
* GAE, the ring insert and the minibatch gather are HIP kernels (avlen_gae_scan, avlen_extmem_insert,
  avlen_minibatch_gather); `gather_minibatch` hands avlen_amd.ppo.PPO an index into the ring instead of
  the (300, T*N_mb, dim) tensor the reference materialises per minibatch (K21).
"""
from collections import defaultdict
import ctypes as C
import torch

from . import _lib as L
from . import config as CFG
from .engine import P


class _CopyAxis:
    """`memory[:, step]` of the reference's (total, copies, N, dim) tensor: every copy is the same."""
    def __init__(self, em):
        self._em = em

    def __getitem__(self, idx):
        if isinstance(idx, tuple) and len(idx) == 2 and idx[0] == slice(None):
            return self._em.memory
        raise IndexError("external memory is stored once: index it as memory[:, step]")

    @property
    def shape(self):
        m = self._em.memory
        return (m.shape[0], self._em.num_copies, m.shape[1], m.shape[2])


class _RingTensor(torch.Tensor):
    """The (total, N, dim) ring of an eval-time `ExternalMemory(num_copies=1)`.  The reference keeps (total, copies, N, dim) and
    its eval loop reads `test_em.memory[:, 0]` (ppo_trainer.py:1912-1959): that exact index pattern -- (all slots, an int copy) --
    returns the whole ring here (every copy is identical by construction, rollout_storage.py:924-933); anything else is ordinary
    tensor indexing."""

    @staticmethod
    def __new__(cls, t):
        return torch.Tensor._make_subclass(cls, t)

    def __getitem__(self, idx):
        if isinstance(idx, tuple) and len(idx) == 2 and isinstance(idx[0], slice) and idx[0] == slice(None) and isinstance(idx[1], int):
            if idx[1] not in (0, -1):                    # the reference's tensor has ONE copy: memory[:, 1] raises there too
                raise IndexError(f"index {idx[1]} is out of bounds for the copy axis of an ExternalMemory with 1 copy")
            return self.as_subclass(torch.Tensor)
        r = super().__getitem__(idx)
        # derived views are plain tensors: only the ring itself answers to the copy-axis pattern
        return r.as_subclass(torch.Tensor) if isinstance(r, _RingTensor) else r


class ExternalMemory:
    def __init__(self, num_envs, total_size, capacity, dim, num_copies=1, num_steps=150, device="cpu"):
        self.num_envs, self.total_size, self.capacity, self.dim = num_envs, total_size, capacity, dim
        self.num_copies, self.num_steps = num_copies, num_steps
        self.masks = torch.zeros(num_envs, total_size, device=device)
        self.memory = self._wrap(torch.zeros(total_size, num_envs, dim, device=device))
        self.idx = 0
        self.env_id = 0
        self._masks_replay = None                      # (num_steps, N, total): allocated by the first insert_replay

    def _wrap(self, t):
        # the trainer's own eval-time objects (num_copies=1) answer `memory[:, 0]` like the reference's 4-D tensor; the rollout
        # storage's rings (num_copies = T + 1, reached through RolloutStorage.external_memory_*[:, step]) stay plain tensors
        return _RingTensor(t) if self.num_copies == 1 else t

    @property
    def masks_replay(self):
        if self._masks_replay is None:
            self._masks_replay = torch.zeros(self.num_steps, self.num_envs, self.total_size, device=self.memory.device)
        return self._masks_replay

    def insert_replay(self, em_features):
        """rollout_storage.py:943-952: a whole stored sequence (k, dim) of ONE environment becomes slots 0..k-1 of ring column
        `env_id`; step i of the replayed sequence sees slots 0..i-1."""
        mr = self.masks_replay
        mr[:, self.env_id, :] = 0.0
        k = em_features.size(0)
        self.memory[:k, self.env_id, :].copy_(em_features)
        # step i sees slots 0..i-1: a strictly lower-triangular (num_steps, total) pattern, written in one op
        tri = torch.ones(self.num_steps, self.total_size, device=mr.device).tril(-1)
        mr[:, self.env_id, :] = tri
        self.env_id = (self.env_id + 1) % self.num_envs

    def pop_at(self, idx):
        """rollout_storage.py:954-956 (eval: an environment is paused, base_trainer.py:186-289): drop column `idx`."""
        keep = [i for i in range(self.masks.shape[0]) if i != idx]
        self.masks = self.masks[keep].contiguous()
        self.memory = self._wrap(self.memory.as_subclass(torch.Tensor)[:, keep].contiguous())
        self.num_envs = len(keep)                      # the insert kernel sizes its grid by it
        self._masks_replay = None                      # sized for the old environment count

    def insert(self, em_features, not_done_masks, masks_out=None):
        f = em_features if em_features.is_contiguous() else em_features.contiguous()
        nd = not_done_masks.float().contiguous()
        L.call("avlen_extmem_insert", P(self.memory), P(self.masks), P(f), f.shape[1], P(nd),
               P(masks_out) if masks_out is not None else None, self.idx, self.total_size, self.capacity,
               self.num_envs, self.dim, L.stream())
        self.idx = (self.idx + 1) % self.total_size

    def fill_op(self, op, em_features, not_done_masks, masks_out=None):
        """Describe this ring's insert in `op` (an _lib.ExtMemOp) for a batched `avlen_extmem_insert_multi` launch and advance
        the ring index; returns the tensors that must stay alive until the launch is enqueued."""
        f = em_features if em_features.is_contiguous() else em_features.contiguous()
        nd = not_done_masks if (not_done_masks.dtype == torch.float32 and not_done_masks.is_contiguous()) \
            else not_done_masks.float().contiguous()
        op.memory, op.masks, op.feats, op.ld_feats = self.memory.data_ptr(), self.masks.data_ptr(), f.data_ptr(), f.shape[1]
        op.not_done = nd.data_ptr()
        op.masks_out = masks_out.data_ptr() if masks_out is not None else None
        op.idx, op.total, op.capacity, op.N, op.dim = self.idx, self.total_size, self.capacity, self.num_envs, self.dim
        self.idx = (self.idx + 1) % self.total_size
        return f, nd

    def to(self, device):
        self.masks, self.memory = self.masks.to(device), self._wrap(self.memory.as_subclass(torch.Tensor).to(device))
        if self._masks_replay is not None:
            self._masks_replay = self._masks_replay.to(device)


class RolloutStorage:
    def __init__(self, num_steps, num_envs, observation_space, action_space, recurrent_hidden_state_size,
                 use_external_memory, external_memory_size, external_memory_capacity, external_memory_option_size,
                 external_memory_option_capacity, external_memory_vln_size, external_memory_vln_capacity,
                 external_memory_dim_goal, external_memory_dim_vln, external_memory_dim_option,
                 external_memory_dim_dialog, num_recurrent_layers=1, max_dialog_len=20, query_count_emb_size=32,
                 use_state_memory=False, device="cuda", skip_sensors=("audiogoal",), uint8_sensors=("rgb",)):
        T, N, dev = num_steps, num_envs, torch.device(device)
        z = lambda *s, **k: torch.zeros(*s, device=dev, **k)
        self.num_steps, self.num_envs, self.device = T, N, dev
        # Observation staging (reference: rollout_storage.py:58-63 keeps every sensor as fp32, 404 KB per env-step):
        # * `skip_sensors`: the raw `audiogoal` waveform (2 x 16000 fp32 = 128 KB per env-step) is copied into the reference's
        #   buffers but no network reads it (the goal sensor is `spectrogram`): it is not kept in HBM;
        # * `uint8_sensors`: RGB is uint8 0..255 at the sensor; it stays uint8 here (49 KB instead of 196 KB per stored frame,
        #   and 4x less to read back in the PPO update) and the tower prologue converts / divides exactly as on fp32 pixels.
        #   `insert` accepts uint8 (device, or pinned host: copied asynchronously) or integer-valued fp32 frames.
        self.uint8_sensors = tuple(k for k in uint8_sensors if k in observation_space.spaces)
        self.observations = {k: z(T + 1, N, *sp.shape, dtype=torch.uint8 if k in self.uint8_sensors else torch.float32)
                             for k, sp in observation_space.spaces.items() if k not in skip_sensors}
        if num_recurrent_layers < 1:
            num_recurrent_layers = 1
        self.recurrent_hidden_states = z(T + 1, num_recurrent_layers, N, recurrent_hidden_state_size)
        self.all_dialog = z(T, N, max_dialog_len, dtype=torch.long)
        self.query_state, self.last_query_info = z(T, N, query_count_emb_size), z(T, N, query_count_emb_size)
        self.agent_step = z(T, N)
        self.rewards, self.value_preds, self.returns = z(T, N, 1), z(T + 1, N, 1), z(T + 1, N, 1)
        self.advantages = z(T, N, 1)
        self.action_log_probs = z(T, N, 1)
        discrete = action_space.__class__.__name__ == "ActionSpace"
        ashape = 1 if discrete else action_space.shape[0]
        adt = torch.long if discrete else torch.float32
        self.actions, self.actions_option = z(T, N, ashape, dtype=adt), z(T, N, ashape, dtype=adt)
        self.prev_actions = z(T + 1, N, ashape, dtype=adt)
        self.masks, self.masks_vln = z(T + 1, N, 1), z(T + 1, N, 1)
        self.o_actions = z(T, N)
        self.o_masks, self.ucnt_gt, self.rl_masks = (z(T, N, dtype=torch.long) for _ in range(3))
        self.action_probs = z(T, N, 4)
        self.use_external_memory, self.use_state_memory = use_external_memory, use_state_memory
        self.em_size, self.em_capacity = external_memory_size, external_memory_capacity
        self.em_option_size, self.em_option_capacity = external_memory_option_size, external_memory_option_capacity
        self.em_vln_size, self.em_vln_capacity = external_memory_vln_size, external_memory_vln_capacity
        self.em_dim_goal, self.em_dim_vln = external_memory_dim_goal, external_memory_dim_vln
        self.em_dim_dialog, self.em_dim_option = external_memory_dim_dialog, external_memory_dim_option
        self.em_masks, self.em_vln_masks = z(T + 1, N, self.em_size), z(T + 1, N, self.em_vln_size)
        mk = lambda size, cap, dim: ExternalMemory(N, size, cap, dim, num_copies=T + 1, num_steps=T, device=dev)
        self.em = self.em_option = self.em_vln = self.em_vln_dialog = None
        if use_external_memory:
            self.em = mk(self.em_size, self.em_capacity, self.em_dim_goal)
            self.em_option = mk(self.em_option_size, self.em_option_capacity, self.em_dim_option)
            self.em_vln = mk(self.em_vln_size, self.em_vln_capacity, self.em_dim_vln)
        if use_state_memory:
            self.em_vln_dialog = mk(self.em_vln_size, self.em_vln_capacity, self.em_dim_dialog)
        self.step = 0
        self.env_id = 0                                  # insert_replay cursor (rollout_storage.py:174-176)
        self._plans = {}
        self._src_plans = {}
        self._em_ops = (L.ExtMemOp * 4)()

    def to(self, device):
        dev = torch.device(device)
        for k, v in list(self.__dict__.items()):
            if torch.is_tensor(v):
                setattr(self, k, v.to(dev))
        self.observations = {k: v.to(dev) for k, v in self.observations.items()}
        for em in (self.em, self.em_option, self.em_vln, self.em_vln_dialog):
            if em is not None:
                em.to(dev)
        self.device = dev
        self._plans = {}
        self._src_plans = {}

    # ---------------------------------------------------------------- insert (rollout_storage.py:214-297)
    _enc_leader = None

    def attach_encoders(self, leader):
        """See policy.share_encoders(..., rollouts=): `insert` starts the leader's shared encoders for the batch it is given."""
        self._enc_leader = leader

    def insert(self, observations, recurrent_hidden_states, actions, actions_option, action_log_probs, value_preds,
               rewards, not_done_masks, not_done_masks_vln, em_features, em_features_option, em_features_vln,
               em_features_dialog, all_dialog, o_action, o_mask, rl_masks, ucnt_gt, action_prob, query_state,
               last_query_info, agent_step):
        s = self.step
        dev = self.device
        lead = self._enc_leader
        if lead is not None and lead._enc_early is None and "rgb" in self.observations:
            # share_encoders(..., rollouts=self): the next act_option will read this batch out of slot s + 1 -- its towers start now
            lead.prefetch_encoders(observations, will_be={k: self.observations[k][s + 1] for k in ("rgb", "depth", "spectrogram")
                                                          if k in self.observations})
        # All of the step's storage writes go out as ONE batched copy launch.  The destination side (views, pointers,
        # byte counts) depends only on the step index and is planned once per step slot; a source that is already a
        # contiguous device tensor of the right dtype and size is passed by pointer, anything else (host values, dtype
        # conversions, broadcasts -- the reference's `tensor[step].copy_(value)` semantics) is converted first.
        have_o, have_opt = o_action is not None, actions_option is not None
        plan = self._plans.get((s, have_o, have_opt))
        if plan is None:
            dsts = [self.observations[k][s + 1] for k in self.observations]
            dsts += [self.recurrent_hidden_states[s + 1], self.all_dialog[s], self.query_state[s], self.last_query_info[s],
                     self.agent_step[s]]
            if have_o:
                dsts += [self.o_masks[s], self.ucnt_gt[s], self.rl_masks[s], self.o_actions[s], self.action_probs[s]]
            dsts.append(self.actions[s])
            if have_opt:
                dsts.append(self.actions_option[s])
            dsts += [self.prev_actions[s + 1], self.action_log_probs[s], self.value_preds[s], self.rewards[s],
                     self.masks[s + 1], self.masks_vln[s + 1]]
            n = len(dsts)
            plan = (dsts, (C.c_void_p * n)(*[d.data_ptr() for d in dsts]),
                    (C.c_int64 * n)(*[d.numel() * d.element_size() for d in dsts]),
                    [(d.dtype, d.numel()) for d in dsts])
            self._plans[(s, have_o, have_opt)] = plan
        dsts, dst_ptrs, sizes, meta = plan
        srcs = [observations[k] for k in self.observations]
        srcs += [recurrent_hidden_states, all_dialog, query_state, last_query_info, agent_step]
        if have_o:
            srcs += [o_mask, ucnt_gt, rl_masks, o_action, action_prob]
        srcs.append(actions)
        if have_opt:
            srcs.append(actions_option)
        srcs += [actions, action_log_probs, value_preds, rewards, not_done_masks, not_done_masks_vln]
        # Fast path: the same source tensors (address, dtype, element count) as the last time this step slot was written -- a
        # trainer that hands over views of persistent buffers, as the HIP-graph outputs are -- reuse the validated pointer array.
        try:
            sig = tuple((v.data_ptr(), v.dtype, v.numel(), v.is_contiguous()) for v in srcs)
        except AttributeError:                           # a host scalar / list among the sources
            sig = None
        fast = self._src_plans.get((s, have_o, have_opt)) if sig is not None else None
        if fast is not None and fast[0] == sig and dev.type == "cuda":
            L.call("avlen_multi_copy", fast[1], dst_ptrs, sizes, len(sig), L.stream())
        else:
            keep, ptrs, direct = [], [], True
            for v, d, (dt, n) in zip(srcs, dsts, meta):
                if not (torch.is_tensor(v) and v.is_cuda and v.dtype == dt and v.numel() == n and v.is_contiguous()):
                    v = (v if torch.is_tensor(v) else torch.as_tensor(v)).to(dev, dtype=dt, non_blocking=True)
                    v = v.reshape(d.shape) if v.numel() == n else v.expand_as(d)
                    v = v.contiguous()
                    keep.append(v)                       # alive until the launch below has been enqueued
                    direct = False
                ptrs.append(v.data_ptr())
            if dev.type == "cuda":
                arr = (C.c_void_p * len(ptrs))(*ptrs)
                L.call("avlen_multi_copy", arr, dst_ptrs, sizes, len(ptrs), L.stream())
                if direct and sig is not None:           # every source was usable in place: remember the pointer array
                    self._src_plans[(s, have_o, have_opt)] = (sig, arr)
        if dev.type != "cuda":
            for v, d in zip(srcs, dsts):
                d.copy_(v if torch.is_tensor(v) else torch.as_tensor(v))
        nd, ndv = self.masks[s + 1], self.masks_vln[s + 1]
        if dev.type == "cuda":                         # the step's ring inserts as ONE launch
            ops, n, alive = self._em_ops, 0, []
            if self.use_external_memory:
                alive.append(self.em.fill_op(ops[0], em_features, nd, self.em_masks[s + 1]))
                alive.append(self.em_option.fill_op(ops[1], em_features_option, nd))
                alive.append(self.em_vln.fill_op(ops[2], em_features_vln, ndv, self.em_vln_masks[s + 1]))
                n = 3
            if self.use_state_memory:
                alive.append(self.em_vln_dialog.fill_op(ops[n], em_features_dialog, ndv))
                n += 1
            if n:
                L.call("avlen_extmem_insert_multi", ops, n, L.stream())
            if self.use_state_memory and not self.use_external_memory:
                self.em_vln_masks[s + 1].copy_(self.em_vln_dialog.masks)
        else:
            if self.use_external_memory:
                self.em.insert(em_features, nd, self.em_masks[s + 1])
                self.em_option.insert(em_features_option, nd)
                self.em_vln.insert(em_features_vln, ndv, self.em_vln_masks[s + 1])
            if self.use_state_memory:
                self.em_vln_dialog.insert(em_features_dialog, ndv)
                if not self.use_external_memory:
                    self.em_vln_masks[s + 1].copy_(self.em_vln_dialog.masks)
        self.step = s + 1

    # ---------------------------------------------------------------- replay / dialog pre-training side
    def insert_replay(self, observations, recurrent_hidden_states, actions, actions_option, action_log_probs, value_preds,
                      rewards, not_done_masks, not_done_masks_vln, em_features, em_features_dialog, all_dialog, o_action, o_mask,
                      action_prob, query_state, agent_step):
        """rollout_storage.py:300-371: one environment's whole stored dialog episode (num_steps rows) becomes column `env_id` of
        every buffer (REPLAY_STORE path, ppo_trainer.py:913-951).  Mirrors the reference statement by statement, including that
        `em_features` feeds both the goal and the vln ring."""
        T, e, dev = self.num_steps, self.env_id, self.device
        c = lambda dst, src: dst.copy_(src if torch.is_tensor(src) else torch.as_tensor(src))
        for k in observations:
            if k in self.observations:
                c(self.observations[k][:T, e], observations[k])
        c(self.recurrent_hidden_states[:T, :, e, :], recurrent_hidden_states)
        c(self.all_dialog[:T, e], all_dialog)
        c(self.query_state[:T, e], query_state)
        c(self.agent_step[:T, e], agent_step)
        if o_action is not None:
            c(self.o_masks[:T, e], o_mask)
            c(self.o_actions[:T, e], o_action)
            c(self.action_probs[:T, e], action_prob)
        c(self.actions[:T, e], actions)
        if actions_option is not None:
            c(self.actions_option[:T, e], actions_option)
        c(self.prev_actions[:T, e], actions)
        c(self.action_log_probs[:T, e], action_log_probs)
        c(self.value_preds[:T, e], value_preds)
        c(self.rewards[:T, e], rewards)
        c(self.masks[:T, e], not_done_masks)
        c(self.masks_vln[:T, e], not_done_masks_vln)
        f = (em_features if torch.is_tensor(em_features) else torch.as_tensor(em_features)).to(dev)
        if self.use_external_memory:
            self.em.insert_replay(f)
            self.em_masks[:T, e, :].copy_(self.em.masks_replay[:, e, :])
            self.em_vln.insert_replay(f)
            self.em_vln_masks[:T, e, :].copy_(self.em_vln.masks_replay[:, e, :])
        if self.use_state_memory:
            fd = (em_features_dialog if torch.is_tensor(em_features_dialog) else torch.as_tensor(em_features_dialog)).to(dev)
            self.em_vln_dialog.insert_replay(fd)
            self.em_vln_masks[:T, e, :].copy_(self.em_vln.masks_replay[:, e, :] if self.use_external_memory
                                              else self.em_vln_dialog.masks_replay[:, e, :])
        self.env_id = e + 1
        self.step = T

    def dialog_batching(self, memories=True):
        """rollout_storage.py:414-588: every environment, steps [0, step), flattened T-major -- the 17-tuple `PPO.update_dialog`
        consumes.  The memories come out as the reference's (em_size, T*N, dim) tensors (every copy of a ring is identical);
        memories=False leaves their three slots None (avlen_amd's own update_dialog reads the rings in place through a row index:
        at N = 64, T = 150 the expanded copies are ~12 GB of transient traffic nobody reads)."""
        T, N = self.step, self.num_envs
        fl = lambda x: x[:T].reshape((T * N,) + tuple(x.shape[2:]))
        mem = lambda em: (em.memory.unsqueeze(1).expand(-1, T, -1, -1).reshape(em.total_size, T * N, em.dim)
                          if (em is not None and memories) else None)
        obs = defaultdict(list)
        for k, v in self.observations.items():
            obs[k] = fl(v)
        use_em, use_sm = self.use_external_memory, self.use_state_memory
        return (obs, self.recurrent_hidden_states[0], fl(self.actions), fl(self.prev_actions), fl(self.value_preds),
                fl(self.returns), fl(self.masks), fl(self.action_log_probs), mem(self.em) if use_em else None,
                mem(self.em_vln) if use_em else None, mem(self.em_vln_dialog) if use_sm else None,
                fl(self.em_masks) if use_em else None, fl(self.em_vln_masks) if (use_em or use_sm) else None, fl(self.all_dialog),
                fl(self.agent_step), self.num_steps, self.num_envs)

    def after_update(self):
        s = self.step
        pairs = [(v[0], v[s]) for v in self.observations.values()]
        pairs.append((self.recurrent_hidden_states[0], self.recurrent_hidden_states[s]))
        pairs += [(buf[0], buf[s]) for buf in (self.masks, self.masks_vln, self.prev_actions, self.em_masks,
                                                 self.em_vln_masks)]
        L.multi_copy(pairs)
        self.step = 0

    # ---------------------------------------------------------------- GAE (rollout_storage.py:394-412)
    def compute_returns(self, next_value, use_gae, gamma, tau):
        nv = next_value.detach().float().contiguous()
        if not use_gae:                                  # rollout_storage.py:406-412
            L.call("avlen_discounted_returns", P(self.rewards), P(self.masks), P(nv), P(self.returns), self.step, self.num_envs,
                   float(gamma), L.stream())
            return
        L.call("avlen_gae_scan", P(self.rewards), P(self.value_preds), P(self.masks), P(nv), P(self.returns),
               P(self.advantages), self.step, self.num_envs, float(gamma), float(tau), L.stream())

    # ---------------------------------------------------------------- accessors used by the trainer
    @property
    def external_memory_goal(self):
        return _CopyAxis(self.em)

    @property
    def external_memory_option(self):
        return _CopyAxis(self.em_option)

    @property
    def external_memory_vln(self):
        return _CopyAxis(self.em_vln)

    @property
    def external_memory_vln_dialog(self):
        return _CopyAxis(self.em_vln_dialog)

    @property
    def external_memory_masks(self):
        return self.em_masks

    @property
    def external_memory_goal_idx(self):
        return self.em.idx

    @property
    def external_memory_option_idx(self):
        return self.em_option.idx

    @property
    def external_memory_vln_idx(self):
        return self.em_vln.idx

    @property
    def external_memory_vln_dialog_idx(self):
        return self.em_vln_dialog.idx

    @property
    def external_memory_vln_masks(self):
        return self.em_vln_masks

    # ---------------------------------------------------------------- minibatches
    def _gather(self, src, env, T):
        """src (T_alloc, N, ...) -> (T*n_mb, ...) rows ordered t-major, like _flatten_helper."""
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

    def gather_minibatch(self, env, advantages=None, in_place=False, feature_rows=False):
        """The tensors PPO.update needs for the env subset `env` (int64, device), WITHOUT materialising the
        (em_size, T*n_mb, dim) memory: rows index the ring through `mem_index`.
        feature_rows: also `stored` = (option ring as (total * N, dim) rows, row index of sample (t, env[j])): the feature row the
        rollout's forward inserted at step t sits in ring slot (idx - T + t) mod total -- total = capacity + T, so none of the
        rollout's T rows has been overwritten."""
        T = self.step
        g = lambda x: self._gather(x, env, T)
        adv = self.advantages if advantages is None else advantages
        n_mb = env.numel()
        mem_index = env.to(torch.int32).repeat(T).contiguous()            # row t*n_mb+j reads ring column env[j]
        obs = {}
        if in_place:
            # the three large sensors (470 KB per stored step at 257x101) are not gathered: the encoders read row
            # t*N + env[j] of the (T+1, N, ...) storage directly (policy.RowsOf -> avlen_*_fwd_indexed)
            from .policy import RowsOf
            rows = (torch.arange(T, device=env.device, dtype=torch.int32).view(T, 1) * self.num_envs +
                    env.to(torch.int32).view(1, n_mb)).reshape(-1).contiguous()
            for k, v in self.observations.items():
                big = k in ("rgb", "depth", "spectrogram") and (v.dtype == torch.float32 or (k == "rgb" and v.dtype == torch.uint8))
                obs[k] = RowsOf(v.view((-1,) + tuple(v.shape[2:])), rows) if big else g(v)
        else:
            obs = {k: g(v) for k, v in self.observations.items()}
        stored = None
        if feature_rows:
            em = self.em_option
            assert T <= em.total_size, "the rollout's rows must still be in the ring"
            slot = (torch.arange(T, device=env.device, dtype=torch.int32) + (em.idx - T)) % em.total_size
            srow = (slot.view(T, 1) * em.num_envs + env.to(torch.int32).view(1, n_mb)).reshape(-1).contiguous()
            stored = (em.memory.as_subclass(torch.Tensor).view(em.total_size * em.num_envs, em.dim), srow)
        return {
            "obs": obs, "stored": stored,
            "actions_option": g(self.actions_option), "prev_actions": g(self.prev_actions),
            "value_preds": g(self.value_preds), "returns": g(self.returns), "masks": g(self.masks),
            "old_log_probs": g(self.action_log_probs), "adv": g(adv), "rl_masks": g(self.rl_masks),
            "ucnt_gt": g(self.ucnt_gt), "em_masks": g(self.em_masks), "query_state": g(self.query_state),
            "last_query_info": g(self.last_query_info), "mem_index": mem_index, "T": T, "n_mb": n_mb,
        }

    def recurrent_generator(self, advantages, num_mini_batch):
        """API-compatible generator (rollout_storage.py:591-810): yields the reference's 22-tuple, including the
        materialised (em_size, T*n_mb, dim) memories."""
        N = self.rewards.size(1)
        assert N >= num_mini_batch, (
            "Trainer requires the number of processes ({}) to be greater than or equal to the number of "
            "trainer mini batches ({}).".format(N, num_mini_batch))
        per = N // num_mini_batch
        perm = torch.randperm(N)
        T = self.step
        for start in range(0, N, per):
            env = perm[start:start + per].to(self.device)
            n_mb = env.numel()
            g = lambda x: self._gather(x, env, T)
            obs = defaultdict(list)
            for k, v in self.observations.items():
                obs[k] = g(v)
            rec = self.recurrent_hidden_states[0][:, env]
            mem = lambda em: (em.memory[:, env].unsqueeze(1).expand(-1, T, -1, -1).reshape(em.total_size, T * n_mb, em.dim)
                              if em is not None else None)
            yield (obs, rec, g(self.actions), g(self.actions_option), g(self.prev_actions), g(self.value_preds),
                   g(self.returns), g(self.masks), g(self.action_log_probs), g(advantages), g(self.rl_masks),
                   g(self.ucnt_gt), mem(self.em), mem(self.em_option), mem(self.em_vln), mem(self.em_vln_dialog),
                   g(self.em_masks), g(self.em_vln_masks), g(self.all_dialog), g(self.query_state),
                   g(self.last_query_info), g(self.agent_step))


CFG.add_ranges(RolloutStorage, ('insert', 'compute_returns', 'after_update'), "RolloutStorage.")
