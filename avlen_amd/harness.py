"""Synthetic-observation driver that reproduces the reference trainer's call pattern on the hot path
(ss_baselines/savi/ppo/ppo_trainer.py:323-897 `_collect_rollout_step`, :1045-1093 `_update_agent`) without
the simulator: pi_q.act_option -> pi_g.act -> pi_l.act_dialog -> RolloutStorage.insert per step, then
get_value_option -> compute_returns (GAE) -> DDPPO.update -> after_update per cycle.

Used by bench.py and __graft_entry__.smoke().  Observations follow SURVEY.md §8(d): 128x128 RGB-D,
binaural spectrogram, pose, beliefs, 77-token dialog; everything is generated once and is resident in
HBM before timing starts ("simulator output").
"""
import math
import time
import numpy as np
import torch

from . import policy as P
from .ppo import DDPPO
from .rollout_storage import RolloutStorage
from .spaces import savi_observation_space, ActionSpace, SMT_KW


def sinusoid_table(n, d):
    pos = torch.arange(n).unsqueeze(1)
    div = torch.exp(torch.arange(0, d, 2) * (-math.log(10000.0) / d))
    pe = torch.zeros(n, d)
    pe[:, 0::2] = torch.sin(pos * div)
    pe[:, 1::2] = torch.cos(pos * div)
    return pe


class Workload:
    def __init__(self, num_envs, num_steps=150, spectrogram=(257, 101, 2), precision="bf16x3", pretraining=True,
                 em_capacity=150, ppo_epoch=2, num_mini_batch=2, device="cuda", seed=0, sampling="race",
                 with_dialog_policy=True, with_goal_policy=True, use_graphs=True, share_encoders=True, weight_seed=0,
                 launch_ahead=True, belief_predictor=False, cached_views=False, distractor=False,
                 dialog_tokens="after_option", dialog_process="fresh", belief_async=False):
        self.N, self.T, self.dev = num_envs, num_steps, torch.device(device)
        self.spec = spectrogram
        # WHEN the step's dialog tokens exist.  "after_option" (default) = the reference's data flow: `current_dialog` and
        # `agent_step` are written by the host loop that follows `act_option` (new query -> Speaker -> clip.tokenize,
        # ppo_trainer.py:347, 449-593), so the host WAITS for pi_q's sampled actions, and the text tower and the dialog half of pi_l
        # are issued only then; pi_g and pi_l's state-encoder half stay ahead.  "ahead" = the tokens are known at the start of the step
        # (text tower launched beside the visual towers): a synthetic ordering the reference trainer cannot provide.
        assert dialog_tokens in ("after_option", "ahead") and dialog_process in ("fresh", "reference")
        self.dialog_tokens = dialog_tokens
        # WHAT the tokens are.  "fresh": every env presents a new random dialog every step (SURVEY 8d's synthetic input; the memo of
        # the text tower never hits).  "reference": the trainer's dialog process -- an env that is not in a dialog and samples
        # a_q == 1 gets a new dialog, keeps it for NUM_DIALOG_STEPS = 3 steps (agent_step 0, 1, 2), then leaves the dialog; every other
        # env presents all-zero tokens (ppo_trainer.py:347, 463-469, 582-587, 763-765).  Needs the tokens after act_option.
        self.dialog_process = dialog_process
        assert dialog_process == "fresh" or dialog_tokens == "after_option"
        # launch_ahead: all three policies' forwards are enqueued before the first host-side wait (the two / three extra calls a trainer
        # adds per step): pi_q on the caller's stream, pi_g and pi_l's state-encoder half on ONE side stream, pi_g first
        # (with pi_l's half first, pi_g ran beside the text tower and crawled on the CUs it left: 318-322 -> 302 ms per cycle, round 4),
        # the text tower on the caller's stream behind pi_q.  Four busy streams is the limit of the process's four hardware queues.
        self.launch_ahead = launch_ahead
        # The reference trainer slices fresh views of the storage every step (ppo_trainer.py:375-391): so does this driver, unless
        # `cached_views` keeps the view objects per step slot (saves the host ~30 tensor-indexing calls per step).
        self.cached_views = cached_views
        self.distractor = distractor
        # Policy.prefetch_encoders on the new observation before `insert` copies it (the towers hide the storage bookkeeping)
        # (with a BeliefPredictor too: its update of the new observation's beliefs is enqueued FIRST, the towers behind it)
        self._early_enc = use_graphs and share_encoders and precision in ("bf16", "bf16x3")
        self._next_views = None
        self.trace = None                               # tools/step_timeline.py: a list collects (mark, host time) pairs of every step
        # ONE set of side streams per process (policy.process_stream): the runtime maps streams to its 4 hardware queues in creation
        # order, so a second Workload with fresh streams can land pi_g's stream on the text tower's queue (seen as records of one
        # bench run that differ by 10 % for no other reason) -- and torch's stream pool wraps around after 32 creations
        self._side = [P.process_stream("harness0"), P.process_stream("followers"), P.process_stream("harness_text")] \
            if launch_ahead else None                   # [1] = EncoderGroup.side_stream(): the followers' (and the audio piece's) stream
        osp, asp = savi_observation_space(spectrogram), ActionSpace(4)
        torch.manual_seed(weight_seed)          # identical initial weights on every rank (data-parallel replicas)
        kw = dict(SMT_KW, precision=precision, sampling=sampling, use_graphs=use_graphs)
        # distractor variant (BASELINE configs[4]): has_distractor_sound -> use_category_input for all three policies
        # (ddppo_trainer.py:306,329,349,371); pi_l ignores it (policy.py:728-732)
        self.pi_q = P.AudioNavOptionPolicy(osp, asp, pretraining=pretraining, use_category_input=distractor,
                                           query_count_emb_size=32, **kw).to(self.dev)
        self.pi_g = (P.AudioNavSMTPolicy(osp, asp, pretraining=False, use_category_input=distractor, **kw).to(self.dev)
                     if with_goal_policy else None)
        self.pi_l = (P.AudioNavDialogPolicy(osp, asp, pretraining=False, use_category_input=distractor, num_steps=3,
                                            **kw).to(self.dev) if with_dialog_policy else None)
        self.seq = None
        if share_encoders and precision in ("bf16", "bf16x3") and self.pi_g is not None and self.pi_l is not None:
            grp = P.share_encoders(self.pi_q, self.pi_g, self.pi_l)
            assert self._side is None or self._side[1] is grp.side_stream()
            if launch_ahead and use_graphs and sampling == "race" and dialog_tokens == "after_option":
                # the step's launch-ahead calls as recorded command lists (one C call per phase): avlen_amd/sequencer.py
                from .sequencer import StepSequencer
                self.seq = StepSequencer(self.pi_q, self.pi_g, self.pi_l)
        self.agent = DDPPO(self.pi_q, clip_param=0.2, ppo_epoch=ppo_epoch, num_mini_batch=num_mini_batch,
                           value_loss_coef=0.5, entropy_coef=0.05, lr=2.5e-4, eps=1e-5, max_grad_norm=0.2,
                           use_normalized_advantage=False)
        self.agent.init_distributed(find_unused_params=True)
        T, N = self.T, self.N
        ems = em_capacity + T                       # ddppo_trainer.py:649-669: size = capacity + num_steps
        dg = self.pi_g.net.memory_dim if self.pi_g is not None else self.pi_q.net.memory_dim - 32
        self.rollouts = RolloutStorage(T, N, osp, asp, 512, True, ems, em_capacity, ems, em_capacity, 3, 3, dg, 276,
                                       self.pi_q.net.memory_dim, 256, num_recurrent_layers=-1, max_dialog_len=77, use_state_memory=True,
                                       device=self.dev)
        if share_encoders and not launch_ahead and use_graphs and not belief_predictor and precision in ("bf16", "bf16x3") and \
                self.pi_g is not None and self.pi_l is not None:
            # the one-line integration (`share_only`): share_encoders(pi_q, pi_g, pi_l, rollouts=rollouts)
            self.rollouts.attach_encoders(self.pi_q)
        self.belief = None
        self._act_buf = None
        self._act_host = None
        self._act_np = {}
        self._host_select = sampling == "race"
        self._small = self._small_ev = self._fwd_ev = None
        self._small_pending = False
        if self._side is not None and self._host_select and dialog_tokens == "after_option":
            # the side stream the after_option flow leaves idle (pi_g and pi_l share the other one, the text tower runs on the
            # caller's): the step's storage writes go there
            self._small = self._side[0]
            self._small_ev, self._fwd_ev = torch.cuda.Event(), torch.cuda.Event()
        self.sampling = sampling
        if belief_predictor:        # use_belief_predictor: True in the interactive yamls (ppo_trainer.py:892); 65x26 spectrogram only
            import types
            from .belief_predictor import BeliefPredictor
            bcfg = types.SimpleNamespace(use_label_belief=True, use_location_belief=True, online_training=True,
                                         current_pred_only=False, weighting_factor=0.5)
            self.belief = BeliefPredictor(bcfg, self.dev, None, None, 512, num_env=num_envs, precision=precision,
                                          load_pretrained=False, use_graphs=use_graphs).to(self.dev)
        # belief_async: the two belief networks write their beliefs into the storage slot on their own stream while the next step's
        # visual towers already run (Policy.late_inputs).  Measured (65x26, 64 envs, bf16x3): 24.9 k env-steps/s asynchronous vs 24.8 k
        # synchronous -- the two ResNet-18s are ~85 chip-wide launches of 5-12 us; on the CUs the persistent tower launch can spare
        # they run 4-8x longer, so the 0.6 ms they cost barely hides.  Off by default.
        self._belief_async = self.belief is not None and use_graphs and launch_ahead and share_encoders and precision == "bf16x3" \
            and belief_async
        if self._belief_async:
            self._belief_stream = P.process_stream("belief_async")
            from . import _lib as L
            L.lib.avlen_set_tower_x3_reserved_cus(32)
        self._make_simulator_output(seed)

    # -- what the CPU simulator + trainer bookkeeping would have produced, resident in HBM ------------------
    def _make_simulator_output(self, seed):
        T, N, dev = self.T, self.N, self.dev
        g = torch.Generator(device="cpu").manual_seed(1000 + seed)
        H, W, _ = self.spec
        r = lambda *s: torch.rand(*s, generator=g)
        self.sim = {
            # uint8 as the simulator's RGB sensor produces it: stays uint8 through the storage into the tower prologue (f2)
            "rgb": torch.randint(0, 256, (T + 1, N, 128, 128, 3), generator=g, dtype=torch.uint8).to(dev),
            "depth": r(T + 1, N, 128, 128, 1).to(dev),
            "spectrogram": torch.log1p(3.0 * torch.randn(T + 1, N, H, W, 2, generator=g).abs()).to(dev),
            "category": torch.nn.functional.one_hot(torch.randint(0, 21, (T + 1, N), generator=g), 21).float().to(dev),
            "category_belief": torch.softmax(torch.randn(T + 1, N, 21, generator=g), -1).to(dev),
            "location_belief": (3.0 * torch.randn(T + 1, N, 2, generator=g)).to(dev),
        }
        pose = torch.stack([r(T + 1, N) * 20 - 10, r(T + 1, N) * 20 - 10, r(T + 1, N) * 2 * math.pi - math.pi,
                            torch.arange(T + 1).float().view(-1, 1).expand(-1, N)], -1)
        self.sim["pose"] = pose.to(dev)
        self.rewards = torch.randn(T, N, 1, generator=g).to(dev)
        self.not_done = (r(T, N, 1) >= 1.0 / 150).float().to(dev)
        self.dones = (1.0 - self.not_done).view(T, N).to(torch.uint8)
        pe = sinusoid_table(1000, 32)
        self.query_state = pe[torch.randint(0, 4, (T, N), generator=g)].to(dev)
        self.last_query_info = pe[torch.randint(0, 150, (T, N), generator=g)].to(dev)
        self.agent_step = torch.randint(0, 3, (T, N), generator=g).float().to(dev)
        toks = torch.zeros(T, N, 77, dtype=torch.long)
        ln = torch.randint(2, 73, (T, N), generator=g)
        body = torch.randint(1, 49406, (T, N, 77), generator=g)
        ar = torch.arange(77).view(1, 1, 77)
        toks = torch.where(ar < ln.unsqueeze(-1), body, toks)
        toks[..., 0] = 49406
        toks.scatter_(-1, ln.unsqueeze(-1), 49407)
        self.dialog = toks.to(dev)
        self.rl_masks = (r(T, N) < 0.7).long().to(dev)
        self.rl_masks[:, 0] = 1
        self.ucnt_gt = torch.randint(0, 2, (T, N), generator=g).to(dev)
        self.o_action = torch.zeros(N, device=dev)
        self.o_mask = torch.ones(N, dtype=torch.long, device=dev)
        self.zero_dialog_feats = torch.zeros(N, 256, device=dev)
        self.zero_probs = torch.zeros(N, 4, device=dev)
        self._views = {}
        for k in self.rollouts.observations:
            self.rollouts.observations[k][0].copy_(self.sim[k][0])
        if self.dialog_process == "reference":
            # host-side query bookkeeping of the trainer (track_query[idx]: 'queried', 'step', 'dialog') + the two tensors it writes
            self._dones_h = self.dones.cpu().numpy()
            self._pool = toks.numpy()                                       # what Speaker + clip.tokenize would return
            self._queried = torch.zeros(N, dtype=torch.bool).numpy()
            self._qstep = torch.zeros(N, dtype=torch.int64).numpy()
            self._qdialog = torch.zeros(N, 77, dtype=torch.int64).numpy()
            self._cur_dialog_h = [torch.zeros(N, 77, dtype=torch.long, pin_memory=True) for _ in range(4)]
            self._cur_astep_h = [torch.zeros(N, pin_memory=True) for _ in range(4)]
            self._cur_dialog = torch.zeros(N, 77, dtype=torch.long, device=dev)     # `current_dialog` (ppo_trainer.py:347)
            self._cur_astep = torch.zeros(N, device=dev)                            # `rollouts.agent_step[step]` (:590)
            self._ring = 0
            self.dialog_stats = {"steps": 0, "new_dialogs": 0, "active_rows": 0}

    def _host_dialog_loop(self, t, a_q):
        """ppo_trainer.py:463-469, 519-587 for the synthetic envs: returns after `current_dialog` / `agent_step` are on their way
        to the device.  a_q: (N,) host int64."""
        q, st = self._queried, self._qstep
        if t > 0:                                         # :390-397: a new episode starts outside any dialog
            ended = self._dones_h[t - 1] != 0
            q[ended] = False
            st[ended] = 0
        new = (~q) & (a_q == 1)
        q |= new
        st[new] = 0
        self._qdialog[new] = self._pool[t][new]
        act = q & (st < 3)
        self._ring = (self._ring + 1) % 4
        dh, ah = self._cur_dialog_h[self._ring], self._cur_astep_h[self._ring]
        dn, an = dh.numpy(), ah.numpy()
        dn[:] = 0
        an[:] = 0
        dn[act] = self._qdialog[act]
        an[act] = st[act]
        st[act] += 1
        self._cur_dialog.copy_(dh, non_blocking=True)
        self._cur_astep.copy_(ah, non_blocking=True)
        fin = q & (st >= 3)                               # :763-765 (after the env step): the dialog is over
        q[fin] = False
        st[fin] = 0
        ds = self.dialog_stats
        ds["steps"] += 1
        ds["new_dialogs"] += int(new.sum())
        ds["active_rows"] += int(act.sum())

    # -- one rollout step (ppo_trainer.py:375-391, 449, 608-636, 864-888) -----------------------------------
    def _step_views(self, t):
        """Every tensor the trainer would slice out of the storage / simulator output at step t (views of persistent
        buffers, built once per step slot)."""
        v = self._views.get(t) if self.cached_views else None
        if v is None:
            ro = self.rollouts
            v = dict(obs={k: x[t] for k, x in ro.observations.items()}, h=ro.recurrent_hidden_states[t],
                     prev=ro.prev_actions[t], masks=ro.masks[t], masks_vln=ro.masks_vln[t],
                     em_masks=ro.external_memory_masks[t], em_vln_masks=ro.external_memory_vln_masks[t],
                     qs=self.query_state[t], lqi=self.last_query_info[t], dialog=self.dialog[t], astep=self.agent_step[t],
                     nxt={k: self.sim[k][t + 1] for k in ro.observations}, rew=self.rewards[t], nd=self.not_done[t],
                     dones=self.dones[t],
                     rl=self.rl_masks[t], ucnt=self.ucnt_gt[t])
            if self.cached_views:
                self._views[t] = v
        return v

    def _forward_all(self, ro, v, t):
        """The three policies on the state `v` (step views) of storage `ro`, in the trainer's order (ppo_trainer.py:449, 608, 625)."""
        obs, h, prev, em_masks = v["obs"], v["h"], v["prev"], v["em_masks"]
        em_opt, em_goal = ro.external_memory_option[:, t], ro.external_memory_goal[:, t]
        em_vln, em_dlg = ro.external_memory_vln[:, t], ro.external_memory_vln_dialog[:, t]
        later = self.dialog_tokens == "after_option"
        ref = self.dialog_process == "reference"
        if ref:                                          # the tensors the host loop fills after act_option
            v = dict(v, dialog=self._cur_dialog, astep=self._cur_astep)
        q_args = (obs, h, prev, v["masks"], em_opt, em_masks, v["qs"], v["lqi"])
        g_args = (obs, h, prev, v["masks"], em_goal, em_masks)
        l_args = (obs, h, prev, v["masks_vln"], em_vln, em_dlg, v["em_vln_masks"], v["dialog"], v["astep"])
        seq = self.seq if (self.launch_ahead and later) else None
        tr = self.trace
        if tr is not None:
            tr.append(("launch_begin", time.perf_counter()))
        if seq is not None:
            seq.launch(q_args, g_args, l_args)           # pi_q | pi_g, pi_l's state-encoder half: one recorded command list
        elif self.launch_ahead:
            if self.pi_l is not None and not later:
                # tokens known ahead (synthetic): the text tower needs only them and runs beside pi_q's graph (the towers)
                self.pi_q.prefetch_act_option(*q_args)
                self.pi_l.prefetch_text(v["dialog"], self._side[2], after_current=False)
            else:
                self.pi_q.prefetch_act_option(*q_args)
            if self.pi_g is not None:                    # pi_g FIRST on the followers' shared stream (see __init__)
                self.pi_g.prefetch_act(*g_args, stream=self._side[1])
            if self.pi_l is not None:
                self.pi_l.prefetch_act_dialog(*l_args, stream=self._side[1] if later else None, dialog_later=later)
        if self.launch_ahead and t + 1 < self.T:
            # the host is about to wait for pi_q's actions: slice the NEXT step's views now (fresh tensor objects every step, as the
            # trainer makes them; only the moment moves off the path between insert and the next forward's launch)
            self._next_views = (t + 1, self._step_views(t + 1))
        if tr is not None:
            tr.append(("launched", time.perf_counter()))
        values, unct, a_opt, lp_opt, h, row_opt, probs_opt = self.pi_q.act_option(*q_args)
        if later and self.pi_l is not None:
            # ppo_trainer.py:463-593: the host loop between act_option and act / act_dialog READS the option actions (a new query
            # -> Speaker -> clip.tokenize -> `current_dialog`): the step's tokens exist only once a_q is on the host -- waited for
            # here in BOTH dialog processes
            a_q_host = self.pi_q.host_actions("option")
            if a_q_host is None:                         # sampling="host" / "device": the actions as the call returned them
                a_q_host = a_opt.cpu()
            if tr is not None:
                tr.append(("a_q_on_host", time.perf_counter()))
            if ref:
                self._host_dialog_loop(t, a_q_host.view(-1).numpy())
            if seq is not None:
                seq.dialog_ready()
            elif self.launch_ahead:
                self.pi_l.dialog_ready()                 # current_dialog / agent_step hold this step's values from here on
            if tr is not None:
                tr.append(("dialog_ready_done", time.perf_counter()))
        dg = ro.em_dim_goal
        o = dict(q_value=values, q_prob=probs_opt, a_q=a_opt, lp_q=lp_opt, h=h, row_q=row_opt, row_g=row_opt[:, :dg],
                 row_l=row_opt[:, :276], row_d=self.zero_dialog_feats, l_prob=self.zero_probs, a_g=a_opt, a_l=a_opt,
                 g_value=values, l_value=values, g_prob=self.zero_probs)
        if self.pi_g is not None:
            o["g_value"], o["a_g"], _, _, o["row_g"], o["g_prob"] = self.pi_g.act(*g_args)
        if self.pi_l is not None:
            o["l_value"], o["a_l"], _, _, o["row_l"], o["row_d"], o["l_prob"] = self.pi_l.act_dialog(*l_args)
        return o

    def policies_on(self, other, t=None):
        """This workload's policies evaluated on ANOTHER workload's current state (its storage, memories, step inputs): the
        bf16-vs-fp32 comparison of bench.py and the harness-vs-oracle tests.  Nothing is stored."""
        t = other.rollouts.step if t is None else t
        assert self.dialog_process == "fresh" and other.dialog_process == "fresh", "the reference dialog process is stateful per workload"
        return self._forward_all(other.rollouts, other._step_views(t), t)

    def rollout_step(self, return_outs=False):
        ro, t = self.rollouts, self.rollouts.step
        self._join_small()
        nv = self._next_views
        self._next_views = None
        v = nv[1] if nv is not None and nv[0] == t else self._step_views(t)
        o = self._forward_all(ro, v, t)
        a_opt, actions = o["a_q"], o["a_q"]
        if self.pi_g is not None:
            actions = o["a_g"]

        def select_on_device(actions=actions):
            if self.pi_l is not None:
                if self._act_buf is None:
                    self._act_buf = torch.empty_like(o["a_l"])
                return torch.where(a_opt == 1, o["a_l"], actions, out=self._act_buf)     # queried envs follow pi_l
            return actions
        late_select = False
        if self.sampling != "host":
            # the simulator needs the step's actions on the host (envs.step, ppo_trainer.py:864).  sampling="race": every policy's
            # sampled actions are already on their way to pinned memory right behind its heads kernel (Policy.host_actions), so the
            # step's actions are selected THERE (64 integers) as soon as pi_l's copy has landed -- the device-side select only feeds
            # the storage and is enqueued behind the next step's towers instead of in front of them (it and its device-to-host
            # copy were ~30 us of idle GPU per step).  Otherwise: select on the device, ONE device-to-host copy.
            if self._act_host is None:
                self._act_host = [torch.empty(o["a_q"].shape, dtype=o["a_q"].dtype, pin_memory=True) for _ in range(4)]
            ah = self._act_host[t & 3]
            hq = self.pi_q.host_actions("option") if self._host_select else None
            hg = self.pi_g.host_actions("goal") if hq is not None and self.pi_g is not None else None
            hl = self.pi_l.host_actions("vln") if hg is not None and self.pi_l is not None else None
            if hl is not None and not return_outs:
                ahn = self._act_np.get(ah.data_ptr())
                if ahn is None:
                    ahn = self._act_np[ah.data_ptr()] = ah.numpy()
                np.copyto(ahn, np.where(hq.numpy() == 1, hl.numpy(), hg.numpy()))       # 64 integers: numpy, not a torch dispatch
                late_select = True
                if self.trace is not None:
                    self.trace.append(("actions_on_host", time.perf_counter()))
            else:
                actions = select_on_device()
                ah.copy_(actions, non_blocking=True)
                P._cur_stream().synchronize()
        else:
            actions = select_on_device()
        if self.belief is not None and not self._belief_async:   # beliefs of the NEW observation, written in place before it is stored
            self.belief.update(v["nxt"], v["dones"])
        if return_outs:                             # graph outputs are overwritten by the next replay
            o = {k: (x.clone() if torch.is_tensor(x) else x) for k, x in o.items()}
            o["actions"] = actions.clone()
        small = self._small if (late_select and self._early_enc and self.launch_ahead) else None
        if small is not None:
            self._fwd_ev.record(P._cur_stream())        # the three forwards of this step are complete behind this point
        started = False
        if self._early_enc and self.launch_ahead:
            # the new observation exists: its towers start now and hide the storage bookkeeping + the next step's launch path
            started = self.pi_q.prefetch_encoders(v["nxt"], will_be={k: ro.observations[k][t + 1] for k in ("rgb", "depth", P.SPECTROGRAM)})   # the three addresses it checks
            if self.trace is not None:
                self.trace.append(("next_towers_launched", time.perf_counter()))
        dlg, astep = (self._cur_dialog, self._cur_astep) if self.dialog_process == "reference" else (v["dialog"], v["astep"])

        def store():
            acts = select_on_device() if late_select else actions
            ro.insert(v["nxt"], o["h"], acts, a_opt, o["lp_q"], o["q_value"], v["rew"], v["nd"], v["nd"], o["row_g"], o["row_q"],
                      o["row_l"], o["row_d"], dlg, self.o_action, self.o_mask, v["rl"], v["ucnt"], o["l_prob"], v["qs"],
                      v["lqi"], astep)
        if small is not None and started:
            # The step's storage writes (device-side action select + the two insert launches) go out on a side stream: enqueued on
            # the main one they would run BEHIND the towers just launched, i.e. between the towers and pi_q's state encoder (~30 us of
            # the next step's critical path); here they run as soon as the persistent tower launch frees a CU -- beside the
            # towers' fc / the audio convs.  The next step's forwards wait for `_small_ev` (top of rollout_step / update).
            small.wait_event(self._fwd_ev)
            with torch.cuda.stream(small):
                store()
                self._small_ev.record(small)
            self._small_pending = True
        else:
            store()
        if self.belief is not None and self._belief_async:
            # asynchronous form: the new observation is stored first, the two belief networks then write their beliefs into the
            # storage slot on their own stream while the next step's visual towers already run (which leave them a few CUs:
            # avlen_set_tower_x3_reserved_cus); the rest of the next forward waits for the event (Policy.late_inputs)
            # The networks read the SIMULATOR's tensors (no dependence on the storage's copy); only the beliefs go into the slot,
            # behind the storage's own write of it (`_small_ev` when the writes ran on the side stream).
            slot = dict(v["nxt"])
            for k in ("location_belief", "category_belief"):
                slot[k] = ro.observations[k][ro.step]
            ev = self.belief.update_async(slot, v["dones"], self._belief_stream,
                                          after=self._small_ev if (small is not None and started) else None)
            self.pi_q.late_inputs(("location_belief", "category_belief"), ev)
        return o if return_outs else None

    def _join_small(self):
        """Order the current stream behind the previous step's storage writes (see `store` in rollout_step)."""
        if self._small_pending:
            P._cur_stream().wait_event(self._small_ev)
            self._small_pending = False

    def finite(self):
        ro = self.rollouts
        return bool(torch.isfinite(ro.value_preds).all()) and bool(torch.isfinite(ro.em_vln_dialog.memory).all()) and \
            bool(torch.isfinite(ro.em.memory).all()) and bool(torch.isfinite(ro.em_option.memory).all())

    # -- _update_agent (ppo_trainer.py:1045-1093) -------------------------------------------------------------
    def update(self):
        ro, s = self.rollouts, self.rollouts.step
        self._join_small()
        if self.belief is not None and self._belief_async and getattr(self.belief, "done", None) is not None:
            torch.cuda.current_stream().wait_event(self.belief.done)       # the last slot's beliefs
            self.pi_q.late_inputs((), None)
        last = {k: v[s] for k, v in ro.observations.items()}
        nv = self.pi_q.get_value_option(last, ro.recurrent_hidden_states[s], ro.prev_actions[s], ro.masks[s],
                                        ro.external_memory_option[:, s], ro.external_memory_masks[s],
                                        ro.query_state[s - 1], ro.last_query_info[s - 1])
        ro.compute_returns(nv, True, 0.99, 0.95)
        out = self.agent.update(ro)
        ro.after_update()
        return out

    def cycle(self, steps=None):
        for _ in range(steps or self.T):
            self.rollout_step()
        return self.update()


class GruWorkload:
    """BASELINE configs[1]: NUM_ENVS=16, `AudioNavBaselinePolicy` (AudioCNN + VisualCNN + single-layer GRU), PPO 4 epochs x 2
    minibatches, in the av_nav call pattern: act -> insert per step, then get_value -> compute_returns -> update ->
    after_update.  Synthetic observations as in `Workload`, resident in HBM."""

    def __init__(self, num_envs=16, num_steps=150, spectrogram=(257, 101, 2), precision="bf16x3", ppo_epoch=4, num_mini_batch=2,
                 device="cuda", seed=0, weight_seed=0, sampling="host", use_graphs=True):
        from . import av_nav
        self.N, self.T, self.dev = num_envs, num_steps, torch.device(device)
        osp, asp = savi_observation_space(spectrogram), ActionSpace(4)
        torch.manual_seed(weight_seed)
        self.pol = P.AudioNavBaselinePolicy(osp, asp, "spectrogram", hidden_size=512, precision=precision,
                                            sampling=sampling, use_graphs=use_graphs).to(self.dev)
        self.agent = av_nav.DDPPO(self.pol, clip_param=0.2, ppo_epoch=ppo_epoch, num_mini_batch=num_mini_batch,
                                  value_loss_coef=0.5, entropy_coef=0.01, lr=2.5e-4, eps=1e-5, max_grad_norm=0.5,
                                  use_normalized_advantage=False)
        self.agent.init_distributed(find_unused_params=True)
        self.rollouts = av_nav.RolloutStorage(num_steps, num_envs, osp, asp, 512, num_recurrent_layers=1, device=self.dev)
        T, N, dev = self.T, self.N, self.dev
        g = torch.Generator(device="cpu").manual_seed(2000 + seed)
        H, W, _ = spectrogram
        r = lambda *s_: torch.rand(*s_, generator=g)
        self.sim = {
            "rgb": torch.randint(0, 256, (T + 1, N, 128, 128, 3), generator=g, dtype=torch.uint8).to(dev),
            "depth": r(T + 1, N, 128, 128, 1).to(dev),
            "spectrogram": torch.log1p(3.0 * torch.randn(T + 1, N, H, W, 2, generator=g).abs()).to(dev),
            "category": torch.nn.functional.one_hot(torch.randint(0, 21, (T + 1, N), generator=g), 21).float().to(dev),
            "category_belief": torch.softmax(torch.randn(T + 1, N, 21, generator=g), -1).to(dev),
            "location_belief": (3.0 * torch.randn(T + 1, N, 2, generator=g)).to(dev),
            "pose": torch.stack([r(T + 1, N) * 20 - 10, r(T + 1, N) * 20 - 10, r(T + 1, N) * 2 * math.pi - math.pi,
                                 torch.arange(T + 1).float().view(-1, 1).expand(-1, N)], -1).to(dev),
        }
        self.rewards = torch.randn(T, N, 1, generator=g).to(dev)
        self.not_done = (r(T, N, 1) >= 1.0 / 150).float().to(dev)
        for k in self.rollouts.observations:
            self.rollouts.observations[k][0].copy_(self.sim[k][0])

    def rollout_step(self):
        ro, t = self.rollouts, self.rollouts.step
        obs = {k: v[t] for k, v in ro.observations.items()}
        v, a, lp, h, _, _ = self.pol.act(obs, ro.recurrent_hidden_states[t], ro.prev_actions[t], ro.masks[t], None, None)
        ro.insert({k: self.sim[k][t + 1] for k in ro.observations}, h, a, lp, v, self.rewards[t], self.not_done[t])

    def update(self):
        ro = self.rollouts
        nv = self.pol.get_value({k: v[-1] for k, v in ro.observations.items()}, ro.recurrent_hidden_states[-1], ro.prev_actions[-1],
                                ro.masks[-1], None, None)
        ro.compute_returns(nv, True, 0.99, 0.95)
        out = self.agent.update(ro)
        ro.after_update()
        return out

    def cycle(self, steps=None):
        for _ in range(steps or self.T):
            self.rollout_step()
        return self.update()

    def finite(self):
        ro = self.rollouts
        return bool(torch.isfinite(ro.value_preds).all()) and bool(torch.isfinite(ro.recurrent_hidden_states).all())
