"""Step sequencer: the launch-ahead flow of one rollout step (`prefetch_act_option` + `prefetch_act` +
`prefetch_act_dialog(dialog_later=True)`, then `dialog_ready`) as TWO calls into the C ABI (`avlen_cmds_run`).

The reference's step (ss_baselines/savi/ppo/ppo_trainer.py:449-636) is a serial chain with two host round trips on it:

    towers -> pi_q -> [host reads a_q, Speaker / clip.tokenize write `current_dialog`] -> CLIP text tower -> pi_l -> [host reads the
    actions, envs.step] -> next towers

On the GPU every forward is a captured HIP graph; what the host has to enqueue at each of the two points is fixed once the argument
buffers of the step are known.  `Policy.prefetch_*` / `dialog_ready` do that work in Python (graph lookup, staging plans, events made
per call, stream context managers: 0.1-0.2 ms each, measured in profiles/r04_step_timeline.txt -- more than the kernels they order).
The sequencer runs them ONCE per set of argument buffers (the "slow" pass: it also captures whatever is not captured yet), then
records what they enqueued as a command list -- graph launches, event records / waits, batched staging copies, on which streams
-- and from then on a step costs: the host generator's noise draws (in the reference's order), one `avlen_cmds_run` per phase,
and the bookkeeping that lets the policies' own `act_option` / `act` / `act_dialog` calls pick the results up (they validate
every argument address, exactly as after a `prefetch_*` call).

Nothing about the arithmetic changes: the same graphs, the same static buffers, the same stream order
(tests/test_gpu_text_cache.py, tests/test_gpu_harness_parity.py hold the storage bit-equal).
"""
import ctypes as C

import torch

from . import _lib as L
from . import config as CFG
from . import policy as P


def _ev():
    e = torch.cuda.Event()
    e.record(P._cur_stream())              # materialises the hipEvent_t (torch creates it at the first record)
    return e


class _Plan:
    __slots__ = ("B", "dev", "gq", "gg", "gl", "gt", "obs3", "grp_key", "key_q", "key_g", "key_l", "outs_q", "heads_q", "outs_g",
                 "heads_g", "out_l1", "outs_l", "heads_l", "ah_q", "ah_g", "ah_l", "a_early", "a_full", "b_variants", "b_build", "emb", "keep",
                 "tok_shape", "main", "aud_ev")


class StepSequencer:
    def __init__(self, pi_q, pi_g, pi_l):
        assert pi_q._enc_group is not None and pi_q._enc_group.leader is pi_q and pi_q._enc_group.members == [pi_q, pi_g, pi_l], \
            "the sequencer drives one EncoderGroup: share_encoders(pi_q, pi_g, pi_l) first"
        self.q, self.g, self.l, self.S = pi_q, pi_g, pi_l, pi_q._enc_group.side_stream()
        self._plans = {}
        self._gen = None
        self._cur = None                   # the running step: ("fast", plan, l_args) or ("slow", key, args, templates)
        self.fast = self.slow = 0
        self.ev = None
        # measured and settled (profiles/r05_sequencer_ab.txt): followers behind pi_q's WHOLE forward instead of behind the shared
        # encoders -- pi_q's chain does not get faster (it is slow because its weights are cold, not because of pi_g beside it) and
        # pi_g then crawls beside the text tower: 1.64 -> 1.89 ms per step; pi_l's dialog half on the side stream behind an event vs
        # on the caller's stream behind the text tower: no difference (1.65 ms both), the caller's stream needs one event less
        self.followers_after_q = False
        self.l2_on_side = False
        self.auto_running = False          # launch() was called by EncoderGroup.auto_sequence, not by the trainer

    # ------------------------------------------------------------------------------------------------------------------
    def _events(self):
        if self.ev is None:
            grp = self.q._enc_group
            if grp.ready is None:
                grp.ready = torch.cuda.Event()
            grp.ready.record(P._cur_stream())
            self.ev = dict(ready=grp.ready, done_q=_ev(), done_g=_ev(), done_l=_ev(), l_half=_ev(), text_read=_ev(), text=_ev())
        return self.ev

    @staticmethod
    def _key(q_args, g_args, l_args):
        obs = q_args[0]
        k = [obs[n].data_ptr() for n in sorted(obs)]
        for a in q_args[1:] + (g_args[4],) + l_args[3:]:
            k.append(a.data_ptr() if torch.is_tensor(a) else a)
        k.append(P._cur_stream().cuda_stream)
        return tuple(k)

    def _eligible(self, q_args, g_args, l_args):
        q, g, l = self.q, self.g, self.l
        return (all(p.sampling == "race" and p.use_graphs and p.precision in ("bf16", "bf16x3") for p in (q, g, l))
                and l.net.text_encoder_override is None and l_args[7] is not None
                and g_args[0] is q_args[0] and l_args[0] is q_args[0] and isinstance(q_args[0], dict)
                and not torch.cuda.is_current_stream_capturing())

    # ------------------------------------------------------------------------------------------------------------------
    def launch(self, q_args, g_args, l_args, explicit=True):
        """The three forwards of the step on their argument tuples (`act_option`'s, `act`'s, `act_dialog`'s positional arguments):
        pi_q's whole forward on the current stream, pi_g's and the half of pi_l's that reads neither the dialog tokens nor
        `agent_step` on the side stream.  Follow with `act_option(...)`, then -- once `all_dialog` / `agent_step` hold the step's
        values -- `dialog_ready()`, then `act(...)` / `act_dialog(...)`."""
        q, g, l = self.q, self.g, self.l
        gen = (q._graph_gen, g._graph_gen, l._graph_gen)
        if gen != self._gen:
            self._plans.clear()
            self._gen = gen
        key = self._key(q_args, g_args, l_args) if self._eligible(q_args, g_args, l_args) else None
        plan = self._plans.get(key) if key is not None else None
        if plan is None:
            self.slow += 1
            self.auto_running = not explicit
            try:
                q.prefetch_act_option(*q_args)
                g.prefetch_act(*g_args, stream=self.S)
                l.prefetch_act_dialog(*l_args, stream=self.S, dialog_later=True)
            finally:
                self.auto_running = False
            tmpl = None
            if key is not None and q._stash is not None and g._stash is not None and l._later is not None and l._later[0] == "vln":
                tmpl = (q._stash, g._stash, l._later, l._deferred)
            self._cur = ("slow", key, (q_args, g_args, l_args), tmpl)
            return
        self.fast += 1
        ev = self.ev
        q._engine(); g._engine(); l._engine()
        l.net._sync_text_cache(l)
        early, q._enc_early = q._enc_early, None
        use_early = early is not None and early[0] is plan.gq and (early[1] is None or early[1] == plan.obs3)
        if use_early and early[2] is not None and early[2] is not plan.aud_ev:
            P._cur_stream().wait_event(early[2])
        B, dev = plan.B, plan.dev
        r0 = torch.get_rng_state()
        q._draw_noise("option", B, dev)                 # the reference's draw order: pi_q, pi_g (, pi_l in dialog_ready)
        r1 = torch.get_rng_state()
        g._draw_noise("goal", B, dev)
        r2 = torch.get_rng_state()
        late = q._late_inputs
        if late is not None:
            # observation entries still being written when the towers started (Policy.late_inputs: the BeliefPredictor's beliefs, on
            # its own stream): everything behind the encoders waits for their event -- the list's first staging copy re-reads them
            P._cur_stream().wait_event(late[1])
            if not use_early:
                late = "full"
        cmds = plan.a_early if use_early else plan.a_full
        L.call("avlen_cmds_run", cmds, len(cmds))
        grp = q._enc_group
        grp.key = grp.ready_key = plan.grp_key
        grp.pending = set()
        grp.static_obs = plan.gq.static[0]
        q._last_lead = plan.gq
        q._stash = ("option", plan.key_q, ((plan.outs_q[0], q_args[1], plan.outs_q[2]), dict(plan.heads_q, rng_spec=(r0, r1))),
                    ev["done_q"])
        q._act_host["option"] = (plan.ah_q, ev["done_q"], True)
        g._stash = ("goal", plan.key_g, ((plan.outs_g[0], g_args[1], plan.outs_g[2]), dict(plan.heads_g, rng_spec=(r1, r2))),
                    ev["done_g"])
        g._act_host["goal"] = (plan.ah_g, ev["done_g"], True)
        l._stash = None
        l._deferred = plan.gl
        o1 = plan.out_l1
        l._later = ("vln", plan.key_l, ((o1[0][0], l_args[1], o1[0][2], o1[0][3]), dict(o1[1])), self.S, l_args[7], l_args[8], False)
        if explicit:
            q._in_prefetch_flow = g._in_prefetch_flow = l._in_prefetch_flow = True
            q._in_prefetch_explicit = g._in_prefetch_explicit = l._in_prefetch_explicit = True
        self._cur = ("fast", plan, l_args)

    def dialog_ready(self):
        """`all_dialog` and `agent_step` (the tensors passed to `launch`) hold the step's values: the frozen text tower on the
        current stream, behind it the dialog half of pi_l on the side stream."""
        cur, self._cur = self._cur, None
        l = self.l
        if cur is None:
            return l.dialog_ready()
        if cur[0] == "slow":
            l.dialog_ready()
            _, key, args, tmpl = cur
            if tmpl is not None and l._stash is not None:
                self._build(key, args, tmpl, l._stash)
            return
        _, plan, l_args = cur
        ev, net = self.ev, l.net
        lt = l._later                                        # (act_dialog of the automatic flow has put the call's own tensors there)
        if lt is None or lt[0] != "vln" or l._deferred is not plan.gl:
            return l.dialog_ready()
        key_l, tok, astep, det = lt[1], lt[4], lt[5], lt[6]
        var = plan.b_variants.get((tok.data_ptr(), astep.data_ptr())) if torch.is_tensor(tok) and torch.is_tensor(astep) else None
        if var is None and not det:
            var = plan.b_build(tok, astep)
        if var is None or det:
            return l.dialog_ready()                          # the Python flow finishes what the command list started
        b_text, b_cmds = var[0], var[1]
        if net._text_read is not None and net._text_read is not ev["text_read"]:
            P._cur_stream().wait_event(net._text_read)       # the embedding's last reader was enqueued by the slow path
        # the text tower first: it is the step's critical path from here (the host has just read pi_q's actions); pi_l's draw -- third
        # in the step, as in the reference -- and its dialog half go out while the tower runs
        L.call("avlen_cmds_run", b_text, len(b_text))
        r0 = torch.get_rng_state()
        l._draw_noise("vln", plan.B, plan.dev)
        r1 = torch.get_rng_state()
        L.call("avlen_cmds_run", b_cmds, len(b_cmds))
        net._text = (tok.data_ptr(), plan.tok_shape, plan.emb, ev["text_read"])
        net._text_key = ("pretext", plan.emb.data_ptr())
        net._text_read = ev["text_read"]
        l._later = l._deferred = None
        l._stash = ("vln", key_l, ((plan.outs_l[0], l_args[1], plan.outs_l[2], plan.outs_l[3]),
                                        dict(plan.heads_l, rng_spec=(r0, r1))), ev["done_l"])
        l._act_host["vln"] = (plan.ah_l, ev["done_l"], True)

    # ------------------------------------------------------------------------------------------------------------------
    def _build(self, key, args, tmpl, l_stash):
        """Record the step the slow pass just enqueued for these argument buffers as command lists.  Every piece is looked up the
        way the policies' own replay path finds it (`_memos`: the resolved launch per set of argument buffers); anything
        unexpected -- a forward that was not cut where the flow needs it, a staging copy that needs a conversion -- leaves the
        buffers on the slow pass."""
        q, g, l = self.q, self.g, self.l
        q_args, g_args, l_args = args
        try:
            mq = q._memos.get(P._memo_key(q, "option", "lead", q_args))
            mg = g._memos.get(P._memo_key(g, "goal", "follow", g_args))
            ml = l._memos.get(P._memo_key(l, "vln", "follow", l_args))
        except (KeyError, AttributeError):
            return
        tok = l_args[7]
        gt = l._graphs.get(("text", tuple(tok.shape))) if tok.dtype == torch.int64 and tok.is_contiguous() else None
        if mq is None or mg is None or ml is None or gt is None or mq.late is None:
            return
        gq, gg, gl = mq.g, mg.g, ml.g
        if gq.graph2 is None or gl.graph2 is None or gg.graph2 is not None or gt.graph2 is not None:
            return
        if mg.lead_static is not gq.static[0] or ml.lead_static is not gq.static[0] or tmpl[3] is not gl:
            return
        if gq.between is not None or gg.between is not None or gl.mid is not None or gg.mid is not None:
            return
        st_q, st_g, later, _ = tmpl
        if not (st_q[2][1].get("finished") and st_g[2][1].get("finished") and l_stash[2][1].get("finished")):
            return
        astep = l_args[8]
        s8 = gl.static[8]
        if not (torch.is_tensor(s8) and torch.is_tensor(astep) and astep.dtype == s8.dtype and astep.is_contiguous()
                and astep.numel() == s8.numel()):
            return
        ev = self._events()
        obs = q_args[0]
        B = obs["rgb"].shape[0]
        pl = _Plan()
        pl.B, pl.dev, pl.gq, pl.gg, pl.gl, pl.gt = B, obs["rgb"].device, gq, gg, gl, gt
        pl.obs3 = tuple(obs[k].data_ptr() for k in ("rgb", "depth", P.SPECTROGRAM))
        pl.grp_key = q._enc_group._key(obs)
        pl.key_q, pl.key_g, pl.key_l = st_q[1], st_g[1], l_stash[1]
        pl.outs_q, pl.heads_q = st_q[2][0], {k: v for k, v in st_q[2][1].items() if k != "rng_spec"}
        pl.outs_g, pl.heads_g = st_g[2][0], {k: v for k, v in st_g[2][1].items() if k != "rng_spec"}
        keep_keys = ("logits", "probs", "value", "unct", "raced", "action_host")       # the graph's static outputs only
        pl.out_l1 = (later[2][0], {k: v for k, v in later[2][1].items() if k in keep_keys})
        pl.outs_l, pl.heads_l = l_stash[2][0], {k: v for k, v in l_stash[2][1].items() if k != "rng_spec"}
        pl.ah_q, pl.ah_g, pl.ah_l = (p._pinned.get(("act_mapped", w, B)) for p, w in ((q, "option"), (g, "goal"), (l, "vln")))
        if pl.ah_q is None or pl.ah_g is None or pl.ah_l is None:
            return
        pl.emb, pl.tok_shape = gt.outs, tuple(tok.shape)
        if not torch.is_tensor(pl.emb):
            return
        M, S = P._cur_stream().cuda_stream, self.S.cuda_stream
        h = lambda e: e.cuda_event
        keep = []

        def copy(stream, srcs, dsts, sizes, n):
            return [(L.CMD_MULTICOPY, n, C.cast(srcs, C.c_void_p).value, C.cast(dsts, C.c_void_p).value,
                     C.cast(sizes, C.c_void_p).value, stream)] if n else []

        def pairs(stream, lst):
            lst = [(d, s) for d, s in lst if d.data_ptr() != s.data_ptr() and d.numel()]
            if not lst:
                return []
            n = len(lst)
            a = ((C.c_void_p * n)(*[s.data_ptr() for _, s in lst]), (C.c_void_p * n)(*[d.data_ptr() for d, _ in lst]),
                 (C.c_int64 * n)(*[d.numel() * d.element_size() for d, _ in lst]))
            keep.append(a)
            return copy(stream, a[0], a[1], a[2], n)

        follow = ([(L.CMD_WAIT, 0, S, h(ev["done_q" if self.followers_after_q else "ready"]), None, None)]
                  + copy(S, mg.srcs, mg.dsts, mg.sizes, mg.n)
                  + [(L.CMD_GRAPH, 0, gg.exec1, S, None, None), (L.CMD_RECORD, 0, h(ev["done_g"]), S, None, None)]
                  + copy(S, ml.srcs, ml.dsts, ml.sizes, ml.n)
                  + [(L.CMD_GRAPH, 0, gl.exec1, S, None, None), (L.CMD_RECORD, 0, h(ev["l_half"]), S, None, None)])
        tail = [(L.CMD_RECORD, 0, h(ev["ready"]), M, None, None), (L.CMD_GRAPH, 0, gq.exec2, M, None, None),
                (L.CMD_RECORD, 0, h(ev["done_q"]), M, None, None)] + follow
        pl.aud_ev = q._enc_group.audio_events()[1] if gq.exec_a is not None else None
        a_early = ([(L.CMD_WAIT, 0, M, h(pl.aud_ev), None, None)] if pl.aud_ev is not None else []) \
            + copy(M, mq.late[0], mq.late[1], mq.late[2], mq.late[3]) + tail
        a_full = copy(M, mq.srcs, mq.dsts, mq.sizes, mq.n) + [(L.CMD_GRAPH, 0, gq.exec1, M, None, None)] \
            + ([(L.CMD_GRAPH, 0, gq.exec_f, M, None, None)] if gq.exec_f is not None else []) \
            + ([(L.CMD_GRAPH, 0, gq.exec_a, M, None, None)] if gq.exec_a is not None else []) + tail
        def arr(cmds):
            a = (L.Cmd * len(cmds))()
            for i, (op, n, x, y, z, w) in enumerate(cmds):
                a[i].op, a[i].n, a[i].a, a[i].b, a[i].c, a[i].d = op, n, x, y, z, w
            return a

        def b_lists(tok, astep):
            # the text tower first (the critical path once pi_q's actions are on the host) ...
            b_text = ([(L.CMD_WAIT, 0, M, h(ev["text_read"]), None, None)]
                      + pairs(M, [(gt.static[0], tok)])
                      + [(L.CMD_GRAPH, 0, gt.exec1, M, None, None)])
            # ... then pi_l's dialog half on the CALLER's stream right behind it (the Python flow puts it on the side stream behind an
            # event; measured: no difference, one event less); its state-encoder half finished on the side stream long before (l_half)
            if self.l2_on_side:
                b = ([(L.CMD_RECORD, 0, h(ev["text"]), M, None, None), (L.CMD_WAIT, 0, S, h(ev["text"]), None, None)]
                     + pairs(S, [(s8, astep)])
                     + [(L.CMD_GRAPH, 0, gl.exec2, S, None, None), (L.CMD_RECORD, 0, h(ev["text_read"]), S, None, None),
                        (L.CMD_RECORD, 0, h(ev["done_l"]), S, None, None)])
            else:
                b = ([(L.CMD_WAIT, 0, M, h(ev["l_half"]), None, None)]
                     + pairs(M, [(s8, astep)])
                     + [(L.CMD_GRAPH, 0, gl.exec2, M, None, None), (L.CMD_RECORD, 0, h(ev["text_read"]), M, None, None),
                        (L.CMD_RECORD, 0, h(ev["done_l"]), M, None, None)])
            return b_text, b

        def b_build(tok, astep):
            """The second phase's two lists for the dialog tokens / agent_step tensors of a call (the recorded ones, or -- automatic
            flow -- whatever act_dialog was handed); None if a staging copy would need a conversion."""
            if not (torch.is_tensor(tok) and tok.is_cuda and tok.dtype == torch.int64 and tok.is_contiguous()
                    and tuple(tok.shape) == pl.tok_shape and torch.is_tensor(astep) and astep.is_cuda and astep.dtype == s8.dtype
                    and astep.is_contiguous() and astep.numel() == s8.numel()):
                return None
            bt, bb = b_lists(tok, astep)
            v = (arr(bt), arr(bb))
            while len(pl.b_variants) >= 64:
                pl.b_variants.pop(next(iter(pl.b_variants)))
            pl.b_variants[(tok.data_ptr(), astep.data_ptr())] = v
            return v

        pl.a_early, pl.a_full = arr(a_early), arr(a_full)
        pl.b_variants, pl.b_build = {}, b_build
        if b_build(tok, astep) is None:
            return
        pl.keep = (keep, mq, mg, ml)                      # the staging arrays the command lists point into
        pl.main = M
        while len(self._plans) >= 1024:
            self._plans.pop(next(iter(self._plans)))
        self._plans[key] = pl


CFG.add_ranges(StepSequencer, ('launch', 'dialog_ready'), "StepSequencer.")
