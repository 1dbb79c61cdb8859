"""Drop-in mirror of ss_baselines/savi/ppo/policy.py on the MI355X HIP library.

Same class names, constructor keywords, method signatures, return tuples and parameter names as the
reference (policy.py:39-276 Policy API; :299-356 policy classes; :379-1114 nets), so
`ss_baselines/savi/ddppo/algo/ddppo_trainer.py:301-512` can construct these instead.  Every tensor
operation below `act* / evaluate_actions* / get_value*` is a HIP kernel launched through the C ABI
(include/avlen_hip.h); PyTorch provides device memory, the stream and the host RNG.

Sampling: `Categorical.sample()` of the reference CPU path consumes the torch CPU generator
(SURVEY App. B).  sampling="race": the host draws the race's noise in the reference's order, the race runs on the device
(avlen_sample_race) -- same actions, no host round trip.  With sampling="host" (default) the (B,A) probabilities are brought to the host and the
action is drawn from that generator in the same order -> bit-exact actions for a fixed seed;
sampling="device" keeps everything on the GPU (like the reference would on a CUDA device).
"""
import contextlib
import ctypes as C
import time
import torch
import torch.nn as nn

from . import _lib as L
from . import config as CFG
from . import engine as E
from . import nets as N

DUAL_GOAL_DELIMITER = ","
_PREC = {"fp32": L.PREC_FP32, "bf16": L.PREC_BF16, "bf16x3": L.PREC_BF16X3, "fp16": L.PREC_FP16}
# precision="bf16x3" (the accurate fast mode): compensated bf16 everywhere except (i) the frozen CLIP text tower, which runs in
# fp16 -- the precision the reference itself runs it in on a CUDA device (clip.load converts the weights to half) -- and (ii) the
# AudioCNN, whose fp16 error (8x below bf16's, measured 1e-4 on the values) is far inside the tolerance
_MODE_MODULES = {"bf16x3": {"clip": "fp16", "audio": "fp16"}}

POSE, SPECTROGRAM, LOCATION_BELIEF, CATEGORY_BELIEF, CATEGORY = "pose", "spectrogram", "location_belief", \
    "category_belief", "category"     # soundspaces/tasks/nav.py cls_uuid values


class RowsOf:
    """Rows `index` (int32, device) of the first axis of `base` -- a batch that is read in place from a larger tensor (the PPO
    minibatch of the rollout storage) instead of being gathered first.  Only the bf16 encoders take it."""

    def __init__(self, base, index):
        assert base.dtype in (torch.float32, torch.uint8) and base.is_contiguous() and index.dtype == torch.int32 \
            and index.is_contiguous()
        self.base, self.index = base, index
        self.shape = (index.numel(),) + tuple(base.shape[1:])
        self.device = base.device

    def materialise(self):
        return self.base[self.index.long()]


def _f32(t):
    if isinstance(t, RowsOf):
        return t
    if t.dtype != torch.float32:
        t = t.float()
    return t if t.is_contiguous() else t.contiguous()


def _img(t):
    """An image sensor as the kernels take it: uint8 pixels stay uint8 (RGB is kept uint8 from the simulator through the
    rollout storage into the tower prologue, which converts and divides exactly as the fp32 path does); anything else fp32."""
    if isinstance(t, RowsOf):
        return t
    if t.dtype == torch.uint8:
        return t if t.is_contiguous() else t.contiguous()
    return _f32(t)


def _u8(t):
    return 1 if (t.base if isinstance(t, RowsOf) else t).dtype == torch.uint8 else 0


def _i64(t):
    if t.dtype != torch.int64:
        t = t.long()
    return t if t.is_contiguous() else t.contiguous()


_MAPPED_ACTIONS = CFG.MAPPED_ACTIONS     # heads kernel stores the sampled actions into mapped pinned memory
_FOLD_TEXT = CFG.FOLD_TEXT               # text_projection folded into dialog_layer in the rollout's text graph
_SPLIT = True        # pi_l's captured forward is cut in two around the text embedding (decided by measurement: DESIGN section 3)


_CAP = None
_STREAM_OBJ = {}
_NAMED = {}


def process_stream(name, priority=0):
    """A process-wide stream by role name.  torch hands out streams from a pool of 32 per device and priority, ROUND-ROBIN and
    without tracking who still uses them: a process that keeps creating streams (one set per policy, per workload, per test)
    eventually gets the pool slot of a stream that is still in use -- e.g. a policy's capture-fork stream that IS the shared
    capture stream, so the captured forward forks onto the very stream it is being captured on (seen as a segmentation fault
    at the first replay of such a graph, after ~30 policies in one process).  Every auxiliary stream of this package therefore
    exists once per process and role."""
    key = (name, priority, torch._C._cuda_getDevice())
    st = _NAMED.get(key)
    if st is None:
        st = _NAMED[key] = torch.cuda.Stream(priority=priority)
    return st


def _cur_stream():
    """_cur_stream() without building a Stream object per call (3 us each, 16 calls per rollout step): the raw
    handle of the current stream is one C call, the Stream object for it is cached."""
    dev = torch._C._cuda_getDevice()
    raw = torch._C._cuda_getCurrentRawStream(dev)
    s = _STREAM_OBJ.get((dev, raw))
    if s is None:
        s = _STREAM_OBJ[(dev, raw)] = torch.cuda.current_stream()
    return s


def _capture_stream():
    """ONE capture stream for every graph (as torch.cuda.graph does): each extra stream shifts the round-robin mapping of
    streams to the 4 hardware queues, and with it which replaying graphs end up serialised behind each other."""
    return process_stream("capture")


class _Graph:
    """One captured forward: static input buffers + the HIP graph + its (static) outputs."""

    def __init__(self, pol, fn, args, by_pointer=()):
        # `by_pointer` args (external-memory rings) are read in place: the graph is keyed on their address
        self.static = [a if (i in by_pointer and (torch.is_tensor(a) or isinstance(a, dict))) else
                       (a.clone() if torch.is_tensor(a) else
                        ({k: v.clone() for k, v in a.items()} if isinstance(a, dict) else a))
                       for i, a in enumerate(args)]
        ws_saved = pol._ws
        pol._ws = E.Workspaces()                         # this graph owns its scratch
        try:
            side = _capture_stream()
            side.wait_stream(_cur_stream())
            with torch.cuda.stream(side):                # warm-up outside capture (lazy HIP init, attributes)
                fn(*self.static)
            _cur_stream().wait_stream(side)
            torch.cuda.synchronize()
            # Captured by hand (not `with torch.cuda.graph`) so that the forward may cut itself into TWO graphs at
            # `split()`: the host can then wait for an outside event (pi_l's text tower) between the two replays.  The leader of an
            # EncoderGroup cuts a third, LINEAR piece out (`split_audio()`: the AudioCNNs, between the towers and the rest): a
            # forked branch inside one graph made its launch cost 54 us on the step's critical path (the towers' launch: 11 us as a
            # linear graph), and as a graph of its own the audio branch can be replayed on another stream.
            # A compensated-bf16 leader cuts a FOURTH piece (`split_fc()`): the towers' fc GEMM, so that the replay can run the
            # AudioCNNs directly behind the towers on the caller's stream and the fc beside them on the side stream.
            self.graph, self.graph_f, self.graph_a, self.graph2, self.between, self.mid = torch.cuda.CUDAGraph(), None, None, None, None, None
            cap = _capture_stream()
            cap.wait_stream(_cur_stream())
            pol._capture = self
            try:
                with torch.cuda.stream(cap):
                    self.graph.capture_begin()
                    self._capturing = self.graph
                    try:
                        self.outs = fn(*self.static)
                    finally:
                        self._capturing.capture_end()
                        self._capturing = None
            finally:
                pol._capture = None
            _cur_stream().wait_stream(cap)
            self.ws = pol._ws
            # raw hipGraphExec_t handles: the step sequencer (sequencer.py) launches them from its recorded command lists
            self.exec1 = self.graph.raw_cuda_graph_exec()
            self.exec_f = self.graph_f.raw_cuda_graph_exec() if self.graph_f is not None else None
            self.exec_a = self.graph_a.raw_cuda_graph_exec() if self.graph_a is not None else None
            self.exec2 = self.graph2.raw_cuda_graph_exec() if self.graph2 is not None else None
        finally:
            pol._ws = ws_saved

    def split(self):
        """Called by the forward under capture: everything enqueued so far becomes graph 1 (+ the audio piece), the rest graph 2
        (same pool)."""
        if self.graph2 is not None:
            return
        self._capturing.capture_end()
        self.graph2 = self._capturing = torch.cuda.CUDAGraph()
        self.graph2.capture_begin(self.graph.pool())

    def split_audio(self):
        """Called by the leader's forward under capture right behind the visual towers: what follows up to `split()` (the
        AudioCNNs) becomes a graph of its own."""
        if self.graph_a is not None or self.graph2 is not None:
            return
        self._capturing.capture_end()
        self.graph_a = self._capturing = torch.cuda.CUDAGraph()
        self.graph_a.capture_begin(self.graph.pool())

    def split_fc(self):
        """Called by a compensated-bf16 leader right behind the tower launch: the towers' fc (and the weight prefetch behind it)
        becomes a graph of its own, in front of the audio piece."""
        if self.graph_f is not None or self.graph_a is not None or self.graph2 is not None:
            return
        self.graph.capture_end()
        self.graph_f = self._capturing = torch.cuda.CUDAGraph()
        self.graph_f.capture_begin(self.graph.pool())

    def replay_first(self):
        """Everything in front of the cut, on the current stream."""
        self.graph.replay()
        if self.graph_f is not None:
            self.graph_f.replay()
        if self.graph_a is not None:
            self.graph_a.replay()

    def staging_pairs(self, args):
        """(dst, src) copies that refresh the static inputs from `args`, or None if one of them is not a plain device copy."""
        pairs = []
        for s, a in zip(self.static, args):
            if torch.is_tensor(s):
                if s.data_ptr() != a.data_ptr():
                    pairs.append((s, a))
            elif isinstance(s, dict) and s is not a:
                pairs += [(s[k], a[k]) for k in s]
        for d, s_ in pairs:
            if not (torch.is_tensor(s_) and s_.is_cuda and s_.dtype == d.dtype and s_.is_contiguous() and s_.numel() == d.numel()):
                return None
        return [(d, s_) for d, s_ in pairs if s_.numel() and s_.data_ptr() != d.data_ptr()]

    def staging_pairs_any(self, args):
        """(dst, src) refreshes of the static inputs from `args` (any source: L.multi_copy falls back to copy_ where needed)."""
        pairs = []
        for s, a in zip(self.static, args):
            if torch.is_tensor(s):
                if s.data_ptr() != a.data_ptr():
                    pairs.append((s, a))
            elif isinstance(s, dict) and s is not a:
                pairs += [(s[k], a[k]) for k in s]
        return pairs

    def replay_only(self):
        if self.graph2 is None:
            if self.between is not None:
                self.between()
            self.graph.replay()
        else:
            self.replay_first()
            if self.between is not None:
                self.between()
            if self.mid is not None:
                self.mid()                               # only ever between the halves of a cut graph (never in front of a whole one)
            self.graph2.replay()
        return self.outs

    def __call__(self, args):
        pairs = []
        for s, a in zip(self.static, args):
            if torch.is_tensor(s):
                if s.data_ptr() != a.data_ptr():
                    pairs.append((s, a))
            elif isinstance(s, dict) and s is not a:
                pairs += [(s[k], a[k]) for k in s]
        L.multi_copy(pairs)                              # one launch for all static-input refreshes
        if self.graph2 is None:
            if self.between is not None:
                self.between()
            self.graph.replay()
        else:
            self.replay_first()
            if self.between is not None:
                self.between()                           # e.g. wait for the text tower's event on this stream
            if self.mid is not None:
                self.mid()
            self.graph2.replay()
        return self.outs


def _sig(a):
    if torch.is_tensor(a):
        return (tuple(a.shape), a.dtype)
    if isinstance(a, dict):
        return tuple((k, tuple(v.shape), v.dtype) for k, v in sorted(a.items()))
    return a


class _Memo:
    """A resolved launch of `_graphed` for one exact set of argument BUFFERS: the graph and the staging copy as ready ctypes
    arrays.  Keyed on (address, dtype, shape, contiguity) of every tensor of the call -- not on Python object identity: a trainer
    that slices fresh views of its storage every step (ppo_trainer.py:375-391) presents new objects over the same memory, and an
    `id()` can be recycled by an unrelated object.  The memo holds no tensors: the copy reads whatever lives at the source
    addresses at replay time, which is exactly what the caller passed."""
    __slots__ = ("g", "srcs", "dsts", "sizes", "n", "lead_static", "late")     # late: (srcs, dsts, sizes, n) without the big sensors


def _tsig(a):
    return (a.data_ptr(), a.dtype, tuple(a.shape), a.is_contiguous())


def _memo_key(pol, which, mode, args):
    obs = args[0]
    skip = (0,) if getattr(pol.net, "reads_rnn_state", False) else (0, 1, 3)
    return (which, mode, getattr(pol.net, "_text_key", None)) + tuple(_tsig(obs[k]) for k in pol.net.obs_keys) + tuple(
        _tsig(a) if torch.is_tensor(a) else a for i, a in enumerate(args) if i not in skip)


def _graphed(pol, which, fn, args, mode=None):
    # rnn_hidden_states (arg 1) and masks (arg 3) are not read by the SMT nets: keep them out of the graph.  The GRU baseline
    # (net.reads_rnn_state) reads both: they become staged inputs and the graph's new hidden state is returned.
    raw = args
    keep_rnn = getattr(pol.net, "reads_rnn_state", False)
    # derived weights (packed convs, bf16 shadows) must be current BEFORE any replay: a load_state_dict / mark_params_changed
    # since the last call only set the dirty flag.  The leader of an EncoderGroup replays the followers' towers too.
    pol._engine()
    if mode == "lead":
        for m_ in pol._enc_group.members:
            if m_ is not pol:
                m_._engine()
    mk = None
    if isinstance(raw[0], dict) and not getattr(pol, "_memo_off", False):
        try:
            mk = _memo_key(pol, which, mode, raw)
        except (KeyError, AttributeError):
            mk = None
    m = pol._memos.get(mk) if mk is not None else None
    if m is not None and (mode != "follow" or pol._enc_group.static_obs is m.lead_static):
        g = m.g
        if mode == "lead":
            pol._enc_group.static_obs = g.static[0]
        early = pol._enc_early
        pol._enc_early = None
        if early is not None and early[0] is g and mode == "lead" and g.graph2 is not None and m.late is not None and \
                (early[1] is None or early[1] == tuple(raw[0][k].data_ptr() for k in ("rgb", "depth", SPECTROGRAM))):
            # prefetch_encoders() already staged this observation's sensors and replayed the encoder half: the small inputs, then
            # the rest of the forward
            if len(early) > 2 and early[2] is not None:
                _cur_stream().wait_event(early[2])       # the audio piece ran on the group's side stream
            if m.late[3]:
                L.call("avlen_multi_copy", m.late[0], m.late[1], m.late[2], m.late[3], L.stream())
            g.mid = getattr(pol, "_mid", None)
            if g.mid is not None:
                g.mid()
            g.graph2.replay()
            outs, heads = g.outs
            outs = list(outs)
            if not keep_rnn:
                outs[1] = raw[1]
            return tuple(outs), dict(heads)
        if m.n:
            L.call("avlen_multi_copy", m.srcs, m.dsts, m.sizes, m.n, L.stream())
        g.between = getattr(pol, "_between", None)
        g.mid = getattr(pol, "_mid", None)
        if mode == "lead":
            pol._last_lead = g
        if pol._defer_second and g.graph2 is not None:
            g.replay_first()                             # the half that does not read the dialog; dialog_ready() replays the rest
            pol._deferred = g
            outs, heads = g.outs
        else:
            outs, heads = g.replay_only()
        outs = list(outs)
        if not keep_rnn:
            outs[1] = raw[1]
        return tuple(outs), dict(heads)
    args = list(args)
    rnn = args[1]
    if keep_rnn:
        args[1], args[3] = _f32(args[1]), _f32(args[3])
    else:
        args[1], args[3] = None, None
    args[0] = {k: (_img(v) if k == "rgb" else _f32(v)) for k, v in args[0].items() if k in pol.net.obs_keys}
    by_ptr = (4, 5) if which == "vln" else (4,)          # ext_memory (+ dialog memory): persistent ring buffers
    for i in by_ptr:
        if torch.is_tensor(args[i]):
            args[i] = _f32(args[i])
    grp = pol._enc_group
    if mode == "follow" and grp.static_obs is not None and _sig(grp.static_obs) == _sig(args[0]):
        # same observation as the leader's call: read the leader graph's static copy in place (no second staging copy)
        args[0] = grp.static_obs
        by_ptr = by_ptr + (0,)
    key = (which, mode, getattr(pol.net, "_text_key", None)) + tuple(_sig(a) for a in args) + tuple(
        (args[i].data_ptr() if torch.is_tensor(args[i]) else id(args[i])) for i in by_ptr)
    g = pol._graphs.get(key)
    if g is None:
        if len(pol._graphs) >= 16:
            torch.cuda.synchronize()                     # a graph that is still executing must not be destroyed under the GPU
            pol._graphs.clear()
            pol._memos.clear()
            pol._graph_gen += 1
        g = pol._graphs[key] = _Graph(pol, fn, args, by_ptr)
    if mode == "lead":
        grp.static_obs = g.static[0]
        pol._last_lead = g
    pol._enc_early = None
    g.between = getattr(pol, "_between", None)
    g.mid = getattr(pol, "_mid", None)
    if mk is not None:
        # memoise when every tensor was used exactly as passed (no dtype / layout conversion made a temporary)
        followed = mode == "follow" and args[0] is grp.static_obs
        same = followed or all(args[0][k] is raw[0][k] for k in args[0])
        same = same and all(args[i] is raw[i] for i in range(len(raw)) if i != 0 and (keep_rnn or i not in (1, 3)))
        pairs = g.staging_pairs(args) if same else None
        if pairs is not None:
            mm = _Memo()
            mm.g = g
            mm.n = len(pairs)
            mm.srcs = (C.c_void_p * max(mm.n, 1))(*[s_.data_ptr() for _, s_ in pairs])
            mm.dsts = (C.c_void_p * max(mm.n, 1))(*[d_.data_ptr() for d_, _ in pairs])
            mm.sizes = (C.c_int64 * max(mm.n, 1))(*[d_.numel() * d_.element_size() for d_, _ in pairs])
            mm.lead_static = grp.static_obs if mode == "follow" else None
            big = set()
            if mode == "lead" and isinstance(g.static[0], dict):
                big = {g.static[0][k].data_ptr() for k in ("rgb", "depth", SPECTROGRAM) if k in g.static[0]}
            lp = [(d_, s_) for d_, s_ in pairs if d_.data_ptr() not in big]
            mm.late = ((C.c_void_p * max(len(lp), 1))(*[s_.data_ptr() for _, s_ in lp]),
                       (C.c_void_p * max(len(lp), 1))(*[d_.data_ptr() for d_, _ in lp]),
                       (C.c_int64 * max(len(lp), 1))(*[d_.numel() * d_.element_size() for d_, _ in lp]), len(lp)) if mode == "lead" else None
            # bounded: a caller that presents freshly allocated observation tensors every step (eval: batch_obs) would otherwise
            # add one memo per distinct address tuple for ever; the oldest entries go first (dicts keep insertion order)
            while len(pol._memos) >= 512:
                pol._memos.pop(next(iter(pol._memos)))
            pol._memos[mk] = mm
    if pol._defer_second and g.graph2 is not None:
        L.multi_copy(g.staging_pairs_any(args))
        g.replay_first()
        pol._deferred = g
        outs, heads = g.outs
    else:
        outs, heads = g(args)
    outs = list(outs)
    if not keep_rnn:
        outs[1] = rnn
    return tuple(outs), dict(heads)


class EncoderGroup:
    """Policies that are evaluated on the SAME observation every step (pi_q, pi_g, pi_l in
    ppo_trainer.py:449-636) can run their visual towers together: when the leader (the policy called first,
    pi_q) runs, all 2*len(members) ResNet towers execute as grouped launches (one launch per layer for all of
    them); a follower called afterwards with the same rgb/depth tensors picks its 128 visual features up
    instead of recomputing them.  Numerically identical to separate calls (same kernels, same operands)."""

    def __init__(self, leader, followers):
        self.members = [leader] + list(followers)
        self.leader = leader
        self.out = {}                  # B -> [tensor (B,128)] per member
        self.static_obs = None         # the leader graph's static observation buffers (followers read them in place)
        self.key = None
        self.pending = set()
        # recorded on the leader's stream as soon as the shared encoders (all towers, all AudioCNNs) of a leader call are enqueued:
        # a follower launched ahead on its own stream waits for THIS, not for the rest of the leader's forward
        self.ready = None
        self.ready_key = None
        for m in self.members:
            m._enc_group = self
        # Launch-ahead WITHOUT trainer changes: when the leader's act* call returns, the group enqueues the followers' forwards itself
        # on a side stream -- with the arguments the followers are about to be called with, predicted from their previous calls
        # (see auto_launch) -- and the followers' own calls pick the results up after validating every address.  AVLEN_AUTO_AHEAD=0
        # switches it off; explicit prefetch_* calls by the caller take precedence.
        self.auto = CFG.AUTO_AHEAD
        self._auto_stream = None
        self.auto_hits = self.auto_misses = 0
        self._side = None
        self._aud_ev = self._stage_ev = None
        self._seq = None                                 # the group's StepSequencer (auto_sequence), built on first use

    def side_stream(self):
        """The ONE side stream of the group: the followers' forwards (launched ahead by the trainer, by auto_launch or by the step
        sequencer) and the leader's audio piece when the encoders are started early (Policy.prefetch_encoders).  One stream for all
        of them: the process has four hardware queues, and the caller's stream, this one and the storage's are three."""
        if self._side is None:
            self._side = process_stream("followers")
        return self._side

    def audio_events(self):
        """(staged, done): persistent events around the leader's audio piece when it runs on the side stream."""
        if self._aud_ev is None:
            self._aud_ev, self._stage_ev = torch.cuda.Event(), torch.cuda.Event()
            self._aud_ev.record(_cur_stream())
            self._stage_ev.record(_cur_stream())
        return self._stage_ev, self._aud_ev

    @staticmethod
    def _key(obs):
        return (obs["rgb"].data_ptr(), obs["depth"].data_ptr(), obs[SPECTROGRAM].data_ptr(), tuple(obs["rgb"].shape))

    def mark(self, obs):
        self.key = self._key(obs)
        self.pending = set(id(m) for m in self.members[1:])
        self.ready_key = None

    def signal(self):
        """The shared encoders of the marked observation are enqueued on the current stream."""
        if self.ready is None:
            self.ready = torch.cuda.Event()
        self.ready.record(_cur_stream())
        self.ready_key = self.key

    def claim(self, pol, obs):
        """True once per leader call for a follower that presents the leader's observation tensors."""
        k = self._key(obs)
        if k == self.key and id(pol) in self.pending:
            self.pending.discard(id(pol))
            return True
        return False

    # ---- automatic launch-ahead --------------------------------------------------------------------------------------------
    @staticmethod
    def _same(a, b):
        if torch.is_tensor(a) and torch.is_tensor(b):
            return a.data_ptr() == b.data_ptr() and a.shape == b.shape and a.dtype == b.dtype
        if isinstance(a, dict) and isinstance(b, dict):
            return a.keys() == b.keys() and all(EncoderGroup._same(a[k], b[k]) for k in a)
        return a is b

    @staticmethod
    def _next_view(cur, prev):
        """The view the caller will pass next if `prev`, `cur` were `base[t - 1]`, `base[t]` of one storage tensor, else None."""
        if not (torch.is_tensor(cur) and torch.is_tensor(prev)) or cur._base is None or cur._base is not prev._base:
            return None
        base = cur._base
        if base.dim() < 1 or cur.shape != base.shape[1:] or cur.dtype != base.dtype:
            return None
        step = base.stride(0) * base.element_size()
        if step <= 0 or cur.data_ptr() - prev.data_ptr() != step:
            return None
        t = (cur.data_ptr() - base.data_ptr()) // step
        return base[t + 1] if 0 <= t + 1 < base.shape[0] else None

    def auto_launch(self, leader, lead_args):
        """Called when the leader's direct act* call has enqueued its forward (and drawn its sampling noise).  Every follower that
        has been called directly before gets its next forward enqueued on the group's side stream.  Its arguments are predicted:
        an argument that had the address of one of the LEADER's arguments last time (the step's observation dict, hidden state,
        previous actions, masks -- the trainer passes the same tensors to all three policies, ppo_trainer.py:449-636) becomes the
        leader's current one; an address that did not move between the follower's last two calls (external-memory rings) is kept;
        `storage[t]` views advance to `storage[t + 1]`.  The follower's real call validates every address (Policy._arg_key): a
        wrong guess costs one discarded forward, never a wrong result."""
        lead_prev = leader._call_hist[-2][1] if len(leader._call_hist) > 1 else None
        if lead_prev is None or not self.auto:
            return
        if self._auto_stream is None:
            self._auto_stream = self.side_stream()
        for f in self.members[1:]:
            p = self._predict(f, lead_prev, lead_args)
            if p is None:
                continue
            which, pred, det = p
            f._auto_pending = True
            f._prefetch(which, *pred, stream=self._auto_stream, dialog_later=(which == "vln" and pred[7] is not None),
                        deterministic=det)

    def _predict(self, f, lead_prev, lead_args):
        """-> (head set, predicted argument list, predicted `deterministic`) of follower f's next call, or None (see auto_launch)."""
        h = f._call_hist
        if len(h) < 2 or h[-1][0] != h[-2][0] or f._stash is not None or f._later is not None:
            return None
        which, last, det = h[-1]                         # the follower is expected to pass the `deterministic` it passed last time
        before = h[-2][1]
        pred = []
        for i, a in enumerate(last):
            j = next((j for j, b in enumerate(lead_prev) if a is not None and self._same(a, b)), None)
            if which == "vln" and i in (7, 8):
                pred.append(a)                           # dialog tokens / agent_step: placeholders, read by the second half only
            elif j is not None and j < len(lead_args):
                pred.append(lead_args[j])
            elif a is None or not torch.is_tensor(a):
                pred.append(a)
            elif self._same(a, before[i]):
                pred.append(a)
            else:
                pred.append(self._next_view(a, before[i]))
                if pred[-1] is None:
                    return None
        return which, pred, det

    def sequencer(self):
        if self._seq is None:
            from .sequencer import StepSequencer
            self._seq = StepSequencer(*self.members)
        return self._seq

    def auto_sequence(self, leader, lead_args):
        """The automatic launch-ahead through the step sequencer (sequencer.py), tried BEFORE the leader's direct act_option enqueues
        anything: with both followers' next calls predicted (as auto_launch predicts them) and all three sampling, the step's three
        forwards go out as one recorded command list -- the leader's own call then picks its forward up like the followers do.
        -> True if the step was launched."""
        if not self.auto or len(self.members) != 3 or not leader._call_hist or leader._in_prefetch_explicit:
            return False
        if not all(m.use_graphs and m.sampling == "race" for m in self.members):
            return False
        lead_prev = leader._call_hist[-1][1]             # (this call has not been recorded yet)
        if len(lead_prev) != len(lead_args):
            return False
        pg = self._predict(self.members[1], lead_prev, lead_args)
        pl = self._predict(self.members[2], lead_prev, lead_args)
        if pg is None or pl is None or pg[0] != "goal" or pl[0] != "vln" or pg[2] or pl[2] or pl[1][7] is None:
            return False
        self.sequencer().launch(tuple(lead_args), tuple(pg[1]), tuple(pl[1]), explicit=False)
        for f in self.members[1:]:
            f._auto_pending = True
        leader._auto_seq_step = True                     # this step's followers are out: _after_act does not launch them again
        return True

    def buffers(self, B, dev):
        if B not in self.out:
            self.out[B] = [torch.empty(B, 128, device=dev) for _ in self.members]
        return self.out[B]

    def audio_buffers(self, B, dev):
        if ("a", B) not in self.out:
            self.out[("a", B)] = [torch.empty(B, 128, device=dev) for _ in self.members]
        return self.out[("a", B)]

    def run_audio(self, pol, spec):
        """All members' AudioCNNs on the shared spectrogram: one cast, one grouped launch per layer."""
        B, H, W = spec.shape[0], spec.shape[1], spec.shape[2]
        bufs = self.audio_buffers(B, spec.device)
        G = len(self.members)
        pa = pol.prec_of("audio")
        if pa not in (L.PREC_BF16, L.PREC_FP16):         # no grouped form in this precision: one call per member
            for m, buf in zip(self.members, bufs):
                eng = m._engine()
                nb = L.lib.avlen_cnn3_workspace_bytes(C.byref(eng["audio"]), B, H, W)
                ws = pol._ws.get("audio_m%d" % self.members.index(m), nb, spec.device)
                L.call("avlen_cnn3_fwd", C.byref(eng["audio"]), E.P(spec), B, H, W, E.P(buf), 128, pa, E.P(ws), nb, L.stream())
            return bufs[0]
        nets = (C.POINTER(L.Cnn3) * G)(*[C.pointer(m._engine()["audio"]) for m in self.members])
        outs = (C.c_void_p * G)(*[b.data_ptr() for b in bufs])
        nb = L.lib.avlen_cnn3_group_workspace_bytes(nets[0], G, B, H, W)
        ws = pol._ws.get("audio_group", nb, spec.device)
        L.call("avlen_cnn3_group_fwd", nets, E.P(spec), G, B, H, W, outs, 128, E.P(ws), nb, L.stream())
        return bufs[0]

    def run_all(self, pol, rgb, depth, phase=3):
        """phase (compensated bf16 only): 1 = the tower launch, 2 = the fc on its outputs, 3 = both."""
        B, dev = rgb.shape[0], rgb.device
        bufs = self.buffers(B, dev)
        G = 2 * len(self.members)
        nets = (C.POINTER(L.ResNet18) * G)()
        imgs, outs = (C.c_void_p * G)(), (C.c_void_p * G)()
        chans, divs, u8 = (C.c_int * G)(), (C.c_float * G)(), (C.c_int * G)()
        for i, m in enumerate(self.members):
            eng = m._engine()
            for j, (key, img, div) in enumerate((("rgb", rgb, 255.0), ("depth", depth, 1.0))):
                g = 2 * i + j
                nets[g] = C.pointer(eng[key])
                imgs[g] = img.data_ptr()
                outs[g] = bufs[i].data_ptr() + 4 * 64 * j
                chans[g], divs[g], u8[g] = img.shape[3], div, _u8(img)
        if pol.prec_of("towers") == L.PREC_BF16X3:
            nb = L.lib.avlen_resnet18_group_x3_workspace_bytes(G, B)
            ws = pol._ws.get("resnet_group_x3", nb, dev)
            L.call("avlen_resnet18_group_fwd_x3_phase", nets, imgs, u8, chans, divs, outs, 128, G, B, rgb.shape[1], None, phase,
                   E.P(ws), nb, L.stream())
            return bufs[0]
        assert phase == 3
        nb = L.lib.avlen_resnet18_group_workspace_bytes(G, B)
        ws = pol._ws.get("resnet_group", nb, dev)
        L.call("avlen_resnet18_group_fwd", nets, imgs, u8, chans, divs, outs, 128, G, B, rgb.shape[1], E.P(ws), nb, L.stream())
        return bufs[0]


def share_encoders(leader, *followers, rollouts=None):
    """Opt-in cross-policy tower batching (see EncoderGroup).  All members must use the same fast precision ("bf16" / "bf16x3").
    `rollouts` (optional, the trainer's RolloutStorage): `rollouts.insert(batch, ...)` then starts the NEXT step's shared encoders on
    `batch` before it copies anything (Policy.prefetch_encoders, validated by address at the next act_option as always) -- the
    towers hide the storage bookkeeping and the next step's launch path without another line in the trainer."""
    assert leader.precision in ("bf16", "bf16x3") and all(m.precision == leader.precision for m in followers), \
        "encoder sharing runs on the grouped fast paths (bf16 / bf16x3), one precision for all members"
    grp = EncoderGroup(leader, followers)
    if rollouts is not None:
        rollouts.attach_encoders(leader)
    return grp


class _Dist:
    """What the trainer reads off the reference's CustomFixedCategorical."""
    def __init__(self, logits, probs):
        self.logits, self.probs = logits, probs


class Policy(nn.Module):
    """policy.py:39-276."""

    def __init__(self, net, dim_actions, dim_actions_option=2, precision="fp32", sampling="host", use_graphs=False):
        super().__init__()
        self.net = net
        self.dim_actions, self.dim_actions_option = dim_actions, dim_actions_option
        d = self.net.output_size
        self.action_distribution_option = N.CategoricalNetParams(d, dim_actions_option)
        self.action_distribution_goal = N.CategoricalNetParams(d, dim_actions)
        self.action_distribution_vln = N.CategoricalNetParams(d, dim_actions)
        self.critic_goal = N.CriticHeadParams(d)
        self.critic_option = N.CriticHeadParams(d)
        self.uncertainty_option = N.CriticHeadParams(d, 2)
        self.critic_vln = N.CriticHeadParams(d)
        self.precision, self.sampling = precision, sampling
        self.module_precision = dict(_MODE_MODULES.get(precision, {}))
        self.use_graphs = use_graphs          # capture each act*/get_value* forward in a HIP graph (static shapes)
        self._graphs = {}
        self._memos = {}
        self._side = None
        self._enc_group = None
        self._shared_mode = None
        self._stash = None
        self._memos = {}                      # resolved launches per exact argument objects (see _Memo)
        self._capture = None                  # the _Graph being captured (lets a forward cut itself in two, see _Graph.split)
        self._post = None                     # act* forwards: closures the net defers behind the heads kernel (see _forward.eager)
        self._between = None                  # host action between the two halves of a split graph
        self._mid = None                      # leader of an EncoderGroup: EncoderGroup.signal between the halves of its cut graph
        self._late_inputs = None              # (observation keys, event): see late_inputs()
        self._last_lead = None                # leader: the captured graph of its last act* forward
        self._enc_early = None                # leader: that graph, if prefetch_encoders() has run its encoder half for the next call
        self._enc_plans = {}                  # staging plans of prefetch_encoders per source address set
        self._defer_second = False            # prefetch_act_dialog(dialog_later=True): replay only the first half of the cut graph
        self._deferred = None                 # ... the graph whose second half dialog_ready() replays
        self._later = None                    # ... (which, arg key, outputs, stream, all_dialog, agent_step) of that prefetch
        self.last_host_action = None          # sampling="host": pinned (B,1) int64 of the most recent draw
        self._act_host = {}                   # head set -> (pinned actions, event or None): see host_actions()
        self._call_hist = []                  # the last two DIRECT act* calls (which, net_args): EncoderGroup.auto_launch
        self._auto_pending = False            # a forward enqueued by EncoderGroup.auto_launch is waiting for its call
        self._param_epoch = 0                 # bumped by mark_params_changed (derived state keyed on the weights: the text memo)
        self._graph_gen = 0                   # bumped whenever captured graphs are dropped (plans that hold their handles die with them)
        self._pinned = {}
        self._poll_views = {}
        self._eng = None
        self._ws = E.Workspaces()
        self._dirty = True
        for m in self.modules():
            m.register_load_state_dict_post_hook(lambda mod, keys, p=self: p.mark_params_changed())

    def __setattr__(self, name, value):
        # the per-call bookkeeping (_stash, _mid, _later, ...: ~45 assignments per rollout step) does not need nn.Module's
        # parameter / buffer / sub-module registration walk
        if name[0] == "_" and not isinstance(value, (torch.Tensor, nn.Module)) and name not in self.__dict__.get("_modules", ()):
            object.__setattr__(self, name, value)
        else:
            super().__setattr__(name, value)

    # ------------------------------------------------------------------ engine state
    TRAINED_PREFIXES = ()

    def forward(self, *x):
        raise NotImplementedError

    def _apply(self, fn, *a, **k):
        r = super()._apply(fn, *a, **k)
        self._eng = None
        # a leader's captured graph bakes in the engine / packed-weight addresses of every member of its EncoderGroup
        for m in ([self] if self._enc_group is None else self._enc_group.members):
            m._graphs = {}
            m._memos = {}
            m._graph_gen += 1
        return r

    def side_streams(self):
        if self._side is None:                           # capture-fork branches: shared by every policy (captures never overlap)
            self._side = [process_stream("fork%d" % i) for i in range(3)]
        return self._side

    def mark_params_changed(self):
        """Call after modifying encoder weights in place (load_state_dict does it automatically)."""
        self._dirty = True
        self._param_epoch += 1

    @property
    def prec(self):
        return _PREC[self.precision]

    # Per-module arithmetic (keys: "towers", "audio", "smt", "clip", "dialog"): modules exchange fp32 tensors, so each one may run
    # in its own mode; unnamed modules use `precision`.
    def uses_x3(self):
        return self.precision == "bf16x3" or "bf16x3" in (self.module_precision or {}).values()

    def prec_of(self, module):
        mp = self.module_precision
        return _PREC[mp[module]] if mp and module in mp else _PREC[self.precision]

    def _engine(self):
        eng = self._eng
        if eng is not None and not self._dirty:          # the common case, several times per step: no parameter walk
            return eng                                   # (moving the module resets _eng in _apply; weight changes set _dirty)
        p0 = next(self.parameters())
        if not p0.is_cuda:
            raise RuntimeError("avlen_amd policies run on an MI355X only: move the policy to a HIP device "
                               "(there is no CPU fallback)")
        if self._eng is None:
            flat = E.FlatParams(self, self.TRAINED_PREFIXES)
            packed = E.Packed(flat.device, flat, lo=self.uses_x3())
            eng = {"flat": flat, "packed": packed}
            self._build_views(eng, packed)
            ptr2name = {p.data_ptr(): n for n, p in self.named_parameters()}
            eng["ptr2name"] = ptr2name
            self._eng = eng
            self._dirty = True
            self._ws.clear()
        if self._dirty:
            self._eng["packed"].refresh()
            self._dirty = False
        return self._eng

    def _build_views(self, eng, packed):
        raise NotImplementedError

    def _heads(self, which):
        eng = self._engine()
        key = "heads_" + which
        if key not in eng:
            act = getattr(self, "action_distribution_" + which).linear
            cr = getattr(self, "critic_" + which).fc
            un = self.uncertainty_option.fc if which == "option" else None
            eng[key] = E.heads_view(act, cr, un)
        return eng[key]

    # ------------------------------------------------------------------ heads + sampling
    def _heads_first(self, which, feats, race=False):
        """logits / probs / value / unct of one head set (one kernel; capturable).  race (sampling="race" forwards): the same launch
        also runs the exponential race on the head set's noise buffer (filled by _draw_noise before the forward is enqueued) and
        returns the sampled action with its log-prob / entropy -- a rollout forward then ends with ONE small kernel, not three."""
        B, d = feats.shape
        A = self.dim_actions_option if which == "option" else self.dim_actions
        dev = feats.device
        h = self._heads(which)
        logits = torch.empty(B, A, device=dev)
        probs = torch.empty(B, A, device=dev)
        value = torch.empty(B, 1, device=dev)
        unct = torch.empty(B, 2, device=dev) if which == "option" else None
        if race:
            nz = self._noise_dev(which, B, A, dev)
            action, logp, ent = self._result_bufs(which, B, dev)
            # the simulator's / query loop's copy of the actions: the kernel stores it straight into mapped pinned memory (one
            # persistent buffer per head set and batch: captured graphs hold its address; read it before the next forward)
            ah = self._pinned.get(("act_mapped", which, B))
            if ah is None and _MAPPED_ACTIONS:
                ah = self._pinned[("act_mapped", which, B)] = torch.zeros(B, 1, dtype=torch.int64, pin_memory=True)
            L.call("avlen_heads_act_host_fwd", C.byref(h), E.P(feats), d, A, E.P(logits), E.P(probs), E.P(value),
                   E.P(unct) if unct is not None else None, E.P(nz), E.P(action), C.c_void_p(ah.data_ptr()) if ah is not None else None,
                   E.P(logp), E.P(ent), B, L.stream())
            return {"logits": logits, "probs": probs, "value": value, "unct": unct, "raced": (action, logp, ent), "action_host": ah}
        L.call("avlen_heads_fwd", C.byref(h), E.P(feats), d, A, E.P(logits), E.P(probs), E.P(value),
               E.P(unct) if unct is not None else None, None, None, None, B, L.stream())
        return {"logits": logits, "probs": probs, "value": value, "unct": unct}

    def _noise_dev(self, which, B, A, dev):
        """The head set's noise buffer as the heads kernel reads it (persistent: captured graphs hold its address): PINNED host
        memory mapped into the device's address space -- the host draws straight into it and the kernel reads its B x A floats over
        the link, so a sampled forward costs no upload launch on its stream (was: pinned ring -> copy_ -> device buffer, ~6 us of
        every forward's critical path)."""
        t = self._pinned.get(("noise_dev", which, B))
        if t is None:
            t = self._pinned[("noise_dev", which, B)] = torch.ones(B, A, pin_memory=True)
        return t

    def _draw_noise(self, which, B, dev):
        """Draw the race's Exp(1) noise of one act* call on the HOST generator (the reference's draw: CustomFixedCategorical.sample ->
        torch.multinomial), in front of the forward that will consume it.  The buffer is shared with the device, so the previous
        forward of this head set must have finished reading it: true by construction in a trainer (the step's actions have been
        waited for), checked here for callers that launch the same head set back to back."""
        A = self.dim_actions_option if which == "option" else self.dim_actions
        r = self._act_host.get(which)
        if r is not None and r[1] is not None and not r[1].query():
            r[1].synchronize()
        self._noise_dev(which, B, A, dev).exponential_(1)
        ah = self._pinned.get(("act_mapped", which, B))
        if ah is not None:
            ah.fill_(-1)                                 # host_actions() polls these entries: each is stored whole by the heads kernel

    def _finish(self, which, feats, out, action=None, deterministic=False, need_sample=True):
        """Action selection (host or device RNG) and the log-prob / entropy of the chosen actions."""
        B, d = feats.shape
        A = self.dim_actions_option if which == "option" else self.dim_actions
        dev = feats.device
        probs = out["probs"]
        if out.get("finished") and action is None and need_sample and not deterministic:
            return out                                   # sampled right behind the forward at prefetch time (sampling="race")
        if action is None and need_sample:
            if deterministic:
                # the reference consumes no generator state here (CustomFixedCategorical.mode; the eval loop's act_dialog,
                # ppo_trainer.py:1917, 2156): a draw made AHEAD of this call for a sampled action that is now not wanted is undone
                self._undo_draw(out)
                self._act_host.pop(which, None)          # host_actions() must not hand out the previous call's sampled actions
                action = probs.argmax(dim=-1, keepdim=True)
            elif self.sampling == "race" and out.get("raced") is not None and out.get("noise_drawn"):
                # the forward's last kernel ran the race on the noise drawn for this call (_draw_noise, _heads_first): the reference's
                # action for the same generator state; no probabilities cross PCIe, no host synchronisation
                action, logp, ent = out["raced"]
                ah = out.get("action_host")              # stored there by the heads kernel itself (mapped pinned memory) ...
                poll = ah is not None                    # ... cleared to -1 by _draw_noise: host_actions() may poll it
                if ah is None:
                    ah = self._host_action(B)            # ... or copied right behind the forward (512 B)
                    ah.copy_(action, non_blocking=True)
                ev = torch.cuda.Event()
                ev.record(_cur_stream())
                self._act_host[which] = (ah, ev, poll)
                self.last_host_action = None
                out.update(action=_i64(action.view(B, 1)), log_prob=logp, entropy_rows=ent)
                return out
            elif self.sampling == "race":
                # (a forward whose noise was not drawn ahead -- e.g. deterministic=False on a get_value-style path: race as its own launch)
                qh, qd = self._noise_bufs(which, B, A, dev)
                qh.exponential_(1)
                qd.copy_(qh, non_blocking=True)
                action = self._result_bufs(which, B, dev)[0]
                L.call("avlen_sample_race", E.P(probs), E.P(qd), E.P(action), B, A, L.stream())
                ah = self._host_action(B)                # the actions start their way to the host right behind the race (512 B)
                ah.copy_(action, non_blocking=True)
                ev = torch.cuda.Event()
                ev.record(_cur_stream())
                self._act_host[which] = (ah, ev, False)
                self.last_host_action = None
            elif self.sampling == "host":
                # == torch.multinomial's exponential race on the host generator; the noise does not depend on the
                # probabilities, so it is drawn BEFORE waiting for them (same generator order as the reference)
                q = torch.empty(probs.shape, dtype=probs.dtype).exponential_(1)
                if out.get("probs_host") is not None:                   # launch-ahead: the copy is already in flight
                    pc, ev = out["probs_host"]
                    ev.synchronize()
                else:
                    pc = probs.cpu()                                    # sync; (B,A) floats
                ah = self._host_action(B)
                torch.argmax(pc / q, dim=-1, keepdim=True, out=ah)
                self.last_host_action = ah               # the trainer's host loop reads the option actions (ppo_trainer.py:463)
                self._act_host[which] = (ah, None, False)
                action = self._result_bufs(which, B, dev)[0]
                action.copy_(ah, non_blocking=True)
            else:
                q = torch.empty_like(probs).exponential_(1)
                action = (probs / q).argmax(-1, keepdim=True)
        if action is not None:
            action = _i64(action.view(B, 1))
            _, logp, ent = self._result_bufs(which, B, dev)
            h = self._heads(which)
            L.call("avlen_heads_fwd", C.byref(h), E.P(feats), d, A, None, None, None, None, E.P(action), E.P(logp),
                   E.P(ent), B, L.stream())
            out.update(action=action, log_prob=logp, entropy_rows=ent)
        return out

    @staticmethod
    def _undo_draw(heads_out):
        """Take back the host-generator draw a launched-ahead forward made (`rng_spec` = generator state before / after it) when its
        sampled action turns out not to be used: a discarded guess, or a call with deterministic=True.  Only the LAST draw of the
        generator can be taken back; if anything else has drawn since, the state is left alone and the caller is told."""
        spec = heads_out.pop("rng_spec", None)
        heads_out.pop("noise_drawn", None)
        if spec is None:
            return
        if torch.equal(torch.get_rng_state(), spec[1]):
            torch.set_rng_state(spec[0])
        else:
            import warnings
            warnings.warn("avlen_amd: a launched-ahead forward drew sampling noise for a call that then asked for deterministic / "
                          "other arguments, and the host generator has advanced since: the draw cannot be undone (pass "
                          "deterministic=True to the prefetch_* call to avoid it)", RuntimeWarning)

    def _result_bufs(self, which, B, dev):
        """Persistent (action, log_prob, entropy_rows) tensors of one head set and batch size: like the graph outputs they are
        overwritten by the next call of the same kind (RolloutStorage.insert copies what it keeps), and their stable addresses let
        the storage reuse its validated copy plan."""
        if not self.use_graphs:                        # eager mode keeps the reference's fresh-tensor-per-call behaviour
            return (torch.empty(B, 1, dtype=torch.int64, device=dev), torch.empty(B, 1, device=dev), torch.empty(B, device=dev))
        key = ("res", which, B)
        r = self._pinned.get(key)
        if r is None or r[0].device != dev:
            r = self._pinned[key] = (torch.empty(B, 1, dtype=torch.int64, device=dev), torch.empty(B, 1, device=dev),
                                     torch.empty(B, device=dev))
        return r

    def _ones(self, R):
        """(R,) int64 ones on the device (rl_masks of a loss that masks nothing), cached."""
        t = self._pinned.get(("ones", R))
        if t is None:
            t = self._pinned[("ones", R)] = torch.ones(R, dtype=torch.int64, device=next(self.parameters()).device)
        return t

    def _noise_bufs(self, which, B, A, dev):
        """(pinned host, device) noise buffers of one head set: a ring of 8 host slots (an upload is consumed long before its slot
        returns), one device buffer per slot."""
        ring = self._pinned.get(("noise_pair", which, B))      # (not _draw_noise's ring of plain pinned tensors)
        if ring is None:
            ring = self._pinned[("noise_pair", which, B)] = [[(torch.empty(B, A, pin_memory=True), torch.empty(B, A, device=dev))
                                                              for _ in range(8)], 0]
        ring[1] = (ring[1] + 1) % 8
        return ring[0][ring[1]]

    def host_actions(self, which="option"):
        """The most recent SAMPLED actions of head set `which` on the host ((B,1) int64, pinned), or None if the last call of that
        head set was deterministic.  sampling="host": drawn there (ring of 8 buffers); sampling="race": stored by the heads kernel
        into ONE mapped buffer per head set and batch size -- read it before the next forward of that head set; this waits for
        that forward only (the trainer's query loop, ppo_trainer.py:463, reads the option actions)."""
        r = self._act_host.get(which)
        if r is None:
            return None
        ah, ev, poll = r
        if ev is not None:
            if poll:
                # the heads kernel stores each sampled action (one aligned 8-byte word per row) straight into this mapped buffer,
                # which _draw_noise set to -1 before the forward went out: the words are complete as soon as none is negative --
                # no completion packet, no signal, no wake-up (hipEventSynchronize: 15-40 us after the kernel's end, measured).
                # Bounded: falls back to the event.
                v = self._poll_views.get(ah.data_ptr())
                if v is None:
                    v = self._poll_views[ah.data_ptr()] = ah.numpy().reshape(-1)
                t_end = None
                while v.min() < 0:
                    if t_end is None:
                        t_end = time.perf_counter() + 0.02
                    elif time.perf_counter() > t_end:
                        ev.synchronize()
                        break
            else:
                ev.synchronize()
        return ah

    def _host_action(self, B):
        """Pinned staging buffer for the sampled actions (ring of 8: an upload is consumed long before its slot returns)."""
        ring = self._pinned.get(("act", B))
        if ring is None:
            ring = self._pinned[("act", B)] = [[torch.empty(B, 1, dtype=torch.int64, pin_memory=True) for _ in range(8)], 0]
        ring[1] = (ring[1] + 1) % 8
        return ring[0][ring[1]]

    def _run_heads(self, which, feats, action=None, deterministic=False, need_sample=True):
        return self._finish(which, feats, self._heads_first(which, feats), action, deterministic, need_sample)

    def _forward(self, which, *net_args, sample=False, speculative=False):
        """net.run(...) + first heads kernel -> (net outputs tuple, heads dict); replayed from a HIP graph when
        use_graphs is set (inputs are copied into the graph's static buffers; outputs are overwritten by the
        next replay, so callers copy what they keep -- RolloutStorage.insert does).  sample: an act* call that will draw an action
        (sampling="race": the noise is drawn and uploaded here, in front of the forward whose last kernel runs the race).
        speculative: the forward is launched AHEAD of its act* call (prefetch_* / automatic launch-ahead): the generator state around
        the draw is kept so that the draw can be taken back (_undo_draw)."""
        race = self.sampling == "race"

        def eager(*args):
            # work of the forward that nothing on the step's critical path waits for (pi_q's memory row) is enqueued BEHIND the heads
            # kernel, whose sampled actions the host is polling for
            self._post = post = []
            try:
                outs = self.net.run(self, *args)
            finally:
                self._post = None
            heads = self._heads_first(which, outs[0], race=race)
            for fn in post:
                fn()
            return outs, heads
        st = self._stash
        if st is not None:
            self._stash = None
            hit = st[0] == which and st[1] == self._arg_key(net_args)
            if self._auto_pending:
                self._auto_pending = False
                grp_ = self._enc_group
                if grp_ is not None:
                    grp_.auto_hits += int(hit); grp_.auto_misses += int(not hit)
            if hit:
                _cur_stream().wait_event(st[3])        # later kernels of the caller read this forward's outputs
                return st[2]
            # a discarded forward may still be running on its own stream, in the SAME captured graph and static buffers the recompute
            # below is about to replay: order the recompute behind it; and give the follower its claim on the leader's encoders back
            # (the discarded forward used it up: the recompute would run its own towers -- same values up to the fc GEMM's tiling)
            _cur_stream().wait_event(st[3])
            if self._enc_group is not None and self._enc_group.leader is not self:
                self._enc_group.pending.add(id(self))
            self._undo_draw(st[2][1])                  # the discarded forward's noise draw never happened
        drawn, rng_spec = False, None
        if sample and race and not self._defer_second:
            rng_before = torch.get_rng_state() if speculative else None
            obs0 = net_args[0]
            B0 = next(iter(obs0.values())).shape[0] if isinstance(obs0, dict) else obs0.shape[0]
            dev0 = next(iter(obs0.values())).device if isinstance(obs0, dict) else obs0.device
            self._draw_noise(which, B0, dev0)
            drawn = True
            if speculative:
                rng_spec = (rng_before, torch.get_rng_state())
        txt = getattr(self.net, "_text", None)
        if which == "vln" and txt is not None:
            tok = net_args[7]
            if self._defer_second:
                pass                                     # the text graph is replayed by dialog_ready(), in stream order before half 2
            elif tok is not None and txt[0] == tok.data_ptr() and txt[1] == tuple(tok.shape):
                ev = txt[3]
                if self.use_graphs:                      # the wait sits between the two halves of the captured forward
                    self._between = lambda: _cur_stream().wait_event(ev)
                else:
                    _cur_stream().wait_event(ev)
            else:
                self.net._text, self.net._text_key = None, None
        if which == "vln" and self.use_graphs and getattr(self.net, "text_cache", False):
            # a captured forward holds the memoised text tower (avlen_clip_text_cached_fwd): its replay checks nothing, so a weight
            # change since the memo was filled (load_state_dict / mark_params_changed) empties the memo here, outside capture,
            # on the stream the replay is about to run on
            self.net._sync_text_cache(self)
        mode, grp = None, self._enc_group
        if grp is not None and self.precision in ("bf16", "bf16x3"):
            if grp.leader is self:
                grp.mark(net_args[0])
                mode = "lead"
                self._mid = grp.signal if self.use_graphs else None      # between the halves of the leader's graph (cut in net.features)
                late = self._late_inputs
                if late is not None and self.use_graphs:
                    # observation entries that are still being written when this forward starts (the BeliefPredictor's beliefs,
                    # updated on its own stream beside the visual towers): the first half -- towers, AudioCNNs -- does not read them;
                    # between the halves the stream waits for their event and their static copies are refreshed
                    keys, ev, obs_now = late[0], late[1], net_args[0]

                    def mid(keys=keys, ev=ev, obs_now=obs_now, grp=grp):
                        _cur_stream().wait_event(ev)
                        so = grp.static_obs
                        if so is not None and so is not obs_now:
                            L.multi_copy([(so[k], _f32(obs_now[k])) for k in keys if k in so])
                        grp.signal()
                    self._mid = mid
            elif grp.claim(self, net_args[0]):
                mode = "follow"
        self._shared_mode = mode
        def tag(o):
            if drawn:
                o[1]["noise_drawn"] = True
            if rng_spec is not None:
                o[1]["rng_spec"] = rng_spec
            return o
        try:
            if not self.use_graphs:
                return tag(eager(*net_args))
            try:
                out = _graphed(self, which, eager, net_args, mode)
            finally:
                self._between = None
                self._mid = None
            if mode == "lead" and grp.ready_key != grp.key:
                grp.signal()                             # the forward was not cut (no capture fork): the whole graph is the wait
            if which == "vln" and getattr(self.net, "_text", None) is not None and self._deferred is None:
                self.net._text_read = torch.cuda.Event()
                self.net._text_read.record(_cur_stream())
            return tag(out)
        finally:
            self._shared_mode = None

    # ------------------------------------------------------------------ launch-ahead (optional)
    @staticmethod
    def _arg_key(args):
        k = []
        for a in args:
            if torch.is_tensor(a):
                k.append((a.data_ptr(), tuple(a.shape)))
            elif isinstance(a, dict):
                k.append(tuple((n, v.data_ptr()) for n, v in sorted(a.items())))
            else:
                k.append(a)
        return tuple(k)

    _in_prefetch_flow = False             # prefetch_* calls are being issued for this policy: no automatic launch-ahead on top
    _in_prefetch_explicit = False         # (same; kept apart from the flag above for EncoderGroup.auto_sequence)
    _auto_seq_step = False                # leader: EncoderGroup.auto_sequence launched this step's followers

    def _prefetch(self, which, *net_args, stream=None, dialog_later=False, deterministic=False):
        """Enqueue the forward of a later act*/get_value* call now (no host synchronisation).  The matching call, made
        with the same tensors, picks the result up instead of launching again; a trainer that evaluates pi_q, pi_g and
        pi_l on one observation (ppo_trainer.py:375-636) can enqueue all three before the first host-side sampling.
        `stream`: run this forward on its own HIP stream (ordered after everything enqueued so far on the current one), so
        that independent policies overlap on the GPU and each one's probabilities reach the host as soon as IT is done.
        `deterministic`: the flag the act* call will pass -- with True no sampling noise is drawn for it (the reference's eval loop
        calls act_dialog(deterministic=True), which consumes no generator state)."""
        self._stash = None
        cur = _cur_stream()
        run_on = stream if stream is not None else cur
        if stream is not None:
            grp = self._enc_group
            obs = net_args[0] if net_args and isinstance(net_args[0], dict) else None
            if (grp is not None and grp.leader is not self and obs is not None and grp.ready_key is not None
                    and grp.ready_key == grp._key(obs)):
                # a follower of the marked observation: the shared encoders' event covers everything it reads (it was recorded on
                # the current stream, behind every earlier write to the storage it reads)
                stream.wait_event(grp.ready)
            else:
                stream.wait_stream(cur)
        ctx = torch.cuda.stream(run_on) if stream is not None else contextlib.nullcontext()
        if dialog_later:
            # pi_l, reference order (ppo_trainer.py:347, 449-593): the step's dialog tokens and agent_step are written AFTER act_option
            # returned.  Only the half of the cut forward that reads neither (encoders + SMT state encoder) goes out now.
            self._deferred = self._later = None
            if self.use_graphs and net_args[7] is not None and self.net.text_encoder_override is None:
                with ctx:
                    if self.net._text is None:           # first call: the forward is captured against the text graph's static output
                        self.net.prefetch_text(self, net_args[7], run_on, after_current=True)
                    self._defer_second = True
                    try:
                        out = self._forward(which, *net_args, sample=True)       # (no draw: the heads sit in the second half)
                    finally:
                        self._defer_second = False
                if self._deferred is not None:
                    self._later = (which, self._arg_key(net_args), out, stream, net_args[7], net_args[8], deterministic)
                    return
                # the forward was not cut (no split capture): it ran whole, with whatever the tensors held -- run it again, whole,
                # at dialog_ready()
            self._later = ("whole", net_args, stream, deterministic)
            return
        with ctx:
            out = self._forward(which, *net_args, sample=not deterministic, speculative=True)
            done = torch.cuda.Event()
            if self.sampling == "race" and not deterministic:
                # the race goes out right behind the forward on ITS stream (noise drawn now: prefetch_* calls are made in the order
                # of the act* calls that follow, so the host generator is consumed in the reference's order)
                out = (out[0], self._finish(which, out[0][0], out[1]))
                out[1]["finished"] = True
            if self.sampling == "host":
                # the probabilities start their way to the host right behind this forward (pinned buffer + event)
                probs = out[1]["probs"]
                key = (which, tuple(probs.shape))
                if key not in self._pinned:
                    self._pinned[key] = torch.empty(probs.shape, dtype=probs.dtype, pin_memory=True)
                pc = self._pinned[key]
                pc.copy_(probs, non_blocking=True)
                done.record(run_on)
                out[1]["probs_host"] = (pc, done)          # one event: forward finished AND probabilities on the host
            else:
                done.record(run_on)
        self._stash = (which, self._arg_key(net_args), out, done)

    def prefetch_act(self, observations, rnn_hidden_states, prev_actions, masks, ext_memory, ext_memory_masks, stream=None,
                     deterministic=False):
        self._mark_explicit()
        self._prefetch("goal", observations, rnn_hidden_states, prev_actions, masks, ext_memory, ext_memory_masks,
                       stream=stream, deterministic=deterministic)

    def prefetch_act_option(self, observations, rnn_hidden_states, prev_actions, masks, ext_memory, ext_memory_masks,
                            query_state, last_query_info, stream=None, deterministic=False):
        self._mark_explicit()
        self._prefetch("option", observations, rnn_hidden_states, prev_actions, masks, ext_memory, ext_memory_masks,
                       query_state, last_query_info, stream=stream, deterministic=deterministic)

    def _mark_explicit(self):
        grp = self._enc_group
        if grp is not None and grp._seq is not None and grp._seq.auto_running:
            return                                       # the group's own sequencer is recording a step of the AUTOMATIC flow
        for m in ([self] if grp is None else grp.members):
            m._in_prefetch_flow = True
            m._in_prefetch_explicit = True

    def prefetch_text(self, all_dialog, stream, after_current=True):
        """pi_l only: start the frozen CLIP text tower for this step's dialog on `stream` right away (see net.prefetch_text).
        after_current=False: do not order it after the work already enqueued on the current stream (the caller guarantees that
        the previous forward that read the embedding has been waited for -- true once its act_dialog returned)."""
        if self.use_graphs and hasattr(self.net, "prefetch_text"):
            self.net.prefetch_text(self, all_dialog, stream, after_current)

    def prefetch_act_dialog(self, observations, rnn_hidden_states, prev_actions, masks, ext_memory, ext_memory_dialog,
                            ext_memory_masks, all_dialog, agent_step, stream=None, dialog_later=False, deterministic=False):
        """dialog_later=True: `all_dialog` and `agent_step` are the tensors the trainer fills only after `act_option` has returned
        (`current_dialog`, `rollouts.agent_step[step]`: ppo_trainer.py:347, 582-593).  Only the half of the forward that reads
        neither is enqueued now; call `dialog_ready()` once both hold this step's values."""
        self._mark_explicit()
        self._prefetch("vln", observations, rnn_hidden_states, prev_actions, masks, ext_memory, ext_memory_dialog,
                       ext_memory_masks, all_dialog, agent_step, stream=stream, dialog_later=dialog_later, deterministic=deterministic)

    def dialog_ready(self, text_stream=None):
        """Second half of `prefetch_act_dialog(..., dialog_later=True)`: the frozen text tower on the tensors passed there (memoised
        per row: only dialogs that changed since the last call run it), agent_step, the dialog state encoder and the heads.  The
        matching `act_dialog` call picks the result up."""
        lt = self._later
        self._later = None
        if lt is None:
            return
        if lt[0] == "whole":
            self._prefetch("vln", *lt[1], stream=lt[2], deterministic=lt[3])
            return
        which, key, out, stream, tokens, agent_step, deterministic = lt
        g, self._deferred = self._deferred, None
        cur = _cur_stream()
        run_on = stream if stream is not None else cur
        ctx = torch.cuda.stream(run_on) if stream is not None else contextlib.nullcontext()
        # The text tower runs on the CALLER's stream -- the one the host loop wrote the tokens on, behind pi_q's forward and nothing
        # else if pi_l was given its own stream -- and pi_l's stream waits for its event before the second half: the tower then
        # overlaps pi_g and pi_l's state-encoder half instead of queueing behind them.
        if text_stream is not None:
            # the tower on a stream of its own (e.g. a high-priority one): ordered after the caller's stream (the tokens' writes)
            text_stream.wait_stream(cur)
            if self.net._text_read is not None:
                text_stream.wait_event(self.net._text_read)
            with torch.cuda.stream(text_stream):
                self.net.prefetch_text(self, tokens, text_stream, after_current=False, same_stream=True)
            (stream if stream is not None else cur).wait_event(self.net._text[3])
        else:
            if stream is not None and self.net._text_read is not None:
                cur.wait_event(self.net._text_read)          # the previous step's reader of the static embedding buffer
            self.net.prefetch_text(self, tokens, cur, after_current=False, same_stream=True)
            if stream is not None:
                stream.wait_event(self.net._text[3])
        with ctx:
            if torch.is_tensor(g.static[8]) and g.static[8].data_ptr() != agent_step.data_ptr():
                L.multi_copy([(g.static[8], _f32(agent_step))])
            race = self.sampling == "race" and not deterministic
            if race:                                     # pi_l's draw: third in the step, as in the reference
                before = torch.get_rng_state()
                self._draw_noise(which, out[1]["probs"].shape[0], out[1]["probs"].device)
                out[1]["noise_drawn"] = True
                out[1]["rng_spec"] = (before, torch.get_rng_state())
            g.graph2.replay()
            self.net._text_read = torch.cuda.Event()
            self.net._text_read.record(run_on)
            done = torch.cuda.Event()
            if race:
                out = (out[0], self._finish(which, out[0][0], out[1]))
                out[1]["finished"] = True
            if self.sampling == "host":
                probs = out[1]["probs"]
                pk = (which, tuple(probs.shape))
                if pk not in self._pinned:
                    self._pinned[pk] = torch.empty(probs.shape, dtype=probs.dtype, pin_memory=True)
                pc = self._pinned[pk]
                pc.copy_(probs, non_blocking=True)
                done.record(run_on)
                out[1]["probs_host"] = (pc, done)
            else:
                done.record(run_on)
        self._stash = (which, key, out, done)

    # ------------------------------------------------------------------ reference API
    def act(self, observations, rnn_hidden_states, prev_actions, masks, ext_memory, ext_memory_masks,
            deterministic=False):
        args = (observations, rnn_hidden_states, prev_actions, masks, ext_memory, ext_memory_masks)
        (features, rnn_hidden_states, ext_memory_feats), h = self._forward("goal", *args, sample=not deterministic)
        h = self._finish("goal", features, h, deterministic=deterministic)
        self._after_act("goal", args, deterministic)
        return h["value"], h["action"], h["log_prob"], rnn_hidden_states, ext_memory_feats, h["probs"]

    def act_option(self, observations, rnn_hidden_states, prev_actions, masks, ext_memory, ext_memory_masks,
                   query_state, last_query_info, deterministic=False):
        args = (observations, rnn_hidden_states, prev_actions, masks, ext_memory, ext_memory_masks, query_state, last_query_info)
        grp = self._enc_group
        if grp is not None and grp.leader is self and not deterministic and self._stash is None and grp.auto \
                and self.precision in ("bf16", "bf16x3"):
            grp.auto_sequence(self, args)                # share_encoders alone: the whole step as one recorded command list
        (features, rnn_hidden_states, ext_memory_feats), h = self._forward("option", *args, sample=not deterministic)
        h = self._finish("option", features, h, deterministic=deterministic)
        self._after_act("option", args, deterministic)
        return (h["value"], h["unct"], h["action"], h["log_prob"], rnn_hidden_states, ext_memory_feats, h["probs"])

    def act_dialog(self, observations, rnn_hidden_states, prev_actions, masks, ext_memory, ext_memory_dialog,
                   ext_memory_masks, all_dialog, agent_step, deterministic=False, without_dialog=False):
        if without_dialog:
            all_dialog = None
        args = (observations, rnn_hidden_states, prev_actions, masks, ext_memory, ext_memory_dialog, ext_memory_masks, all_dialog,
                agent_step)
        lt = self._later
        if lt is not None and self._auto_pending and lt[0] == "vln":
            # the first half was enqueued by EncoderGroup.auto_launch: if it ran on the tensors of THIS call, the text tower and the
            # second half follow now, on the call's own dialog tokens / agent_step
            seq = self._enc_group._seq if self._enc_group is not None else None
            if all_dialog is not None and lt[1][:7] == self._arg_key(args[:7]):
                self._later = (lt[0], self._arg_key(args), lt[2], lt[3], all_dialog, agent_step, deterministic)
                if seq is not None and seq._cur is not None:
                    seq.dialog_ready()                   # the step was launched by the group's sequencer: so is its second phase
                else:
                    self.dialog_ready()
            else:
                if seq is not None:
                    seq._cur = None
                if lt[3] is not None:                    # the guessed first half may still be running in the graph this call replays
                    _cur_stream().wait_stream(lt[3])
                if self._enc_group is not None:          # ... and it used up this follower's claim on the leader's encoders
                    self._enc_group.pending.add(id(self))
                self._later = self._deferred = None
                self._auto_pending = False
                if self._enc_group is not None:
                    self._enc_group.auto_misses += 1
        (features, rnn_hidden_states, ext_memory_feats, ext_memory_dialog_feats), h = self._forward("vln", *args,
                                                                                                     sample=not deterministic)
        h = self._finish("vln", features, h, deterministic=deterministic)
        self._after_act("vln", args, deterministic)
        return (h["value"], h["action"], h["log_prob"], rnn_hidden_states, ext_memory_feats,
                ext_memory_dialog_feats, h["probs"])

    def prefetch_encoders(self, observations, will_be=None):
        """Leader of an EncoderGroup with `use_graphs`: start the shared encoders -- every member's visual towers and AudioCNN -- on
        `observations` NOW.  Meant for the new observation batch right before it is handed to `rollouts.insert` (ppo_trainer.py:
        864-897: env.step -> batch_obs -> insert -> next step's act_option): the towers read nothing but the sensors, so the ~0.45 ms
        they take hides the host's storage bookkeeping and the next step's launch path instead of following them.  The next
        act_option / get_value_option call of this policy must be made on that same observation (the storage slot it was copied
        to): it then stages only the small inputs and replays the rest of the captured forward.  `will_be`: the tensors that call
        will pass (the storage slot's views) -- their addresses are checked then, and a call on anything else runs the whole forward.
        A no-op until that forward has been captured once (first step) or if the batch shape changed."""
        g = self._last_lead
        grp = self._enc_group
        self._enc_early = None
        if g is None or g.graph2 is None or grp is None or grp.leader is not self or not isinstance(g.static[0], dict):
            return False
        so = g.static[0]
        raw = torch._C._cuda_getCurrentRawStream(torch._C._cuda_getDevice())
        try:
            key = (id(g), raw) + tuple(observations[k].data_ptr() for k in so)
        except KeyError:
            return False
        plan = self._enc_plans.get(key)
        if plan is None:
            pairs = []
            for k, d in so.items():
                a = observations[k]
                if not (a.is_cuda and a.is_contiguous() and a.dtype == d.dtype and tuple(a.shape) == tuple(d.shape)):
                    return False                         # a dtype / layout conversion would be needed: let the act* call do everything
                pairs.append((d, a))
            n = len(pairs)
            arrs = ((C.c_void_p * n)(*[a.data_ptr() for _, a in pairs]), (C.c_void_p * n)(*[d.data_ptr() for d, _ in pairs]),
                    (C.c_int64 * n)(*[d.numel() * d.element_size() for d, _ in pairs]))
            # staging copy + the encoder half of the captured forward as ONE call (csrc/sequencer.hip): this runs between "the
            # step's actions are on the host" and "the next towers are in the queue"
            # ... the towers first (the critical path), then the audio piece on the group's side stream: it depends on the staged
            # spectrogram only (stage event) and is waited for by the rest of the forward (audio event, checked in _graphed)
            side, (stage_ev, aud_ev) = grp.side_stream().cuda_stream, grp.audio_events()
            lst = [(L.CMD_MULTICOPY, n) + tuple(C.cast(x, C.c_void_p).value for x in arrs) + (raw,)]
            if g.exec_f is not None and g.exec_a is not None:
                # towers, then the audio piece on THIS stream; the fc piece on the side stream behind the towers' event (stage_ev);
                # the rest of the forward waits for the fc's event (aud_ev: "what ran on the side stream is done")
                lst += [(L.CMD_GRAPH, 0, g.exec1, raw, None, None), (L.CMD_RECORD, 0, stage_ev.cuda_event, raw, None, None),
                        (L.CMD_GRAPH, 0, g.exec_a, raw, None, None),
                        (L.CMD_WAIT, 0, side, stage_ev.cuda_event, None, None), (L.CMD_GRAPH, 0, g.exec_f, side, None, None),
                        (L.CMD_RECORD, 0, aud_ev.cuda_event, side, None, None)]
            elif g.exec_a is not None:
                lst += [(L.CMD_RECORD, 0, stage_ev.cuda_event, raw, None, None), (L.CMD_GRAPH, 0, g.exec1, raw, None, None),
                        (L.CMD_WAIT, 0, side, stage_ev.cuda_event, None, None), (L.CMD_GRAPH, 0, g.exec_a, side, None, None),
                        (L.CMD_RECORD, 0, aud_ev.cuda_event, side, None, None)]
            else:
                lst += [(L.CMD_GRAPH, 0, g.exec1, raw, None, None)]
            cmds = (L.Cmd * len(lst))()
            for i, (op, n_, a_, b_, c_, d_) in enumerate(lst):
                cmds[i].op, cmds[i].n, cmds[i].a, cmds[i].b, cmds[i].c, cmds[i].d = op, n_, a_, b_, c_, d_
            plan = (cmds, arrs, g, aud_ev if g.exec_a is not None else None)
            while len(self._enc_plans) >= 512:
                self._enc_plans.pop(next(iter(self._enc_plans)))
            self._enc_plans[key] = plan
        self._engine()
        for m_ in grp.members:
            if m_ is not self:
                m_._engine()
        L.call("avlen_cmds_run", plan[0], len(plan[0]))
        self._enc_early = (g, None if will_be is None else tuple(will_be[k].data_ptr() for k in ("rgb", "depth", SPECTROGRAM)), plan[3])
        return True

    def late_inputs(self, keys, event):
        """Leader of an EncoderGroup with `use_graphs`: the observation entries `keys` of the NEXT act* call are complete only when
        `event` has fired (e.g. `BeliefPredictor.update_async` writing `location_belief` / `category_belief` into the storage slot
        while this step's visual towers already run).  The towers / AudioCNNs start at once; the rest of the forward -- and the
        followers -- wait for the event.  Pass event=None to clear."""
        self._late_inputs = None if event is None else (tuple(keys), event)

    def _after_act(self, which, args, deterministic=False):
        """Bookkeeping of a direct act* call for the group's automatic launch-ahead (EncoderGroup.auto_launch)."""
        grp = self._enc_group
        if grp is None or not grp.auto or not self.use_graphs or self.precision not in ("bf16", "bf16x3"):
            return
        h = self._call_hist
        h.append((which, args, deterministic))
        if len(h) > 2:
            del h[0]
        if grp.leader is self and not self._in_prefetch_flow:
            if self._auto_seq_step:
                self._auto_seq_step = False
            else:
                grp.auto_launch(self, args)

    def get_value(self, observations, rnn_hidden_states, prev_actions, masks, ext_memory, ext_memory_masks):
        features, _, _ = self.net.run(self, observations, rnn_hidden_states, prev_actions, masks, ext_memory,
                                      ext_memory_masks)
        return self._run_heads("goal", features, need_sample=False)["value"]

    def get_value_option(self, observations, rnn_hidden_states, prev_actions, masks, ext_memory, ext_memory_masks,
                         query_state, last_query_info):
        (features, _, _), h = self._forward("option", observations, rnn_hidden_states, prev_actions, masks, ext_memory,
                                            ext_memory_masks, query_state, last_query_info)
        return h["value"]

    def evaluate_actions(self, observations, rnn_hidden_states, prev_actions, masks, action, ext_memory,
                         ext_memory_masks):
        features, rnn_hidden_states, ext_memory_feats = self.net.run(
            self, observations, rnn_hidden_states, prev_actions, masks, ext_memory, ext_memory_masks)
        h = self._run_heads("goal", features, action=action)
        return h["value"], h["log_prob"], h["entropy_rows"].mean(), rnn_hidden_states, ext_memory_feats

    def evaluate_actions_option(self, observations, rnn_hidden_states, prev_actions, masks, action, ext_memory,
                                ext_memory_masks, query_state, last_query_info):
        """Values only (no autograd graph): training goes through avlen_amd.ppo.PPO.update, which runs the
        fused HIP loss/backward (ppo.py:207-262)."""
        features, rnn_hidden_states, ext_memory_feats = self.net.run(
            self, observations, rnn_hidden_states, prev_actions, masks, ext_memory, ext_memory_masks, query_state,
            last_query_info)
        h = self._run_heads("option", features, action=action)
        return (h["value"], h["unct"], h["log_prob"], h["entropy_rows"].mean(), rnn_hidden_states, ext_memory_feats,
                h["probs"])

    def evaluate_actions_dialog(self, observations, rnn_hidden_states, prev_actions, masks, action, ext_memory,
                                ext_memory_dialog, ext_memory_masks, all_dialog, agent_step, without_dialog=False):
        if without_dialog:
            all_dialog = None
        features, rnn_hidden_states, ext_memory_feats, ext_memory_dialog_feats = self.net.run(
            self, observations, rnn_hidden_states, prev_actions, masks, ext_memory, ext_memory_dialog,
            ext_memory_masks, all_dialog, agent_step)
        h = self._run_heads("vln", features, action=action)
        return (None, h["log_prob"], h["entropy_rows"].mean(), rnn_hidden_states, ext_memory_feats,
                ext_memory_dialog_feats, h["logits"])


# =========================================================================================================
# nets
# =========================================================================================================
class Net(N._Holder):
    @property
    def is_blind(self):
        return False


class _SMTBase(Net):
    """Shared by AudioNavSMTNet / AudioNavDialogNet / AudioNavOptionNet (policy.py:501-1114)."""

    def _init_encoders(self, observation_space, action_space, hidden_size, use_category_input, extra_dims, smt_kwargs):
        self._hidden_size = hidden_size
        self._action_size = action_space.n
        self._use_category_input = use_category_input
        assert SPECTROGRAM in observation_space.spaces and POSE in observation_space.spaces
        sh, sw, sc = observation_space.spaces[SPECTROGRAM].shape
        self.goal_encoder = N.Cnn3Params(sc, (sh, sw), N.audio_geometry(sh, sw), 128)
        self.visual_encoder = N.SMTCNNParams(observation_space)
        self.action_encoder = N._no_fwd(nn.Linear(self._action_size, 16))
        nfeats = self.visual_encoder.feature_dims + 16 + 128
        self._col_cat = nfeats
        if use_category_input:
            nfeats += 21
        pose_dims = observation_space.spaces[POSE].shape[0]
        pose_indices = (nfeats, nfeats + pose_dims)
        nfeats += pose_dims
        self._x_dims = nfeats                     # [visual | action | audio | (category) | pose]
        nfeats += extra_dims
        self._feature_size = nfeats
        self._img = observation_space.spaces["rgb"].shape[0]
        self.obs_keys = ("rgb", "depth", SPECTROGRAM, POSE, CATEGORY_BELIEF, LOCATION_BELIEF) + \
            ((CATEGORY,) if use_category_input else ())
        self.smt_state_encoder = N.SMTStateEncoderParams(nfeats, dim_feedforward=hidden_size,
                                                         pose_indices=pose_indices, **smt_kwargs)

    @property
    def memory_dim(self):
        return self._feature_size

    @property
    def output_size(self):
        return self.smt_state_encoder.hidden_state_size

    @property
    def num_recurrent_layers(self):
        return -1

    def freeze_encoders(self):
        for m in (self.goal_encoder, self.visual_encoder, self.action_encoder):
            for p in m.parameters():
                p.requires_grad = False

    def set_eval_encoders(self):
        self.goal_encoder.eval()
        self.visual_encoder.eval()

    def pretrained_initialization(self, path):
        sd = torch.load(path, map_location="cpu")["state_dict"]
        self.load_state_dict({k[len("actor_critic.net."):]: v for k, v in sd.items() if "actor_critic.net." in k},
                             strict=False)

    # ---- engine
    def build_views(self, eng, packed):
        eng["rgb"] = E.resnet18_view(self.visual_encoder.rgb_encoder, packed)
        eng["depth"] = E.resnet18_view(self.visual_encoder.depth_encoder, packed)
        eng["audio"] = E.cnn3_view(self.goal_encoder, packed, fmt=eng.get("audio_fmt", 0))
        eng["action"] = E.linear_view(self.action_encoder.weight, self.action_encoder.bias)
        eng["smt"] = E.smt_view(self.smt_state_encoder, eng["flat"], packed, lo=packed.lo)

    def features(self, pol, obs, prev_actions, extra=None, stored=None):
        """-> feats (B, F) = [visual 128 | action 16 | audio 128 | (category 21) | pose 4 | (extra)], goal (B,d).
        stored = (rows2d (R, >= 272) fp32, index (B,) int32): the visual / audio columns are READ from rows of the external-memory
        ring written at rollout time instead of re-running the frozen encoders (PPO.update(feature_reuse=True))."""
        eng = pol._engine()
        if stored is not None:
            return self._features_from_rows(pol, eng, obs, prev_actions, extra, stored)
        rgb, depth, spec = _img(obs["rgb"]), _f32(obs["depth"]), _f32(obs[SPECTROGRAM])
        idx = None
        if isinstance(rgb, RowsOf):                      # minibatch rows of the storage, read in place (grouped fast encoders only)
            if pol.prec_of("towers") in (L.PREC_BF16, L.PREC_BF16X3) and pol._shared_mode is None and isinstance(depth, RowsOf) \
                    and isinstance(spec, RowsOf) and depth.index is rgb.index and spec.index is rgb.index:
                idx = rgb.index
                if pol.prec_of("audio") not in (L.PREC_BF16, L.PREC_FP16):  # the indexed AudioCNN exists on the 16-bit paths only
                    spec = spec.materialise()
            else:
                rgb, depth, spec = (t.materialise() if isinstance(t, RowsOf) else t for t in (rgb, depth, spec))
        B = rgb.shape[0]
        dev = rgb.device
        F = self._feature_size
        feats = torch.empty(B, F, device=dev)
        goal = torch.empty(B, self._hidden_size, device=dev)
        st = L.stream()
        prec, prec_a = pol.prec_of("towers"), pol.prec_of("audio")
        S = rgb.shape[1]
        H, W = spec.shape[1], spec.shape[2]
        nb2 = L.lib.avlen_cnn3_workspace_bytes(C.byref(eng["audio"]), B, H, W)
        ws2 = pol._ws.get("audio", nb2, dev)
        cur = _cur_stream()
        fork = torch.cuda.is_current_stream_capturing()
        mode, grp = pol._shared_mode, pol._enc_group
        x3 = prec == L.PREC_BF16X3 and bool(eng["rgb"].conv1.w16lo)
        vis = aud = None                                 # an EncoderGroup's shared feature buffers (copied into `feats` by the assemble launch)
        if prec == L.PREC_BF16 or x3:
            # bf16 fast path: the towers run as grouped launches (rgb+depth of this policy, or -- leader of an
            # EncoderGroup -- all towers of all member policies); the audio CNN is a parallel branch under capture
            # a follower only copies its audio features out of the group's buffers: no parallel branch for that (every branch of a
            # captured graph is one more busy hardware queue at replay, and the process has four)
            # ... and the LEADER's audio branch is not a fork either: it is captured as a linear graph of its own between the towers
            # and the rest (_Graph.split_audio), so each piece launches cheaply and the replay decides the stream it runs on
            fork_aud = fork and mode is None
            s_aud = pol.side_streams()[0] if fork_aud else cur
            if fork_aud:
                s_aud.wait_stream(cur)
            vis_early = None
            if mode == "lead" and fork:
                # capture order = submission order of the replay: the towers' persistent launch is the step's critical path and goes
                # first; the audio piece runs behind / beside it
                if x3 and pol._capture is not None:
                    # towers | fc + weight prefetch | audio: the replay runs the AudioCNNs right behind the towers and the fc (whose
                    # 30 us sat between them) beside them on the side stream
                    vis_early = grp.run_all(pol, rgb, depth, phase=1)
                    pol._capture.split_fc()
                    grp.run_all(pol, rgb, depth, phase=2)
                else:
                    vis_early = grp.run_all(pol, rgb, depth)
                if pol._capture is not None:
                    # the state encoder's weights come into the L2s while the audio piece runs
                    self.prefetch_weights(pol, ("net.smt_state_encoder.",))
                    pol._capture.split_audio()
            with torch.cuda.stream(s_aud):
                if mode == "lead":
                    aud = grp.run_audio(pol, spec)
                elif mode == "follow":
                    aud = grp.audio_buffers(B, dev)[grp.members.index(pol)]
                else:
                    aud = None
                    if idx is not None and isinstance(spec, RowsOf):
                        L.call("avlen_cnn3_fwd_indexed", C.byref(eng["audio"]), E.P(spec.base), E.P(idx), B, H, W, E.P(feats, 144), F,
                               E.P(ws2), nb2, L.stream())
                    else:
                        L.call("avlen_cnn3_fwd", C.byref(eng["audio"]), E.P(spec), B, H, W, E.P(feats, 144), F, prec_a, E.P(ws2), nb2,
                               L.stream())
                # (a group's audio / visual feature rows reach `feats` inside avlen_feature_assemble below)
            if mode == "follow":
                vis = grp.buffers(B, dev)[grp.members.index(pol)]
            elif mode == "lead":
                vis = vis_early if vis_early is not None else grp.run_all(pol, rgb, depth)
                # the followers need nothing else of this forward: cut the captured graph here (the replay records grp.ready
                # between the two halves, see Policy._forward); eager: record it now
                if fork:
                    if pol._capture is not None and pol._capture.graph2 is None:
                        pol._capture.split()
                else:
                    grp.signal()
            else:
                vis = None
                G = 2
                nets = (C.POINTER(L.ResNet18) * G)(C.pointer(eng["rgb"]), C.pointer(eng["depth"]))
                imgs = (C.c_void_p * G)(*((rgb.base.data_ptr(), depth.base.data_ptr()) if idx is not None else
                                         (rgb.data_ptr(), depth.data_ptr())))
                outs = (C.c_void_p * G)(feats.data_ptr(), feats.data_ptr() + 4 * 64)
                chans, divs = (C.c_int * G)(rgb.shape[3], depth.shape[3]), (C.c_float * G)(255.0, 1.0)
                u8 = (C.c_int * G)(_u8(rgb), 0)
                if x3:
                    nbg = L.lib.avlen_resnet18_group_x3_workspace_bytes(G, B)
                    wsg = pol._ws.get("resnet_pair_x3", nbg, dev)
                    L.call("avlen_resnet18_group_fwd_x3", nets, imgs, u8, chans, divs, outs, F, G, B, S,
                           E.P(idx) if idx is not None else None, E.P(wsg), nbg, st)
                else:
                    nbg = L.lib.avlen_resnet18_group_workspace_bytes(G, B)
                    wsg = pol._ws.get("resnet_pair", nbg, dev)
                    if idx is not None:
                        L.call("avlen_resnet18_group_fwd_indexed", nets, imgs, u8, chans, divs, outs, F, G, B, S, E.P(idx), E.P(wsg),
                               nbg, st)
                    else:
                        L.call("avlen_resnet18_group_fwd", nets, imgs, u8, chans, divs, outs, F, G, B, S, E.P(wsg), nbg, st)
            s_rgb = s_dep = s_aud
        else:
            nb = L.lib.avlen_resnet18_workspace_bytes(B)
            ws_rgb, ws_dep = pol._ws.get("resnet_rgb", nb, dev), pol._ws.get("resnet_depth", nb, dev)
            # inside a graph capture the two towers become parallel branches (fork/join on side streams)
            s_rgb, s_dep = (pol.side_streams()[:2] if fork else (cur, cur))
            if fork:
                s_rgb.wait_stream(cur)
                s_dep.wait_stream(cur)
            with torch.cuda.stream(s_rgb):
                L.call("avlen_resnet18_fwd", C.byref(eng["rgb"]), E.P(rgb), _u8(rgb), B, S, rgb.shape[3], 255.0, E.P(feats, 0), F, prec,
                       E.P(ws_rgb), nb, L.stream())
            with torch.cuda.stream(s_dep):
                L.call("avlen_resnet18_fwd", C.byref(eng["depth"]), E.P(depth), 0, B, S, depth.shape[3], 1.0, E.P(feats, 64), F,
                       prec, E.P(ws_dep), nb, L.stream())
            L.call("avlen_cnn3_fwd", C.byref(eng["audio"]), E.P(spec), B, H, W, E.P(feats, 144), F, prec_a, E.P(ws2), nb2, st)
        pa = _i64(prev_actions.view(B, -1)[:, :1])
        cat = _f32(obs[CATEGORY]) if self._use_category_input else None
        pose = _f32(obs[POSE])
        cb, lb = _f32(obs[CATEGORY_BELIEF]), _f32(obs[LOCATION_BELIEF])
        ex = _f32(extra) if extra is not None else None
        pose_col = self._x_dims - 4
        L.call("avlen_feature_assemble", E.P(feats), F, C.byref(eng["action"]), E.P(pa), 128,
               E.P(cat) if cat is not None else None, self._col_cat, E.P(pose), pose_col,
               E.P(ex) if ex is not None else None, ex.shape[1] if ex is not None else 0, self._x_dims, E.P(cb), E.P(lb),
               E.P(goal), self._hidden_size, B, E.P(vis) if vis is not None else None, 128, 128,
               E.P(aud) if aud is not None else None, 128, 128, 144, st)
        if fork:
            for s_ in (s_rgb, s_dep):
                if s_ is not cur:
                    cur.wait_stream(s_)
        return feats, goal

    def prefetch_weights(self, pol, prefixes, plan_only=False):
        """Warm every XCD's L2 with the 16-bit weight planes of the parameters under `prefixes` (avlen_prefetch_l2), on the current
        stream: issued where the stream would otherwise idle, a few tens of microseconds before a fused chain streams them (the
        chain runs 62 us on warm weights and ~100 us on cold ones; in the rollout step they are always cold).  plan_only: return
        (pointer array, byte counts, n) for a launch that does the warming with its spare workgroups (the text tail)."""
        eng = pol._engine()
        flat = eng["flat"]
        key = ("prefetch", prefixes)
        plan = eng.get(key)
        if plan is None:
            offs = [flat.offsets[n] for n in flat.offsets if n.startswith(prefixes)]
            if offs:
                lo, hi = min(o for o, _ in offs), max(o + k for o, k in offs)
                lo -= lo % 8                                 # 16-byte aligned in the 2-byte planes
                planes = [flat.flat16] + ([flat.extra16[2]] if 2 in flat.extra16 and pol.uses_x3() else [])
                n = len(planes)
                plan = ((C.c_void_p * n)(*[t.data_ptr() + 2 * lo for t in planes]), (C.c_int64 * n)(*[2 * (hi - lo)] * n), n)
            else:
                plan = (None, None, 0)
            eng[key] = plan
        if plan_only:
            return plan
        if plan[2]:
            L.call("avlen_prefetch_l2", plan[0], plan[1], plan[2], L.stream())

    def _features_from_rows(self, pol, eng, obs, prev_actions, extra, stored):
        base, index = stored
        B, dev, F, st = index.numel(), base.device, self._feature_size, L.stream()
        feats = torch.empty(B, F, device=dev)
        goal = torch.empty(B, self._hidden_size, device=dev)
        # [visual 128 | action 16 | audio 128] as the rollout's forward computed them for exactly these observations
        L.call("avlen_gather_rows", E.P(base), base.shape[1], E.P(index), E.P(feats), F, B, 272, st)
        pa = _i64(prev_actions.view(B, -1)[:, :1])
        cat = _f32(obs[CATEGORY]) if self._use_category_input else None
        pose, cb, lb = _f32(obs[POSE]), _f32(obs[CATEGORY_BELIEF]), _f32(obs[LOCATION_BELIEF])
        ex = _f32(extra) if extra is not None else None
        L.call("avlen_feature_assemble", E.P(feats), F, C.byref(eng["action"]), E.P(pa), 128,
               E.P(cat) if cat is not None else None, self._col_cat, E.P(pose), self._x_dims - 4,
               E.P(ex) if ex is not None else None, ex.shape[1] if ex is not None else 0, self._x_dims, E.P(cb), E.P(lb),
               E.P(goal), self._hidden_size, B, None, 0, 0, None, 0, 0, 0, st)
        return feats, goal

    def smt(self, pol, feats, goal, ext_memory, ext_memory_masks, save_key="smt", mem_index=None, save=False):
        eng = pol._engine()
        B, F = feats.shape
        dev = feats.device
        cto = 1 if self.smt_state_encoder._pretraining else 0
        mem = _f32(ext_memory) if ext_memory is not None else None
        M = mem.shape[0] if mem is not None else 0
        NC = mem.shape[1] if mem is not None else B
        masks = _f32(ext_memory_masks) if ext_memory_masks is not None else None
        if not cto:
            assert mem is not None and mem.shape[2] == F and masks.shape == (B, M), "ext memory/mask shape"
            assert mem_index is not None or NC == B
        out = torch.empty(B, self._hidden_size, device=dev)
        nb = L.lib.avlen_smt_workspace_bytes(C.byref(eng["smt"]), B, M, F, cto)
        ws = pol._ws.get(save_key, nb, dev)
        L.call("avlen_smt_fwd", C.byref(eng["smt"]), E.P(feats), E.P(mem) if mem is not None else None,
               E.P(mem_index) if mem_index is not None else None, NC, E.P(masks) if masks is not None else None,
               E.P(goal), E.P(out), B, M, F, self._x_dims - 4, cto, 1 if save else 0, pol.prec_of("smt"), E.P(ws), nb, L.stream())
        return out, (ws, nb, B, M, F, cto)


class AudioNavSMTNet(_SMTBase):
    """policy.py:501-674 (pi_g)."""

    def __init__(self, observation_space, action_space, hidden_size=128, use_pretrained=False, pretrained_path="",
                 use_belief_as_goal=True, use_label_belief=True, use_location_belief=True, use_belief_encoding=False,
                 normalize_category_distribution=False, use_category_input=False, **kwargs):
        super().__init__()
        assert use_belief_as_goal and use_label_belief and use_location_belief and not use_belief_encoding \
            and not normalize_category_distribution, "only the AVLEN yaml configuration is implemented"
        self._init_encoders(observation_space, action_space, hidden_size, use_category_input, 0, kwargs)
        if use_pretrained:
            self.pretrained_initialization(pretrained_path)
        self.train()

    def run(self, pol, observations, rnn_hidden_states, prev_actions, masks, ext_memory, ext_memory_masks):
        feats, goal = self.features(pol, observations, prev_actions)
        x_att, _ = self.smt(pol, feats, goal, ext_memory, ext_memory_masks)
        return x_att, rnn_hidden_states, feats


class AudioNavOptionNet(_SMTBase):
    """policy.py:919-1114 (pi_q)."""

    def __init__(self, observation_space, action_space, hidden_size=128, use_pretrained=False, pretrained_path="",
                 use_belief_as_goal=True, use_label_belief=True, use_location_belief=True, use_belief_encoding=False,
                 normalize_category_distribution=False, use_category_input=False, query_count_emb_size=32, **kwargs):
        super().__init__()
        assert use_belief_as_goal and use_label_belief and use_location_belief and not use_belief_encoding \
            and not normalize_category_distribution, "only the AVLEN yaml configuration is implemented"
        self._query_count_emb_size = query_count_emb_size
        kwargs.pop("use_query_count", None)
        self._init_encoders(observation_space, action_space, hidden_size, use_category_input, query_count_emb_size,
                            kwargs)
        self.policy_selector = N._no_fwd(nn.Linear(self.smt_state_encoder.hidden_state_size, 2))   # unused in fwd
        self._qcnt_emb = N._no_fwd(nn.Embedding(2, query_count_emb_size))                          # unused in fwd
        if use_pretrained:
            self.pretrained_initialization(pretrained_path)
        self.train()

    @property
    def qcnt_emb(self):
        return self._qcnt_emb

    def run(self, pol, observations, rnn_hidden_states, prev_actions, masks, ext_memory, ext_memory_masks,
            query_state, last_query_info, mem_index=None, save_key="smt", save=False, stored=None):
        feats, goal = self.features(pol, observations, prev_actions, extra=query_state, stored=stored)      # [x | query_state]
        x_att, saved = self.smt(pol, feats, goal, ext_memory, ext_memory_masks, save_key, mem_index, save)
        B = feats.shape[0]
        lqi = _f32(last_query_info)
        row = torch.empty(B, self._feature_size, device=feats.device)                          # [x | last_query_info]

        def memory_row(feats=feats, lqi=lqi, row=row, B=B):
            L.call("avlen_concat_rows", E.P(feats), feats.shape[1], self._x_dims, E.P(lqi), lqi.shape[1], lqi.shape[1],
                   E.P(row), row.shape[1], B, L.stream())
        post = getattr(pol, "_post", None)
        if post is not None:
            post.append(memory_row)                      # act_option: behind the heads (Policy._forward)
        else:
            memory_row()
        self._last = (feats, goal, saved)
        return x_att, rnn_hidden_states, row


class AudioNavDialogNet(_SMTBase):
    """policy.py:679-916 (pi_l)."""

    def __init__(self, observation_space, action_space, hidden_size=128, use_pretrained=False, pretrained_path="",
                 use_belief_as_goal=True, use_label_belief=True, use_location_belief=True, use_belief_encoding=False,
                 normalize_category_distribution=False, use_category_input=False, num_steps=5, **kwargs):
        super().__init__()
        assert use_belief_as_goal and use_label_belief and use_location_belief and not use_belief_encoding \
            and not normalize_category_distribution, "only the AVLEN yaml configuration is implemented"
        self._num_steps = num_steps
        # policy.py:728-732: the category input is NOT appended for pi_l
        self._init_encoders(observation_space, action_space, hidden_size, False, 0, kwargs)
        self.clip = N.ClipTextParams()
        self.dialog_layer = N._no_fwd(nn.Linear(self.clip.transformer.width, hidden_size))
        self.dialog_state_encoder = N.DialogStateEncoderParams(hidden_size + hidden_size, dim_feedforward=hidden_size,
                                                               **kwargs)
        if use_pretrained:
            self.pretrained_initialization(pretrained_path)
        self.train()

    def build_views(self, eng, packed):
        super().build_views(eng, packed)
        eng["clip"] = E.clip_view(self.clip, eng["flat"], packed, fmt=eng.get("clip_fmt", 0))
        eng["dialog_layer"] = E.linear_view(self.dialog_layer.weight, self.dialog_layer.bias, eng["flat"], fmt=eng.get("clip_fmt", 0))
        eng["dialog"] = E.dialog_view(self.dialog_state_encoder, eng["flat"], lo=packed.lo)
        # rollout text graph in the 16-bit modes: dialog_layer(ln_final(x) @ text_projection) as ONE few-row GEMM on the folded weight
        # (policy.py:847-849 puts nothing between the two); the tower then stops at ln_final (struct copy without the projection)
        if eng["clip"].wstream and _FOLD_TEXT:
            eng["dialog_fold"] = packed.proj_fold(self.dialog_layer, self.clip.text_projection, fmt=eng.get("clip_fmt", 0))
            nop = L.ClipText.from_buffer_copy(eng["clip"])
            nop.text_proj, nop.text_proj_t = None, None
            eng["clip_noproj"] = nop

    def encode_text(self, pol, tokens, project=True):
        """project=False (needs eng["clip_noproj"]): ln_final(EOT rows), (B, width), for the folded dialog_layer."""
        eng = pol._engine()
        tok = _i64(tokens)
        B = tok.shape[0]
        clip = eng["clip"] if project else eng["clip_noproj"]
        out = torch.empty(B, self.clip.text_projection.shape[1 if project else 0], device=tok.device)
        nb = L.lib.avlen_clip_text_workspace_bytes(C.byref(clip), B)
        ws = pol._ws.get("clip", nb, tok.device)
        L.call("avlen_clip_text_fwd", C.byref(clip), E.P(tok), E.P(out), B, pol.prec_of("clip"), E.P(ws), nb, L.stream())
        return out

    # Per-row memo of the frozen text tower (avlen_clip_text_cached_fwd): in the reference a dialog is constant for NUM_DIALOG_STEPS
    # steps after a query and all-zero for every other env (ppo_trainer.py:347, 582-586), so most rows of `all_dialog` equal the
    # previous step's.  What is kept is the tower's output BEFORE ln_final / text_projection / dialog_layer -- dialog_layer is
    # trained by update_dialog and is applied fresh every call.  Keyed on a row's tokens (content), one state block per batch size;
    # emptied when the weights change (mark_params_changed / load_state_dict bump pol._param_epoch).
    text_cache = CFG.TEXT_CACHE

    def _text_state(self, pol, clip, B, dev):
        st = self.__dict__.setdefault("_text_states", {})
        key = (B, bool(clip.text_proj))
        t = st.get(key)
        if t is None or t.device != dev:
            nb = L.lib.avlen_clip_text_cache_bytes(C.byref(clip), B)
            t = st[key] = torch.zeros(nb, dtype=torch.uint8, device=dev)
        return t

    def _sync_text_cache(self, pol):
        """Outside capture, on the stream the next text forward runs on: empty the memo if the weights changed since it was filled."""
        if self.__dict__.get("_text_epoch") != pol._param_epoch:
            self.__dict__["_text_epoch"] = pol._param_epoch
            for t in self.__dict__.get("_text_states", {}).values():
                t.zero_()

    def invalidate_text_cache(self):
        self.__dict__["_text_epoch"] = None

    def encode_text_cached(self, pol, tokens, project=True):
        eng = pol._engine()
        tok = _i64(tokens)
        B = tok.shape[0]
        clip = eng["clip"] if project else eng["clip_noproj"]
        if not torch.cuda.is_current_stream_capturing():
            self._sync_text_cache(pol)
        state = self._text_state(pol, clip, B, tok.device)
        out = torch.empty(B, self.clip.text_projection.shape[1 if project else 0], device=tok.device)
        nb = L.lib.avlen_clip_text_workspace_bytes(C.byref(clip), B + 1)
        ws = pol._ws.get("clip_cached", nb, tok.device)
        L.call("avlen_clip_text_cached_fwd", C.byref(clip), E.P(tok), E.P(state), state.numel(), E.P(out), B, pol.prec_of("clip"),
               E.P(ws), nb, L.stream())
        return out

    def _text_to_dialog(self, pol, tokens):
        """tokens -> dialog_layer(CLIP.encode_text(tokens)) (policy.py:847-849); folded form in the 16-bit modes."""
        eng = pol._engine()
        enc = self.encode_text_cached if self.text_cache else self.encode_text
        if "dialog_fold" in eng and pol.prec_of("clip") in (L.PREC_BF16, L.PREC_FP16):
            fold = eng["dialog_fold"]
            if self._fused_tail(pol, tokens.shape[0]):
                # memoised tower + ONE tail launch (memo rows, ln_final, 16-bit cast, the folded product): avlen_clip_text_dialog_fwd
                tok = _i64(tokens)
                B = tok.shape[0]
                clip = eng["clip_noproj"]
                if not torch.cuda.is_current_stream_capturing():
                    self._sync_text_cache(pol)
                state = self._text_state(pol, clip, B, tok.device)
                out = torch.empty(B, fold.out_f, device=tok.device)
                nb = L.lib.avlen_clip_text_workspace_bytes(C.byref(clip), B + 1)
                ws = pol._ws.get("clip_cached", nb, tok.device)
                # ... whose spare workgroups warm the L2s with the dialog state encoder's weights: its chain is next
                warm = self.prefetch_weights(pol, ("net.dialog_state_encoder.",), plan_only=True)
                L.call("avlen_clip_text_dialog_fwd", C.byref(clip), C.byref(fold), E.P(tok), E.P(state), state.numel(), E.P(out), B,
                       pol.prec_of("clip"), E.P(ws), nb, warm[0], warm[1], warm[2], L.stream())
                return out
            return self._dialog_embed(pol, enc(pol, tokens, project=False), fold)
        return self._dialog_embed(pol, enc(pol, tokens))

    def _fused_tail(self, pol, B):
        """The rollout's text graph ends in avlen_clip_text_dialog_fwd's one tail launch (which also warms the dialog chain's weights)."""
        eng = pol._engine()
        fold = eng.get("dialog_fold")
        return (fold is not None and pol.prec_of("clip") in (L.PREC_BF16, L.PREC_FP16) and self.text_cache and B + 1 <= 512
                and fold.out_f % 16 == 0 and fold.out_f <= 256 and fold.in_f == 512)

    text_encoder_override = None      # tests: callable(tokens)->(B,512) replacing the CLIP tower (unpinned)
    _text = None                      # (tokens ptr, shape, static embedding, event) of the last prefetch_text
    _text_key = None                  # part of the graph key: a forward captured against the static embedding buffer
    _text_read = None                 # event: the last forward that read the static embedding has been enqueued up to here

    def _dialog_embed(self, pol, e, dl=None):
        """dialog_layer(CLIP embedding) (policy.py:849): (B, 512) -> (B, d).  dl: another Linear view (the folded one)."""
        eng = pol._engine()
        B, d, dev, st = e.shape[0], self._hidden_size, e.device, L.stream()
        d_emb = torch.empty(B, d, device=dev)
        dl = eng["dialog_layer"] if dl is None else dl
        pc = pol.prec_of("clip")
        if pc in (L.PREC_BF16, L.PREC_FP16) and dl.w16:
            fmt = 1 if pc == L.PREC_FP16 else 0
            e16 = torch.empty(B, e.shape[1], device=dev, dtype=torch.bfloat16)
            nbg = L.lib.avlen_gemm_bf16_workspace_bytes(B, d)
            wsg = pol._ws.get("dlg_gemm", nbg, dev)
            L.call("avlen_cast_h16", E.P(e), e.shape[1], E.P(e16), e.shape[1], B, e.shape[1], fmt, st)
            L.call("avlen_gemm_h16", E.P(e16), e.shape[1], dl.w16, dl.ld16, E.P(d_emb), d, None, 0, dl.b, None, 0, B, d,
                   e.shape[1], 0, fmt, E.P(wsg), nbg, st)
        else:
            nbg = L.lib.avlen_gemm_workspace_bytes(B, d, e.shape[1], 1)
            wsg = pol._ws.get("dlg_gemm", nbg, dev)
            L.call("avlen_gemm", E.P(e), e.shape[1], 0, dl.w, dl.in_f, 0, E.P(d_emb), d, dl.b, None, 0, B, d,
                   e.shape[1], 0, pol.prec_of("clip"), 1, 0.0, E.P(wsg), nbg, st)
        return d_emb

    def prefetch_text(self, pol, tokens, stream, after_current=True, same_stream=False):
        """Enqueue CLIP.encode_text(tokens) NOW on `stream`: the text tower depends on nothing but the dialog tokens, so it
        can run under the visual towers of the same step instead of after them.  The next run() with the same token tensor
        reads the embedding from this call's static buffer.  same_stream: `stream` is the current stream and every reader of the
        embedding buffer runs on it too (dialog_ready): stream order is the only ordering needed."""
        if self.text_encoder_override is not None or tokens is None:
            return
        tok = _i64(tokens)
        cur = _cur_stream()
        if same_stream:
            pass
        elif after_current:
            stream.wait_stream(cur)                      # after every earlier reader of the embedding buffer
        elif self._text_read is not None:
            stream.wait_event(self._text_read)           # the last forward that read the embedding buffer
        if self._text is not None:
            # the previous replay of the text graph (same captured graph, same static buffers, possibly another stream -- e.g. the
            # priming replay of an automatically launched first half) must have finished
            stream.wait_event(self._text[3])
        with (contextlib.nullcontext() if same_stream else torch.cuda.stream(stream)):
            key = ("text", tuple(tok.shape))
            g = pol._graphs.get(key)
            pol._engine()                                # derived weights current (a weight change also empties the text memo below)
            if self.text_cache:
                self._sync_text_cache(pol)
            if g is None:
                # the graph ends with dialog_layer: what the forward picks up is the (B, d) dialog embedding
                g = pol._graphs[key] = _Graph(pol, lambda t: self._text_to_dialog(pol, t), [tok])
            emb = g([tok])
            ev = torch.cuda.Event()
            ev.record(stream)
        self._text = (tokens.data_ptr(), tuple(tokens.shape), emb, ev)
        self._text_key = ("pretext", emb.data_ptr())

    def _text_ready(self, all_dialog):
        t = self._text
        if t is None or all_dialog is None:
            return None
        if torch.cuda.is_current_stream_capturing():
            return t[2]                                  # capture: the kernels read the static embedding buffer in place
        if t[0] != all_dialog.data_ptr() or t[1] != tuple(all_dialog.shape):
            return None                                  # other tokens (or a graph warm-up on clones): encode inside this forward
        return t[2]

    def run(self, pol, observations, rnn_hidden_states, prev_actions, masks, ext_memory, ext_memory_dialog,
            ext_memory_masks, all_dialog, agent_step):
        eng = pol._engine()
        e = None
        cur = _cur_stream()
        fork = torch.cuda.is_current_stream_capturing()
        s_txt = pol.side_streams()[2] if fork else cur
        pre = self._text_ready(all_dialog)
        embedded = pre is not None                       # e is already dialog_layer(embedding)
        if pre is not None:                              # embedding enqueued earlier by prefetch_text (static buffer)
            e = pre
        elif all_dialog is not None:                     # frozen CLIP text tower: a parallel branch under capture
            if fork:
                s_txt.wait_stream(cur)
            with torch.cuda.stream(s_txt):
                if self.text_encoder_override is not None:
                    e = _f32(self.text_encoder_override(all_dialog))
                else:                                    # the same arithmetic as the prefetched text graph (folded in the 16-bit modes)
                    e = self._text_to_dialog(pol, all_dialog)
                    embedded = True
        feats, goal = self.features(pol, observations, prev_actions)
        x_att, _ = self.smt(pol, feats, goal, ext_memory, ext_memory_masks)
        B, d = x_att.shape
        dev = x_att.device
        st = L.stream()
        d_emb = None
        if all_dialog is not None:
            if fork and pre is None:
                cur.wait_stream(s_txt)
            if fork and pre is not None and getattr(pol, "_capture", None) is not None and _SPLIT:
                pol._capture.split()                     # everything above does not need the text embedding
                # the dialog half starts right behind the text tower, which has swept the caches: the dialog encoder's fused chain
                # wants its weights warm -- the text graph's tail launch has done that with its spare workgroups, or 10 us of
                # prefetch here buy ~25 us of the chain
                if not (self.text_encoder_override is None and self._fused_tail(pol, x_att.shape[0])):
                    self.prefetch_weights(pol, ("net.dialog_state_encoder.",))
            # prefetch_text already applied dialog_layer inside the text graph; otherwise do it here
            d_emb = e if embedded else self._dialog_embed(pol, e)
        memd = _f32(ext_memory_dialog)
        mk = _f32(ext_memory_masks)
        M = memd.shape[0]
        step = _f32(agent_step.view(-1))
        out = torch.empty(B, d, device=dev)
        nb = L.lib.avlen_dialog_workspace_bytes(C.byref(eng["dialog"]), B, M)
        ws = pol._ws.get("dialog", nb, dev)
        L.call("avlen_dialog_fwd", C.byref(eng["dialog"]), E.P(x_att), E.P(memd), E.P(mk),
               E.P(d_emb) if d_emb is not None else None, E.P(step), E.P(goal), E.P(out), B, M, pol.prec_of("dialog"), E.P(ws), nb, st)
        return out, rnn_hidden_states, feats, out


    # ---- training path of PPO.update_dialog (ppo.py:99-154; csrc: train_resnet.hip, train_gru.hip, modules.hip) -----------
    def train_forward(self, pol, obs, prev_actions, ext_memory, mem_index, ext_memory_dialog, ext_memory_masks, all_dialog,
                      agent_step):
        """evaluate_actions_dialog's forward (policy.py:807-865) with every activation kept: towers, AudioCNN, feature row, SMT
        encoder, frozen CLIP -> dialog_layer, dialog encoder.  -> (x_att_dialog (R, d), saved)."""
        eng = pol._engine()
        rgb, depth, spec = _img(obs["rgb"]), _f32(obs["depth"]), _f32(obs[SPECTROGRAM])
        R, S, dev, st, prec = rgb.shape[0], rgb.shape[1], rgb.device, L.stream(), pol.prec
        F = self._feature_size
        feats = torch.empty(R, F, device=dev)
        goal = torch.empty(R, self._hidden_size, device=dev)
        sv = {"R": R}
        # visual towers: the preprocessed (x/255, 2x2 mean) 64x64 images are the networks' inputs
        for key, img, div, col in (("rgb", rgb, 255.0, 0), ("depth", depth, 1.0, 64)):
            x0 = torch.empty(R, 64, 64, img.shape[3], device=dev)
            L.call("avlen_preprocess_image", E.P(img), _u8(img), E.P(x0), R, S, img.shape[3], div, st)
            nb = L.lib.avlen_resnet18_train_workspace_bytes(C.byref(eng[key]), R, 64, 64, prec)
            ws = pol._ws.get("train_" + key, nb, dev)
            L.call("avlen_resnet18_train_fwd", C.byref(eng[key]), E.P(x0), R, 64, 64, E.P(feats, col), F, prec, E.P(ws), nb, st)
            sv[key] = (x0, ws, nb)
        H, W = spec.shape[1], spec.shape[2]
        nb = L.lib.avlen_cnn3_train_workspace_bytes(C.byref(eng["audio"]), R, H, W, prec)
        ws = pol._ws.get("train_audio", nb, dev)
        L.call("avlen_cnn3_train_fwd", C.byref(eng["audio"]), E.P(spec), R, H, W, E.P(feats, 144), F, prec, E.P(ws), nb, st)
        sv["audio"] = (spec, ws, nb, H, W)
        pa = _i64(prev_actions.view(R, -1)[:, :1])
        pose, cb, lb = _f32(obs[POSE]), _f32(obs[CATEGORY_BELIEF]), _f32(obs[LOCATION_BELIEF])
        L.call("avlen_feature_assemble", E.P(feats), F, C.byref(eng["action"]), E.P(pa), 128, None, self._col_cat, E.P(pose),
               self._x_dims - 4, None, 0, self._x_dims, E.P(cb), E.P(lb), E.P(goal), self._hidden_size, R, None, 0, 0, None, 0, 0, 0, st)
        x_att, smt_saved = self.smt(pol, feats, goal, ext_memory, ext_memory_masks, "smt_train", mem_index, save=True)
        d_emb = e = None
        if all_dialog is not None:
            e = _f32(self.text_encoder_override(all_dialog) if self.text_encoder_override is not None
                     else self.encode_text(pol, all_dialog))                 # frozen CLIP tower: no gradient
            d_emb = self._dialog_embed(pol, e)
        memd, mk, step = _f32(ext_memory_dialog), _f32(ext_memory_masks), _f32(agent_step.reshape(-1))
        M = memd.shape[0]
        nb = L.lib.avlen_dialog_train_workspace_bytes(C.byref(eng["dialog"]), R, M)
        ws = pol._ws.get("train_dialog", nb, dev)
        out = torch.empty(R, self._hidden_size, device=dev)
        L.call("avlen_dialog_train_fwd", C.byref(eng["dialog"]), E.P(x_att), E.P(memd), E.P(mk), E.P(d_emb) if d_emb is not None
               else None, E.P(step), E.P(goal), E.P(out), R, M, prec, E.P(ws), nb, st)
        sv.update(feats=feats, goal=goal, pa=pa, smt=smt_saved, e=e, dialog=(ws, nb, M), keep=(memd, mk, step, x_att, d_emb))
        return out, sv

    def train_backward(self, pol, g, sv, d_out):
        """Gradient of every trained parameter of pi_l given d_out = dL/d(x_att_dialog)."""
        eng = pol._engine()
        R, dev, st, prec, d = sv["R"], d_out.device, L.stream(), pol.prec, self._hidden_size
        F = self._feature_size
        ws, nb, M = sv["dialog"]
        has_dialog = sv["e"] is not None
        d_x_att = torch.empty(R, d, device=dev)
        d_demb = torch.empty(R, d, device=dev) if has_dialog else None
        L.call("avlen_dialog_bwd", C.byref(eng["dialog"]), C.byref(g["dialog"]), E.P(sv["goal"]), E.P(d_out), int(has_dialog),
               E.P(d_x_att), E.P(d_demb) if has_dialog else None, R, M, prec, E.P(ws), nb, st)
        if has_dialog:
            e = sv["e"]
            nbl = L.lib.avlen_linear_bwd_workspace_bytes()
            wsl = pol._ws.get("linear_bwd", nbl, dev)
            L.call("avlen_linear_bwd", C.byref(eng["dialog_layer"]), C.byref(g["dialog_layer"]), E.P(e), e.shape[1], E.P(d_demb), d,
                   None, 0, R, prec, E.P(wsl), nbl, st)
        ws, nb, B, Ms, Fs, cto = sv["smt"]
        d_x = torch.empty(R, F, device=dev)
        L.call("avlen_smt_bwd", C.byref(eng["smt"]), C.byref(g["smt"]), E.P(sv["goal"]), E.P(d_x_att), B, Ms, Fs, self._x_dims - 4,
               cto, prec, E.P(d_x), F, E.P(ws), nb, st)
        for key, col in (("rgb", 0), ("depth", 64)):
            x0, wst, nbt = sv[key]
            L.call("avlen_resnet18_train_bwd", C.byref(eng[key]), C.byref(g[key]), E.P(x0), E.P(d_x, col), F, R, 64, 64, None, prec,
                   E.P(wst), nbt, st)
        L.call("avlen_action_encoder_bwd", E.P(d_x, 128), F, E.P(sv["pa"]), C.byref(g["action"]), R, st)
        spec, wsa, nba, H, W = sv["audio"]
        L.call("avlen_cnn3_train_bwd", C.byref(eng["audio"]), C.byref(g["audio"]), E.P(spec), E.P(sv["feats"], 144), E.P(d_x, 144),
               F, R, H, W, prec, E.P(wsa), nba, st)


class AudioNavBaselineNet(Net):
    """policy.py:379-498: [AudioCNN 512 | VisualCNN 512 | category 21] -> masked GRU (config 2)."""
    reads_rnn_state = True            # a captured forward stages rnn_hidden_states / masks and returns the new hidden state

    def __init__(self, observation_space, hidden_size, goal_sensor_uuid, extra_rgb=False, use_mlp_state_encoder=False):
        super().__init__()
        assert goal_sensor_uuid == SPECTROGRAM and not extra_rgb and not use_mlp_state_encoder, \
            "avlen_amd implements the spectrogram-goal GRU baseline"
        self.goal_sensor_uuid, self._hidden_size = goal_sensor_uuid, hidden_size
        self._label = CATEGORY in observation_space.spaces
        self.obs_keys = ("rgb", "depth", SPECTROGRAM) + ((CATEGORY,) if self._label else ())
        H, W, _ = observation_space.spaces["rgb"].shape
        self.visual_encoder = N.Cnn3Params(4, (H, W), N.VISUAL_GEOMETRY, hidden_size)
        sh, sw, sc = observation_space.spaces[SPECTROGRAM].shape
        self.audio_encoder = N.Cnn3Params(sc, (sh, sw), N.audio_geometry(sh, sw), hidden_size)
        self._rnn_in = 2 * hidden_size + (observation_space.spaces[CATEGORY].shape[0] if self._label else 0)
        self.state_encoder = N.RNNStateEncoderParams(self._rnn_in, hidden_size)
        self.train()

    @property
    def output_size(self):
        return self._hidden_size

    @property
    def num_recurrent_layers(self):
        return 1

    def build_views(self, eng, packed):
        # precision="bf16x3" (the mode that keeps values / probabilities / hidden states within 1e-3 of fp32): both CNNs on
        # compensated bf16 pairs (three MFMAs per product; fp16 operands measured 3e-3 on the hidden state against the reference's
        # goldens), the GRU and every Linear product in exact fp32; "bf16": bf16 operands everywhere
        eng["visual"] = E.cnn3_view(self.visual_encoder, packed)
        eng["audio"] = E.cnn3_view(self.audio_encoder, packed)
        eng["gru"] = E.gru_view(self.state_encoder.rnn)

    def run(self, pol, observations, rnn_hidden_states, prev_actions, masks, ext_memory=None, ext_memory_masks=None):
        eng = pol._engine()
        rgb, depth, spec = _img(observations["rgb"]), _f32(observations["depth"]), _f32(observations[SPECTROGRAM])
        R = rgb.shape[0]
        dev = rgb.device
        st = L.stream()
        F = self._rnn_in
        x = torch.empty(R, F, device=dev)
        H, W = spec.shape[1], spec.shape[2]
        nb = L.lib.avlen_cnn3_workspace_bytes(C.byref(eng["audio"]), R, H, W)
        ws = pol._ws.get("audio", nb, dev)
        pc = pol.prec                                    # both CNNs (bf16x3: compensated pairs on the fp32-staged implicit GEMM)
        pg = L.PREC_FP32 if pol.prec == L.PREC_BF16X3 else pol.prec
        L.call("avlen_cnn3_fwd", C.byref(eng["audio"]), E.P(spec), R, H, W, E.P(x, 0), F, pc, E.P(ws), nb, st)
        S = rgb.shape[1]
        rgbd = torch.empty(R, S, S, 4, device=dev)
        L.call("avlen_rgbd_concat", E.P(rgb), _u8(rgb), E.P(depth), E.P(rgbd), R, S * S, st)
        nb2 = L.lib.avlen_cnn3_workspace_bytes(C.byref(eng["visual"]), R, S, S)
        ws2 = pol._ws.get("visual", nb2, dev)
        L.call("avlen_cnn3_fwd", C.byref(eng["visual"]), E.P(rgbd), R, S, S, E.P(x, self._hidden_size), F, pc,
               E.P(ws2), nb2, st)
        if self._label:
            cat = _f32(observations[CATEGORY])
            L.call("avlen_copy_rows", E.P(cat), cat.shape[1], E.P(x, 2 * self._hidden_size), F, R, cat.shape[1], st)
        h0 = _f32(rnn_hidden_states)
        Nn = h0.shape[1]
        T = R // Nn
        mk = _f32(masks.view(-1))
        out = torch.empty(R, self._hidden_size, device=dev)
        h_out = torch.empty(1, Nn, self._hidden_size, device=dev)
        nb3 = L.lib.avlen_gru_workspace_bytes(C.byref(eng["gru"]), T, Nn)
        ws3 = pol._ws.get("gru", nb3, dev)
        L.call("avlen_gru_fwd", C.byref(eng["gru"]), E.P(x), E.P(h0), E.P(mk), E.P(out), E.P(h_out), T, Nn, pg,
               E.P(ws3), nb3, st)
        return out, h_out, None


    # ---- training path (avlen_amd/av_nav.py:PPO.update; csrc/train_gru.hip)
    def _train_dims(self, obs, h0):
        rgb, spec = obs["rgb"], obs[SPECTROGRAM]
        R, Nn = rgb.shape[0], h0.shape[1]
        assert R % Nn == 0, "rows must be T*N (T-major)"
        return R // Nn, Nn, spec.shape[1], spec.shape[2], rgb.shape[1]

    def train_forward(self, pol, observations, rnn_hidden_states, masks):
        """evaluate_actions' forward over a (T*N)-row T-major minibatch with every activation kept for `train_backward`
        (policy.py:451-477 + rnn_state_encoder.py:92-143).  -> out (R, hidden), workspace, dims."""
        eng = pol._engine()
        rgb, depth, spec = _img(observations["rgb"]), _f32(observations["depth"]), _f32(observations[SPECTROGRAM])
        h0 = _f32(rnn_hidden_states)
        T, Nn, Ha, Wa, S = self._train_dims(observations, h0)
        dev = rgb.device
        cat = _f32(observations[CATEGORY]) if self._label else None
        nb = L.lib.avlen_baseline_train_workspace_bytes(C.byref(eng["audio"]), C.byref(eng["visual"]), C.byref(eng["gru"]), T, Nn,
                                                        Ha, Wa, S, pol.prec)
        ws = pol._ws.get("train", nb, dev)
        out = torch.empty(T * Nn, self._hidden_size, device=dev)
        mk = _f32(masks.view(-1))
        L.call("avlen_baseline_train_fwd", C.byref(eng["audio"]), C.byref(eng["visual"]), C.byref(eng["gru"]), E.P(spec), E.P(rgb),
               _u8(rgb), E.P(depth), E.P(cat) if cat is not None else None, cat.shape[1] if cat is not None else 0, E.P(h0), E.P(mk), E.P(out),
               None, T, Nn, Ha, Wa, S, pol.prec, E.P(ws), nb, L.stream())
        return out, (ws, nb), (T, Nn, Ha, Wa, S, spec, mk)

    def train_backward(self, pol, g, observations, masks, d_out, ws, dims):
        eng = pol._engine()
        T, Nn, Ha, Wa, S, spec, mk = dims
        L.call("avlen_baseline_train_bwd", C.byref(eng["audio"]), C.byref(eng["visual"]), C.byref(eng["gru"]), C.byref(g["audio"]),
               C.byref(g["visual"]), C.byref(g["gru"]), E.P(spec), E.P(mk), E.P(d_out), T, Nn, Ha, Wa, S, pol.prec, E.P(ws[0]),
               ws[1], L.stream())


# =========================================================================================================
# policies (policy.py:299-356)
# =========================================================================================================
class _NetPolicy(Policy):
    def _build_views(self, eng, packed):
        eng["clip_fmt"] = 1 if self.prec_of("clip") == L.PREC_FP16 else 0
        eng["audio_fmt"] = 1 if self.prec_of("audio") == L.PREC_FP16 else 0
        self.net.build_views(eng, packed)


class AudioNavBaselinePolicy(_NetPolicy):
    TRAINED_PREFIXES = ("net.", "action_distribution_goal.", "critic_goal.")

    def __init__(self, observation_space, action_space, goal_sensor_uuid, hidden_size=512, extra_rgb=False,
                 use_mlp_state_encoder=False, **eng_kw):
        super().__init__(AudioNavBaselineNet(observation_space=observation_space, hidden_size=hidden_size,
                                             goal_sensor_uuid=goal_sensor_uuid, extra_rgb=extra_rgb,
                                             use_mlp_state_encoder=use_mlp_state_encoder), action_space.n, **eng_kw)

    def grad_views(self, eng):
        """Gradient structs for avlen_baseline_train_bwd / avlen_ppo_loss_heads_bwd: same struct types as the parameter views,
        pointers into the flat gradient buffer at the CANONICAL tensors (conv OIHW, fc (out, C*H*W))."""
        if "grads" not in eng:
            flat = eng["flat"]
            gp = lambda name: C.c_void_p(flat.grad_ptr(name))

            def cnn(view, prefix):
                g = L.Cnn3()
                for i, idx in enumerate((0, 2, 4)):
                    v = view.conv[i]
                    g.conv[i] = L.Conv(gp(f"{prefix}.cnn.{idx}.weight"), gp(f"{prefix}.cnn.{idx}.bias"), v.cin, v.cout, v.kh,
                                       v.kw, v.stride, v.pad)
                g.fc = L.Linear(gp(f"{prefix}.cnn.6.weight"), gp(f"{prefix}.cnn.6.bias"), view.fc.out_f, view.fc.in_f)
                return g
            r = "net.state_encoder.rnn."
            gru = L.Gru(gp(r + "weight_ih_l0"), gp(r + "weight_hh_l0"), gp(r + "bias_ih_l0"), gp(r + "bias_hh_l0"),
                        eng["gru"].in_f, eng["gru"].hidden)
            eng["grads"] = {"audio": cnn(eng["audio"], "net.audio_encoder"), "visual": cnn(eng["visual"], "net.visual_encoder"),
                            "gru": gru, "heads": E.grad_struct_like(self._heads("goal"), eng["ptr2name"], flat)}
        return eng["grads"]


def _split_engine_kwargs(kwargs):
    return {k: kwargs.pop(k) for k in ("precision", "sampling", "use_graphs") if k in kwargs}


class AudioNavSMTPolicy(_NetPolicy):
    TRAINED_PREFIXES = ()                      # pi_g is frozen on this path (ddppo_trainer.py:417-419)

    def __init__(self, observation_space, action_space, hidden_size=128, **kwargs):
        ek = _split_engine_kwargs(kwargs)
        super().__init__(AudioNavSMTNet(observation_space, action_space, hidden_size=hidden_size, **kwargs),
                         action_space.n, **ek)


class AudioNavDialogPolicy(_NetPolicy):
    # frozen on the interactive rollout path; the parameters `PPO.update_dialog` reaches (ppo.py:99-154: everything under the
    # vln action logits except the frozen CLIP tower, ddppo_trainer.py:401-403) come first in the flat buffer
    TRAINED_PREFIXES = ("net.visual_encoder.", "net.goal_encoder.", "net.action_encoder.", "net.smt_state_encoder.",
                        "net.dialog_layer.", "net.dialog_state_encoder.", "action_distribution_vln.")

    def grad_views(self, eng):
        if "grads" not in eng:
            flat, p2n = eng["flat"], eng["ptr2name"]
            gp = lambda n: C.c_void_p(flat.grad_ptr(n))
            lin = lambda v, n: L.Linear(gp(n + ".weight"), gp(n + ".bias"), v.out_f, v.in_f)
            eng["grads"] = {
                "rgb": E.resnet18_grad_view(eng["rgb"], flat, "net.visual_encoder.rgb_encoder."),
                "depth": E.resnet18_grad_view(eng["depth"], flat, "net.visual_encoder.depth_encoder."),
                "audio": E.cnn3_grad_view(eng["audio"], flat, "net.goal_encoder."),
                "action": lin(eng["action"], "net.action_encoder"),
                "smt": E.grad_struct_like(eng["smt"], p2n, flat),
                "dialog_layer": lin(eng["dialog_layer"], "net.dialog_layer"),
                "dialog": E.grad_struct_like(eng["dialog"], p2n, flat),
                "heads": E.grad_struct_like(self._heads("vln"), p2n, flat)}
        return eng["grads"]

    def __init__(self, observation_space, action_space, hidden_size=128, **kwargs):
        ek = _split_engine_kwargs(kwargs)
        super().__init__(AudioNavDialogNet(observation_space, action_space, hidden_size=hidden_size, **kwargs),
                         action_space.n, **ek)


class AudioNavOptionPolicy(_NetPolicy):
    # parameters reached by the gradient of PPO.update (policy.py:1035-1036 detaches the encoders)
    TRAINED_PREFIXES = ("net.smt_state_encoder.", "action_distribution_option.", "critic_option.",
                        "uncertainty_option.")

    def __init__(self, observation_space, action_space, hidden_size=128, **kwargs):
        ek = _split_engine_kwargs(kwargs)
        super().__init__(AudioNavOptionNet(observation_space, action_space, hidden_size=hidden_size, **kwargs), 2, **ek)


# SURVEY section 5 (the reference keeps pth_time / env_time around these calls, ppo_trainer.py:326-328, 726-734, 896): roctx ranges for
# rocprofv3 --marker-trace, installed only with AVLEN_ROCTX=1
CFG.add_ranges(Policy, ("act", "act_option", "act_dialog", "get_value", "get_value_option", "evaluate_actions", "evaluate_actions_option",
                        "evaluate_actions_dialog", "prefetch_act", "prefetch_act_option", "prefetch_act_dialog", "dialog_ready",
                        "prefetch_encoders"))
