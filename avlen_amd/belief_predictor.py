"""Drop-in for ss_baselines/savi/models/belief_predictor.py (BeliefPredictor, :56-206): same constructor, attribute and
state_dict names (`predictor.*`, `classifier.*`), same `update(observations, dones)` contract (writes
`observations['location_belief']` / `['category_belief']` in place).

What differs is where it runs: both spectrogram networks and the per-environment filter (exponential averaging of the
predicted goal location in the odometry frame, label averaging, resets on `dones`, the "sound stopped" branches) execute
on the MI355X through the C ABI (`avlen_resnet18_any_fwd`, `avlen_resnet18_tv_fwd`, `avlen_belief_update`); the reference
copies the network outputs and every pose to the host and loops over environments in numpy.  No CPU fallback.
"""
import ctypes as C
import logging

import torch
import torch.nn as nn

from . import _lib as L
from . import engine as E
from . import nets as N
from .policy import SPECTROGRAM, POSE, LOCATION_BELIEF, CATEGORY_BELIEF, CATEGORY, _f32

_TWO_STREAMS = True          # the two networks run on two streams (decided by measurement)
_SHARED_CAPTURE = True
# The two networks as TWO graphs replayed on two streams (classifier on a side stream, predictor + filter on the caller's): as two
# branches of ONE captured graph they execute back to back (kernel trace of a replay: the predictor's first kernel starts when the
# classifier's last one ends -- 350 + 435 us instead of max(350, 435)); separate graphs on separate streams do overlap.
_TWO_GRAPHS = True
_PRED_FIRST = True
LABEL_PREDICTOR_PATH = "data/pretrained_weights/semantic_audionav/savi/label_predictor.pth"    # belief_predictor.py:96


class BeliefPredictor(nn.Module):
    NUM_LABELS = 21

    def __init__(self, belief_config, device, input_size, pose_indices, hidden_state_size, num_env=1,
                 has_distractor_sound=False, precision="fp32", load_pretrained=True, use_graphs=False):
        super().__init__()
        self.config = belief_config
        self.device = torch.device(device)
        self.predict_label = belief_config.use_label_belief
        self.predict_location = belief_config.use_location_belief
        self.has_distractor_sound = has_distractor_sound
        self.precision = precision
        self.prec = {"fp32": L.PREC_FP32, "bf16": L.PREC_BF16, "bf16x3": L.PREC_BF16X3}[precision]
        # inference of the two auxiliary ResNets under the bf16x3 policies: fp16 operands (as their AudioCNN; 8x closer to fp32 than
        # bf16, on the launch-per-layer 16-bit kernels) -- the staged compensated kernels cost 4 ms per step here.  The online
        # regression keeps self.prec.
        self.prec_inf = L.PREC_FP16 if precision == "bf16x3" else self.prec
        self._fmt = 1 if precision == "bf16x3" else 0
        if self.predict_location:
            if not belief_config.online_training:
                # belief_predictor.py:74-77 swaps in an ImageNet-pretrained torchvision resnet18 with a 23-way head
                raise NotImplementedError("BeliefPredictor with online_training=False (pretrained torchvision location "
                                          "predictor) is not part of the accelerated path")
            self.predictor = N.ResNet18Params(23 if has_distractor_sound else 2)
            self.predictor.fc = N._no_fwd(nn.Linear(4608, 2))
        if self.predict_label:
            self.classifier = N.TvResNet18Params(2, self.NUM_LABELS)
        self.num_env = num_env
        if belief_config.online_training:
            self.regressor_criterion = nn.MSELoss()
            self.optimizer = None
        self._eng = None
        self._ws = E.Workspaces()
        self._state = None
        # use_graphs: the whole update (both networks + the filter, ~75 launches) is captured once per batch shape and
        # replayed; inputs are staged into the graph's static buffers by one batched copy, beliefs copied out by another
        self.use_graphs = use_graphs
        self._graph = None
        if load_pretrained:
            self.load_pretrained_weights()

    # ---- reference API -------------------------------------------------------------------------
    def load_pretrained_weights(self):
        if self.predict_label:
            state_dict = torch.load(LABEL_PREDICTOR_PATH, map_location="cpu")
            cleaned = {k[len("predictor."):]: v for k, v in state_dict["audiogoal_predictor"].items() if "predictor." in k}
            self.classifier.load_state_dict(cleaned)
            self._eng = None
            logging.info("Loaded pretrained label classifier")

    def freeze_encoders(self):
        if self.config.online_training:
            if self.config.use_label_belief:
                for p in self.classifier.parameters():
                    p.requires_grad = False
        elif self.config.use_label_belief or self.config.use_location_belief:
            for p in self.parameters():
                p.requires_grad = False
        logging.info("Freezing belief predictor weights")

    def set_eval_encoders(self):
        if self.config.use_label_belief:
            self.classifier.eval()
        if self.config.use_location_belief:
            self.predictor.eval()

    @property
    def last_pointgoal(self):
        """Host view of the filter state, shaped like the reference's list (None = no estimate yet)."""
        s = self._filter_state(self.num_env)
        has, val = s["has_pg"].cpu(), s["last_pg"].cpu()
        return [val[i].numpy() if has[i] else None for i in range(self.num_env)]

    @property
    def last_label(self):
        s = self._filter_state(self.num_env)
        has, val = s["has_label"].cpu(), s["last_label"].cpu()
        return [val[i].numpy() if has[i] else None for i in range(self.num_env)]

    # ---- engine --------------------------------------------------------------------------------
    def _apply(self, fn, *a, **k):
        self._eng, self._state, self._graph = None, None, None
        r = super()._apply(fn, *a, **k)
        p = next(self.parameters(), None)
        if p is not None:
            self.device = p.device
        return r

    def load_state_dict(self, *a, **k):
        self._eng, self._graph = None, None
        return super().load_state_dict(*a, **k)

    def refresh_weights(self):
        """Re-derive the packed / BatchNorm-folded weights after the parameters changed (optimizer step, checkpoint load)."""
        if self._eng is not None:
            self._eng["packed"].refresh()

    def _engine(self, H, W):
        key = (H, W)
        if self._eng is None or self._eng["key"] != key:
            dev = next(self.parameters()).device
            assert dev.type == "cuda", "avlen_amd runs on the MI355X only (no CPU fallback)"
            packed = E.Packed(dev)
            eng = {"key": key, "packed": packed}
            if self.predict_location:
                h, w = H, W
                for _ in range(3):                                   # three stride-2 stages (3x3, pad 1)
                    h, w = (h - 1) // 2 + 1, (w - 1) // 2 + 1
                assert 128 * h * w == self.predictor.fc.in_features, (
                    "predictor.fc expects %d features, a %dx%d spectrogram yields %d" %
                    (self.predictor.fc.in_features, H, W, 128 * h * w))
                eng["predictor"] = E.resnet18_any_view(self.predictor, packed, 128, h * w, fmt=self._fmt)
            if self.predict_label:
                eng["classifier"] = E.resnet18_tv_view(self.classifier, packed, fmt=self._fmt)
            packed.refresh()
            self._eng = eng
        return self._eng

    # ---- online regression of the location predictor (ppo_trainer.py:996-1015) ---------------------
    def _train_state(self):
        if getattr(self, "_flat", None) is None or not self._flat.intact(self.predictor):
            self._flat = E.FlatParams(self.predictor, ("",))          # every predictor parameter is trained
            self._eng = None                                           # views hold parameter addresses: rebuild
            dev = self._flat.flat.device
            n = self._flat.n_trained
            self._adam = {"m": torch.zeros(n, device=dev), "v": torch.zeros(n, device=dev), "step": 0,
                          "norm_sq": torch.zeros(1, dtype=torch.float64, device=dev)}
            self._grad_view = None
        return self._flat

    def regression_step(self, obs_batch, acc, apply=True):
        """One optimiser step on a batch of stored observations: predictor forward (activations kept), masked MSE against the
        transformed point goal, backward through the GroupNorm ResNet-18, Adam (lr / eps / betas of `self.optimizer` when the
        trainer set one, torch defaults otherwise).  acc (3,) device floats += (loss, correct rows, masked rows)."""
        flat = self._train_state()
        x = self._predictor_input(obs_batch)
        spec = _f32(obs_batch[SPECTROGRAM])
        gts = _f32(obs_batch[POINTGOAL])
        B, H, W, _ = x.shape
        eng = self._engine(H, W)
        if self._grad_view is None:
            self._grad_view = resnet18_grad_view(eng["predictor"], flat)
        net, dev, st = eng["predictor"], x.device, L.stream()
        nb = L.lib.avlen_resnet18_train_workspace_bytes(C.byref(net), B, H, W, self.prec)
        ws = self._ws.get("train", nb, dev)
        preds, d_preds = torch.empty(B, 2, device=dev), torch.empty(B, 2, device=dev)
        flat.grad.zero_()
        L.call("avlen_resnet18_train_fwd", C.byref(net), E.P(x), B, H, W, E.P(preds), 2, self.prec, E.P(ws), nb, st)
        L.call("avlen_belief_regression_loss", E.P(preds), E.P(spec), spec[0].numel(), E.P(gts), gts.shape[1], E.P(d_preds),
               E.P(acc), B, st)
        L.call("avlen_resnet18_train_bwd", C.byref(net), C.byref(self._grad_view), E.P(x), E.P(d_preds), 2, B, H, W, None,
               self.prec, E.P(ws), nb, st)
        self.reduce_gradients(flat)
        if not apply:                                      # gradient only (tests): flat.grad holds it
            return preds
        ad = self._adam
        ad["step"] += 1
        pg = self.optimizer.param_groups[0] if self.optimizer is not None else {"lr": 1e-3, "eps": 1e-8, "betas": (0.9, 0.999)}
        b1, b2 = pg.get("betas", (0.9, 0.999))
        # plain Adam: the trainer does not clip this gradient (no norm accumulator -> no scaling)
        L.call("avlen_adam_step", E.P(flat.flat), E.P(flat.grad), E.P(ad["m"]), E.P(ad["v"]), flat.n_trained, float(pg["lr"]),
               float(b1), float(b2), float(pg["eps"]), ad["step"], 0.0, None, st)
        self._eng["packed"].refresh()                      # packed conv / fc copies follow the stepped weights
        return preds

    def reduce_gradients(self, flat):
        pass

    def _side_stream(self):
        if getattr(self, "_side", None) is None:
            from .policy import process_stream
            self._side = process_stream("belief_side")
        return self._side

    def _filter_state(self, B):
        if self._state is None or self._state["B"] != B:
            dev = next(self.parameters()).device
            z = lambda *s, dt=torch.float32: torch.zeros(*s, dtype=dt, device=dev)
            self._state = {"B": B, "last_pg": z(B, 2), "has_pg": z(B, dt=torch.int32), "last_label": z(B, self.NUM_LABELS),
                           "has_label": z(B, dt=torch.int32), "spec_sum": z(B), "pg": z(B, 2), "labels": z(B, self.NUM_LABELS)}
        return self._state

    def _predictor_input(self, observations):
        spec = _f32(observations[SPECTROGRAM])
        if not self.has_distractor_sound:
            return spec
        cat = _f32(observations[CATEGORY])
        B, H, W, Cs = spec.shape
        x = torch.empty(B, H, W, Cs + cat.shape[1], device=spec.device)
        L.call("avlen_belief_input", E.P(spec), E.P(cat), E.P(x), B, H * W, Cs, cat.shape[1], L.stream())
        return x

    def _run(self, which, x, out):
        B, H, W, Cin = x.shape
        eng = self._engine(observations_hw(x)[0], observations_hw(x)[1])
        fn = "avlen_resnet18_any" if which == "predictor" else "avlen_resnet18_tv"
        nb = getattr(L.lib, fn + "_workspace_bytes")(B, H, W)
        ws = self._ws.get(which, nb, x.device)
        L.call(fn + "_fwd", C.byref(eng[which]), E.P(x), B, H, W, Cin, E.P(out), out.shape[1], self.prec_inf, E.P(ws), nb, L.stream())
        return out

    def cnn_forward(self, observations):
        """belief_predictor.py:126-137: (B, 2) point goals."""
        x = self._predictor_input(observations)
        return self._run("predictor", x, torch.empty(x.shape[0], 2, device=x.device))

    def update(self, observations, dones):
        """belief_predictor.py:139-206."""
        if not (self.predict_label or self.predict_location):
            return
        spec = _f32(observations[SPECTROGRAM])
        assert spec.is_cuda, "avlen_amd runs on the MI355X only (no CPU fallback)"
        d = None
        if dones is not None:
            d = dones if torch.is_tensor(dones) else torch.as_tensor(dones)
            if d.device != spec.device or d.dtype != torch.uint8 or not d.is_contiguous():
                d = d.to(device=spec.device, dtype=torch.uint8).contiguous()
        obs = {SPECTROGRAM: spec}
        if self.predict_location:
            obs[POSE] = _f32(observations[POSE])
            if self.has_distractor_sound:
                obs[CATEGORY] = _f32(observations[CATEGORY])
        outs = {}
        if self.predict_location:
            outs[LOCATION_BELIEF] = observations[LOCATION_BELIEF]
        if self.predict_label:
            outs[CATEGORY_BELIEF] = observations[CATEGORY_BELIEF]
        for t in outs.values():
            assert t.dtype == torch.float32 and t.is_contiguous() and t.is_cuda
        if not self.use_graphs:
            return self._update_eager(obs, d, outs)
        key = tuple((k, tuple(v.shape)) for k, v in sorted(obs.items()))
        if _TWO_GRAPHS and _TWO_STREAMS and self.predict_location and self.predict_label:
            return self._update_two_graphs(obs, d, outs, key)
        g = self._graph
        if g is None or g["key"] != key:
            self._engine(spec.shape[1], spec.shape[2])                    # packed weights exist before capture
            self._filter_state(spec.shape[0])
            st_in = {k: v.clone() for k, v in obs.items()}
            st_d = torch.zeros(spec.shape[0], dtype=torch.uint8, device=spec.device)
            st_out = {k: torch.zeros_like(v) for k, v in outs.items()}
            saved = {k: v.clone() for k, v in self._state.items() if torch.is_tensor(v)}
            from .policy import process_stream
            side = process_stream("belief_warmup")
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):                                 # warm-up outside capture
                self._update_eager(st_in, st_d, st_out)
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            graph = torch.cuda.CUDAGraph()
            if _SHARED_CAPTURE:                                           # every extra stream shifts the stream -> hardware-queue map
                from .policy import _capture_stream
                with torch.cuda.graph(graph, stream=_capture_stream()):
                    self._update_eager(st_in, st_d, st_out)
            else:
                with torch.cuda.graph(graph):
                    self._update_eager(st_in, st_d, st_out)
            for k, v in saved.items():                                    # the warm-up must not advance the filter
                self._state[k].copy_(v)
            g = self._graph = {"key": key, "graph": graph, "in": st_in, "dones": st_d, "out": st_out,
                               "zero": torch.zeros_like(st_d)}
        pairs = [(g["in"][k], v) for k, v in obs.items()]
        pairs.append((g["dones"], d if d is not None else g["zero"]))
        L.multi_copy(pairs)
        g["graph"].replay()
        L.multi_copy([(outs[k], g["out"][k]) for k in outs])

    def update_async(self, observations, dones, stream, after=None):
        """`update` on `stream`: the beliefs are written into `observations[...]` -- pass the views of the rollout storage's slot for
        the two belief entries (the sensors may be the simulator's own tensors) -- while the caller goes on to launch the next
        step's policies; `self.done` (an event) fires when they are in place: hand it to the leader policy
        (`policy.late_inputs((location_belief, category_belief), predictor.done)`), whose visual towers then run beside the two
        belief networks instead of after them.  Ordered after `after` (an event: e.g. the storage's copy of the observation into
        the slot, which must not overwrite the beliefs) or, without it, after everything enqueued so far on the caller's stream."""
        if after is not None:
            stream.wait_event(after)
        else:
            stream.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(stream):
            self.update(observations, dones)
            if getattr(self, "done", None) is None:
                self.done = torch.cuda.Event()
            self.done.record(stream)
        return self.done

    def _update_two_graphs(self, obs, d, outs, key):
        """update() with the classifier and the predictor captured as separate graphs (see _TWO_GRAPHS)."""
        spec = obs[SPECTROGRAM]
        g = self._graph
        if g is None or g["key"] != key or "cls" not in g:
            self._engine(spec.shape[1], spec.shape[2])
            s = self._filter_state(spec.shape[0])
            st_in = {k: v.clone() for k, v in obs.items()}
            st_d = torch.zeros(spec.shape[0], dtype=torch.uint8, device=spec.device)
            st_out = {k: torch.zeros_like(v) for k, v in outs.items()}
            run_cls = lambda: self._run("classifier", st_in[SPECTROGRAM], s["labels"])
            run_pred = lambda: self._run("predictor", self._predictor_input(st_in), s["pg"])
            from .policy import _capture_stream
            cap = _capture_stream()
            cap.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(cap):                                  # warm-up outside capture (the filter state is not touched)
                run_cls(); run_pred()
            torch.cuda.current_stream().wait_stream(cap)
            torch.cuda.synchronize()
            graphs = {}
            for name, fn in (("cls", run_cls), ("pred", run_pred)):
                graphs[name] = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graphs[name], stream=cap):
                    fn()
            g = self._graph = {"key": key, "cls": graphs["cls"], "pred": graphs["pred"], "in": st_in, "dones": st_d, "out": st_out,
                               "zero": torch.zeros_like(st_d), "ev_in": torch.cuda.Event(), "ev_cls": torch.cuda.Event()}
        cur, side = torch.cuda.current_stream(), self._side_stream()
        pairs = [(g["in"][k], v) for k, v in obs.items()]
        pairs.append((g["dones"], d if d is not None else g["zero"]))
        L.multi_copy(pairs)
        g["ev_in"].record(cur)
        side.wait_event(g["ev_in"])
        # the longer network (the predictor: ~45 nodes, ~360 us) is launched first: a graph launch costs the host ~2 us per node, and
        # the second graph's launch hides behind the first one's execution
        if _PRED_FIRST:
            g["pred"].replay()
        with torch.cuda.stream(side):
            g["cls"].replay()
            g["ev_cls"].record(side)
        if not _PRED_FIRST:
            g["pred"].replay()
        cur.wait_event(g["ev_cls"])
        s, st_in, st_out = self._state, g["in"], g["out"]
        loc, catb = st_out.get(LOCATION_BELIEF), st_out.get(CATEGORY_BELIEF)
        pose = st_in[POSE]
        sp = st_in[SPECTROGRAM]
        L.call("avlen_belief_update", E.P(s["pg"]), 2, E.P(s["labels"]), self.NUM_LABELS, E.P(pose), pose.shape[1], E.P(sp), sp[0].numel(),
               E.P(g["dones"]), E.P(s["last_pg"]), E.P(s["has_pg"]), E.P(s["last_label"]), E.P(s["has_label"]),
               E.P(loc) if loc is not None else None, E.P(catb) if catb is not None else None, E.P(s["spec_sum"]), sp.shape[0],
               self.NUM_LABELS, float(self.config.weighting_factor), int(bool(self.config.current_pred_only)), L.stream())
        L.multi_copy([(outs[k], g["out"][k]) for k in outs])

    def _update_eager(self, obs, d, outs):
        spec = obs[SPECTROGRAM]
        B = spec.shape[0]
        s = self._filter_state(B)
        pg = labels = pose = None
        # the two networks are independent until the filter: they run on two streams (two branches of the captured graph);
        # at these sizes (64 spectrograms of 65x26) every kernel fills a fraction of the chip
        side = None
        if self.predict_location and self.predict_label and _TWO_STREAMS:
            side = self._side_stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                labels = self._run("classifier", spec, s["labels"])
        elif self.predict_label:
            labels = self._run("classifier", spec, s["labels"])
        if self.predict_label and labels is None:
            labels = self._run("classifier", spec, s["labels"])
        if self.predict_location:
            pg = self._run("predictor", self._predictor_input(obs), s["pg"])
            pose = obs[POSE]
        if side is not None:
            torch.cuda.current_stream().wait_stream(side)
        loc, catb = outs.get(LOCATION_BELIEF), outs.get(CATEGORY_BELIEF)
        ptr = lambda t: E.P(t) if t is not None else None
        L.call("avlen_belief_update", ptr(pg), 2, ptr(labels), self.NUM_LABELS, ptr(pose), pose.shape[1] if pose is not None else 0,
               E.P(spec), spec[0].numel(), ptr(d), E.P(s["last_pg"]), E.P(s["has_pg"]), E.P(s["last_label"]), E.P(s["has_label"]),
               ptr(loc), ptr(catb), E.P(s["spec_sum"]), B, self.NUM_LABELS, float(self.config.weighting_factor),
               int(bool(self.config.current_pred_only)), L.stream())


def observations_hw(x):
    return x.shape[1], x.shape[2]


resnet18_grad_view = E.resnet18_grad_view


POINTGOAL = "pointgoal_with_gps_compass"        # IntegratedPointGoalGPSAndCompassSensor.cls_uuid


def train_belief_predictor(bp, rollouts, num_epoch=5, num_mini_batch=1):
    """Mirror of the trainer method (ppo_trainer.py:959-1030; called after every PPO update when `online_training` is on,
    ddppo_trainer.py:977-978): `num_epoch` passes over the stored steps, one optimiser step per minibatch of the location
    predictor on (spectrogram -> transformed point goal) with silent rows masked out.  -> (mean loss, prediction accuracy)."""
    adv = torch.zeros_like(rollouts.returns)
    acc = torch.zeros(3, device=rollouts.returns.device)
    for _ in range(num_epoch):
        for sample in rollouts.recurrent_generator(adv, num_mini_batch):
            bp.regression_step(sample[0], acc)
    loss, correct, n = (float(x) for x in acc.cpu())
    return loss / (num_epoch * num_mini_batch), (correct / n if n else 0)


class BeliefPredictorDDP(BeliefPredictor):
    """belief_predictor.py:208-210: the data-parallel variant -- one all-reduce (RCCL) of the flat predictor gradient per
    regression step, parameters broadcast from rank 0 at `init_distributed`."""

    def init_distributed(self, find_unused_params=True):
        import torch.distributed as distrib
        self._dist = distrib.is_available() and distrib.is_initialized()
        if self._dist:
            distrib.broadcast(self._train_state().flat, src=0)
            self._eng = None

    def reduce_gradients(self, flat):
        if getattr(self, "_dist", False):
            import torch.distributed as distrib
            distrib.all_reduce(flat.grad)
            flat.grad.mul_(1.0 / distrib.get_world_size())
