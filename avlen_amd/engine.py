"""Host-side engine shared by the policies: flat parameter storage, packed weights, C-struct views of
the parameters, workspaces.  PyTorch is used for device memory only."""
import ctypes as C
import torch

from . import _lib as L
from . import config as CFG

ALIGN = 64          # floats


def _align(n):
    return (n + ALIGN - 1) // ALIGN * ALIGN


class FlatParams:
    """All parameters of a policy live in ONE fp32 device buffer (trained ones first, so gradient
    clipping, Adam and the RCCL all-reduce each touch one contiguous range); nn.Parameters keep their
    reference names but become views into it."""

    def __init__(self, module, trained_prefixes):
        named = [(n, p) for n, p in module.named_parameters()]
        tr = [(n, p) for n, p in named if n.startswith(tuple(trained_prefixes))]
        rest = [(n, p) for n, p in named if not n.startswith(tuple(trained_prefixes))]
        dev = named[0][1].device
        off, self.offsets = 0, {}
        for n, p in tr:
            self.offsets[n] = (off, p.numel())
            off += _align(p.numel())
        self.n_trained = off
        for n, p in rest:
            self.offsets[n] = (off, p.numel())
            off += _align(p.numel())
        self.flat = torch.zeros(off, dtype=torch.float32, device=dev)
        self.grad = torch.zeros(self.n_trained, dtype=torch.float32, device=dev)
        self.trained_names = [n for n, _ in tr]
        with torch.no_grad():
            for n, p in tr + rest:
                o, k = self.offsets[n]
                view = self.flat[o:o + k].view(p.shape)
                view.copy_(p.data.to(torch.float32))
                p.data = view
            for n, p in tr:                      # torch-side tools see the HIP-written gradients
                p.grad = self.grad_view(n, p.shape)
        self.device = dev
        self._ptrs = {n: p.data_ptr() for n, p in named}
        # bf16 shadow of the whole buffer (same offsets): the w16 operands of the bf16 fast path
        self.flat16 = torch.zeros(off, dtype=torch.bfloat16, device=dev)
        # further 16-bit shadows, built on first use: fmt 1 = fp16 (AVLEN_PREC_FP16 modules), fmt 2 = the LOW plane of the
        # compensated bf16 pair, bf16(w - bf16(w)) (AVLEN_PREC_BF16X3 modules)
        self.extra16 = {}

    def refresh16(self, trained_only=False):
        n = self.n_trained if trained_only else self.flat.numel()
        if n:
            L.call("avlen_cast_bf16", P(self.flat), n, P(self.flat16), n, 1, n, L.stream())
            for fmt, buf in self.extra16.items():
                L.call("avlen_cast_h16", P(self.flat), n, P(buf), n, 1, n, fmt, L.stream())

    def shadow_ptr(self, t, fmt=0):
        """16-bit shadow address of parameter tensor t (a view into flat) in format fmt (0 bf16, 1 fp16, 2 bf16 low plane), or None."""
        off = (t.data_ptr() - self.flat.data_ptr()) // 4
        if off < 0 or off >= self.flat.numel():
            return None
        if fmt == 0:
            return C.c_void_p(self.flat16.data_ptr() + 2 * off)
        if fmt not in self.extra16:
            self.extra16[fmt] = torch.zeros(self.flat.numel(), dtype=torch.int16, device=self.device)
        return C.c_void_p(self.extra16[fmt].data_ptr() + 2 * off)

    def intact(self, module):
        for n, p in module.named_parameters():
            if self._ptrs.get(n) != p.data_ptr():
                return False
        return True

    def grad_view(self, name, shape):
        o, k = self.offsets[name]
        return self.grad[o:o + k].view(shape)

    def grad_ptr(self, name):
        o, _ = self.offsets[name]
        return self.grad.data_ptr() + 4 * o


class FlatAdam:
    """Adam's moments for the trained range of a FlatParams: two flat fp32 buffers the HIP optimiser step walks in one launch,
    exposed through the torch optimiser object the trainer checkpoints.

    The reference saves and restores `agent.optimizer.state_dict()` (ddppo_trainer.py:812-817, 857-862; ppo_trainer.py:1184-1187,
    1224).  `optimizer.state[p]` of every trained parameter therefore holds torch.optim.Adam's own keys -- `step`, `exp_avg`,
    `exp_avg_sq` -- as VIEWS into the flat buffers (the parameters themselves are views into FlatParams.flat the same way), so
    `state_dict()` sees the live moments, and a `load_state_dict()` (which replaces the entries by copies of the checkpoint's
    tensors) is folded back into the flat buffers by a post-hook, or at the next optimiser step if the policy had no engine yet.
    As in the reference, parameters the loss never reaches have no entry, and the state is empty before the first step."""

    def __init__(self, optimizer, module, with_norm=False):
        self.optimizer, self.module, self.with_norm = optimizer, module, with_norm
        self.m = self.v = self.norm_sq = None
        self.step = 0
        self._step_t = torch.zeros((), dtype=torch.float32)      # ONE host scalar shared by every entry (torch keeps `step` on the host)
        self._flat = None
        optimizer.register_load_state_dict_post_hook(self._loaded)

    def _loaded(self, optimizer):
        eng = getattr(self.module, "_eng", None)
        if eng is not None and eng["flat"].flat.is_cuda:
            self.state(eng["flat"], adopt=True)
        else:
            self._flat = None                                    # adopted by the first state() call

    def state(self, flat, adopt=False):
        """The moments for `flat` (allocated on first use / when the policy moved); entries the optimiser holds that are not views
        into them (a loaded checkpoint) are copied in first."""
        fresh = self.m is None or self.m.numel() != flat.n_trained or self.m.device != flat.flat.device
        if fresh:
            dev = flat.flat.device
            self.m, self.v = torch.zeros(flat.n_trained, device=dev), torch.zeros(flat.n_trained, device=dev)
            if self.with_norm:
                self.norm_sq = torch.zeros(1, dtype=torch.float64, device=dev)
        if fresh or adopt or self._flat is not flat:
            self._bind(flat)
        return self

    def _bind(self, flat):
        name_of = {id(p): n for n, p in self.module.named_parameters()}
        trained = set(flat.trained_names)
        lo, hi = self.m.data_ptr(), self.m.data_ptr() + 4 * self.m.numel()
        loaded_step = None
        st_all = self.optimizer.state
        for group in self.optimizer.param_groups:
            for p in group["params"]:
                n = name_of.get(id(p))
                if n not in trained:
                    continue
                o, k = flat.offsets[n]
                st = st_all.get(p)
                if st and "exp_avg" in st and not (lo <= st["exp_avg"].data_ptr() < hi and st["exp_avg"].is_cuda):
                    with torch.no_grad():
                        self.m[o:o + k].copy_(st["exp_avg"].reshape(-1))
                        self.v[o:o + k].copy_(st["exp_avg_sq"].reshape(-1))
                    s = int(float(st["step"]))
                    loaded_step = s if loaded_step is None else max(loaded_step, s)
                st_all[p] = {"step": self._step_t, "exp_avg": self.m[o:o + k].view(p.shape),
                             "exp_avg_sq": self.v[o:o + k].view(p.shape)}
        if loaded_step is not None:
            self.step = loaded_step
        self._step_t.fill_(self.step)
        self._flat = flat

    def advance(self):
        self.step += 1
        self._step_t.fill_(self.step)
        return self.step


class Workspaces:
    def __init__(self):
        self._bufs = {}

    def get(self, key, nbytes, device):
        b = self._bufs.get(key)
        if b is None or b.numel() < nbytes or b.device != device:
            b = torch.empty(int(nbytes), dtype=torch.uint8, device=device)
            self._bufs[key] = b
        return b

    def clear(self):
        self._bufs.clear()


def P(t, off_floats=0):
    """raw device pointer of tensor t (+ offset in floats)"""
    return C.c_void_p(t.data_ptr() + 4 * off_floats)


def linear_view(w, b, flat=None, packed=None, fmt=0, lo=False):
    """fmt: format of the 16-bit shadow w16 (0 bf16, 1 fp16); lo: also the low plane of the compensated bf16 pair (w16lo)."""
    v = L.Linear(P(w), P(b) if b is not None else None, w.shape[0], w.shape[1])
    if flat is not None and w.shape[1] % 8 == 0:
        v.w16, v.ld16 = flat.shadow_ptr(w, fmt), w.shape[1]
        if lo:
            v.w16lo = flat.shadow_ptr(w, 2)
    elif packed is not None:
        # in_f not a multiple of 8 (the distractor variant's fusion input: 297 / 329 + 12 columns): the bf16 GEMMs need
        # 16-byte rows, so the shadow is a zero-padded derived copy [out_f][ld16]
        v.w16, v.ld16 = packed.linear_pad(w)
        if lo:
            v.w16lo, _ = packed.linear_pad(w, 2)
    return v


def affine_view(m):
    return L.Affine(P(m.weight), P(m.bias))


class Packed:
    """Conv / post-flatten fc weights re-laid out for the NHWC implicit-GEMM kernels.  The packed copies
    are derived data: rebuilt (by HIP kernels) whenever the canonical parameters may have changed."""

    def __init__(self, device, flat=None, lo=False):
        # lo: also build the LOW planes of the compensated bf16 pairs (w16lo / w16flo: AVLEN_PREC_BF16X3 fast paths)
        self.device, self.bufs, self.jobs, self.flat, self.lo = device, [], [], flat, lo

    def conv(self, conv, has_bias, bf16=True, fmt=0):
        """fmt: format of the 16-bit copies (0 bf16, 1 fp16)."""
        O, I, KH, KW = conv.weight.shape
        if not bf16:                                  # fp32-staged kernels only (any channel count)
            buf = torch.empty(O * KH * KW * I, dtype=torch.float32, device=self.device)
            self.bufs.append(buf)
            self.jobs.append(("conv32", conv.weight, buf, None, (O, I, KH, KW), 0))
            return L.Conv(P(buf), P(conv.bias) if has_bias else None, I, O, KH, KW, conv.stride[0], conv.padding[0])
        buf = torch.empty(O * KH * KW * I, dtype=torch.float32, device=self.device)
        c16 = 8
        while c16 < I:                                # bf16 conv path: power-of-two channel count >= 8 (zero padded)
            c16 *= 2
        buf16 = torch.empty(O * KH * KW * c16, dtype=torch.bfloat16, device=self.device)
        self.bufs += [buf, buf16]
        self.jobs.append(("conv", conv.weight, buf, buf16, (O, I, KH, KW), (c16, fmt)))
        v = L.Conv(P(buf), P(conv.bias) if has_bias else None, I, O, KH, KW, conv.stride[0], conv.padding[0])
        v.w16, v.cin16 = P(buf16), c16
        if self.lo and fmt == 0:
            buf16lo = torch.empty_like(buf16)
            self.bufs.append(buf16lo)
            self.jobs.append(("conv_lo", conv.weight, None, buf16lo, (O, I, KH, KW), c16))
            v.w16lo = P(buf16lo)
        if O % 16 == 0 and (KH * KW * c16) % 32 == 0 and c16 >= 32:       # fragment-order copy (register-resident weights: tower_tail)
            buff = torch.empty_like(buf16)
            self.bufs.append(buff)
            self.jobs.append(("frag", buf16, None, buff, (O, KH * KW * c16), 0))
            v.w16f = P(buff)
            if self.lo and fmt == 0:
                bufflo = torch.empty_like(buf16)
                self.bufs.append(bufflo)
                self.jobs.append(("frag", buf16lo, None, bufflo, (O, KH * KW * c16), 0))
                v.w16flo = P(bufflo)
        if I * conv.stride[0] == 8 and KW % conv.stride[0] == 0 and conv.padding[0] == 0 and I < 8:
            bufc = torch.empty(O * KH * KW * I, dtype=torch.bfloat16, device=self.device)      # compact: super-pixel form
            self.bufs.append(bufc)
            self.jobs.append(("conv16c", conv.weight, None, bufc, (O, I, KH, KW), (I, fmt)))
            v.w16c = P(bufc)
        return v

    def conv_bn(self, conv, bn, fmt=0):
        """conv followed by an eval-mode BatchNorm2d, folded: w' = w * gamma / sqrt(var + eps) per output channel,
        b' = beta - mean * gamma / sqrt(var + eps)  (torchvision BasicBlock, belief_predictor.py:79-81)."""
        O, I, KH, KW = conv.weight.shape
        buf = torch.empty(O * KH * KW * I, dtype=torch.float32, device=self.device)
        wf = torch.empty_like(conv.weight)
        bias = torch.empty(O, dtype=torch.float32, device=self.device)
        c16 = 8
        while c16 < I:
            c16 *= 2
        buf16 = torch.empty(O * KH * KW * c16, dtype=torch.bfloat16, device=self.device)
        self.bufs += [buf, wf, bias, buf16]
        self.jobs.append(("convbn", conv.weight, buf, (bn, wf, bias, buf16), (O, I, KH, KW), (c16, fmt)))
        v = L.Conv(P(buf), P(bias), I, O, KH, KW, conv.stride[0], conv.padding[0])
        v.w16, v.cin16 = P(buf16), c16
        return v

    def fc_after_flatten(self, lin, C_, HW, fmt=0):
        O = lin.weight.shape[0]
        buf = torch.empty(O * C_ * HW, dtype=torch.float32, device=self.device)
        want_lo = self.lo and fmt == 0
        # the low plane lies right behind the high plane: one common distance for every tower's fc (grouped compensated GEMM)
        both = torch.empty((2 if want_lo else 1) * O * C_ * HW, dtype=torch.bfloat16, device=self.device)
        buf16 = both[:O * C_ * HW]
        self.bufs += [buf, both]
        self.jobs.append(("fc", lin.weight, buf, buf16, (O, C_, HW), fmt))
        v = L.Linear(P(buf), P(lin.bias), O, C_ * HW)
        v.w16, v.ld16 = P(buf16), C_ * HW
        if want_lo:
            buf16lo = both[O * C_ * HW:]
            self.jobs.append(("fc_lo", lin.weight, None, buf16lo, (O, C_, HW), 2))
            v.w16lo = P(buf16lo)
        return v

    def linear_pad(self, w, fmt=0):
        """16-bit copy (fmt 0 bf16, 2 low plane) of a Linear weight with rows padded to a multiple of 8 columns (pad columns zero)."""
        out_f, in_f = w.shape
        ld = (in_f + 7) // 8 * 8
        buf16 = torch.zeros(out_f, ld, dtype=torch.bfloat16, device=self.device)
        self.bufs.append(buf16)
        self.jobs.append(("pad16", w, fmt, buf16, (out_f, in_f), ld))
        return P(buf16), ld

    def refresh_pads(self):
        """Only the padded Linear shadows (the ones among them that are TRAINED move with every optimiser step)."""
        st = L.stream()
        for kind, w, buf, buf16, dims, c16 in self.jobs:
            if kind == "pad16":
                L.call("avlen_cast_h16", P(w), dims[1], P(buf16), c16, dims[0], dims[1], buf or 0, st)

    def proj_fold(self, lin, proj, fmt=0):
        """Linear `lin` (out, k) behind a bias-free projection `proj` (in, k): lin(x @ proj) = x @ (lin.weight @ proj.T).T + b.
        -> L.Linear over the folded fp32 weight (out, in) with its 16-bit shadow (fmt 0 bf16, 1 fp16); refreshed with the other
        derived copies."""
        out_f, in_f = lin.weight.shape[0], proj.shape[0]
        wf = torch.empty(out_f, in_f, dtype=torch.float32, device=self.device)
        wf16 = torch.empty(out_f, in_f, dtype=torch.bfloat16, device=self.device)
        self.bufs += [wf, wf16]
        self.jobs.append(("projfold", (lin.weight, proj), wf, wf16, (out_f, in_f), fmt))
        v = L.Linear(P(wf), P(lin.bias) if lin.bias is not None else None, out_f, in_f)
        v.w16, v.ld16 = P(wf16), in_f
        return v

    def ln_fold(self, lin_w, lin_b, ln, fmt=0):
        """LayerNorm `ln` folded into the Linear (lin_w, lin_b) that follows it (avlen_ln_fold_weights); fmt 1: fp16 weights."""
        N_, K = lin_w.shape
        w16f = torch.empty(N_, K, dtype=torch.bfloat16, device=self.device)
        s = torch.empty(N_, dtype=torch.float32, device=self.device)
        c = torch.empty(N_, dtype=torch.float32, device=self.device)
        self.bufs += [w16f, s, c]
        self.jobs.append(("fold", lin_w, (lin_b, ln.weight, ln.bias, s, c), w16f, (N_, K), fmt))
        return L.LnFold(P(w16f), P(s), P(c))

    def refresh(self):
        st = L.stream()
        for kind, w, buf, buf16, dims, c16 in self.jobs:
            if kind == "fold":
                b, g, be, s, c = buf
                L.call("avlen_ln_fold_weights_h16", P(w), P(b) if b is not None else None, P(g), P(be), P(buf16), dims[1], P(s),
                       P(c), dims[0], dims[1], c16, st)
            elif kind == "pad16":
                L.call("avlen_cast_h16", P(w), dims[1], P(buf16), c16, dims[0], dims[1], buf or 0, st)
            elif kind == "projfold":                  # derived data: Linear weight x projection (exact fp32 MFMA), then its 16-bit shadow
                lw, pj = w                            # [out][k], [in][k]: folded[out][in] = sum_k lw[out][k] pj[in][k]
                k = lw.shape[1]
                nbg = L.lib.avlen_gemm_workspace_bytes(dims[0], dims[1], k, 1)
                wsg = torch.empty(int(nbg), dtype=torch.uint8, device=self.device)
                L.call("avlen_gemm", P(lw), k, 0, P(pj), pj.shape[1], 0, P(buf), dims[1], None, None, 0, dims[0], dims[1], k, 0,
                       L.PREC_FP32, 1, 0.0, P(wsg), nbg, st)
                self.bufs.append(wsg)                 # alive until the stream has run the product
                L.call("avlen_cast_h16", P(buf), dims[1], P(buf16), dims[1], dims[0], dims[1], c16, st)
            elif kind == "t32":                       # derived data: a transposed fp32 copy
                with torch.no_grad():
                    buf.copy_(w.t())
            elif kind == "conv32":
                L.call("avlen_pack_conv_weight", P(w), P(buf), *dims, st)
            elif kind == "convbn":
                bn, wf, bias, w16 = buf16
                with torch.no_grad():                 # weight preparation (derived data), not the compute path
                    scale = bn.weight / torch.sqrt(bn.running_var + bn.eps)
                    wf.copy_(w * scale[:, None, None, None])
                    bias.copy_(bn.bias - bn.running_mean * scale)
                L.call("avlen_pack_conv_weight", P(wf), P(buf), *dims, st)
                L.call("avlen_pack_conv_weight_h16", P(wf), P(w16), *dims, c16[0], c16[1], st)
            elif kind == "frag":                      # queued after the conv's own job: w16 is up to date
                L.call("avlen_pack_conv_weight_frag", P(w), P(buf16), dims[0], dims[1], st)
            elif kind == "clipstream":
                L.call("avlen_clip_pack_stream", C.byref(w), P(buf16), c16, st)
            elif kind == "conv_lo":
                L.call("avlen_pack_conv_weight_h16", P(w), P(buf16), *dims, c16, 2, st)
            elif kind == "conv16c":
                L.call("avlen_pack_conv_weight_h16", P(w), P(buf16), *dims, c16[0], c16[1], st)
            elif kind == "conv":
                L.call("avlen_pack_conv_weight", P(w), P(buf), *dims, st)
                L.call("avlen_pack_conv_weight_h16", P(w), P(buf16), *dims, c16[0], c16[1], st)
            elif kind == "fc_lo":
                L.call("avlen_pack_fc_after_flatten_h16", P(w), P(buf16), *dims, 2, st)
            else:
                L.call("avlen_pack_fc_after_flatten", P(w), P(buf), *dims, st)
                L.call("avlen_pack_fc_after_flatten_h16", P(w), P(buf16), *dims, c16, st)
        if self.flat is not None:
            self.flat.refresh16()


def resnet18_view(net, packed):
    s = L.ResNet18()
    s.conv1 = packed.conv(net.conv1, False)
    s.bn1 = affine_view(net.bn1)
    i = 0
    for layer in (net.layer1, net.layer2, net.layer3, net.layer4):
        for blk in layer:
            b = s.block[i]
            b.conv1 = packed.conv(blk.conv1, False)
            b.conv2 = packed.conv(blk.conv2, False)
            b.bn1, b.bn2 = affine_view(blk.bn1), affine_view(blk.bn2)
            if blk.downsample is not None:
                b.down = packed.conv(blk.downsample[0], False)
                b.bnd = affine_view(blk.downsample[1])
                b.has_down = 1
            else:
                b.has_down = 0
            i += 1
    s.fc = packed.fc_after_flatten(net.fc, 128, 64)
    return s


def resnet18_any_view(net, packed, fc_c, fc_hw, fmt=0):
    """CustomResNet at a non-64x64 input (BeliefPredictor.predictor): fp32-staged kernels, fc packed for (fc_c, fc_hw).
    fmt: format of the 16-bit weight copies (0 bf16, 1 fp16: AVLEN_PREC_FP16 calls)."""
    s = L.ResNet18()
    s.conv1 = packed.conv(net.conv1, False, fmt=fmt)
    s.bn1 = affine_view(net.bn1)
    i = 0
    for layer in (net.layer1, net.layer2, net.layer3, net.layer4):
        for blk in layer:
            b = s.block[i]
            b.conv1 = packed.conv(blk.conv1, False, fmt=fmt)
            b.conv2 = packed.conv(blk.conv2, False, fmt=fmt)
            b.bn1, b.bn2 = affine_view(blk.bn1), affine_view(blk.bn2)
            b.has_down = 0
            if blk.downsample is not None:
                b.down = packed.conv(blk.downsample[0], False, fmt=fmt)
                b.bnd = affine_view(blk.downsample[1])
                b.has_down = 1
            i += 1
    s.fc = packed.fc_after_flatten(net.fc, fc_c, fc_hw, fmt=fmt)
    return s


def resnet18_tv_view(net, packed, fmt=0):
    """torchvision resnet18 with its BatchNorms folded into the convs (BeliefPredictor.classifier); fmt as resnet18_any_view."""
    s = L.ResNet18()
    s.conv1 = packed.conv_bn(net.conv1, net.bn1, fmt)
    i = 0
    for layer in (net.layer1, net.layer2, net.layer3, net.layer4):
        for blk in layer:
            b = s.block[i]
            b.conv1 = packed.conv_bn(blk.conv1, blk.bn1, fmt)
            b.conv2 = packed.conv_bn(blk.conv2, blk.bn2, fmt)
            b.has_down = 0
            if blk.downsample is not None:
                b.down = packed.conv_bn(blk.downsample[0], blk.downsample[1], fmt)
                b.has_down = 1
            i += 1
    s.fc = linear_view(net.fc.weight, net.fc.bias)
    return s


def cnn3_view(net, packed, fmt=0):
    """fmt: format of the 16-bit weight shadows (0 bf16, 1 fp16: AVLEN_PREC_FP16 calls)."""
    s = L.Cnn3()
    s.half_fmt = fmt
    for i, idx in enumerate((0, 2, 4)):
        s.conv[i] = packed.conv(net.cnn[idx], True, fmt=fmt)
    h, w = net.out_hw
    s.fc = packed.fc_after_flatten(net.cnn[6], 64, h * w, fmt=fmt)
    return s


def mha_view(m, flat=None, fmt=0, lo=False):
    return L.Mha(linear_view(m.in_proj_weight, m.in_proj_bias, flat, fmt=fmt, lo=lo),
                 linear_view(m.out_proj.weight, m.out_proj.bias, flat, fmt=fmt, lo=lo))


def transformer_view(t, d, nhead, flat=None, lo=False):
    s = L.Transformer()
    e, q = t.encoder.layers[0], t.decoder.layers[0]
    lv = lambda m: linear_view(m.weight, m.bias, flat, lo=lo)
    s.enc = L.EncLayer(mha_view(e.self_attn, flat, lo=lo), lv(e.linear1), lv(e.linear2), affine_view(e.norm1), affine_view(e.norm2))
    s.enc_norm = affine_view(t.encoder.norm)
    s.dec = L.DecLayer(mha_view(q.self_attn, flat, lo=lo), mha_view(q.multihead_attn, flat, lo=lo), lv(q.linear1), lv(q.linear2),
                       affine_view(q.norm1), affine_view(q.norm2), affine_view(q.norm3))
    s.dec_norm = affine_view(t.decoder.norm)
    s.d, s.nhead = d, nhead
    return s


def smt_view(enc, flat=None, packed=None, lo=False):
    s = L.Smt()
    s.pose = linear_view(enc.pose_encoder.weight, enc.pose_encoder.bias)
    s.fus0 = linear_view(enc.fusion_encoder[0].weight, enc.fusion_encoder[0].bias, flat, packed, lo=lo)
    s.fus2 = linear_view(enc.fusion_encoder[2].weight, enc.fusion_encoder[2].bias, flat, lo=lo)
    s.tr = transformer_view(enc.transformer, enc._dim_feedforward, enc._nhead, flat, lo=lo)
    return s


def dialog_view(enc, flat=None, lo=False):
    s = L.Dialog()
    s.fus0 = linear_view(enc.fusion_encoder[0].weight, enc.fusion_encoder[0].bias, flat, lo=lo)
    s.fus2 = linear_view(enc.fusion_encoder[2].weight, enc.fusion_encoder[2].bias, flat, lo=lo)
    s.tr = transformer_view(enc.dialog_transformer, enc._dim_feedforward, enc._nhead, flat, lo=lo)
    s.pe = P(enc.pos_encode.pe)
    s.pe_len = enc.pos_encode.pe.shape[0]
    return s


def clip_view(clip, flat=None, packed=None, fmt=0):
    """fmt: format of the tower's 16-bit weight shadows (0 bf16, 1 fp16: AVLEN_PREC_FP16 calls)."""
    s = L.ClipText()
    s.half_fmt = fmt
    s.tok_emb, s.pos_emb = P(clip.token_embedding.weight), P(clip.positional_embedding)
    for i, blk in enumerate(clip.transformer.resblocks):
        b = s.block[i]
        b.ln1, b.ln2 = affine_view(blk.ln_1), affine_view(blk.ln_2)
        b.attn = mha_view(blk.attn, flat, fmt)
        b.fc = linear_view(blk.mlp.c_fc.weight, blk.mlp.c_fc.bias, flat, fmt=fmt)
        b.proj = linear_view(blk.mlp.c_proj.weight, blk.mlp.c_proj.bias, flat, fmt=fmt)
        if packed is not None and blk.attn.in_proj_weight.shape[1] % 8 == 0:
            b.attn_fold = packed.ln_fold(blk.attn.in_proj_weight, blk.attn.in_proj_bias, blk.ln_1, fmt)
            b.fc_fold = packed.ln_fold(blk.mlp.c_fc.weight, blk.mlp.c_fc.bias, blk.ln_2, fmt)
    s.ln_final = affine_view(clip.ln_final)
    s.text_proj = P(clip.text_projection)
    s.vocab, s.ctx = clip.vocab_size, clip.context_length
    s.width, s.heads, s.layers = clip.transformer.width, clip.heads, clip.transformer.layers
    s.out_dim = clip.text_projection.shape[1]
    if packed is not None:
        # [out][width] copy of the projection: the few-row form (one wave per output column) reads rows, not columns
        tp = torch.empty(clip.text_projection.shape[1], clip.text_projection.shape[0], dtype=torch.float32, device=packed.device)
        packed.bufs.append(tp)
        packed.jobs.append(("t32", clip.text_projection, tp, None, None, 0))
        s.text_proj_t = P(tp)
    if packed is not None and CFG.CLIP_STREAM:
        # per-wave weight streams of the one-launch sequence-stationary tower (csrc/clip_tower.hip): 0.76 ms for 64 dialogs against
        # 0.93 ms for the launch-per-GEMM tower (AVLEN_CLIP_STREAM=0 keeps that one).  0 bytes: shape not supported.
        nb = L.lib.avlen_clip_stream_bytes(C.byref(s))
        if nb:
            buf = torch.empty(nb, dtype=torch.uint8, device=packed.device)
            packed.bufs.append(buf)
            packed.jobs.append(("clipstream", s, None, buf, None, fmt))
            s.wstream = P(buf)
    return s


def gru_view(rnn):
    return L.Gru(P(rnn.weight_ih_l0), P(rnn.weight_hh_l0), P(rnn.bias_ih_l0), P(rnn.bias_hh_l0),
                 rnn.weight_ih_l0.shape[1], rnn.weight_hh_l0.shape[1])


def heads_view(action_lin, critic_lin, unct_lin=None):
    h = L.Heads()
    h.action = linear_view(action_lin.weight, action_lin.bias)
    h.critic = linear_view(critic_lin.weight, critic_lin.bias)
    if unct_lin is not None:
        h.unct = linear_view(unct_lin.weight, unct_lin.bias)
        h.has_unct = 1
    else:
        h.has_unct = 0
    return h


def resnet18_grad_view(net_view, flat, prefix=""):
    """avlen_resnet18 gradient struct for avlen_resnet18_train_bwd: same dims as `net_view`, pointers at the CANONICAL gradient
    tensors inside `flat.grad` (names as in CustomResNet.state_dict(), smt_resnet.py:56-149)."""
    gp = lambda n: C.c_void_p(flat.grad_ptr(prefix + n))
    g = L.ResNet18()

    def conv(v, name):
        return L.Conv(gp(name), None, v.cin, v.cout, v.kh, v.kw, v.stride, v.pad)
    g.conv1 = conv(net_view.conv1, "conv1.weight")
    g.bn1 = L.Affine(gp("bn1.weight"), gp("bn1.bias"))
    for i in range(8):
        b, name = net_view.block[i], f"layer{i // 2 + 1}.{i % 2}."
        gb = g.block[i]
        gb.conv1, gb.conv2 = conv(b.conv1, name + "conv1.weight"), conv(b.conv2, name + "conv2.weight")
        gb.bn1 = L.Affine(gp(name + "bn1.weight"), gp(name + "bn1.bias"))
        gb.bn2 = L.Affine(gp(name + "bn2.weight"), gp(name + "bn2.bias"))
        gb.has_down = b.has_down
        if b.has_down:
            gb.down = conv(b.down, name + "downsample.0.weight")
            gb.bnd = L.Affine(gp(name + "downsample.1.weight"), gp(name + "downsample.1.bias"))
    g.fc = L.Linear(gp("fc.weight"), gp("fc.bias"), net_view.fc.out_f, net_view.fc.in_f)
    return g


def cnn3_grad_view(net_view, flat, prefix):
    """avlen_cnn3 gradient struct (canonical conv OIHW / fc (out, C*H*W) tensors inside flat.grad)."""
    gp = lambda n: C.c_void_p(flat.grad_ptr(prefix + n))
    g = L.Cnn3()
    for i, idx in enumerate((0, 2, 4)):
        v = net_view.conv[i]
        g.conv[i] = L.Conv(gp(f"cnn.{idx}.weight"), gp(f"cnn.{idx}.bias"), v.cin, v.cout, v.kh, v.kw, v.stride, v.pad)
    g.fc = L.Linear(gp("cnn.6.weight"), gp("cnn.6.bias"), net_view.fc.out_f, net_view.fc.in_f)
    return g


def grad_struct_like(view, name_of_ptr, flat):
    """Build a struct of the same ctypes type whose pointers address the matching ranges of flat.grad.
    `name_of_ptr`: {param data_ptr -> name}.  Pointers into a packed projection (row slices) are not
    needed here: every view above points at the start of a parameter tensor."""
    out = type(view)()
    for fname, ftype in view._fields_:
        val = getattr(view, fname)
        if isinstance(val, C.Structure):
            setattr(out, fname, grad_struct_like(val, name_of_ptr, flat))
        elif isinstance(val, C.Array):
            arr = getattr(out, fname)
            for i in range(len(val)):
                arr[i] = grad_struct_like(val[i], name_of_ptr, flat)
        elif ftype is C.c_void_p:
            if val is None:
                setattr(out, fname, None)
            else:
                name = name_of_ptr.get(val)
                setattr(out, fname, C.c_void_p(flat.grad_ptr(name)) if name in flat.trained_names else None)
        else:
            setattr(out, fname, val)
    return out
