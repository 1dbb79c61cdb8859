"""Deterministic fixture tensors (TEST INFRASTRUCTURE).

Weights and inputs are a closed-form function of (tensor name, flat index) computed with exact
integer arithmetic (a 64-bit multiplicative hash -> top 24 bits -> fp32), so the golden
generator in the build container and the tests on the GPU box regenerate bit-identical
tensors without storing them (SURVEY.md §8c "fixture recipe").
"""
import zlib
import numpy as np
import torch

_M1 = np.uint64(0x9E3779B97F4A7C15)
_M2 = np.uint64(0xBF58476D1CE4E5B9)
_M3 = np.uint64(0x94D049BB133111EB)


def unit(name, n):
    """n fp32 values in [0,1) as exact multiples of 2^-24."""
    seed = np.uint64(zlib.crc32(name.encode()))
    with np.errstate(over="ignore"):
        x = (np.arange(n, dtype=np.uint64) + np.uint64(1)) * _M1 + seed * _M3
        x ^= x >> np.uint64(30)
        x *= _M2
        x ^= x >> np.uint64(27)
        x *= _M3
        x ^= x >> np.uint64(31)
    return ((x >> np.uint64(40)).astype(np.float32)) * np.float32(2.0 ** -24)


def sym(name, shape, scale=1.0):
    """fp32 tensor uniform in [-scale, scale)."""
    n = int(np.prod(shape)) if len(shape) else 1
    v = (unit(name, n) - np.float32(0.5)) * np.float32(2.0 * scale)
    return torch.from_numpy(v.astype(np.float32)).reshape(tuple(shape))


def uni(name, shape, lo=0.0, hi=1.0):
    n = int(np.prod(shape)) if len(shape) else 1
    v = unit(name, n) * np.float32(hi - lo) + np.float32(lo)
    return torch.from_numpy(v.astype(np.float32)).reshape(tuple(shape))


def ints(name, shape, n_values):
    n = int(np.prod(shape)) if len(shape) else 1
    v = np.floor(unit(name, n) * np.float32(n_values)).astype(np.int64)
    return torch.from_numpy(np.minimum(v, n_values - 1)).reshape(tuple(shape))


def weight_for(name, shape, tag=""):
    """A sane fixture value for a parameter called `name` (reference naming)."""
    key = tag + name
    shape = tuple(shape)
    leaf = name.split(".")[-1]
    if name.endswith("positional_embedding"):
        return sym(key, shape, 0.02)
    if name.endswith("token_embedding.weight"):
        return sym(key, shape, 0.05)
    if name.endswith("text_projection"):
        return sym(key, shape, (3.0 / shape[0]) ** 0.5)
    if name.endswith("pos_encode.pe"):
        return None                              # buffer: keep the module's own value
    if len(shape) == 1 and leaf == "weight":      # every 1-D "weight" is a norm scale
        return 1.0 + sym(key, shape, 0.2)
    if len(shape) >= 2:
        fan_in = int(np.prod(shape[1:]))
        return sym(key, shape, (6.0 / fan_in) ** 0.5)
    return sym(key, shape, 0.1)                    # biases, in_proj_bias, ...


def state_dict_for(spec, tag=""):
    """spec: {name: shape}.  Returns {name: tensor} (buffers that must keep their own value omitted)."""
    out = {}
    for k, shp in spec.items():
        w = weight_for(k, shp, tag)
        if w is not None:
            out[k] = w
    return out


def observations(tag, B, spectrogram=(65, 26), step=0):
    """Synthetic observation batch with the reference's sensor shapes/dtypes."""
    H, W = spectrogram
    pose = torch.stack([
        sym(f"{tag}.pose.x", (B,), 10.0), sym(f"{tag}.pose.y", (B,), 10.0),
        sym(f"{tag}.pose.h", (B,), 3.14159), ints(f"{tag}.pose.t", (B,), 40).float() + step], 1)
    cb = torch.softmax(sym(f"{tag}.cb", (B, 21), 2.0), 1)
    cat = torch.zeros(B, 21)
    cat[torch.arange(B), ints(f"{tag}.cat", (B,), 21)] = 1.0
    return {
        "rgb": ints(f"{tag}.rgb", (B, 128, 128, 3), 256).float(),
        "depth": uni(f"{tag}.depth", (B, 128, 128, 1)),
        "spectrogram": torch.log1p(3.0 * uni(f"{tag}.spec", (B, H, W, 2), 0.0, 2.0)),
        "category": cat,
        "category_belief": cb,
        "location_belief": sym(f"{tag}.lb", (B, 2), 3.0),
        "pose": pose,
    }


def memory(tag, M, B, dim, pose_start):
    """External memory rows (M,B,dim) with pose-like values in [pose_start, pose_start+4)."""
    m = sym(f"{tag}.mem", (M, B, dim), 1.0)
    m[..., pose_start + 0] = sym(f"{tag}.mem.x", (M, B), 10.0)
    m[..., pose_start + 1] = sym(f"{tag}.mem.y", (M, B), 10.0)
    m[..., pose_start + 2] = sym(f"{tag}.mem.h", (M, B), 3.14159)
    m[..., pose_start + 3] = ints(f"{tag}.mem.t", (M, B), 40).float()
    return m


def mask_patterns(tag, B, M):
    """(B,M) 0/1 masks covering empty / full / partial rows."""
    mk = (unit(f"{tag}.mask", B * M).reshape(B, M) < 0.6).astype(np.float32)
    mk[0, :] = 0.0
    if B > 1:
        mk[1, :] = 1.0
    return torch.from_numpy(mk)


def dialog_tokens(tag, B, ctx=77, vocab=49408):
    """(B,77) int64: SOT, random ids, EOT (= the largest id, so argmax finds it), zeros after."""
    t = torch.zeros(B, ctx, dtype=torch.long)
    ln = ints(f"{tag}.dlen", (B,), ctx - 4) + 2
    body = ints(f"{tag}.dtok", (B, ctx), vocab - 2)
    for b in range(B):
        L = int(ln[b])
        t[b, 0] = vocab - 2
        t[b, 1:L] = body[b, 1:L]
        t[b, L] = vocab - 1
    return t


def stub_text_embedding(tokens, d=512):
    """A fixed, cheap stand-in for CLIP.encode_text used ONLY to pin pi_l's non-CLIP arithmetic
    against the reference (the real CLIP package is absent)."""
    proj = sym("stub_clip.proj", (tokens.shape[1], d), 1.0)
    return torch.sin((tokens.float() / 1000.0) @ proj)


def belief_scenario(tag, N=3, T=7, spectrogram=(65, 26)):
    """Per-step inputs of BeliefPredictor.update (belief_predictor.py:139-206): T steps x N envs with every branch of the
    filter exercised -- silent before any estimate, silent after one, episode resets while sounding and while silent,
    dones=None (ddppo_trainer.py:729) and dones as a list (ppo_trainer.py:892)."""
    silent = {(0, 1), (3, 1), (4, 1), (2, 2), (5, 0), (6, 2)}
    done = {(3, 0), (4, 1), (5, 2), (6, 2)}
    steps = []
    for t in range(T):
        o = observations(f"{tag}.t{t}", N, spectrogram, step=t)
        for (tt, i) in silent:
            if tt == t and i < N:
                o["spectrogram"][i] = 0.0
        o["location_belief"] = torch.zeros(N, 2)
        o["category_belief"] = torch.zeros(N, 21)
        dones = None if t == 0 else [(t, i) in done for i in range(N)]
        steps.append((o, dones))
    return steps


def delta_stats(post, pre, keys):
    """What the `cycle_*` goldens keep of a PPO.update's parameter step, per float tensor in `keys`: the L2 norm of the step
    (post - pre) and a SIGNED weighted checksum of it, sum_i w_i * (post - pre)_i with closed-form weights w (sym of the tensor's
    name): a transposed / shifted / sign-flipped step of the right size changes the checksum, which sum(|p|) does not see.
    float64 accumulation; `pre` is the closed-form fixture state (regenerated wherever the test runs)."""
    import numpy as np
    l2, chk = [], []
    for k in keys:
        d = (post[k].detach().double().cpu() - pre[k].detach().double().cpu()).reshape(-1)
        w = sym("chk." + k, (d.numel(),), 1.0).double()
        l2.append(float(d.norm()))
        chk.append(float((d * w).sum()))
    return np.array(l2), np.array(chk)
