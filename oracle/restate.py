"""CPU restatement of the AVLEN / SAVi PPO hot path (TEST INFRASTRUCTURE ONLY).

This file is the *oracle*: a plain-PyTorch, fp32, CPU restatement of the algorithm the
reference executes on the rollout-and-update path.  It is written from the reference's
behaviour (file:line cited per function, relative to /root/reference) as pure functions over
a ``state_dict`` that uses the reference's parameter names, so that the reference's own
modules can be loaded into it for pinning (``oracle/make_goldens.py`` -> ``tests/golden``).

Pinned: every function below is compared with the reference's own modules executed in the
build container (see tests/test_oracle_vs_golden.py).  NOT pinned ("parity unpinned"):
``clip_encode_text`` -- OpenAI CLIP is an un-vendored, un-pinned third-party dependency
(README.md:61) that is absent from /root/reference; the function restates CLIP's public
text-transformer definition and is anchored on the call site policy.py:847-849; it is cross-checked
against Hugging Face transformers' independent CLIP text model (tests/test_clip_independent.py).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this file.
The product package ``avlen_amd`` never does.
"""
import math
import torch
import torch.nn.functional as F

# --------------------------------------------------------------------------------------
# helpers
# --------------------------------------------------------------------------------------


def _lin(sd, p, x):
    return x @ sd[p + ".weight"].t() + sd[p + ".bias"]


def _layer_norm(sd, p, x, eps=1e-5):
    mu = x.mean(-1, keepdim=True)
    var = ((x - mu) ** 2).mean(-1, keepdim=True)
    return (x - mu) / torch.sqrt(var + eps) * sd[p + ".weight"] + sd[p + ".bias"]


def _group_norm(x, groups, w, b, eps=1e-5):
    # x: (B,C,H,W); statistics per (sample, group) over (C/groups, H, W)   [smt_resnet.py:30-33]
    B, C, H, W = x.shape
    xg = x.reshape(B, groups, -1)
    mu = xg.mean(-1, keepdim=True)
    var = ((xg - mu) ** 2).mean(-1, keepdim=True)
    xn = ((xg - mu) / torch.sqrt(var + eps)).reshape(B, C, H, W)
    return xn * w.view(1, C, 1, 1) + b.view(1, C, 1, 1)


# --------------------------------------------------------------------------------------
# a1/a2/a3: visual encoder                       smt_cnn.py:78-115, smt_resnet.py:37-53,132-146
# --------------------------------------------------------------------------------------


def resize_center_crop_64(x):
    """common/utils.py:467-557 for a square input whose side is a multiple of 64:
    area interpolation to 64x64 is an exact k x k block mean; the centre crop is the identity."""
    B, C, H, W = x.shape
    assert H == W and H % 64 == 0
    k = H // 64
    return x.reshape(B, C, 64, k, 64, k).mean(dim=(3, 5))


def _basic_block(sd, p, x, stride, has_down):
    idt = x
    out = F.conv2d(x, sd[p + ".conv1.weight"], None, stride=stride, padding=1)
    out = torch.relu(_group_norm(out, 16, sd[p + ".bn1.weight"], sd[p + ".bn1.bias"]))
    out = F.conv2d(out, sd[p + ".conv2.weight"], None, stride=1, padding=1)
    out = _group_norm(out, 16, sd[p + ".bn2.weight"], sd[p + ".bn2.bias"])
    if has_down:
        idt = F.conv2d(x, sd[p + ".downsample.0.weight"], None, stride=stride)
        idt = _group_norm(idt, 16, sd[p + ".downsample.1.weight"], sd[p + ".downsample.1.bias"])
    return torch.relu(out + idt)


def custom_resnet18(sd, p, x):
    """smt_resnet.py:132-146.  x: (B,Cin,64,64) -> (B,64)."""
    x = F.conv2d(x, sd[p + ".conv1.weight"], None, stride=1, padding=3)
    x = torch.relu(_group_norm(x, 16, sd[p + ".bn1.weight"], sd[p + ".bn1.bias"]))
    for li, stride in ((1, 1), (2, 2), (3, 2), (4, 2)):
        x = _basic_block(sd, f"{p}.layer{li}.0", x, stride, has_down=(li != 1))
        x = _basic_block(sd, f"{p}.layer{li}.1", x, 1, has_down=False)
    x = x.flatten(1)                                        # NCHW flatten order (C,H,W)
    return _lin(sd, p + ".fc", x)


def smt_cnn(sd, p, obs):
    """smt_cnn.py:78-115.  obs['rgb'] (B,128,128,3) 0..255, obs['depth'] (B,128,128,1) -> (B,128)."""
    rgb = obs["rgb"].permute(0, 3, 1, 2) / 255.0
    dep = obs["depth"].permute(0, 3, 1, 2)
    f_rgb = custom_resnet18(sd, p + ".rgb_encoder", resize_center_crop_64(rgb))
    f_dep = custom_resnet18(sd, p + ".depth_encoder", resize_center_crop_64(dep))
    return torch.cat([f_rgb, f_dep], 1)


# --------------------------------------------------------------------------------------
# a4/a5: audio / visual 3-conv CNNs                audio_cnn.py:44-49,136-151; visual_cnn.py
# --------------------------------------------------------------------------------------


def audio_cnn_geometry(h, w):
    """audio_cnn.py:44-49: kernel/stride choice depends on the spectrogram size."""
    if h < 30 or w < 30:
        return [(5, 2), (3, 2), (3, 1)]
    return [(8, 4), (4, 2), (3, 1)]


def audio_cnn(sd, p, spec):
    """audio_cnn.py:136-151.  spec (B,H,W,2) -> (B,out)."""
    x = spec.permute(0, 3, 1, 2)
    geo = audio_cnn_geometry(spec.shape[1], spec.shape[2])
    for i, (k, s) in zip((0, 2, 4), geo):
        x = F.conv2d(x, sd[f"{p}.cnn.{i}.weight"], sd[f"{p}.cnn.{i}.bias"], stride=s)
        if i != 4:
            x = torch.relu(x)
    x = x.flatten(1)
    return torch.relu(_lin(sd, p + ".cnn.6", x))


def visual_cnn(sd, p, obs):
    """visual_cnn.py:165-190 (GRU baseline).  rgb/255 ++ depth -> conv8s4,4s2,3s2 -> fc -> relu."""
    x = torch.cat([obs["rgb"].permute(0, 3, 1, 2) / 255.0, obs["depth"].permute(0, 3, 1, 2)], 1)
    for i, s in zip((0, 2, 4), (4, 2, 2)):
        x = F.conv2d(x, sd[f"{p}.cnn.{i}.weight"], sd[f"{p}.cnn.{i}.bias"], stride=s)
        if i != 4:
            x = torch.relu(x)
    x = x.flatten(1)
    return torch.relu(_lin(sd, p + ".cnn.6", x))


# --------------------------------------------------------------------------------------
# a6: masked GRU                                   av_nav/models/rnn_state_encoder.py:80-143
# --------------------------------------------------------------------------------------


def gru_cell(sd, p, x, h):
    gi = x @ sd[p + ".weight_ih_l0"].t() + sd[p + ".bias_ih_l0"]
    gh = h @ sd[p + ".weight_hh_l0"].t() + sd[p + ".bias_hh_l0"]
    H = h.shape[1]
    r = torch.sigmoid(gi[:, :H] + gh[:, :H])
    z = torch.sigmoid(gi[:, H:2 * H] + gh[:, H:2 * H])
    n = torch.tanh(gi[:, 2 * H:] + r * gh[:, 2 * H:])
    return (1 - z) * n + z * h


def rnn_state_encoder(sd, p, x, hidden, masks):
    """x (B,in) with hidden (1,B,H), masks (B,1): one step.  x (T*N,in): a sequence whose hidden
    state is multiplied by masks[t] before step t (equivalent to the reference's split-at-zeros
    loop, rnn_state_encoder.py:92-143, because a mask of 1 is a no-op)."""
    n = hidden.shape[1]
    h = hidden[0]
    if x.shape[0] == n:
        h = gru_cell(sd, p + ".rnn", x, h * masks)
        return h, h.unsqueeze(0)
    t = x.shape[0] // n
    xs, ms = x.view(t, n, -1), masks.view(t, n, 1)
    outs = []
    for i in range(t):
        h = gru_cell(sd, p + ".rnn", xs[i], h * ms[i])
        outs.append(h)
    return torch.cat(outs, 0), h.unsqueeze(0)


# --------------------------------------------------------------------------------------
# torch.nn.Transformer (post-norm, relu), restated          smt_state_encoder.py:88-96,160-166
# --------------------------------------------------------------------------------------


def mha(sd, p, q_in, kv_in, nhead, key_padding_mask=None, causal=False):
    """q_in (Sq,B,d), kv_in (Sk,B,d); packed in_proj (3d,d); key_padding_mask (B,Sk) True=ignore."""
    Sq, B, d = q_in.shape
    Sk = kv_in.shape[0]
    W, bi = sd[p + ".in_proj_weight"], sd[p + ".in_proj_bias"]
    q = q_in @ W[:d].t() + bi[:d]
    k = kv_in @ W[d:2 * d].t() + bi[d:2 * d]
    v = kv_in @ W[2 * d:].t() + bi[2 * d:]
    hd = d // nhead
    q = q.reshape(Sq, B, nhead, hd).permute(1, 2, 0, 3) * (1.0 / math.sqrt(hd))
    k = k.reshape(Sk, B, nhead, hd).permute(1, 2, 0, 3)
    v = v.reshape(Sk, B, nhead, hd).permute(1, 2, 0, 3)
    s = q @ k.transpose(-1, -2)                                  # (B,h,Sq,Sk)
    if key_padding_mask is not None:
        s = s.masked_fill(key_padding_mask[:, None, None, :], float("-inf"))
    if causal:
        cm = torch.ones(Sq, Sk, dtype=torch.bool, device=s.device).triu(1)
        s = s.masked_fill(cm, float("-inf"))
    a = torch.softmax(s, -1)
    o = (a @ v).permute(2, 0, 1, 3).reshape(Sq, B, d)
    return o @ sd[p + ".out_proj.weight"].t() + sd[p + ".out_proj.bias"]


def transformer(sd, p, src, tgt, nhead, kpm):
    """nn.Transformer(d, nhead, 1 enc, 1 dec, ff, dropout=0, relu), norm_first=False, with
    src_key_padding_mask = memory_key_padding_mask = kpm (B,S) and final enc/dec LayerNorms."""
    e = p + ".encoder.layers.0"
    x = src
    x = _layer_norm(sd, e + ".norm1", x + mha(sd, e + ".self_attn", x, x, nhead, kpm))
    ff = _lin(sd, e + ".linear2", torch.relu(_lin(sd, e + ".linear1", x)))
    x = _layer_norm(sd, e + ".norm2", x + ff)
    mem = _layer_norm(sd, p + ".encoder.norm", x)
    dl = p + ".decoder.layers.0"
    y = tgt
    y = _layer_norm(sd, dl + ".norm1", y + mha(sd, dl + ".self_attn", y, y, nhead))
    y = _layer_norm(sd, dl + ".norm2", y + mha(sd, dl + ".multihead_attn", y, mem, nhead, kpm))
    ff = _lin(sd, dl + ".linear2", torch.relu(_lin(sd, dl + ".linear1", y)))
    y = _layer_norm(sd, dl + ".norm3", y + ff)
    return _layer_norm(sd, p + ".decoder.norm", y)


# --------------------------------------------------------------------------------------
# a11: SMT state encoder                                   smt_state_encoder.py:109-276
# --------------------------------------------------------------------------------------


def relative_pose(pose_a, pose_b):
    """smt_state_encoder.py:238-265: pose_b - pose_a expressed in pose_a's frame."""
    ha, hb = -pose_a[..., 2], -pose_b[..., 2]
    dx, dy = pose_b[..., 0] - pose_a[..., 0], pose_b[..., 1] - pose_a[..., 1]
    r = torch.sqrt(dx * dx + dy * dy)
    phi = torch.atan2(dy, dx) - ha
    dh = hb - ha
    dh = -torch.atan2(torch.sin(dh), torch.cos(dh))
    return torch.stack([r * torch.cos(phi), r * torch.sin(phi), dh], -1)


def format_pose(rel_xyh, t):
    """smt_state_encoder.py:267-276 -> (x, y, cos h, sin h, exp(-t))."""
    return torch.stack([rel_xyh[..., 0], rel_xyh[..., 1], torch.cos(rel_xyh[..., 2]),
                        torch.sin(rel_xyh[..., 2]), torch.exp(-t)], -1)


def smt_state_encoder(sd, p, x, memory, memory_masks, goal, pose_indices, nhead=8, pretraining=False):
    """smt_state_encoder.py:109-188.  x (B,F), memory (M,B,F), memory_masks (B,M) 1=valid, goal (B,d)."""
    B = x.shape[0]
    ones = torch.ones(B, 1, device=x.device)
    if pretraining:                                               # :126-129
        masks = torch.cat([torch.zeros_like(memory_masks), ones], 1)
    else:
        masks = torch.cat([memory_masks, ones], 1)
    pi, pj = pose_indices
    xp, mp = x[..., pi:pj], memory[..., pi:pj]
    x_fmt = format_pose(relative_pose(xp[..., :3], xp[..., :3]), xp[..., 3])
    m_fmt = format_pose(relative_pose(xp[None, :, :3], mp[..., :3]), mp[..., 3])
    x = torch.cat([x[..., :pi], _lin(sd, p + ".pose_encoder", x_fmt), x[..., pj:]], -1)
    memory = torch.cat([memory[..., :pi], _lin(sd, p + ".pose_encoder", m_fmt), memory[..., pj:]], -1)
    seq = torch.cat([memory, x[None]], 0)                          # (M+1,B,F')
    seq = _lin(sd, p + ".fusion_encoder.2", torch.relu(_lin(sd, p + ".fusion_encoder.0", seq)))
    kpm = (1 - masks) > 0
    return transformer(sd, p + ".transformer", seq, goal[None], nhead, kpm)[-1]


# --------------------------------------------------------------------------------------
# a12: dialog state encoder                                dialog_state_encoder.py:114-155
# --------------------------------------------------------------------------------------


def sinusoid_table(max_len, d):
    """dialog_state_encoder.py:24-29 / ddppo_trainer.py:506-512."""
    pos = torch.arange(max_len).unsqueeze(1)
    div = torch.exp(torch.arange(0, d, 2) * (-math.log(10000.0) / d))
    pe = torch.zeros(max_len, d)
    pe[:, 0::2] = torch.sin(pos * div)
    pe[:, 1::2] = torch.cos(pos * div)
    return pe


def dialog_state_encoder(sd, p, x_att, memory_state, memory_masks, d_emb, agent_step, goal, nhead=8):
    B = x_att.shape[0]
    masks = torch.cat([memory_masks, torch.ones(B, 1, device=x_att.device)], 1)
    seq = torch.cat([memory_state, x_att[None]], 0)
    S = seq.shape[0]
    if d_emb is not None:
        seq = torch.cat([seq, d_emb[None].expand(S, -1, -1)], -1)
        seq = _lin(sd, p + ".fusion_encoder.2", torch.relu(_lin(sd, p + ".fusion_encoder.0", seq)))
    pe = sd[p + ".pos_encode.pe"] if (p + ".pos_encode.pe") in sd else sinusoid_table(100, seq.shape[-1])[:, None]
    seq = seq + pe[agent_step.long(), 0, :][None]
    kpm = (1 - masks) > 0
    return transformer(sd, p + ".dialog_transformer", seq, goal[None], nhead, kpm)[-1]


# --------------------------------------------------------------------------------------
# a19: CLIP text encoder (PUBLIC DEFINITION; PARITY UNPINNED — see module docstring)
# --------------------------------------------------------------------------------------


def clip_encode_text(sd, p, tokens, nhead=8):
    """tokens (B,77) int64 -> (B,512).  x = tok_emb + pos_emb; 12 pre-norm residual blocks with a
    causal mask and QuickGELU MLP; ln_final; take the row of the EOT token (= argmax id); project.
    openai/CLIP is absent (parity unpinned against it); cross-checked against Hugging Face transformers' CLIPTextModelWithProjection
    with the same weights in tests/test_clip_independent.py."""
    x = sd[p + ".token_embedding.weight"][tokens] + sd[p + ".positional_embedding"]
    x = x.permute(1, 0, 2)                                          # (S,B,d)
    i = 0
    while f"{p}.transformer.resblocks.{i}.ln_1.weight" in sd:
        b = f"{p}.transformer.resblocks.{i}"
        h = _layer_norm(sd, b + ".ln_1", x)
        x = x + mha(sd, b + ".attn", h, h, nhead, causal=True)
        h = _layer_norm(sd, b + ".ln_2", x)
        h = _lin(sd, b + ".mlp.c_fc", h)
        h = h * torch.sigmoid(1.702 * h)
        x = x + _lin(sd, b + ".mlp.c_proj", h)
        i += 1
    x = _layer_norm(sd, p + ".ln_final", x.permute(1, 0, 2))
    eot = tokens.argmax(-1)
    return x[torch.arange(x.shape[0]), eot] @ sd[p + ".text_projection"]


# --------------------------------------------------------------------------------------
# a7-a10: nets                                              policy.py:451-477,602-674,807-865,1031-1114
# --------------------------------------------------------------------------------------


def _one_hot(a, n):
    oh = torch.zeros(a.shape[0], n, device=a.device)
    oh.scatter_(1, a.long(), 1)
    return oh


def smt_features(sd, obs, prev_actions, use_category_input=False):
    """policy.py:662-674: [visual 128 | action 16 | audio 128 | (category 21) | pose 4]."""
    parts = [smt_cnn(sd, "net.visual_encoder", obs),
             _lin(sd, "net.action_encoder", _one_hot(prev_actions, 4)),
             audio_cnn(sd, "net.goal_encoder", obs["spectrogram"])]
    if use_category_input:
        parts.append(obs["category"])
    parts.append(obs["pose"])
    return torch.cat(parts, 1)


def belief_vector(obs, d=256, normalize=False):
    """policy.py:605-618."""
    B = obs["pose"].shape[0]
    b = torch.zeros(B, d, device=obs["pose"].device)
    cb = obs["category_belief"]
    b[:, :21] = torch.softmax(cb, 1) if normalize else cb
    b[:, 21:23] = obs["location_belief"]
    return b


def smt_net(sd, obs, prev_actions, ext_memory, ext_memory_masks, pretraining=False, use_category_input=False):
    """AudioNavSMTNet.forward, policy.py:602-625 -> (x_att, x)."""
    x = smt_features(sd, obs, prev_actions, use_category_input)
    pi = x.shape[1] - 4
    x_att = smt_state_encoder(sd, "net.smt_state_encoder", x, ext_memory, ext_memory_masks,
                              belief_vector(obs), (pi, pi + 4), pretraining=pretraining)
    return x_att, x


def option_net(sd, obs, prev_actions, ext_memory, ext_memory_masks, query_state, last_query_info,
               pretraining=False, use_category_input=False):
    """AudioNavOptionNet.forward, policy.py:1031-1065 -> (x_att, x_for_memory).
    NB policy.py:1035-1036: [x | query_state] is built under no_grad, so no gradient reaches
    the visual / audio / action encoders through this net."""
    x = smt_features(sd, obs, prev_actions, use_category_input).detach()
    pi = x.shape[1] - 4
    xq = torch.cat([x, query_state], 1)
    x_att = smt_state_encoder(sd, "net.smt_state_encoder", xq, ext_memory, ext_memory_masks,
                              belief_vector(obs), (pi, pi + 4), pretraining=pretraining)
    return x_att, torch.cat([x, last_query_info], 1)


def dialog_net(sd, obs, prev_actions, ext_memory, ext_memory_dialog, ext_memory_masks, all_dialog, agent_step,
               clip_fn=None):
    """AudioNavDialogNet.forward, policy.py:807-865 -> (x_att_dialog, x).  clip_fn(tokens)->(B,512)
    defaults to the restated CLIP text encoder over sd['net.clip.*']."""
    x = smt_features(sd, obs, prev_actions, False)
    pi = x.shape[1] - 4
    goal = belief_vector(obs)
    x_att = smt_state_encoder(sd, "net.smt_state_encoder", x, ext_memory, ext_memory_masks, goal, (pi, pi + 4))
    d_emb = None
    if all_dialog is not None:
        with torch.no_grad():
            e = (clip_fn or (lambda t: clip_encode_text(sd, "net.clip", t)))(all_dialog).float()
        d_emb = _lin(sd, "net.dialog_layer", e)
    xd = dialog_state_encoder(sd, "net.dialog_state_encoder", x_att, ext_memory_dialog, ext_memory_masks,
                              d_emb, agent_step, goal)
    return xd, x


def baseline_net(sd, obs, hidden, masks):
    """AudioNavBaselineNet.forward, policy.py:451-477: [audio 512 | visual 512 | category 21] -> GRU."""
    x = torch.cat([audio_cnn(sd, "net.audio_encoder", obs["spectrogram"]),
                   visual_cnn(sd, "net.visual_encoder", obs), obs["category"]], 1)
    return rnn_state_encoder(sd, "net.state_encoder", x, hidden, masks)


# --------------------------------------------------------------------------------------
# a13/a14: heads + categorical                             common/utils.py:44-72; policy.py:70-276
# --------------------------------------------------------------------------------------


def categorical(logits):
    logp = logits - torch.logsumexp(logits, -1, keepdim=True)
    return logp, torch.exp(logp)


def sample_host(probs, generator=None):
    """Categorical.sample on the CPU generator == torch.multinomial(probs,1,True): an exponential
    race, argmax(probs / Exp(1))."""
    q = torch.empty_like(probs).exponential_(1, generator=generator)
    return (probs / q).argmax(-1, keepdim=True)


def heads(sd, which, feats, action=None, deterministic=False, generator=None):
    """which in {'goal','option','vln'} -> dict(value, unct, logits, probs, action, log_prob, entropy)."""
    logits = _lin(sd, f"action_distribution_{which}.linear", feats)
    logp, probs = categorical(logits)
    out = {"logits": logits, "probs": probs, "value": _lin(sd, f"critic_{which}.fc", feats)}
    if which == "option":
        out["unct"] = _lin(sd, "uncertainty_option.fc", feats)
    if action is None:
        action = probs.argmax(-1, keepdim=True) if deterministic else sample_host(probs, generator)
    out["action"] = action
    out["log_prob"] = logp.gather(1, action.long())
    out["entropy"] = -(probs * logp).sum(-1).mean()
    return out


# --------------------------------------------------------------------------------------
# a15: storage arithmetic                                  rollout_storage.py:394-412, 930-941
# --------------------------------------------------------------------------------------


def gae_returns(rewards, value_preds, masks, next_value, gamma, tau, steps=None):
    """rollout_storage.py:394-405.  rewards (T,N,1), value_preds (T+1,N,1), masks (T+1,N,1)."""
    T = rewards.shape[0] if steps is None else steps
    v = value_preds.clone()
    v[T] = next_value
    ret = torch.zeros_like(v)
    gae = torch.zeros_like(v[0])
    for s in reversed(range(T)):
        delta = rewards[s] + gamma * v[s + 1] * masks[s + 1] - v[s]
        gae = delta + gamma * tau * masks[s + 1] * gae
        ret[s] = gae + v[s]
    return ret, v


class ExtMemoryRing:
    """ExternalMemory.insert, rollout_storage.py:907-941, ONE copy (all copies are identical)."""
    def __init__(self, num_envs, total_size, capacity, dim):
        self.total_size, self.capacity = total_size, capacity
        self.masks = torch.zeros(num_envs, total_size)
        self.memory = torch.zeros(total_size, num_envs, dim)
        self.idx = 0

    def insert(self, feats, not_done):
        self.memory[self.idx].copy_(feats)
        overflow = self.masks.sum(1) == self.capacity
        self.masks[overflow, self.idx - self.capacity] = 0.0
        self.masks[:, self.idx] = 1.0
        self.masks *= not_done
        self.idx = (self.idx + 1) % self.total_size


# --------------------------------------------------------------------------------------
# a16: PPO loss / step                                     ppo.py:157-303
# --------------------------------------------------------------------------------------


def ppo_losses(values, unct, action_log_probs, entropy, old_log_probs, adv, rl_masks, value_preds, returns,
               unct_gt, clip=0.2):
    """ppo.py:219-262 -> (value_loss, action_loss, unct_loss, values_mean, returns_mean)."""
    ratio = torch.exp(action_log_probs - old_log_probs)
    m = rl_masks.unsqueeze(1).to(ratio.dtype)
    surr1 = ratio * adv * m
    surr2 = torch.clamp(ratio, 1.0 - clip, 1.0 + clip) * adv * m
    action_loss = -torch.min(surr1, surr2).sum() / m.sum()
    vclip = value_preds + (values - value_preds).clamp(-clip, clip)
    value_loss = 0.5 * torch.max((values - returns) ** 2, (vclip - returns) ** 2).mean()
    lp = unct - torch.logsumexp(unct, -1, keepdim=True)
    unct_loss = -lp.gather(1, unct_gt.long().view(-1, 1)).mean()
    return value_loss, action_loss, unct_loss, values.mean(), returns.mean()


def total_loss(value_loss, action_loss, entropy, unct_loss, value_coef=0.5, entropy_coef=0.05, unct_coef=0.5):
    return value_loss * value_coef + action_loss - entropy * entropy_coef + unct_coef * unct_loss


def clip_grad_norm(grads, max_norm):
    """nn.utils.clip_grad_norm_ (ppo.py:297-300): global L2 norm, scale by max/(norm+1e-6) clamped to 1."""
    tot = torch.sqrt(sum((g.double() ** 2).sum() for g in grads)).float()
    coef = torch.clamp(max_norm / (tot + 1e-6), max=1.0)
    return [g * coef for g in grads], tot


def adam_step(p, g, m, v, step, lr, eps, b1=0.9, b2=0.999):
    """torch.optim.Adam (no weight decay, no amsgrad), one tensor."""
    m.mul_(b1).add_(g, alpha=1 - b1)
    v.mul_(b2).addcmul_(g, g, value=1 - b2)
    bc1, bc2 = 1 - b1 ** step, 1 - b2 ** step
    denom = (v.sqrt() / math.sqrt(bc2)).add_(eps)
    p.addcdiv_(m, denom, value=-lr / bc1)


# --------------------------------------------------------------------------------------
# a18 / f1: BeliefPredictor                           belief_predictor.py:56-206, 213-230
# --------------------------------------------------------------------------------------


def belief_cnn_forward(sd, obs, has_distractor_sound=False, p="predictor"):
    """belief_predictor.py:126-137: custom_resnet18 on the spectrogram at its own size (no resize) -> (B,2)."""
    spec = obs["spectrogram"].permute(0, 3, 1, 2)
    if has_distractor_sound:
        lab = obs["category"]
        spec = torch.cat([spec, lab.reshape(lab.shape + (1, 1)).expand(lab.shape + spec.shape[-2:])], 1)
    return custom_resnet18(sd, p, spec)


def _batch_norm_eval(sd, p, x, eps=1e-5):
    s = sd[p + ".weight"] / torch.sqrt(sd[p + ".running_var"] + eps)
    return x * s.view(1, -1, 1, 1) + (sd[p + ".bias"] - sd[p + ".running_mean"] * s).view(1, -1, 1, 1)


def tv_resnet18(sd, p, x):
    """torchvision.models.resnet18 in eval mode (third party, un-vendored; PARITY UNPINNED: restated from the public
    architecture -- conv7x7/2, BN, ReLU, maxpool3x3/2, 4 stages x 2 BasicBlocks (64,128,256,512), avgpool, fc --
    anchored on the call sites belief_predictor.py:79-81,179; cross-checked against Hugging Face transformers'
    ResNetForImageClassification with the same weights in tests/test_tv_resnet18_independent.py)."""
    x = torch.relu(_batch_norm_eval(sd, p + ".bn1", F.conv2d(x, sd[p + ".conv1.weight"], None, stride=2, padding=3)))
    x = F.max_pool2d(x, 3, 2, 1)
    for li, stride in ((1, 1), (2, 2), (3, 2), (4, 2)):
        for bi in (0, 1):
            q = f"{p}.layer{li}.{bi}"
            s = stride if bi == 0 else 1
            idt = x
            out = torch.relu(_batch_norm_eval(sd, q + ".bn1", F.conv2d(x, sd[q + ".conv1.weight"], None, stride=s, padding=1)))
            out = _batch_norm_eval(sd, q + ".bn2", F.conv2d(out, sd[q + ".conv2.weight"], None, stride=1, padding=1))
            if q + ".downsample.0.weight" in sd:
                idt = _batch_norm_eval(sd, q + ".downsample.1", F.conv2d(x, sd[q + ".downsample.0.weight"], None, stride=s))
            x = torch.relu(out + idt)
    return _lin(sd, p + ".fc", x.mean(dim=(2, 3)))


def _base_to_odom(pg, pose):
    import numpy as np                                      # belief_predictor.py:213-220
    angle = -pose[2]
    d = np.linalg.norm(pg)
    theta = np.arctan2(pg[1], pg[0])
    return np.array([pose[0] + d * np.cos(theta + angle), pose[1] + d * np.sin(theta + angle)])


def _odom_to_base(pg, pose):
    import numpy as np                                      # belief_predictor.py:223-230
    angle = -pose[2]
    delta = pg - pose[:2]
    dth = np.arctan2(delta[1], delta[0]) - angle
    d = np.linalg.norm(delta)
    return np.array([d * np.cos(dth), d * np.sin(dth)])


class BeliefFilter:
    """The per-environment loop of BeliefPredictor.update (belief_predictor.py:146-206) over given network outputs."""

    def __init__(self, num_env, weighting_factor=0.5, current_pred_only=False):
        self.last_pointgoal = [None] * num_env
        self.last_label = [None] * num_env
        self.w, self.current_pred_only = weighting_factor, current_pred_only

    def update(self, obs, dones, pointgoals=None, labels=None):
        import numpy as np
        B = obs["spectrogram"].shape[0]
        if pointgoals is not None:
            pgs = pointgoals.cpu().numpy()
            for i in range(B):
                pose = obs["pose"][i].cpu().numpy()
                if dones is not None and dones[i]:
                    self.last_pointgoal[i] = None
                if obs["spectrogram"][i].sum().item() != 0:
                    base = np.array([-pgs[i][1], pgs[i][0]])
                    if self.last_pointgoal[i] is None or self.current_pred_only:
                        avg = base
                    else:
                        avg = (1 - self.w) * base + self.w * _odom_to_base(self.last_pointgoal[i], pose)
                    self.last_pointgoal[i] = _base_to_odom(avg, pose)
                elif self.last_pointgoal[i] is None:
                    avg = np.array([10, 10])
                else:
                    avg = _odom_to_base(self.last_pointgoal[i], pose)
                obs["location_belief"][i].copy_(torch.from_numpy(np.asarray(avg)))
        if labels is not None:
            labs = labels[:, :21].cpu().numpy()
            for i in range(B):
                if dones is not None and dones[i]:
                    self.last_label[i] = None
                if obs["spectrogram"][i].sum().item() != 0:
                    if self.last_label[i] is None or self.current_pred_only:
                        avg = labs[i]
                    else:
                        avg = (1 - self.w) * labs[i] + self.w * self.last_label[i]
                    self.last_label[i] = avg
                elif self.last_label[i] is None:
                    avg = np.ones(21) / 21
                else:
                    avg = self.last_label[i]
                obs["category_belief"][i].copy_(torch.from_numpy(np.asarray(avg)))
