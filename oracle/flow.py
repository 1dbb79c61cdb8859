"""CPU restatement of the rollout -> GAE -> PPO.update cycle (TEST INFRASTRUCTURE ONLY).

Drives oracle/restate.py in the reference's call order (ppo_trainer.py:323-897 for a rollout
step, ppo_trainer.py:1045-1093 + ppo.py:157-289 for the update) on synthetic observations.
Used by tests (pinned against tests/golden/cycle_*.npz generated from the reference's own
PPO / RolloutStorage / Policy classes) and as bench.py's ``cpu_baseline`` ("port").
"""
import math
import torch
import restate as R


class Storage:
    """RolloutStorage restated (rollout_storage.py:21-297) with ONE copy of each external memory."""

    def __init__(self, T, N, obs0, em_size, em_cap, dim_goal=276, dim_option=308, vln_size=3, dim_dialog=256, dim_vln=276):
        self.T, self.N, self.step = T, N, 0
        self.obs = {k: torch.zeros(T + 1, N, *v.shape[1:]) for k, v in obs0.items()}
        for k, v in obs0.items():
            self.obs[k][0].copy_(v)
        z = torch.zeros
        self.rewards, self.value_preds, self.returns = z(T, N, 1), z(T + 1, N, 1), z(T + 1, N, 1)
        self.action_log_probs = z(T, N, 1)
        self.actions = z(T, N, 1, dtype=torch.long)
        self.actions_option = z(T, N, 1, dtype=torch.long)
        self.prev_actions = z(T + 1, N, 1, dtype=torch.long)
        self.masks, self.masks_vln = z(T + 1, N, 1), z(T + 1, N, 1)
        self.rl_masks = z(T, N, dtype=torch.long)
        self.ucnt_gt = z(T, N, dtype=torch.long)
        self.query_state, self.last_query_info = z(T, N, 32), z(T, N, 32)
        self.agent_step = z(T, N)
        self.all_dialog = z(T, N, 77, dtype=torch.long)
        self.em_masks = z(T + 1, N, em_size)
        self.em_vln_masks = z(T + 1, N, vln_size)
        self.em = R.ExtMemoryRing(N, em_size, em_cap, dim_goal)
        self.em_option = R.ExtMemoryRing(N, em_size, em_cap, dim_option)
        self.em_vln = R.ExtMemoryRing(N, vln_size, vln_size, dim_vln)
        self.em_vln_dialog = R.ExtMemoryRing(N, vln_size, vln_size, dim_dialog)

    def insert(self, obs, actions, actions_option, logp, values, rewards, not_done, not_done_vln,
               f_goal, f_option, f_vln, f_dialog, dialog, rl_masks, ucnt_gt, query_state, last_query_info,
               agent_step):
        s = self.step
        for k in self.obs:
            self.obs[k][s + 1].copy_(obs[k])
        self.all_dialog[s].copy_(dialog)
        self.query_state[s].copy_(query_state)
        self.last_query_info[s].copy_(last_query_info)
        self.agent_step[s].copy_(agent_step)
        self.ucnt_gt[s].copy_(ucnt_gt)
        self.rl_masks[s].copy_(rl_masks)
        self.actions[s].copy_(actions)
        self.actions_option[s].copy_(actions_option)
        self.prev_actions[s + 1].copy_(actions)
        self.action_log_probs[s].copy_(logp)
        self.value_preds[s].copy_(values)
        self.rewards[s].copy_(rewards)
        self.masks[s + 1].copy_(not_done)
        self.masks_vln[s + 1].copy_(not_done_vln)
        self.em.insert(f_goal, not_done)
        self.em_masks[s + 1].copy_(self.em.masks)
        self.em_option.insert(f_option, not_done)
        self.em_vln.insert(f_vln, not_done_vln)
        self.em_vln_dialog.insert(f_dialog, not_done_vln)
        self.em_vln_masks[s + 1].copy_(self.em_vln.masks)
        self.step = s + 1

    def after_update(self):
        s = self.step
        for k in self.obs:
            self.obs[k][0].copy_(self.obs[k][s])
        for buf in (self.masks, self.masks_vln, self.prev_actions, self.em_masks, self.em_vln_masks):
            buf[0].copy_(buf[s])
        self.step = 0

    def minibatches(self, advantages, num_mini_batch):
        """recurrent_generator (rollout_storage.py:591-810): randperm over envs, N/num_mini_batch
        envs x all steps each, flattened T-major."""
        N, T = self.N, self.step
        assert N >= num_mini_batch
        per = N // num_mini_batch
        perm = torch.randperm(N)
        for start in range(0, N, per):
            ind = perm[start:start + per]
            fl = lambda x: x[:T, ind].reshape(T * len(ind), *x.shape[2:])
            yield {
                "obs": {k: fl(v) for k, v in self.obs.items()},
                "actions_option": fl(self.actions_option), "prev_actions": fl(self.prev_actions),
                "value_preds": fl(self.value_preds), "returns": fl(self.returns), "masks": fl(self.masks),
                "old_log_probs": fl(self.action_log_probs), "adv": fl(advantages),
                "rl_masks": fl(self.rl_masks), "ucnt_gt": fl(self.ucnt_gt),
                # every copy of the ring is identical, so step t sees the final ring + its own mask
                "em_option": self.em_option.memory[:, None, ind].expand(-1, T, -1, -1).reshape(
                    self.em_option.total_size, T * len(ind), -1),
                "em_masks": fl(self.em_masks),
                "query_state": fl(self.query_state), "last_query_info": fl(self.last_query_info),
            }


TRAINED_PREFIXES = ("net.smt_state_encoder.", "action_distribution_option.", "critic_option.", "uncertainty_option.")


class OptionAgent:
    """pi_q + its PPO optimiser state (ppo.py:31-303), restated."""

    def __init__(self, sd, pretraining, lr=2.5e-4, eps=1e-5, clip=0.2, epochs=2, mini_batches=2,
                 value_coef=0.5, entropy_coef=0.05, max_grad_norm=0.2, unct_coef=0.5, use_category_input=False):
        self.sd, self.pretraining, self.use_category_input = sd, pretraining, use_category_input
        self.lr, self.eps, self.clip, self.epochs, self.mb = lr, eps, clip, epochs, mini_batches
        self.vc, self.ec, self.gn, self.uc = value_coef, entropy_coef, max_grad_norm, unct_coef
        # only parameters that receive a gradient get Adam state (grad None -> skipped by torch.optim)
        self.trained = [k for k in sd if k.startswith(TRAINED_PREFIXES)]
        self.m = {k: torch.zeros_like(sd[k]) for k in self.trained}
        self.v = {k: torch.zeros_like(sd[k]) for k in self.trained}
        self.t = 0

    def forward(self, obs, prev_actions, em, em_masks, query_state, last_query_info):
        return R.option_net(self.sd, obs, prev_actions, em, em_masks, query_state, last_query_info,
                            pretraining=self.pretraining, use_category_input=self.use_category_input)

    def act(self, obs, prev_actions, em, em_masks, qs, lqi, generator=None):
        with torch.no_grad():
            feats, row = self.forward(obs, prev_actions, em, em_masks, qs, lqi)
            h = R.heads(self.sd, "option", feats, generator=generator)
        return h, row

    def value(self, obs, prev_actions, em, em_masks, qs, lqi):
        with torch.no_grad():
            feats, _ = self.forward(obs, prev_actions, em, em_masks, qs, lqi)
            return R._lin(self.sd, "critic_option.fc", feats)

    def update(self, st, gamma=0.99, tau=0.95):
        """_update_agent + PPO.update.  Returns the reference's 6-tuple."""
        s = st.step
        nv = self.value({k: v[s] for k, v in st.obs.items()}, st.prev_actions[s],
                        st.em_option.memory, st.em_masks[s], st.query_state[s - 1], st.last_query_info[s - 1])
        ret, vp = R.gae_returns(st.rewards, st.value_preds, st.masks, nv, gamma, tau, steps=s)
        st.returns.copy_(ret)
        st.value_preds.copy_(vp)
        adv = st.returns[:-1] - st.value_preds[:-1]
        acc = [0.0] * 6
        for _ in range(self.epochs):
            for b in st.minibatches(adv, self.mb):
                for k in self.trained:
                    self.sd[k].requires_grad_(True)
                    self.sd[k].grad = None
                feats, _ = self.forward(b["obs"], b["prev_actions"], b["em_option"], b["em_masks"],
                                        b["query_state"], b["last_query_info"])
                h = R.heads(self.sd, "option", feats, action=b["actions_option"])
                vl, al, ul, vm, rm = R.ppo_losses(h["value"], h["unct"], h["log_prob"], h["entropy"],
                                                  b["old_log_probs"], b["adv"], b["rl_masks"], b["value_preds"],
                                                  b["returns"], b["ucnt_gt"], self.clip)
                loss = R.total_loss(vl, al, h["entropy"], ul, self.vc, self.ec, self.uc)
                loss.backward()
                with torch.no_grad():
                    grads, _ = R.clip_grad_norm([self.sd[k].grad for k in self.trained], self.gn)
                    self.t += 1
                    for k, g in zip(self.trained, grads):
                        R.adam_step(self.sd[k], g, self.m[k], self.v[k], self.t, self.lr, self.eps)
                for k in self.trained:
                    self.sd[k].requires_grad_(False)
                for i, x in enumerate((vl, al, h["entropy"], vm, rm, ul)):
                    acc[i] += float(x.detach())
        n = self.epochs * self.mb
        st.after_update()
        return acc[0] / n, acc[1] / n, acc[2] / n, acc[3], acc[4], acc[5] / n


def sinus_pe(n, d=32):
    return R.sinusoid_table(n, d)


# --------------------------------------------------------------------------------------------------
# CPU baseline for bench.py ("port"): the three-policy rollout + pi_q update, reference call order
# --------------------------------------------------------------------------------------------------
def clip_text_spec(width=512, layers=12, ctx=77, vocab=49408, out=512, prefix="net.clip."):
    s = {prefix + "token_embedding.weight": (vocab, width), prefix + "positional_embedding": (ctx, width),
         prefix + "ln_final.weight": (width,), prefix + "ln_final.bias": (width,),
         prefix + "text_projection": (width, out)}
    for i in range(layers):
        b = f"{prefix}transformer.resblocks.{i}."
        s.update({b + "attn.in_proj_weight": (3 * width, width), b + "attn.in_proj_bias": (3 * width,),
                  b + "attn.out_proj.weight": (width, width), b + "attn.out_proj.bias": (width,),
                  b + "ln_1.weight": (width,), b + "ln_1.bias": (width,), b + "ln_2.weight": (width,),
                  b + "ln_2.bias": (width,), b + "mlp.c_fc.weight": (4 * width, width), b + "mlp.c_fc.bias": (4 * width,),
                  b + "mlp.c_proj.weight": (width, 4 * width), b + "mlp.c_proj.bias": (width,)})
    return s


def cpu_baseline(specs, N=16, T=4, spectrogram=(257, 101), pretraining=True, em_size=300, threads=None, mini_batches=2):
    """Time ONE bounded rollout(T steps, 3 policies incl. CLIP text) + pi_q update (2 epochs x 2 minibatches)
    of the oracle on the host cores.  Returns (env_steps_per_second, seconds, threads)."""
    import time
    import fixtures as fx
    if threads:
        torch.set_num_threads(threads)
    key = "option_257" if spectrogram[0] >= 30 and spectrogram[1] >= 30 else "option"
    mk = lambda spec: fx.state_dict_for({k: tuple(v) for k, v in spec.items()})
    sd_q = mk(specs[key])
    # pi_g / pi_l share the encoder shapes of the chosen spectrogram size: swap in the audio FC shape
    fcw = tuple(specs[key]["net.goal_encoder.cnn.6.weight"])
    conv_keys = [k for k in specs[key] if k.startswith("net.goal_encoder.")]
    def with_audio(base):
        s = dict(base)
        for k in conv_keys:
            s[k] = specs[key][k]
        return s
    sd_g = mk(with_audio(specs["goal"]))
    sl = with_audio(specs["dialog"])
    sl.update(clip_text_spec())
    sd_l = mk(sl)
    agent = OptionAgent(sd_q, pretraining=pretraining, mini_batches=mini_batches)
    obs0 = fx.observations("cpu.obs0", N, spectrogram)
    st = Storage(T, N, obs0, em_size, em_size // 2)
    pe = sinus_pe(1000)
    toks = fx.dialog_tokens("cpu.tok", N)
    zeros = torch.zeros
    t0 = time.perf_counter()
    for t in range(T):
        so = {k: v[st.step] for k, v in st.obs.items()}
        qs, lqi = pe[fx.ints(f"cpu.qc{t}", (N,), 4)], pe[fx.ints(f"cpu.lq{t}", (N,), 150)]
        h, row = agent.act(so, st.prev_actions[st.step], st.em_option.memory, st.em_masks[st.step], qs, lqi)
        with torch.no_grad():
            fg, xg = R.smt_net(sd_g, so, st.prev_actions[st.step], st.em.memory, st.em_masks[st.step])
            hg = R.heads(sd_g, "goal", fg)
            fl, xl = R.dialog_net(sd_l, so, st.prev_actions[st.step], st.em_vln.memory, st.em_vln_dialog.memory,
                                  st.em_vln_masks[st.step], toks, fx.ints(f"cpu.as{t}", (N,), 3).float())
            hl = R.heads(sd_l, "vln", fl)
        actions = torch.where(h["action"] == 1, hl["action"], hg["action"])
        nd = torch.from_numpy((fx.unit(f"cpu.nd{t}", N) >= 1 / 150).astype("float32")).view(N, 1)
        st.insert(fx.observations(f"cpu.obs{t + 1}", N, spectrogram, step=t + 1), actions, h["action"], h["log_prob"],
                  h["value"], fx.sym(f"cpu.r{t}", (N, 1)), nd, nd, xg, row, xl, fl, toks,
                  torch.ones(N, dtype=torch.long), fx.ints(f"cpu.ug{t}", (N,), 2), qs, lqi, zeros(N))
    agent.update(st)
    dt = time.perf_counter() - t0
    return N * T / dt, dt, torch.get_num_threads()


# --------------------------------------------------------------------------------------------------
# GRU baseline (BASELINE configs[1]): common/rollout_storage.py + av_nav/ppo/ppo.py restated
# --------------------------------------------------------------------------------------------------
class PlainStorage:
    """ss_baselines/common/rollout_storage.py:16-235."""

    def __init__(self, T, N, obs0, hidden=512):
        self.T, self.N, self.step = T, N, 0
        self.obs = {k: torch.zeros(T + 1, N, *v.shape[1:]) for k, v in obs0.items()}
        for k, v in obs0.items():
            self.obs[k][0].copy_(v)
        z = torch.zeros
        self.hidden = z(T + 1, 1, N, hidden)
        self.rewards, self.value_preds, self.returns = z(T, N, 1), z(T + 1, N, 1), z(T + 1, N, 1)
        self.action_log_probs = z(T, N, 1)
        self.actions, self.prev_actions = z(T, N, 1, dtype=torch.long), z(T + 1, N, 1, dtype=torch.long)
        self.masks = torch.ones(T + 1, N, 1)

    def insert(self, obs, hidden, actions, logp, values, rewards, masks):
        s = self.step
        for k in obs:
            self.obs[k][s + 1].copy_(obs[k])
        self.hidden[s + 1].copy_(hidden)
        self.actions[s].copy_(actions)
        self.prev_actions[s + 1].copy_(actions)
        self.action_log_probs[s].copy_(logp)
        self.value_preds[s].copy_(values)
        self.rewards[s].copy_(rewards)
        self.masks[s + 1].copy_(masks)
        self.step = (s + 1) % self.T

    def after_update(self):
        for k in self.obs:
            self.obs[k][0].copy_(self.obs[k][-1])
        self.hidden[0].copy_(self.hidden[-1])
        self.masks[0].copy_(self.masks[-1])
        self.prev_actions[0].copy_(self.prev_actions[-1])

    def compute_returns(self, next_value, use_gae, gamma, tau):
        if use_gae:
            ret, vp = R.gae_returns(self.rewards, self.value_preds, self.masks, next_value, gamma, tau)
            self.returns.copy_(ret)
            self.value_preds.copy_(vp)
        else:
            self.returns[-1] = next_value
            for t in reversed(range(self.T)):
                self.returns[t] = self.returns[t + 1] * gamma * self.masks[t + 1] + self.rewards[t]

    def minibatches(self, advantages, num_mini_batch):
        N, T = self.N, self.T
        per = N // num_mini_batch
        perm = torch.randperm(N)
        for start in range(0, N, per):
            ind = perm[start:start + per]
            fl = lambda x: x[:T, ind].reshape(T * len(ind), *x.shape[2:])
            yield {"obs": {k: fl(v) for k, v in self.obs.items()}, "h0": self.hidden[0][:, ind], "actions": fl(self.actions),
                   "value_preds": fl(self.value_preds), "returns": fl(self.returns), "masks": fl(self.masks),
                   "old_log_probs": fl(self.action_log_probs), "adv": fl(advantages)}


class BaselineAgent:
    """AudioNavBaselinePolicy + av_nav PPO (ppo.py:16-165), restated: every policy parameter that receives a gradient
    (net.* and the goal heads) is stepped by Adam after a global clip-norm."""

    PREFIXES = ("net.", "action_distribution_goal.", "critic_goal.")

    def __init__(self, sd, clip_param=0.2, ppo_epoch=4, num_mini_batch=2, value_loss_coef=0.5, entropy_coef=0.01, lr=7e-4,
                 eps=1e-5, max_grad_norm=0.5, use_normalized_advantage=False):
        self.sd = sd
        self.clip, self.epochs, self.mb, self.vc, self.ec = clip_param, ppo_epoch, num_mini_batch, value_loss_coef, entropy_coef
        self.lr, self.eps, self.gn, self.norm_adv = lr, eps, max_grad_norm, use_normalized_advantage
        self.trained = [k for k in sd if k.startswith(self.PREFIXES)]
        self.m = {k: torch.zeros_like(sd[k]) for k in self.trained}
        self.v = {k: torch.zeros_like(sd[k]) for k in self.trained}
        self.t = 0

    def act(self, obs, hidden, masks, generator=None):
        with torch.no_grad():
            x, h = R.baseline_net(self.sd, obs, hidden, masks)
            return R.heads(self.sd, "goal", x, generator=generator), h

    def value(self, obs, hidden, masks):
        with torch.no_grad():
            x, _ = R.baseline_net(self.sd, obs, hidden, masks)
            return R._lin(self.sd, "critic_goal.fc", x)

    def update(self, st):
        adv = st.returns[:-1] - st.value_preds[:-1]
        if self.norm_adv:
            adv = (adv - adv.mean()) / (adv.std() + 1e-5)
        acc = [0.0, 0.0, 0.0]
        for _ in range(self.epochs):
            for b in st.minibatches(adv, self.mb):
                for k in self.trained:
                    self.sd[k].requires_grad_(True)
                    self.sd[k].grad = None
                x, _ = R.baseline_net(self.sd, b["obs"], b["h0"], b["masks"])
                h = R.heads(self.sd, "goal", x, action=b["actions"])
                ratio = torch.exp(h["log_prob"] - b["old_log_probs"])
                surr1 = ratio * b["adv"]
                surr2 = torch.clamp(ratio, 1.0 - self.clip, 1.0 + self.clip) * b["adv"]
                action_loss = -torch.min(surr1, surr2).mean()
                vpc = b["value_preds"] + (h["value"] - b["value_preds"]).clamp(-self.clip, self.clip)
                value_loss = 0.5 * torch.max((h["value"] - b["returns"]).pow(2), (vpc - b["returns"]).pow(2)).mean()
                (value_loss * self.vc + action_loss - h["entropy"] * self.ec).backward()
                with torch.no_grad():
                    grads, _ = R.clip_grad_norm([self.sd[k].grad for k in self.trained], self.gn)
                    self.t += 1
                    for k, g in zip(self.trained, grads):
                        R.adam_step(self.sd[k], g, self.m[k], self.v[k], self.t, self.lr, self.eps)
                for k in self.trained:
                    self.sd[k].requires_grad_(False)
                for i, x_ in enumerate((value_loss, action_loss, h["entropy"])):
                    acc[i] += float(x_.detach())
        n = self.epochs * self.mb
        return acc[0] / n, acc[1] / n, acc[2] / n
