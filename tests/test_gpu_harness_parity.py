"""The BENCHED configuration of avlen_amd.harness.Workload (what bench.py times: bf16, HIP graphs, shared grouped towers,
launch-ahead on side streams incl. the CLIP text tower as its own graph and pi_l's split graph, fresh storage views every step,
257x101 spectrogram) against the CPU oracle (oracle/restate.py + oracle/flow.py) on the same weights and inputs, step by step:

* every step's pi_q / pi_g / pi_l values, action probabilities and memory rows vs the oracle evaluated on the SAME state;
* the sampled actions are exactly what the host generator yields for the product's probabilities in the reference's draw order
  (pi_q, pi_g, pi_l), and the generator ends in the same state;
* the rollout storage (rings, masks, values, actions, observations) equals the oracle's storage fed with the same outputs, bit for bit;
* GAE returns, the PPO.update 6-tuple and the parameter step vs the oracle's update.

fp32 parity mode and the bf16x3 mode run the same harness (graphs on) to 1e-3; bf16 tolerances are the measured bf16 envelope,
stated per assertion.
A wiring mistake between the captured graphs (stale static buffer, wrong stream order, a memset node firing at the wrong point)
produces O(1) differences here, not 1e-2 ones."""
import numpy as np
import pytest
import torch

import restate as R
import flow
from avlen_amd.harness import Workload

pytestmark = pytest.mark.gpu


def _cpu(x):
    if isinstance(x, dict):
        return {k: _cpu(v) for k, v in x.items()}
    return x.detach().cpu()


def _err(a, b):
    return float((a.float().cpu() - b.float()).abs().max())


# "bf16x3" = the accurate fast mode (compensated bf16 towers / state encoders, fp16 CLIP text tower and AudioCNN): held to the
# SAME 1e-3 tolerances as the fp32 parity mode (north_star: "within 1e-3 on logits/values")
CASES = [("bf16", True, False), ("bf16", False, False), ("fp32", True, False), ("fp32", False, False), ("bf16", True, True),
         ("fp32", False, True), ("bf16x3", True, False), ("bf16x3", False, False), ("bf16x3", True, True)]


@pytest.mark.parametrize("precision,pre,distractor", CASES)
def test_benched_harness_matches_oracle(precision, pre, distractor):
    _harness_case(precision, pre, distractor)


def test_benched_harness_mixed_backward_update_matches_oracle():
    """bf16x3 at scale runs the SMT BACKWARD on plain bf16 operands (from `avlen_set_x3_mixed_backward_rows` token rows on, 64 k by
    default: every real 2nd-stage minibatch; the forward -- logits, ratio, losses -- stays compensated).  Forced here from the
    first row, on the large-M GEMM route it belongs to: the same rollout + PPO.update against the oracle; the parameter step of the
    whole update (2 epochs x 2 minibatches through Adam) stays within 3e-2 relative L2 of the oracle's (measured 1.0e-2; the
    compensated backward: 2.4e-3 measured, 1e-2 asserted above)."""
    from avlen_amd import _lib as L
    try:
        L.lib.avlen_set_big_m(32)
        L.lib.avlen_set_x3_mixed_backward_rows(1)
        _harness_case("bf16x3", False, False, step_tol=3e-2)
    finally:
        L.lib.avlen_set_big_m(0)
        L.lib.avlen_set_x3_mixed_backward_rows(-1)


def _harness_case(precision, pre, distractor, step_tol=None):
    N, T, CAP = 4, 5, 3
    bf = precision == "bf16"
    wl = Workload(N, T, spectrogram=(257, 101, 2), precision=precision, pretraining=pre, em_capacity=CAP, seed=3,
                  use_graphs=True, share_encoders=True, launch_ahead=True, cached_views=False, distractor=distractor)
    sd_q, sd_g, sd_l = (_cpu(p.state_dict()) for p in (wl.pi_q, wl.pi_g, wl.pi_l))
    ro = wl.rollouts
    dq, dg = wl.pi_q.net.memory_dim, wl.pi_g.net.memory_dim
    assert (dq, dg) == ((329, 297) if distractor else (308, 276))
    obs0 = {k: _cpu(v[0]) for k, v in ro.observations.items()}
    st = flow.Storage(T, N, obs0, CAP + T, CAP, dim_goal=dg, dim_option=dq, dim_vln=276)
    agent = flow.OptionAgent(sd_q, pretraining=pre, use_category_input=distractor)
    # tolerances: fp32 parity mode = north_star's 1e-3; bf16 = measured envelope of bf16 operands through the 20-conv towers
    tv, tp, tr = (6e-2, 2e-2, 6e-2) if bf else (1e-3, 1e-3, 1e-3)
    worst = dict(v=0.0, p=0.0, row=0.0)
    torch.manual_seed(99)
    for t in range(T):
        so = {k: v[t] for k, v in st.obs.items()}
        pa = st.prev_actions[t]
        qs, lqi = _cpu(wl.query_state[t]), _cpu(wl.last_query_info[t])
        toks, astep = _cpu(wl.dialog[t]), _cpu(wl.agent_step[t])
        with torch.no_grad():
            fq, row_q = agent.forward(so, pa, st.em_option.memory, st.em_masks[t], qs, lqi)
            hq = R.heads(sd_q, "option", fq, deterministic=True)
            fg, row_g = R.smt_net(sd_g, so, pa, st.em.memory, st.em_masks[t], use_category_input=distractor)
            hg = R.heads(sd_g, "goal", fg, deterministic=True)
            fl, row_l = R.dialog_net(sd_l, so, pa, st.em_vln.memory, st.em_vln_dialog.memory, st.em_vln_masks[t], toks, astep)
            hl = R.heads(sd_l, "vln", fl, deterministic=True)
        rng0 = torch.get_rng_state()
        o = wl.rollout_step(return_outs=True)
        torch.cuda.synchronize()
        rng1 = torch.get_rng_state()
        for name, h, row, rown in (("q", hq, row_q, "row_q"), ("g", hg, row_g, "row_g"), ("l", hl, row_l, "row_l")):
            ev, ep = _err(o[name + "_value"], h["value"]), _err(o[name + "_prob"], h["probs"])
            er = _err(o[rown], row) / float(row.abs().max())
            worst.update(v=max(worst["v"], ev), p=max(worst["p"], ep), row=max(worst["row"], er))
            assert ev < tv * max(1.0, float(h["value"].abs().max())), (t, name, "value", ev)
            assert ep < tp, (t, name, "prob", ep)
            assert er < tr, (t, name, "memory row", er)
        ed = _err(o["row_d"], fl) / float(fl.abs().max())
        assert ed < (8e-2 if bf else 1e-3), (t, "dialog state row", ed)
        # sampling: the reference's draws (policy.py:86-123 -> torch.multinomial on the host generator), in its order, on the
        # probabilities the product computed
        torch.set_rng_state(rng0)
        for name in ("q", "g", "l"):
            a = R.sample_host(o[name + "_prob"].cpu())
            assert torch.equal(a.view(-1), o["a_" + name].cpu().view(-1)), (t, name)
        assert torch.equal(torch.get_rng_state(), rng1)
        actions = torch.where(o["a_q"] == 1, o["a_l"], o["a_g"])
        assert torch.equal(actions.cpu(), o["actions"].cpu())
        # the oracle's storage, fed with the product's outputs, must equal the product's storage bit for bit
        nxt = {k: _cpu(wl.sim[k][t + 1]) for k in st.obs}
        nd = _cpu(wl.not_done[t])
        st.insert(nxt, _cpu(o["actions"]), _cpu(o["a_q"]), _cpu(o["lp_q"]), _cpu(o["q_value"]), _cpu(wl.rewards[t]), nd, nd,
                  _cpu(o["row_g"]), _cpu(o["row_q"]), _cpu(o["row_l"]), _cpu(o["row_d"]), toks, _cpu(wl.rl_masks[t]),
                  _cpu(wl.ucnt_gt[t]), qs, lqi, astep)
        for mine, ref in ((ro.em.memory, st.em.memory), (ro.em_option.memory, st.em_option.memory),
                          (ro.em_vln.memory, st.em_vln.memory), (ro.em_vln_dialog.memory, st.em_vln_dialog.memory),
                          (ro.em_masks, st.em_masks), (ro.em_vln_masks, st.em_vln_masks), (ro.value_preds, st.value_preds),
                          (ro.prev_actions, st.prev_actions), (ro.actions_option, st.actions_option), (ro.masks, st.masks),
                          (ro.action_log_probs, st.action_log_probs), (ro.rl_masks, st.rl_masks),
                          (ro.observations["spectrogram"], st.obs["spectrogram"]), (ro.observations["rgb"], st.obs["rgb"])):
            assert torch.equal(mine.cpu().to(ref.dtype), ref), (t, tuple(ref.shape))
    print(f"{precision} pre={pre} distractor={distractor}: worst |value| {worst['v']:.3g} |prob| {worst['p']:.3g} "
          f"row (rel) {worst['row']:.3g}")
    # update: same host RNG draw order for the minibatch permutations
    sd0 = {k: v.clone() for k, v in sd_q.items() if k.startswith(flow.TRAINED_PREFIXES)}
    torch.manual_seed(123)
    ours = wl.update()
    torch.cuda.synchronize()
    mine_ret = ro.returns.cpu()
    torch.manual_seed(123)
    ref = agent.update(st)
    np.testing.assert_allclose(mine_ret[:T].numpy(), st.returns[:T].numpy(), rtol=5e-2 if bf else 1e-3, atol=5e-2 if bf else 1e-3)
    # (value_loss, action_loss, entropy, values_debug, return_debug, unct_loss); [3] and [4] are SUMS over the updates of small
    # cancelling means (ppo.py:282-289): absolute tolerance only
    ours, ref = np.array(ours), np.array(ref)
    np.testing.assert_allclose(ours[[0, 1, 2, 5]], ref[[0, 1, 2, 5]], rtol=8e-2 if bf else 2e-3, atol=2e-2 if bf else 2e-4)
    np.testing.assert_allclose(ours[[3, 4]], ref[[3, 4]], atol=1e-1 if bf else 2e-3)
    new = _cpu(wl.pi_q.state_dict())
    num = den = 0.0
    for k in sd0:
        d_ref, d_ours = (sd_q[k].detach() - sd0[k]).double(), (new[k] - sd0[k]).double()
        num += float(((d_ours - d_ref) ** 2).sum())
        den += float((d_ref ** 2).sum())
    rel = (num / den) ** 0.5
    print(f"parameter step: relative L2 difference {rel:.3g} over {len(sd0)} trained tensors")
    assert den > 0 and rel < (step_tol if step_tol is not None else (0.35 if bf else (1e-2 if precision == "bf16x3" else 2e-2))), rel
    # encoders are not touched by the update (policy.py:1035-1036)
    k = "net.visual_encoder.rgb_encoder.conv1.weight"
    assert torch.equal(new[k], sd_q[k])


def test_fresh_dicts_and_tensors_every_call_with_graphs():
    """a caller that builds a fresh observation dict AND fresh tensors for every call (the reference's eval loop:
    batch_obs -> new tensors each step) while passing persistent memory objects must never replay on stale observations.  The
    launch memo is keyed on buffer addresses, so a recycled dict id cannot alias; a recycled ADDRESS is harmless because the staging
    copy reads the current contents."""
    wl = Workload(3, 2, spectrogram=(65, 26, 2), precision="fp32", pretraining=False, em_capacity=2, seed=1, use_graphs=True,
                  share_encoders=False, launch_ahead=False, with_goal_policy=False, with_dialog_policy=False)
    pol, ro = wl.pi_q, wl.rollouts
    eager = Workload(3, 2, spectrogram=(65, 26, 2), precision="fp32", pretraining=False, em_capacity=2, seed=1, use_graphs=False,
                     share_encoders=False, launch_ahead=False, with_goal_policy=False, with_dialog_policy=False).pi_q
    prev, mem, mk = ro.prev_actions[0], ro.external_memory_option[:, 0], ro.external_memory_masks[0]
    qs, lqi = wl.query_state[0], wl.last_query_info[0]
    outs = []
    for i in range(6):                               # A, B, A, ... : dicts and tensors are freed between calls
        src = {k: wl.sim[k][i % 3] for k in ro.observations}
        obs = {k: v.clone() for k, v in src.items()}
        v = pol.act_option(obs, None, prev, None, mem, mk, qs, lqi, deterministic=True)[0].clone()
        ref = eager.act_option({k: v_.clone() for k, v_ in src.items()}, None, prev, None, mem, mk, qs, lqi,
                               deterministic=True)[0]
        assert torch.equal(v, ref), i
        outs.append(v)
        del obs
    assert not torch.equal(outs[0], outs[1])
    # load_state_dict on a policy that already holds captured graphs: the next replay must use the new weights
    sd = {k: v.clone() for k, v in pol.state_dict().items()}
    sd["critic_option.fc.bias"] += 1.0
    sd["net.visual_encoder.rgb_encoder.conv1.weight"] *= 0.5
    pol.load_state_dict(sd)
    eager.load_state_dict(sd)
    obs = {k: wl.sim[k][0].clone() for k in ro.observations}
    v_new = pol.get_value_option(obs, None, prev, None, mem, mk, qs, lqi).clone()
    v_ref = eager.get_value_option(obs, None, prev, None, mem, mk, qs, lqi)
    assert torch.equal(v_new, v_ref) and not torch.allclose(v_new, outs[0])


@pytest.mark.parametrize("native", [False, True])
def test_ddppo_update_through_rccl_one_rank(native):
    """A real DDPPO.update with an initialised `nccl` (= RCCL) process group of one rank: init_distributed broadcasts the flat
    parameter buffer, every optimiser step all-reduces the flat gradient ON THE DEVICE.  With one rank the reduction is the
    identity, so the result must equal the same update without a process group.  native: the reduction goes through the C ABI's own
    RCCL binding (avlen_comm_unique_id / avlen_comm_init_rank / avlen_grad_allreduce) instead of torch.distributed.all_reduce."""
    import os
    import socket
    import torch.distributed as dist
    from avlen_amd.ppo import DDPPO as _D
    keep = _D.native_allreduce
    _D.native_allreduce = native
    try:
        _rccl_one_rank(native, os, socket, dist)
    finally:
        _D.native_allreduce = keep


def _rccl_one_rank(native, os, socket, dist):
    outs = []
    for with_group in (False, True):
        if with_group:
            s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
            os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
            dist.init_process_group("nccl", rank=0, world_size=1)
        try:
            wl = Workload(4, 3, spectrogram=(65, 26, 2), precision="fp32", pretraining=False, em_capacity=3, seed=2, use_graphs=False,
                          share_encoders=False, launch_ahead=False, with_goal_policy=False, with_dialog_policy=False)
            assert wl.agent._distributed == with_group
            assert (wl.agent._comm is not None) == (with_group and native)
            torch.manual_seed(5)
            for _ in range(3):
                wl.rollout_step()
            out = wl.update()
            torch.cuda.synchronize()
            outs.append((out, wl.pi_q.state_dict()["net.smt_state_encoder.fusion_encoder.2.weight"].clone()))
        finally:
            if with_group:
                dist.destroy_process_group()
    # identity reduction; the loss log itself is accumulated with float atomics (1e-7 run-to-run)
    np.testing.assert_allclose(np.array(outs[0][0]), np.array(outs[1][0]), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(outs[0][1].cpu().numpy(), outs[1][1].cpu().numpy(), rtol=1e-5, atol=1e-7)


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_uint8_rgb_end_to_end_is_bit_identical(precision):
    """RGB kept uint8 from the sensor through the rollout storage into the tower prologue.  The features computed from
    uint8 pixels equal the ones computed from the same pixels stored as fp32 BIT FOR BIT (fp32 mode; bf16 mode up to the atomic
    order of the fused GroupNorm statistics), `insert` takes pinned-host uint8 frames, and the PPO minibatch reads the uint8 rows of
    the storage in place."""
    from avlen_amd import policy as P
    wl = Workload(4, 3, spectrogram=(65, 26, 2), precision=precision, pretraining=True, em_capacity=3, seed=9, use_graphs=False,
                  share_encoders=False, launch_ahead=False, with_goal_policy=False, with_dialog_policy=False)
    ro, pol = wl.rollouts, wl.pi_q
    assert ro.observations["rgb"].dtype == torch.uint8 and "audiogoal" not in ro.observations
    obs8 = {k: v[0] for k, v in ro.observations.items()}
    obs32 = dict(obs8, rgb=obs8["rgb"].float())
    pa = ro.prev_actions[0]
    f8, _ = pol.net.features(pol, obs8, pa)
    f8 = f8[:, :276].clone()             # [visual 128 | action 16 | audio 128 | pose 4]; the caller fills the columns behind (extra=None here)
    f32, _ = pol.net.features(pol, obs32, pa)
    f32 = f32[:, :276]
    torch.cuda.synchronize()
    if precision == "fp32":
        assert torch.equal(f8, f32)
    else:
        assert float((f8 - f32).abs().max()) < 3e-2 * float(f32.abs().max())
    assert float(f8[:, :64].abs().max()) > 0
    # insert from a pinned host uint8 frame (asynchronous H2D) and from an integer-valued fp32 device frame
    for _ in range(3):
        wl.rollout_step()
    host = torch.randint(0, 256, (4, 128, 128, 3), dtype=torch.uint8).pin_memory()
    nxt = {k: wl.sim[k][1] for k in ro.observations}
    nxt["rgb"] = host
    z = torch.zeros
    ro.step = 0
    args = (z(1, 4, 512, device="cuda"), z(4, 1, dtype=torch.long, device="cuda"), z(4, 1, dtype=torch.long, device="cuda"),
            z(4, 1, device="cuda"), z(4, 1, device="cuda"), z(4, 1, device="cuda"), torch.ones(4, 1, device="cuda"),
            torch.ones(4, 1, device="cuda"), z(4, 276, device="cuda"), z(4, 308, device="cuda"), z(4, 276, device="cuda"),
            z(4, 256, device="cuda"), z(4, 77, dtype=torch.long, device="cuda"), z(4, device="cuda"),
            torch.ones(4, dtype=torch.long, device="cuda"), torch.ones(4, dtype=torch.long, device="cuda"),
            z(4, dtype=torch.long, device="cuda"), z(4, 4, device="cuda"), z(4, 32, device="cuda"), z(4, 32, device="cuda"),
            z(4, device="cuda"))
    ro.insert(nxt, *args)
    torch.cuda.synchronize()
    assert torch.equal(ro.observations["rgb"][1].cpu(), host)
    ro.step = 0
    nxt["rgb"] = host.cuda().float()
    ro.insert(nxt, *args)
    torch.cuda.synchronize()
    assert torch.equal(ro.observations["rgb"][1].cpu(), host)
    # minibatch rows read in place from the uint8 storage
    ro.step = 3
    b = ro.gather_minibatch(torch.tensor([1, 3], device="cuda"), in_place=True)
    if precision == "bf16":
        assert isinstance(b["obs"]["rgb"], P.RowsOf) and b["obs"]["rgb"].base.dtype == torch.uint8
    b0 = ro.gather_minibatch(torch.tensor([1, 3], device="cuda"), in_place=False)
    assert b0["obs"]["rgb"].dtype == torch.uint8
    assert torch.equal(b0["obs"]["rgb"], ro.observations["rgb"][:3][:, [1, 3]].reshape(6, 128, 128, 3))
