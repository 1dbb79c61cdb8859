"""The per-row memo of the frozen CLIP text tower (avlen_clip_text_cached_fwd; AudioNavDialogNet.encode_text_cached) and the
reference's dialog data flow (ss_baselines/savi/ppo/ppo_trainer.py:347, 449-625; policy.py:844-851):

* `current_dialog` is all-zero for an env without a query, holds the tokenised instruction for NUM_DIALOG_STEPS = 3 steps after a
  query, then is zero again -- the memo must return, bit for bit, what the uncached tower returns over such a scenario while running
  the 12 blocks only for the rows that changed;
* the tokens exist only after `act_option` returned: the harness' default ("after_option": pi_l's dialog half issued by
  `dialog_ready()`) must produce exactly what the tokens-known-ahead ordering produces.
"""
import numpy as np
import pytest
import torch

from avlen_amd import policy as P
from avlen_amd.spaces import savi_observation_space, ActionSpace, SMT_KW

pytestmark = pytest.mark.gpu


def _dialog(gen, ln):
    t = torch.zeros(77, dtype=torch.long)
    t[:ln] = torch.randint(1, 49406, (ln,), generator=gen)
    t[0] = 49406
    t[ln] = 49407
    return t


def _policy(mode):
    torch.manual_seed(3)
    pol = P.AudioNavDialogPolicy(savi_observation_space((65, 26, 2)), ActionSpace(4), pretraining=False, num_steps=3,
                                 precision=mode, **SMT_KW).to("cuda")
    pol._engine()
    return pol


def _tower_rows(net, B, project):
    """rows the last cached call sent through the 12 blocks (header word 1 of the state block)."""
    st = net._text_states[(B, project)]
    return int(st[:12].view(torch.int32)[1])


@pytest.mark.parametrize("mode", ["bf16x3", "bf16"])
def test_memo_is_bit_equal_to_the_uncached_tower_over_a_query_scenario(mode):
    """query -> 3 steps -> pause, two envs out of phase, a repeated instruction, an env whose dialog is replaced mid-way."""
    pol = _policy(mode)
    net = pol.net
    assert net.text_cache and pol._engine()["clip"].wstream
    gen = torch.Generator().manual_seed(7)
    B = 6
    d = [_dialog(gen, ln) for ln in (9, 33, 70, 17, 48)]
    Z = torch.zeros(77, dtype=torch.long)
    #            env0  env1  env2  env3  env4  env5        expected rows through the tower
    steps = [([Z, Z, Z, Z, Z, Z], 1),                      # first call: only the shared all-zero row
             ([d[0], Z, Z, Z, Z, Z], 1),                   # env 0 queries
             ([d[0], Z, d[1], Z, Z, Z], 1),                # env 2 queries one step later
             ([d[0], Z, d[1], Z, Z, Z], 0),                # nothing changed: no tower work at all
             ([Z, Z, d[1], Z, Z, d[2]], 1),                # env 0's dialog is over (zero row: no tower), env 5 queries (5 row tiles)
             ([Z, d[0], Z, Z, Z, d[2]], 1),                # env 1 gets the instruction env 0 had: its row changed -> recomputed
             ([d[3], d[0], Z, d[4], Z, d[3]], 3),          # three new rows at once (two of them the same instruction)
             ([Z, Z, Z, Z, Z, Z], 0)]                      # pause: every row back to the shared zero embedding
    for project in (True, False):
        if not project and "clip_noproj" not in pol._engine():
            continue
        net.invalidate_text_cache()
        for i, (rows, want) in enumerate(steps):
            tok = torch.stack(rows).cuda()
            got = net.encode_text_cached(pol, tok, project=project).clone()
            ref = net.encode_text(pol, tok, project=project).clone()
            torch.cuda.synchronize()
            assert torch.isfinite(got).all()
            assert torch.equal(got, ref), (mode, project, i, float((got - ref).abs().max()))
            assert _tower_rows(net, B, project) == want, (mode, project, i, _tower_rows(net, B, project), want)
    # a weight change empties the memo: every non-zero row + the zero row run again, and the result follows the new weights
    tok = torch.stack([d[0], Z, d[1], Z, Z, d[2]]).cuda()
    a = net.encode_text_cached(pol, tok).clone()
    with torch.no_grad():
        pol.net.clip.ln_final.weight.mul_(1.5)
        pol.net.clip.transformer.resblocks[3].mlp.c_fc.bias.add_(0.05)
    pol.mark_params_changed()
    b = net.encode_text_cached(pol, tok).clone()
    ref = net.encode_text(pol, tok).clone()
    torch.cuda.synchronize()
    assert _tower_rows(net, B, True) == 4
    assert torch.equal(b, ref) and not torch.equal(a, b)


def _storage_snapshot(wl):
    wl._join_small()                                        # the last step's storage writes ran on the harness' side stream
    ro = wl.rollouts
    keys = ("value_preds", "actions", "actions_option", "action_log_probs", "action_probs", "all_dialog", "agent_step")
    snap = {k: getattr(ro, k).clone() for k in keys}
    snap["em_goal"] = ro.em.memory.clone()
    snap["em_option"] = ro.em_option.memory.clone()
    snap["em_vln"] = ro.em_vln.memory.clone()
    snap["em_vln_dialog"] = ro.em_vln_dialog.memory.clone()
    return snap


def _run(N, T, **kw):
    from avlen_amd.harness import Workload
    torch.manual_seed(99)                                   # host RNG of the action sampling
    wl = Workload(N, T, spectrogram=(65, 26, 2), em_capacity=T, **kw)
    return wl


@pytest.mark.parametrize("mode", ["bf16x3", "fp32"])
def test_after_option_ordering_equals_tokens_ahead(mode):
    """Same arithmetic, different issue order: the storage after a rollout is bit-equal (the update's loss scalars are sums whose
    last bits depend on the atomics' arrival order in fp32 mode: compared to 1e-6)."""
    N, T = 4, 6
    outs = []
    for order in ("ahead", "after_option"):
        wl = _run(N, T, precision=mode, dialog_tokens=order, share_encoders=(mode != "fp32"))
        for _ in range(T):
            wl.rollout_step()
        snap = _storage_snapshot(wl)
        losses = [float(x) for x in wl.update()]
        torch.cuda.synchronize()
        outs.append((snap, losses))
        del wl
    (s0, l0), (s1, l1) = outs
    for k in s0:
        assert torch.equal(s0[k], s1[k]), (mode, k)
    assert np.allclose(l0, l1, rtol=1e-6, atol=1e-7), (l0, l1)


def test_reference_dialog_process_memo_on_equals_memo_off():
    """The trainer's dialog process (tokens persist 3 steps after a_q == 1, zeros otherwise) through the whole harness: the stored
    rollout with the memo equals the one without it, and the memo did skip most of the tower work."""
    N, T = 8, 12
    outs = []
    for cache in (True, False):
        wl = _run(N, T, precision="bf16x3", dialog_process="reference")
        wl.pi_l.net.text_cache = cache
        for _ in range(T):
            wl.rollout_step()
        snap = _storage_snapshot(wl)
        torch.cuda.synchronize()
        outs.append((snap, dict(wl.dialog_stats)))
        del wl
    (s0, st0), (s1, st1) = outs
    assert st0 == st1 and st0["steps"] == T and st0["new_dialogs"] > 0
    # a dialog lasts 3 steps: active rows ~ 3 x new dialogs (fewer at the end of the rollout / at episode ends)
    assert st0["new_dialogs"] <= st0["active_rows"] <= 3 * st0["new_dialogs"]
    for k in s0:
        assert torch.equal(s0[k], s1[k]), k


def test_memo_at_the_benched_batch_is_independent_of_how_many_rows_changed():
    """64 dialogs of the benched length mix: a call in which 5 rows changed reproduces, bit for bit, what the full uncached batch
    gives -- the tower's column split must not depend on the number of dialogs it is asked to compute."""
    pol = _policy("bf16x3")
    net = pol.net
    gen = torch.Generator().manual_seed(9)
    lens = [int(x) for x in torch.randint(2, 73, (64,), generator=gen)]
    tok = torch.stack([_dialog(gen, ln) for ln in lens]).cuda()
    net.invalidate_text_cache()
    first = net.encode_text_cached(pol, tok).clone()
    tok2 = tok.clone()
    for r, ln in ((3, 70), (17, 5), (40, 33), (41, 64), (63, 18)):
        tok2[r] = _dialog(gen, ln).cuda()
    got = net.encode_text_cached(pol, tok2).clone()
    torch.cuda.synchronize()
    assert _tower_rows(net, 64, True) == 5
    ref = net.encode_text(pol, tok2).clone()
    assert torch.equal(first, net.encode_text(pol, tok).clone()) and torch.equal(got, ref)


def test_automatic_launch_ahead_equals_explicit_and_plain_calls():
    """share_encoders alone (no prefetch_* calls in the trainer): the leader's act_option enqueues the followers' forwards itself on
    predicted arguments, the followers' calls validate the addresses and pick the results up.  Same storage, bit for bit, as the
    explicit launch-ahead flow and as the fully serial flow (AVLEN_AUTO_AHEAD off), incl. the rollout wrap-around where the guess
    is wrong; and the guesses are right on all other steps."""
    N, T = 4, 5
    outs = {}
    for name, kw, auto in (("explicit", dict(launch_ahead=True), True), ("auto", dict(launch_ahead=False), True),
                           ("serial", dict(launch_ahead=False), False)):
        wl = _run(N, T, precision="bf16x3", **kw)
        wl.pi_q._enc_group.auto = auto
        for _ in range(T):
            wl.rollout_step()
        first = _storage_snapshot(wl)
        wl.rollouts.after_update()                          # (no optimiser step: its loss sums are not bit-reproducible run to run)
        for _ in range(3):                                  # past the wrap-around: slot 0 again after after_update
            wl.rollout_step()
        torch.cuda.synchronize()
        grp = wl.pi_q._enc_group
        outs[name] = (first, _storage_snapshot(wl), grp.auto_hits, grp.auto_misses, torch.get_rng_state())
        del wl
    for name in ("auto", "serial"):
        for part in (0, 1):
            for k in outs["explicit"][part]:
                a_, b_ = outs["explicit"][part][k], outs[name][part][k]
                if not torch.equal(a_, b_):
                    d_ = (a_.double() - b_.double()).abs().reshape(a_.shape[0], -1).max(1).values
                    raise AssertionError((name, part, k, [(i, float(x)) for i, x in enumerate(d_) if x > 0][:12]))
        assert torch.equal(outs["explicit"][4], outs[name][4]), name          # the host generator ends in the same state
    assert outs["serial"][2] == 0 and outs["explicit"][2] == 0
    # two followers x (T + 3) steps; no guess on the first two steps (no history), wrong or absent around the wrap-around
    assert outs["auto"][2] >= 2 * (T + 3) - 8 and outs["auto"][3] <= 2, outs["auto"][2:4]


def test_encoders_started_before_insert_equal_the_plain_order():
    """`Policy.prefetch_encoders(new_obs)` right before `rollouts.insert(new_obs, ...)` (the towers hide the storage bookkeeping and
    the next step's launch path) against the plain order: same storage bit for bit over a rollout, the wrap-around and the update's
    `get_value_option`."""
    N, T = 4, 5
    snaps = []
    if True:
        for early in (True, False):
            wl = _run(N, T, precision="bf16x3")
            assert wl._early_enc
            wl._early_enc = early
            for _ in range(T):
                wl.rollout_step()
            ro = wl.rollouts
            wl._join_small()
            last = {k: v[ro.step] for k, v in ro.observations.items()}
            nv = wl.pi_q.get_value_option(last, ro.recurrent_hidden_states[ro.step], ro.prev_actions[ro.step], ro.masks[ro.step],
                                          ro.external_memory_option[:, ro.step], ro.external_memory_masks[ro.step],
                                          ro.query_state[ro.step - 1], ro.last_query_info[ro.step - 1]).clone()
            ro.after_update()
            for _ in range(2):
                wl.rollout_step()
            torch.cuda.synchronize()
            snaps.append((_storage_snapshot(wl), nv))
            del wl
    (a, va), (b, vb) = snaps
    assert torch.equal(va, vb)
    for k in a:
        assert torch.equal(a[k], b[k]), k


def test_actions_selected_on_the_host_equal_the_device_select():
    """sampling="race": the step's actions for the simulator are selected on the host from the three policies' pinned copies
    (`Policy.host_actions`); the device-side select feeds the storage later.  Both must be the device select of the plain order,
    step by step, over a rollout, the wrap-around and beyond."""
    N, T = 4, 5
    snaps = []
    if True:
        for sel in (True, False):
            wl = _run(N, T, precision="bf16x3")
            assert wl._host_select and wl.sampling == "race"
            wl._host_select = sel
            host = []
            for i in range(T + 2):
                if i == T:
                    wl.rollouts.after_update()
                t = wl.rollouts.step
                wl.rollout_step()
                host.append(wl._act_host[t & 3].clone())
                torch.cuda.synchronize()
                assert torch.equal(host[-1], wl.rollouts.actions[t].cpu()), (sel, i)       # what the simulator got == what was stored
            snaps.append((_storage_snapshot(wl), host))
            del wl
    (a, ha), (b, hb) = snaps
    for x, y in zip(ha, hb):
        assert torch.equal(x, y)
    for k in a:
        assert torch.equal(a[k], b[k]), k


def test_storage_writes_on_the_side_stream_equal_the_plain_order():
    """The step's storage writes (device-side action select + insert) on a side stream beside the next step's towers against the
    same writes on the caller's stream: same storage bit for bit over a rollout, the update's value call, the wrap-around."""
    N, T = 4, 5
    snaps = []
    if True:
        for side in (True, False):
            wl = _run(N, T, precision="bf16x3")
            assert wl._small is not None
            if not side:
                wl._small = None
            for _ in range(T):
                wl.rollout_step()
            assert wl._small_pending == side
            wl._join_small()
            ro = wl.rollouts
            wl._join_small()
            last = {k: v[ro.step] for k, v in ro.observations.items()}
            nv = wl.pi_q.get_value_option(last, ro.recurrent_hidden_states[ro.step], ro.prev_actions[ro.step], ro.masks[ro.step],
                                          ro.external_memory_option[:, ro.step], ro.external_memory_masks[ro.step],
                                          ro.query_state[ro.step - 1], ro.last_query_info[ro.step - 1]).clone()
            ro.after_update()
            for _ in range(3):
                wl.rollout_step()
            torch.cuda.synchronize()
            snaps.append((_storage_snapshot(wl), nv))
            del wl
    (a, va), (b, vb) = snaps
    assert torch.equal(va, vb)
    for k in a:
        assert torch.equal(a[k], b[k]), k


@pytest.mark.parametrize("process", ["fresh", "reference"])
def test_sequencer_command_lists_equal_the_python_launch_ahead_flow(process):
    """avlen_amd/sequencer.py: after one pass through `prefetch_act_option` / `prefetch_act` / `prefetch_act_dialog` / `dialog_ready`
    per set of argument buffers, a step's launches are two recorded command lists run by `avlen_cmds_run`.  Scheduling only: the
    storage after two rollouts (the second one entirely on the recorded lists), the wrap-around in between, the sampled actions
    and the host generator's end state equal the Python flow's bit for bit."""
    N, T = 4, 5
    snaps = []
    for use_seq in (True, False):
        wl = _run(N, T, precision="bf16x3", dialog_process=process)       # reference: tokens written by the host loop after a_q
        assert wl.seq is not None
        seq = wl.seq
        if not use_seq:
            wl.seq = None
        for i in range(2 * T + 2):
            if i in (T, 2 * T):
                wl._join_small()
                wl.rollouts.after_update()
            wl.rollout_step()
        wl._join_small()
        torch.cuda.synchronize()
        snaps.append((_storage_snapshot(wl), torch.get_rng_state().clone()))
        if use_seq:
            assert seq.slow == T and seq.fast == T + 2, (seq.slow, seq.fast)     # every step slot recorded once, then replayed
        del wl
    (a, ra), (b, rb) = snaps
    assert torch.equal(ra, rb)
    for k in a:
        assert torch.equal(a[k], b[k]), k


def test_one_line_integration_with_the_storage_hook_equals_without():
    """`share_encoders(pi_q, pi_g, pi_l, rollouts=rollouts)`: `rollouts.insert(batch, ...)` starts the next step's shared encoders on
    the batch it is given.  Plain act* calls, no prefetch_* anywhere: same storage bit for bit as without the hook over a rollout,
    the update's value call, the wrap-around."""
    N, T = 4, 5
    snaps = []
    if True:
        for hook in (True, False):
            wl = _run(N, T, precision="bf16x3", launch_ahead=False)
            assert wl.rollouts._enc_leader is wl.pi_q
            if not hook:
                wl.rollouts._enc_leader = None
            for _ in range(T):
                wl.rollout_step()
            ro = wl.rollouts
            assert (wl.pi_q._enc_early is not None) == hook                    # the last insert started slot T's encoders
            last = {k: v[ro.step] for k, v in ro.observations.items()}
            nv = wl.pi_q.get_value_option(last, ro.recurrent_hidden_states[ro.step], ro.prev_actions[ro.step], ro.masks[ro.step],
                                          ro.external_memory_option[:, ro.step], ro.external_memory_masks[ro.step],
                                          ro.query_state[ro.step - 1], ro.last_query_info[ro.step - 1]).clone()
            ro.after_update()
            for _ in range(3):
                wl.rollout_step()
            torch.cuda.synchronize()
            snaps.append((_storage_snapshot(wl), nv))
            del wl
    (a, va), (b, vb) = snaps
    assert torch.equal(va, vb)
    for k in a:
        assert torch.equal(a[k], b[k]), k


@pytest.mark.parametrize("mode", ["bf16x3"])
def test_replayed_graph_follows_a_clip_weight_change(mode):
    """a plain act_dialog with use_graphs=True and no prefetch captures the MEMOISED tower inside the forward graph; the
    replay must not hand out embeddings computed with the old CLIP weights for rows whose tokens did not change."""
    import json
    import os
    import fixtures as fx
    torch.manual_seed(4)
    pol = P.AudioNavDialogPolicy(savi_observation_space((65, 26, 2)), ActionSpace(4), pretraining=False, num_steps=3, precision=mode,
                                 use_graphs=True, **SMT_KW).to("cuda")
    B, dev = 3, "cuda"
    gen = torch.Generator().manual_seed(11)
    obs = {k: v.to(dev) for k, v in fx.observations("tc", B).items()}
    prev = torch.zeros(B, 1, dtype=torch.long, device=dev)
    mem, memd = torch.zeros(3, B, 276, device=dev), torch.zeros(3, B, 256, device=dev)
    mk = torch.zeros(B, 3, device=dev)
    tok = torch.stack([_dialog(gen, 12), torch.zeros(77, dtype=torch.long), _dialog(gen, 40)]).to(dev)
    astep = torch.zeros(B, device=dev)
    call = lambda p: [t.clone() for t in p.act_dialog(obs, None, prev, mk[:, :1], mem, memd, mk, tok, astep, deterministic=True)
                      if torch.is_tensor(t)]
    first = call(pol)
    again = call(pol)                                            # replay: every row comes out of the memo
    assert all(torch.equal(a, b) for a, b in zip(first, again))
    with torch.no_grad():                                        # "a checkpoint with another CLIP tower"
        sd = {k: v.clone() for k, v in pol.state_dict().items()}
        for k in sd:
            if k.startswith("net.clip.transformer.resblocks.5.") or k == "net.clip.ln_final.bias":
                sd[k] = sd[k] + 0.02 * torch.randn(sd[k].shape, generator=gen).to(dev)
    pol.load_state_dict(sd)
    moved = call(pol)                                            # same tokens, same graph
    torch.cuda.synchronize()
    fresh = P.AudioNavDialogPolicy(savi_observation_space((65, 26, 2)), ActionSpace(4), pretraining=False, num_steps=3,
                                   precision=mode, use_graphs=False, **SMT_KW).to("cuda")
    fresh.load_state_dict(sd)
    fresh.net.text_cache = False                                 # the uncached tower, eager
    want = call(fresh)
    torch.cuda.synchronize()
    assert not torch.equal(moved[0], first[0])
    # (the memoised path ends in one fused tail launch, the uncached one in ln_final + cast + product launches: same operands,
    # another summation order in the last 512-long product)
    for a, b in zip(moved, want):
        assert float((a.float() - b.float()).abs().max()) <= 2e-5 * max(1.0, float(b.float().abs().max())), (a, b)
    old = max(float((a.float() - b.float()).abs().max()) for a, b in zip(first, want))
    assert old > 1e-3, old                                       # ... whereas the stale embedding is far away


def test_deterministic_follower_under_auto_ahead_keeps_the_host_generator_in_step():
    """the eval loop samples pi_q and pi_g but calls act_dialog(deterministic=True) (ppo_trainer.py:1917-2156), which
    consumes no generator state.  With sampling="race" and the automatic launch-ahead, no speculative draw may be left behind:
    the sampled actions and the generator's end state equal the plain host-sampling flow without any launch-ahead."""
    import functools
    N, T = 4, 7
    res = []
    for kw in (dict(sampling="race", share_encoders=True, launch_ahead=False),
               dict(sampling="host", share_encoders=False, launch_ahead=False)):
        wl = _run(N, T, precision="bf16x3", **kw)
        wl.pi_l.act_dialog = functools.partial(wl.pi_l.act_dialog, deterministic=True)
        for _ in range(T):
            wl.rollout_step()
        torch.cuda.synchronize()
        grp = wl.pi_q._enc_group
        res.append((wl.rollouts.actions_option.clone(), wl.rollouts.actions.clone(), torch.get_rng_state().clone(),
                    None if grp is None else (grp.auto_hits, grp.auto_misses)))
        del wl
    (q0, a0, r0, auto), (q1, a1, r1, _) = res
    assert auto is not None and auto[0] >= 2 * (T - 3), auto          # the followers really were launched ahead
    assert torch.equal(q0, q1) and torch.equal(a0, a1)
    assert torch.equal(r0, r1), "the host generator drifted: a speculative draw for a deterministic call was not undone"
