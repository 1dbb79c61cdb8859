"""The trainer checkpoints `agent.optimizer.state_dict()` and restores it with `load_state_dict()` on a requeue
(ss_baselines/savi/ddppo/algo/ddppo_trainer.py:812-817, 857-862; ss_baselines/savi/ppo/ppo_trainer.py:1184-1187, 1224).  The
HIP Adam steps flat moment buffers; `engine.FlatAdam` exposes them as torch.optim.Adam's own state entries.

CPU: the binding / adoption logic on a small module.  GPU: save -> NEW agent -> load: the moments and the step counter arrive bit
for bit, and the resumed update() is the uninterrupted second update (to the last-bit noise of the update's own float atomics:
two uninterrupted runs differ by the same 1e-9), while an agent without the optimiser state lands 1e-4 away."""
import copy
import io

import numpy as np
import pytest
import torch
import torch.nn as nn

from avlen_amd import engine as E


class _Tiny(nn.Module):
    def __init__(self):
        super().__init__()
        self.frozen = nn.Linear(3, 5)
        self.a = nn.Linear(5, 7)
        self.b = nn.Linear(7, 2)
        self._eng = None


def _tiny():
    torch.manual_seed(5)
    m = _Tiny()
    flat = E.FlatParams(m, ("a.", "b."))
    opt = torch.optim.Adam(m.parameters(), lr=1e-3, eps=1e-5)
    return m, flat, opt


def test_state_is_empty_before_the_first_step_and_keyed_like_torch_adam():
    m, flat, opt = _tiny()
    ad = E.FlatAdam(opt, m)
    assert opt.state_dict()["state"] == {}                      # as a fresh torch.optim.Adam
    ad.state(flat)
    ad.advance()
    sd = opt.state_dict()
    names = [n for n, _ in m.named_parameters()]
    trained = {i for i, n in enumerate(names) if n.startswith(("a.", "b."))}
    assert set(sd["state"]) == trained                           # no entry for parameters the loss never reaches
    for i in trained:
        e = sd["state"][i]
        assert set(e) == {"step", "exp_avg", "exp_avg_sq"} and float(e["step"]) == 1.0
        assert e["exp_avg"].shape == dict(m.named_parameters())[names[i]].shape
    # the entries are VIEWS: what the (HIP) optimiser step writes into the flat buffers is what a checkpoint sees
    ad.m.fill_(0.25)
    assert all(float(e["exp_avg"].min()) == 0.25 for e in opt.state_dict()["state"].values())


@pytest.mark.parametrize("engine_ready", [True, False])
def test_round_trip_through_torch_save(engine_ready):
    m, flat, opt = _tiny()
    ad = E.FlatAdam(opt, m).state(flat)
    g = torch.Generator().manual_seed(1)
    ad.m.copy_(torch.randn(ad.m.shape, generator=g))
    ad.v.copy_(torch.rand(ad.v.shape, generator=g))
    for _ in range(3):
        ad.advance()
    buf = io.BytesIO()
    torch.save(opt.state_dict(), buf)
    buf.seek(0)
    m2, flat2, opt2 = _tiny()
    ad2 = E.FlatAdam(opt2, m2)
    if engine_ready:                                             # moments already allocated (a running agent that reloads)
        ad2.state(flat2)
        ad2.advance()
    opt2.load_state_dict(torch.load(buf))
    ad2.state(flat2)                                             # what the next optimiser step calls
    assert ad2.step == 3 and float(ad2._step_t) == 3.0
    for n in flat.trained_names:
        o, k = flat.offsets[n]
        o2, _ = flat2.offsets[n]
        assert torch.equal(ad.m[o:o + k], ad2.m[o2:o2 + k]) and torch.equal(ad.v[o:o + k], ad2.v[o2:o2 + k])
    # and the entries are views into the new agent's buffers again
    p = dict(m2.named_parameters())["a.weight"]
    assert opt2.state[p]["exp_avg"].data_ptr() == ad2.m.data_ptr() + 4 * flat2.offsets["a.weight"][0]
    assert ad2.advance() == 4


@pytest.mark.gpu
@pytest.mark.parametrize("precision", ["fp32", "bf16x3"])
def test_requeued_agent_continues_bit_for_bit(precision):
    from avlen_amd.harness import Workload
    from avlen_amd.ppo import DDPPO
    N, T = 4, 6
    torch.manual_seed(21)
    wl = Workload(N, T, spectrogram=(65, 26, 2), precision=precision, pretraining=False, em_capacity=4, seed=2)
    for _ in range(T):
        wl.rollout_step()
    ro, s = wl.rollouts, wl.rollouts.step
    wl._join_small()                                        # the last step's storage writes ran on the harness' side stream
    nv = wl.pi_q.get_value_option({k: v[s] for k, v in ro.observations.items()}, ro.recurrent_hidden_states[s], ro.prev_actions[s],
                                  ro.masks[s], ro.external_memory_option[:, s], ro.external_memory_masks[s], ro.query_state[s - 1],
                                  ro.last_query_info[s - 1])
    ro.compute_returns(nv, True, 0.99, 0.95)
    first = wl.agent.update(ro)
    assert all(np.isfinite(first))
    torch.cuda.synchronize()
    # --- the checkpoint the trainer writes ---
    buf = io.BytesIO()
    torch.save({"state_dict": wl.pi_q.state_dict(), "optim_state": wl.agent.optimizer.state_dict()}, buf)
    rng = torch.get_rng_state()
    m_saved, v_saved = wl.agent._adam.m.clone(), wl.agent._adam.v.clone()
    osd = wl.agent.optimizer.state_dict()
    trained = [n for n, _ in wl.pi_q.named_parameters() if n.startswith(wl.pi_q.TRAINED_PREFIXES)]
    assert len(osd["state"]) == len(trained) > 0
    assert all(float(e["step"]) == 4.0 for e in osd["state"].values())              # 2 epochs x 2 minibatches
    # --- uninterrupted: the second update ---
    second = wl.agent.update(ro)
    torch.cuda.synchronize()
    want = {k: v.detach().clone() for k, v in wl.pi_q.state_dict().items()}

    def resumed(load_optimizer):
        buf.seek(0)
        ck = torch.load(buf)
        wl.pi_q.load_state_dict(ck["state_dict"])
        agent = DDPPO(wl.pi_q, clip_param=0.2, ppo_epoch=2, num_mini_batch=2, value_loss_coef=0.5, entropy_coef=0.05, lr=2.5e-4,
                      eps=1e-5, max_grad_norm=0.2, use_normalized_advantage=False)
        agent.init_distributed(find_unused_params=True)
        if load_optimizer:
            agent.optimizer.load_state_dict(ck["optim_state"])
            ad = agent._adam.state(wl.pi_q._engine()["flat"])
            assert ad.step == 4 and torch.equal(ad.m, m_saved) and torch.equal(ad.v, v_saved)      # the state itself: bit for bit
        torch.set_rng_state(rng)
        out = agent.update(ro)
        torch.cuda.synchronize()
        return out, {k: v.detach().clone() for k, v in wl.pi_q.state_dict().items()}, agent

    out, got, agent = resumed(True)
    np.testing.assert_allclose(out, second, rtol=1e-4, atol=1e-6)     # the loss LOG is accumulated with float atomics (1e-6 run to run)
    # the update accumulates head gradients and loss sums with float atomics: two runs of the SAME update differ in the last bits
    # (measured 4e-9 .. 1.2e-7 -- one fp32 ulp -- on parameters of O(1) that move by 2.5e-4 per step)
    worst = max(float((got[k].double() - want[k].double()).abs().max()) for k in want)
    assert worst < 1e-6, worst
    assert all(float(e["step"]) == 8.0 for e in agent.optimizer.state_dict()["state"].values())
    # control: without the optimiser state Adam restarts from zero moments and the step differs
    _, cold, _ = resumed(False)
    apart = max(float((cold[k].double() - want[k].double()).abs().max()) for k in trained)
    assert apart > 1e-5, apart
    print(f"resumed vs uninterrupted: {worst:.2e}; without the optimiser state: {apart:.2e}")
