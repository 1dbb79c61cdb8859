"""GPU + host timeline of one rollout step of the harness' default flow (dialog tokens issued after act_option, race sampling),
averaged: when each piece ends on its stream, relative to the step's first host call.
Usage: python tools/step_timeline.py [envs] [fresh|reference]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from avlen_amd.harness import Workload
from avlen_amd import policy as P

N = int(sys.argv[1]) if len(sys.argv) > 1 else 64
proc = sys.argv[2] if len(sys.argv) > 2 else "fresh"
wl = Workload(N, 150, spectrogram=(257, 101, 2), precision=os.environ.get("AVLEN_PREC", "bf16x3"), pretraining=True,
              dialog_process=proc, sampling=os.environ.get("AVLEN_SAMPLING", "race"))
for _ in range(20):
    wl.rollout_step()
torch.cuda.synchronize()
E = lambda: torch.cuda.Event(enable_timing=True)
acc, host = {}, {}
STEPS = 60
ref = proc == "reference"
for it in range(STEPS):
    ro, t = wl.rollouts, wl.rollouts.step
    v = wl._step_views(t)
    if ref:
        v = dict(v, dialog=wl._cur_dialog, astep=wl._cur_astep)
    obs, h, prev, em_masks = v["obs"], v["h"], v["prev"], v["em_masks"]
    em_opt, em_goal = ro.external_memory_option[:, t], ro.external_memory_goal[:, t]
    em_vln, em_dlg = ro.external_memory_vln[:, t], ro.external_memory_vln_dialog[:, t]
    cur = torch.cuda.current_stream()
    ev, hs = {}, {}
    def mark(name, stream):
        e = E(); e.record(stream); ev[name] = e
    def hmark(name):
        hs[name] = time.perf_counter()
    h0 = time.perf_counter()
    mark("t0", cur)
    wl.pi_q.prefetch_act_option(obs, h, prev, v["masks"], em_opt, em_masks, v["qs"], v["lqi"])
    mark("q_end", cur); hmark("launch_q")
    gs = wl._side[wl._g_stream]
    ls = None if wl._l_main else wl._side[0 if wl._l_where == "own" else wl._g_stream]
    def launch_g():
        wl.pi_g.prefetch_act(obs, h, prev, v["masks"], em_goal, em_masks, stream=gs)
        mark("g_end", gs); hmark("launch_g")
    def launch_l1():
        wl.pi_l.prefetch_act_dialog(obs, h, prev, v["masks_vln"], em_vln, em_dlg, v["em_vln_masks"], v["dialog"], v["astep"], stream=ls,
                                    dialog_later=True)
        mark("l_half1_end", cur if ls is None else ls); hmark("launch_l1")
    for fn in ((launch_l1, launch_g) if wl._l_first else (launch_g, launch_l1)):      # the harness' order
        fn()
    values, unct, a_opt, lp_opt, h2, row_opt, probs_opt = wl.pi_q.act_option(obs, h, prev, v["masks"], em_opt, em_masks, v["qs"], v["lqi"])
    hmark("act_q_done")
    if ref:
        wl._host_dialog_loop(t, wl.pi_q.host_actions("option").view(-1).numpy())
        hmark("host_loop_done")
    mark("txt_start", cur)
    wl.pi_l.dialog_ready()
    mark("txt_end", cur); mark("l_end", cur if ls is None else ls); hmark("dialog_ready")
    _, a_goal, _, _, row_goal, _ = wl.pi_g.act(obs, h2, prev, v["masks"], em_goal, em_masks)
    hmark("act_g_done")
    _, a_vln, _, _, row_vln, row_dlg, probs_vln = wl.pi_l.act_dialog(obs, h2, prev, v["masks_vln"], em_vln, em_dlg, v["em_vln_masks"],
                                                                     v["dialog"], v["astep"])
    hmark("act_l_done")
    host_sel = wl._host_select and wl.sampling == "race"
    if host_sel:                                      # the harness' default: select on the host from the policies' pinned copies
        hq, hg, hl = wl.pi_q.host_actions("option"), wl.pi_g.host_actions("goal"), wl.pi_l.host_actions("vln")
        ah = torch.where(hq == 1, hl, hg)
        mark("actions", cur)
    else:
        actions = torch.where(a_opt == 1, a_vln, a_goal)
        mark("actions", cur)
        if wl.sampling != "host":
            ah = torch.empty(actions.shape, dtype=actions.dtype, pin_memory=True)
            ah.copy_(actions, non_blocking=True)
            cur.synchronize()
    hmark("actions_on_host")
    if wl._early_enc:
        wl.pi_q.prefetch_encoders(v["nxt"], will_be={k: ro.observations[k][t + 1] for k in ("rgb", "depth", P.SPECTROGRAM)})   # the three addresses it checks
        hmark("next_towers_launched")
    if host_sel:
        actions = torch.where(a_opt == 1, a_vln, a_goal)
    ro.insert(v["nxt"], h2, actions, a_opt, lp_opt, values, v["rew"], v["nd"], v["nd"], row_goal, row_opt, row_vln, row_dlg,
              v["dialog"], wl.o_action, wl.o_mask, v["rl"], v["ucnt"], probs_vln, v["qs"], v["lqi"], v["astep"])
    mark("insert_end", cur); hmark("insert_done")
    torch.cuda.synchronize()
    for k in ev:
        if k != "t0":
            acc[k] = acc.get(k, 0.0) + ev["t0"].elapsed_time(ev[k])
    for k, x in hs.items():
        host[k] = host.get(k, 0.0) + (x - h0) * 1e3
print(f"N={N} dialog_process={proc} sampling={wl.sampling}")
print("GPU event times since step start (ms):", {k: round(x / STEPS, 3) for k, x in acc.items()})
print("host times since step start (ms):", {k: round(x / STEPS, 3) for k, x in host.items()})
if ref:
    print("dialog stats:", wl.dialog_stats)
