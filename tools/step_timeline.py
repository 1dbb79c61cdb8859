"""GPU + host timeline of one rollout step (averaged): when each policy's graph starts/ends on its stream, relative to the
step's first host call.  Usage: python tools/step_timeline.py [envs]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from avlen_amd.harness import Workload

N = int(sys.argv[1]) if len(sys.argv) > 1 else 64
wl = Workload(N, 150, spectrogram=(257, 101, 2), precision=os.environ.get("AVLEN_PREC", "bf16x3"), pretraining=True)
for _ in range(20):
    wl.rollout_step()
torch.cuda.synchronize()
E = lambda: torch.cuda.Event(enable_timing=True)
acc = {}
host = {}
STEPS = 60
for it in range(STEPS):
    ro, t = wl.rollouts, wl.rollouts.step
    v = wl._step_views(t)
    obs, h, prev, em_masks = v["obs"], v["h"], v["prev"], v["em_masks"]
    em_opt, em_goal = ro.external_memory_option[:, t], ro.external_memory_goal[:, t]
    em_vln, em_dlg = ro.external_memory_vln[:, t], ro.external_memory_vln_dialog[:, t]
    cur = torch.cuda.current_stream()
    ev = {}
    def mark(name, stream):
        e = E(); e.record(stream); ev[name] = e
    h0 = time.perf_counter()
    mark("t0", cur)
    wl.pi_q.prefetch_act_option(obs, h, prev, v["masks"], em_opt, em_masks, v["qs"], v["lqi"])
    mark("q_end", cur); host_q = time.perf_counter()
    mark("txt_start", wl._side[2])
    wl.pi_l.prefetch_text(v["dialog"], wl._side[2], after_current=False)
    mark("txt_end", wl._side[2]); host_t = time.perf_counter()
    wl.pi_g.prefetch_act(obs, h, prev, v["masks"], em_goal, em_masks, stream=wl._side[wl._g_stream])
    mark("g_end", wl._side[wl._g_stream]); host_g = time.perf_counter()
    l_stream = None if wl._l_main else wl._side[1]       # the harness default: pi_l behind pi_q on the current stream
    wl.pi_l.prefetch_act_dialog(obs, h, prev, v["masks_vln"], em_vln, em_dlg, v["em_vln_masks"], v["dialog"], v["astep"],
                                stream=l_stream)
    mark("l_end", cur if l_stream is None else l_stream); host_l = time.perf_counter()
    values, unct, a_opt, lp_opt, h2, row_opt, probs_opt = wl.pi_q.act_option(obs, h, prev, v["masks"], em_opt, em_masks, v["qs"], v["lqi"])
    host_aq = time.perf_counter()
    _, a_goal, _, _, row_goal, _ = wl.pi_g.act(obs, h2, prev, v["masks"], em_goal, em_masks)
    host_ag = time.perf_counter()
    _, a_vln, _, _, row_vln, row_dlg, probs_vln = wl.pi_l.act_dialog(obs, h2, prev, v["masks_vln"], em_vln, em_dlg, v["em_vln_masks"],
                                                                     v["dialog"], v["astep"])
    host_al = time.perf_counter()
    actions = torch.where(a_opt == 1, a_vln, a_goal)
    ro.insert(v["nxt"], h2, actions, a_opt, lp_opt, values, v["rew"], v["nd"], v["nd"], row_goal, row_opt, row_vln, row_dlg,
              v["dialog"], wl.o_action, wl.o_mask, v["rl"], v["ucnt"], probs_vln, v["qs"], v["lqi"], v["astep"])
    mark("insert_end", cur); host_ins = time.perf_counter()
    torch.cuda.synchronize()
    for k in ("q_end", "txt_start", "txt_end", "g_end", "l_end", "insert_end"):
        acc[k] = acc.get(k, 0.0) + ev["t0"].elapsed_time(ev[k])
    for k, x in (("launch_q", host_q), ("launch_txt", host_t), ("launch_g", host_g), ("launch_l", host_l), ("act_q_done", host_aq),
                 ("act_g_done", host_ag), ("act_l_done", host_al), ("insert_done", host_ins)):
        host[k] = host.get(k, 0.0) + (x - h0) * 1e3
    if ro.step == 0:
        pass
print("GPU event times since step start (ms):", {k: round(x / STEPS, 3) for k, x in acc.items()})
print("host times since step start (ms):", {k: round(x / STEPS, 3) for k, x in host.items()})
