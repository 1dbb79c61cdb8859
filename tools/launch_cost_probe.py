import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from avlen_amd.harness import Workload
wl = Workload(64, 150, spectrogram=(257, 101, 2), precision="bf16", pretraining=True)
wl.cycle(); wl.cycle()
torch.cuda.synchronize()
for use_memo in (True, False, True, False):
    if not use_memo:
        for p in (wl.pi_q, wl.pi_g, wl.pi_l):
            p._memos = {}
            p._memo_off = True
    else:
        for p in (wl.pi_q, wl.pi_g, wl.pi_l):
            p._memo_off = False
        wl.cycle()
    tq = 0.0
    ro = wl.rollouts
    for t in range(100):
        v = wl._step_views(ro.step)
        obs, h, prev, em_masks = v["obs"], v["h"], v["prev"], v["em_masks"]
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        wl.pi_q.prefetch_act_option(obs, h, prev, v["masks"], ro.external_memory_option[:, ro.step], em_masks, v["qs"], v["lqi"])
        tq += time.perf_counter() - t0
        wl.pi_q._stash = None
        torch.cuda.synchronize()
    print("memo" if use_memo else "no memo", "prefetch_act_option host time", round(tq / 100 * 1e6, 1), "us")
