import sys, os, time
sys.path.insert(0, "/root/repo")
os.environ.setdefault("GPU_MAX_HW_QUEUES", "4")
import torch
from avlen_amd.harness import Workload
from avlen_amd import _lib as L
wl = Workload(64, 150, spectrogram=(257, 101, 2), precision="bf16", pretraining=True)
wl.cycle()
S = lambda: (torch.cuda.synchronize(), time.perf_counter())[1]
for bm in (16384, 4096, 1024, 16384, 4096):
    L.lib.avlen_set_big_m(bm)
    for _ in range(150):
        wl.rollout_step()
    ro, s = wl.rollouts, wl.rollouts.step
    last = {k: v[s] for k, v in ro.observations.items()}
    nv = wl.pi_q.get_value_option(last, ro.recurrent_hidden_states[s], ro.prev_actions[s], ro.masks[s], ro.external_memory_option[:, s],
                                  ro.external_memory_masks[s], ro.query_state[s - 1], ro.last_query_info[s - 1])
    ro.compute_returns(nv, True, 0.99, 0.95)
    t0 = S(); out = wl.agent.update(ro); t1 = S()
    ro.after_update()
    print("big_m", bm, "agent.update %.2f ms" % ((t1 - t0) * 1e3), [round(float(x), 4) for x in out[:3]])
