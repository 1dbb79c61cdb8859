"""Wall time of the pieces of one benched cycle (synchronised host timers): steady rollout step, the first step after an update,
get_value + returns, agent.update, after_update."""
import sys, os, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "4")
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from avlen_amd.harness import Workload

wl = Workload(64, 150, spectrogram=(257, 101, 2), precision=os.environ.get("AVLEN_PREC", "bf16x3"), pretraining=True)
wl.cycle(); wl.cycle()
S = lambda: (torch.cuda.synchronize(), time.perf_counter())[1]
for rep in range(2):
    t0 = S(); wl.rollout_step(); t1 = S()
    for _ in range(9):
        wl.rollout_step()
    t2 = S()
    for _ in range(140):
        wl.rollout_step()
    t3 = S()
    ro, s = wl.rollouts, wl.rollouts.step
    last = {k: v[s] for k, v in ro.observations.items()}
    nv = wl.pi_q.get_value_option(last, ro.recurrent_hidden_states[s], ro.prev_actions[s], ro.masks[s], ro.external_memory_option[:, s],
                                  ro.external_memory_masks[s], ro.query_state[s - 1], ro.last_query_info[s - 1])
    t4 = S(); ro.compute_returns(nv, True, 0.99, 0.95); t5 = S()
    out = wl.agent.update(ro); t6 = S()
    ro.after_update(); t7 = S()
    print("first step after update %.2f ms | next 9 steps %.2f ms each | 140 steps %.3f ms each | get_value %.2f | returns %.2f | "
          "agent.update %.2f | after_update %.2f ms" % ((t1 - t0) * 1e3, (t2 - t1) / 9 * 1e3, (t3 - t2) / 140 * 1e3, (t4 - t3) * 1e3,
                                                      (t5 - t4) * 1e3, (t6 - t5) * 1e3, (t7 - t6) * 1e3))
