"""Host-side timeline of the FREE-RUNNING rollout step of the harness' default flow (no synchronisation added): the harness
stamps perf_counter at its phase boundaries (Workload.trace).  With sampling="race" the host polls the mapped action buffers,
so "a_q_on_host" / "actions_on_host" are the GPU's completion times of pi_q's / pi_l's heads kernels to within the poll period.
Usage: python tools/step_timeline.py [envs] [fresh|reference] [steps]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "4")
import torch
from avlen_amd.harness import Workload

N = int(sys.argv[1]) if len(sys.argv) > 1 else 64
proc = sys.argv[2] if len(sys.argv) > 2 else "fresh"
STEPS = int(sys.argv[3]) if len(sys.argv) > 3 else 100
wl = Workload(N, 150, spectrogram=(257, 101, 2), precision=os.environ.get("AVLEN_PREC", "bf16x3"), pretraining=True, dialog_process=proc)
wl.cycle()
for _ in range(20):
    wl.rollout_step()
torch.cuda.synchronize()
wl.trace = tr = []
for _ in range(STEPS):
    wl.rollout_step()
torch.cuda.synchronize()
wl.trace = None
names = ["launch_begin", "launched", "a_q_on_host", "dialog_ready_done", "actions_on_host", "next_towers_launched"]
steps, cur = [], {}
for k, t in tr:
    if k == "launch_begin" and cur:
        steps.append(cur); cur = {}
    cur[k] = t
steps = [s for s in steps if all(n in s for n in names)]
period = [(b["actions_on_host"] - a["actions_on_host"]) * 1e6 for a, b in zip(steps, steps[1:])]
med = lambda x: sorted(x)[len(x) // 2]
print(f"N={N} dialog_process={proc}: {len(steps)} steps, step period (actions_on_host -> actions_on_host) median {med(period):.1f} us, mean {sum(period) / len(period):.1f} us")
seg = [("launch_begin", "launched", "host: launch q | g | l1"), ("launched", "a_q_on_host", "host waits for a_q"),
       ("a_q_on_host", "dialog_ready_done", "host: (dialog loop +) text tower + pi_l half 2 launched"),
       ("dialog_ready_done", "actions_on_host", "host waits for a_g, a_l; selects"),
       ("actions_on_host", "next_towers_launched", "host: next towers launched")]
for a, b, what in seg:
    d = [(s[b] - s[a]) * 1e6 for s in steps]
    print(f"  {what:58s} median {med(d):7.1f} us")
d = [(b["launch_begin"] - a["next_towers_launched"]) * 1e6 for a, b in zip(steps, steps[1:])]
print(f"  {'host: insert + views (towers running)':58s} median {med(d):7.1f} us")
d = [(b["a_q_on_host"] - a["next_towers_launched"]) * 1e6 for a, b in zip(steps, steps[1:])]
print(f"  GPU: next towers launched -> a_q on host (towers + fc + staging + pi_q)   median {med(d):7.1f} us")
d = [(s["actions_on_host"] - s["dialog_ready_done"]) * 1e6 for s in steps]
print(f"  GPU: text tower launched -> actions on host (text tower + pi_l half 2)    median {med(d):7.1f} us")
if wl.seq is not None:
    print("sequencer: fast", wl.seq.fast, "slow", wl.seq.slow)
if proc == "reference":
    print("dialog stats:", wl.dialog_stats)
