"""How long does hipGraphLaunch hold the host, and does the GPU start before it returns?  (pi_q / pi_g / pi_l graphs)"""
import sys, os, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from avlen_amd.harness import Workload
wl = Workload(64, 150)
for _ in range(10):
    wl.rollout_step()
torch.cuda.synchronize()
for pol, name in ((wl.pi_q, "pi_q"), (wl.pi_g, "pi_g"), (wl.pi_l, "pi_l")):
    for key, g in pol._graphs.items():
        host, tot = [], []
        for _ in range(20):
            torch.cuda.synchronize()
            t0 = time.perf_counter(); g.graph.replay(); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
            host.append(t1 - t0); tot.append(t2 - t0)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for _ in range(20):
            g.graph.replay()
        e1.record(); torch.cuda.synchronize()
        n = len(g.graph.debug_dump.__doc__ or "") * 0
        print(f"{name} {key[0]}: launch call {1e6*sorted(host)[10]:.0f} us on the host; launch->done {1e6*sorted(tot)[10]:.0f} us; "
              f"back-to-back GPU time {e0.elapsed_time(e1)/20*1e3:.0f} us")
