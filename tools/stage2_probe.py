"""cfg4-like sanity / timing: pi_q with pretraining=False (memory history used: M = 300 slots in rollout AND in the PPO
update), N envs on one GPU.  Prints env-steps/s and the PPO losses of two cycles."""
import sys, os, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from avlen_amd.harness import Workload

N = int(sys.argv[1]) if len(sys.argv) > 1 else 32
wl = Workload(N, 150, pretraining=False)
for i in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(wl.T):
        wl.rollout_step()
    torch.cuda.synchronize(); t1 = time.perf_counter()
    out = wl.update()
    torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"cycle {i}: rollout {1e3*(t1-t0):.0f} ms, update {1e3*(t2-t1):.0f} ms -> {N*wl.T/(t2-t0):.0f} env-steps/s; "
          f"losses {[round(float(x), 5) for x in out]}; mem {torch.cuda.max_memory_allocated()/2**30:.1f} GiB", flush=True)
