"""Per-kernel times of the grouped tower call (6 towers x N images) in bf16x3 and bf16, from rocprofv3 or events."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from avlen_amd.harness import Workload
N = int(sys.argv[1]) if len(sys.argv) > 1 else 64
for mode in ("bf16x3", "bf16"):
    wl = Workload(N, 2, precision=mode, use_graphs=True)
    grp = wl.pi_q._enc_group
    obs = {k: v[0] for k, v in wl.rollouts.observations.items()}
    fn = lambda: grp.run_all(wl.pi_q, obs["rgb"], obs["depth"])
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        fn()
    e1.record()
    torch.cuda.synchronize()
    print(mode, "towers grouped call: %.1f us" % (e0.elapsed_time(e1) / 20 * 1e3), flush=True)
