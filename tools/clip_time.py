"""Times of the one-launch CLIP text tower's kernels for the benched token mix (64 dialogs), eager calls.  Under rocprofv3
--kernel-trace --stats the per-kernel table is the result; alone it prints the wall time per call."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from avlen_amd import policy as P
from avlen_amd.harness import Workload
from avlen_amd.spaces import savi_observation_space, ActionSpace, SMT_KW
N = int(sys.argv[1]) if len(sys.argv) > 1 else 64
wl = Workload(N, 2, precision="bf16x3", use_graphs=False, share_encoders=False, launch_ahead=False, with_goal_policy=False)
tok = wl.dialog[0]
if os.environ.get("SPLIT4") is not None:
    from avlen_amd import _lib as L
    L.lib.avlen_set_clip_tower_split4_wgs(int(os.environ["SPLIT4"]))
pol = wl.pi_l
f = lambda: pol.net.encode_text(pol, tok)
for _ in range(3):
    f()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    f()
e1.record()
torch.cuda.synchronize()
print("encode_text(%d dialogs): %.1f us per call (events)" % (N, e0.elapsed_time(e1) / 20 * 1e3))
