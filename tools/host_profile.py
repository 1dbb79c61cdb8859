"""cProfile of the host side of the rollout loop (where the Python time of one step goes)."""
import sys, os, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from avlen_amd.harness import Workload
wl = Workload(64, 150, spectrogram=(257, 101, 2), precision=os.environ.get("AVLEN_PREC", "bf16x3"), pretraining=True)
wl.cycle()
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(150):
    wl.rollout_step()
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr); st.sort_stats("cumulative").print_stats(45)
