import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from avlen_amd.harness import Workload
wl = Workload(4, 3, spectrogram=(65, 26, 2), precision="bf16", pretraining=False, em_capacity=4, seed=5, use_graphs=True, share_encoders=True)
pol, net = wl.pi_l, wl.pi_l.net
side = torch.cuda.Stream()
for t in range(4):
    pol.prefetch_text(wl.dialog[t % 3], side, after_current=True)
    torch.cuda.synchronize()
    print("replay", t, "emb norm", float(net._text[2].norm()), flush=True)
# eager twice on the same workspace
for t in range(3):
    e = net.encode_text(pol, wl.dialog[t]); torch.cuda.synchronize()
    print("eager", t, float(e.norm()), flush=True)
