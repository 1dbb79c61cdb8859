import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from avlen_amd.harness import Workload
wl = Workload(4, 3, spectrogram=(65, 26, 2), precision="bf16", pretraining=False, em_capacity=4, seed=5, use_graphs=True, share_encoders=True)
torch.manual_seed(11)
net = wl.pi_l.net
for t in range(3):
    wl.rollout_step()
    torch.cuda.synchronize()
    emb = net._text[2].clone()
    ref = net.encode_text(wl.pi_l, wl.dialog[t]).clone()
    torch.cuda.synchronize()
    print(t, "emb vs eager:", float((emb - ref).abs().max()), "emb norm", float(emb.norm()), "ptr ok", net._text[0] == wl.dialog[t].data_ptr())
    prev = net.encode_text(wl.pi_l, wl.dialog[max(t - 1, 0)])
    print("   emb vs eager(prev tokens):", float((emb - prev).abs().max()))
