import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from avlen_amd.harness import Workload
N = int(sys.argv[1]) if len(sys.argv) > 1 else 4
wl = Workload(N, 3, spectrogram=(65, 26, 2), precision="bf16", pretraining=False, em_capacity=4, seed=5, use_graphs=True, share_encoders=True)
pol, net = wl.pi_l, wl.pi_l.net
side = torch.cuda.Stream()
ro = wl.rollouts
mode = sys.argv[2] if len(sys.argv) > 2 else "q"
NIT = int(os.environ.get('NIT', 6))
for t in range(NIT):
    v = wl._step_views(0)
    obs, h, prev, em_masks = v["obs"], v["h"], v["prev"], v["em_masks"]
    if mode in ("q", "qg"):
        wl.pi_q.prefetch_act_option(obs, h, prev, v["masks"], ro.external_memory_option[:, 0], em_masks, v["qs"], v["lqi"])
    pol.prefetch_text(wl.dialog[t % 3], side, after_current=False)
    if mode in ("g", "qg"):
        wl.pi_g.prefetch_act(obs, h, prev, v["masks"], ro.external_memory_goal[:, 0], em_masks, stream=wl._side[0])
    torch.cuda.synchronize()
    wl.pi_q._stash = None; wl.pi_g._stash = None
    print(round(float(net._text[2].norm()), 2), end=" ", flush=True)
eng = pol._engine()
g = pol._graphs[("text", tuple(wl.dialog[0].shape))]
print("static tokens equal:", torch.equal(g.static[0], wl.dialog[5 % 3]))
e = net.encode_text(pol, wl.dialog[5 % 3]); torch.cuda.synchronize()
print("eager now:", float(e.norm()))
g.graph.replay(); torch.cuda.synchronize()
print("replay alone now:", float(g.outs.norm()))
