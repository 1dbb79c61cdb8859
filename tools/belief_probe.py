"""GPU probe: determinism of the belief networks (eager twice, fresh instance, graph replay) and per-step cost."""
import sys, os, types, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import torch
import fixtures as fx
from avlen_amd.belief_predictor import BeliefPredictor

cfg = types.SimpleNamespace(use_label_belief=True, use_location_belief=True, online_training=True, current_pred_only=False,
                            weighting_factor=0.5)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 3


def make(ug=False, prec="bf16"):
    torch.manual_seed(0)
    bp = BeliefPredictor(cfg, "cuda", None, None, 512, num_env=N, precision=prec, load_pretrained=False, use_graphs=ug)
    return bp.cuda()


obs = {k: v.cuda() for k, v in fx.observations("probe", N).items()}
a = make()
p1 = a.cnn_forward(obs).clone(); p2 = a.cnn_forward(obs).clone()
junk = torch.randn(64 << 20, device="cuda")            # perturb the allocator state
b = make()
p3 = b.cnn_forward(obs).clone()
print("predictor eager twice equal:", torch.equal(p1, p2), " fresh instance equal:", torch.equal(p1, p3), (p1 - p3).abs().max().item())
s = a._filter_state(N)
l1 = a._run("classifier", obs["spectrogram"], s["labels"]).clone(); l2 = a._run("classifier", obs["spectrogram"], s["labels"]).clone()
sb = b._filter_state(N)
l3 = b._run("classifier", obs["spectrogram"], sb["labels"]).clone()
print("classifier eager twice equal:", torch.equal(l1, l2), " fresh instance equal:", torch.equal(l1, l3), (l1 - l3).abs().max().item())
for ug in (False, True):
    bp = make(ug)
    o = {k: v.clone() for k, v in obs.items()}
    for _ in range(3):
        bp.update(o, None)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(50):
        bp.update(o, None)
    torch.cuda.synchronize()
    print(f"update graphs={ug}: {(time.perf_counter() - t0) / 50 * 1e6:.0f} us/step at N={N}")
