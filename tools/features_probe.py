import sys, os, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from avlen_amd import policy as P
from avlen_amd.spaces import savi_observation_space, ActionSpace, SMT_KW
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4800
pol = P.AudioNavOptionPolicy(savi_observation_space((257, 101, 2)), ActionSpace(4), pretraining=True, use_category_input=False,
                             query_count_emb_size=32, precision="bf16", **SMT_KW).cuda()
obs = {"rgb": torch.rand(B, 128, 128, 3, device="cuda") * 255, "depth": torch.rand(B, 128, 128, 1, device="cuda"),
       "spectrogram": torch.rand(B, 257, 101, 2, device="cuda"), "pose": torch.rand(B, 4, device="cuda"),
       "category_belief": torch.rand(B, 21, device="cuda"), "location_belief": torch.rand(B, 2, device="cuda")}
pa = torch.zeros(B, 1, dtype=torch.long, device="cuda"); qs = torch.zeros(B, 32, device="cuda")
for i in range(3):
    torch.cuda.synchronize(); t = time.perf_counter()
    pol.net.features(pol, obs, pa, extra=qs)
    torch.cuda.synchronize(); print("features B=%d: %.2f ms" % (B, (time.perf_counter() - t) * 1e3))
