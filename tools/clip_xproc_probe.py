"""Cross-process determinism probe of the text tower: prints a digest of encode_text on the benched token mix (seeded weights and
tokens) -- run it in two processes and compare the lines."""
import os, sys, hashlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from avlen_amd.harness import Workload
wl = Workload(64, 4, precision="bf16x3", use_graphs=False, share_encoders=False, launch_ahead=False, with_goal_policy=False)
pol = wl.pi_l
for n in (64, 16):
    tok = wl.dialog[1][:n].contiguous()
    e = pol.net.encode_text(pol, tok)
    torch.cuda.synchronize()
    b = e.cpu().numpy().tobytes()
    rows = [hashlib.sha1(e[i].cpu().numpy().tobytes()).hexdigest()[:6] for i in range(n)]
    print(f"n={n} digest {hashlib.sha1(b).hexdigest()[:16]} rows {' '.join(rows[:12])}")
    print("   tokens digest", hashlib.sha1(tok.cpu().numpy().tobytes()).hexdigest()[:16],
          "weights digest", hashlib.sha1(pol.net.clip.token_embedding.weight.detach().cpu().numpy().tobytes()).hexdigest()[:16])
