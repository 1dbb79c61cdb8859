"""Lab: where the bf16x3 memory row differs from the reference goldens (per column group)."""
import os, sys, json
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import fixtures as fx
from conftest import golden, param_specs
from test_gpu_policy_parity import build, load_fixture, cu
specs = param_specs()
for tag, kind, dim in (("opt_p0_m300", "option", 308), ("goal_m300", "goal", 276)):
    for prec in ("fp32", "bf16x3", "bf16"):
        pol = build(kind, precision=prec, pretraining=False)
        load_fixture(pol, kind, specs); pol.cuda()
        B, M = 3, 300
        g = golden("policy_" + tag)
        obs = cu(fx.observations(tag, B))
        mem, mk = fx.memory(tag, M, B, dim, 272).cuda(), fx.mask_patterns(tag, B, M).cuda()
        pa = fx.ints(tag + ".pa", (B, 1), 4).cuda()
        h0, ones = torch.zeros(1, B, 512, device="cuda"), torch.ones(B, 1, device="cuda")
        if kind == "option":
            qs, lqi = fx.sym(tag + ".qs", (B, 32)).cuda(), fx.sym(tag + ".lqi", (B, 32)).cuda()
            act = fx.ints(tag + ".a", (B, 1), 2).cuda()
            v, u, lp, ent, _, row, probs = pol.evaluate_actions_option(obs, h0, pa, ones, act, mem, mk, qs, lqi)
        else:
            act = fx.ints(tag + ".a", (B, 1), 4).cuda()
            v, lp, ent, _, row = pol.evaluate_actions(obs, h0, pa, ones, act, mem, mk)
        r, gr = row.cpu().numpy(), g["row"]
        d = np.abs(r - gr)
        grp = {"rgb": (0, 64), "depth": (64, 128), "action": (128, 144), "audio": (144, 272), "pose": (272, 276), "rest": (276, dim)}
        print(tag, prec, "value err %.3e" % float(np.abs(v.cpu().numpy() - g["value"]).max()),
              {k: "%.2e / scale %.2f" % (d[:, a:b].max() if b > a else 0, np.abs(gr[:, a:b]).max() if b > a else 0) for k, (a, b) in grp.items()})
