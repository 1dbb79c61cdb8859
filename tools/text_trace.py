"""The CLIP text tower's graph of one rollout step, replayed alone: run under `rocprofv3 --kernel-trace` and summarise with
tools/text_trace.py --report <results.db>: the kernel sequence of ONE replay with durations and gaps."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 2 and sys.argv[1] == "--report":
    import sqlite3
    c = sqlite3.connect(sys.argv[2])
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if "kernel_dispatch" in t][0]; ks = [t for t in tabs if "kernel_symbol" in t][0]
    rows = c.execute(f"select s.kernel_name, d.start, d.end from {kd} d join {ks} s on d.kernel_id=s.id order by d.start").fetchall()
    # the last replay: find the last occurrence of the first kernel of the text graph (the token gather)
    names = [r[0] for r in rows]
    per = int(os.environ.get("PER", "0"))
    if not per:
        # period = distance between the last two occurrences of the last kernel name
        last = names[-1]
        idx = [i for i, n in enumerate(names) if n == last]
        cnt = {}
        per = None
        for gap in range(20, 400):
            if len(names) > 2 * gap and names[-gap:] == names[-2 * gap:-gap]:
                per = gap; break
    seq = rows[-per:]
    t0 = seq[0][1]
    tot = 0.0
    agg = {}
    for i, (n, s, e) in enumerate(seq):
        short = n.split("(")[0][:70]
        gap = (s - seq[i - 1][2]) / 1e3 if i else 0.0
        if i < int(os.environ.get("SHOW", "24")):
            print(f"{(s - t0) / 1e3:8.1f} us  +{(e - s) / 1e3:6.1f}  gap {gap:5.1f}  {short}")
        a = agg.setdefault(short, [0, 0.0, 0.0]); a[0] += 1; a[1] += (e - s) / 1e3; a[2] += max(gap, 0.0)
        tot += (e - s) / 1e3
    span = (seq[-1][2] - t0) / 1e3
    print(f"kernels per replay {per}, busy {tot:.1f} us, span {span:.1f} us")
    for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print(f"  {a[0]:4d} x {a[1] / a[0]:6.2f} us = {a[1]:7.1f} us  (gaps before: {a[2]:6.1f})  {k}")
    sys.exit(0)
import torch
from avlen_amd.harness import Workload
N = int(sys.argv[1]) if len(sys.argv) > 1 else 64
wl = Workload(N, 150, spectrogram=(257, 101, 2), precision=os.environ.get("AVLEN_PREC", "bf16x3"), pretraining=True)
for _ in range(3):
    wl.rollout_step()
torch.cuda.synchronize()
pol = wl.pi_l
v = wl._step_views(wl.rollouts.step)
side = torch.cuda.Stream()
for _ in range(12):
    pol.prefetch_text(v["dialog"], side, after_current=True)
    torch.cuda.synchronize()
print("done")
