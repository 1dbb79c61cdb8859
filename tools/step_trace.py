"""Kernel timeline of ONE rollout step from a rocprofv3 rocpd database of `bench.py --steps 1 --warmup 1 --rollout R`: every kernel
of step `k` of the second cycle's rollout with its start (us since the step's first kernel), duration and queue.
Usage: python tools/step_trace.py <results.db> <R> [k=10]"""
import sqlite3, sys, re
c = sqlite3.connect(sys.argv[1]).cursor()
R = int(sys.argv[2]); k = int(sys.argv[3]) if len(sys.argv) > 3 else 10
cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
qcol = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else None)
rows = list(c.execute(f"select start, end, name{', ' + qcol if qcol else ''} from kernels order by start"))
marks = [r[0] for r in rows if "ppo_loss" in r[2]]
lo, hi = marks[3] + 1e6, marks[4] - 1e5
sel = [r for r in rows if r[0] >= lo and r[1] <= hi]
# a step starts with the tower head launch of the lead policy (one per step; the update is outside the window)
heads = [i for i, r in enumerate(sel) if "tower_head" in r[2] or "tower_x3_kernel" in r[2]]
a, b = heads[k], heads[k + 1]
# the text graph of this step started before the head: include kernels back to the previous insert
t0 = sel[a][0]
for r in sel[max(0, a - 8):b]:
    n = re.sub(r"\(anonymous namespace\)::", "", r[2]); n = re.sub(r"^void ", "", n).split("(")[0][:48]
    print(f"{(r[0]-t0)/1e3:8.1f} {(r[1]-r[0])/1e3:7.1f}  q{r[3] if qcol else '?'}  {n}")
