"""Per-rollout-step kernel breakdown from a rocprofv3 rocpd database of `bench.py --steps 1 --warmup 1 --rollout R`:
takes the kernels between the 4th and 5th `ppo_loss_kernel` launches (= the second cycle's rollout).
Usage: python tools/step_breakdown.py <results.db> <R>"""
import sqlite3, sys, re, collections
c = sqlite3.connect(sys.argv[1]).cursor()
R = int(sys.argv[2])
rows = list(c.execute("select start, end, name from kernels order by start"))
marks = [s for s, e, n in rows if "ppo_loss" in n]
lo, hi = marks[3] + 1e6, marks[4] - 1e5
sel = [r for r in rows if r[0] >= lo and r[1] <= hi]
busy = sum(e - s for s, e, n in sel)
print(f"rollout region {(hi-lo)/1e6:.1f} ms, {len(sel)/R:.0f} launches/step, kernel time {busy/R/1e3:.0f} us/step")
agg = collections.defaultdict(lambda: [0, 0])
for s, e, n in sel:
    n = re.sub(r"\(anonymous namespace\)::", "", n); n = re.sub(r"^void ", "", n).split("(")[0][:58]
    agg[n][0] += 1; agg[n][1] += e - s
print("| kernel | launches/step | us/step |\n|---|---|---|")
for n, (k, t) in sorted(agg.items(), key=lambda x: -x[1][1])[:34]:
    print(f"| `{n}` | {k/R:.1f} | {t/R/1e3:.1f} |")
