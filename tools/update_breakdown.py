"""Kernel breakdown of the PPO update (everything after the last `gae_kernel` launch) from a rocprofv3 rocpd database.
Usage: python tools/update_breakdown.py <results.db>"""
import sqlite3, sys, re, collections
c = sqlite3.connect(sys.argv[1]).cursor()
rows = list(c.execute("select start, end, name from kernels order by start"))
gae = [s for s, e, n in rows if "gae_kernel" in n]
lo = gae[-1]
sel = [r for r in rows if r[0] >= lo]
busy = sum(e - s for s, e, n in sel)
print(f"update region {(rows[-1][1]-lo)/1e6:.1f} ms wall, kernel time {busy/1e6:.1f} ms, {len(sel)} launches")
agg = collections.defaultdict(lambda: [0, 0])
for s, e, n in sel:
    n = re.sub(r"\(anonymous namespace\)::", "", n); n = re.sub(r"^void ", "", n).split("(")[0][:58]
    agg[n][0] += 1; agg[n][1] += e - s
print("| kernel | launches | ms | % |\n|---|---|---|---|")
for n, (k, t) in sorted(agg.items(), key=lambda x: -x[1][1])[:24]:
    print(f"| `{n}` | {k} | {t/1e6:.2f} | {100*t/busy:.1f} |")
