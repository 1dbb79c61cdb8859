"""The kernels of ONE minibatch pass of the PPO update in launch order with durations (from a rocprofv3 rocpd database; the first
pass after the last `gae_kernel`; kernels under `min_us` are folded).  Usage: python tools/update_sequence.py <results.db> [min_us]"""
import sqlite3, sys, re
c = sqlite3.connect(sys.argv[1]).cursor()
min_us = float(sys.argv[2]) if len(sys.argv) > 2 else 100.0
rows = list(c.execute("select start, end, name from kernels order by start"))
lo = [s for s, e, n in rows if "gae_kernel" in n][-1]
sel = [r for r in rows if r[0] >= lo]
adam = [i for i, r in enumerate(sel) if "adam" in r[2]]
sel = sel[:adam[0] + 1] if adam else sel
t0 = sel[0][0]; small = 0.0; nsmall = 0
for s, e, n in sel:
    n = re.sub(r"\(anonymous namespace\)::", "", n); n = re.sub(r"^void ", "", n).split("(")[0][:70]
    us = (e - s) / 1e3
    if us < min_us:
        small += us; nsmall += 1
        continue
    if nsmall:
        print(f"            ... {nsmall} kernels under {min_us:.0f} us: {small:.0f} us"); small = 0.0; nsmall = 0
    print(f"{(s - t0) / 1e3:10.1f} {us:9.1f}  {n}")
if nsmall: print(f"            ... {nsmall} kernels under {min_us:.0f} us: {small:.0f} us")
