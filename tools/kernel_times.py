"""Print per-launch duration statistics of kernels whose name contains a pattern, from a rocprofv3 rocpd database.
Usage: python tools/kernel_times.py <results.db> <pattern>"""
import sqlite3, sys, statistics
c = sqlite3.connect(sys.argv[1]).cursor()
rows = [r[0] / 1e3 for r in c.execute("select end-start from kernels where name like ? order by start", (f"%{sys.argv[2]}%",))]
rows.sort()
n = len(rows)
print(f"{sys.argv[2]}: n={n} median={statistics.median(rows):.1f}us p10={rows[n//10]:.1f} p90={rows[(9*n)//10]:.1f} max={rows[-1]:.1f} sum={sum(rows)/1e3:.2f}ms")
