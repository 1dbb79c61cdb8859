"""Summarise a rocprofv3 run (rocpd `*_results.db`, or a `*_kernel_stats.csv`) as the markdown table kept under profiles/.
Usage: python tools/prof_summary.py <results.db|kernel_stats.csv> [title ...]"""
import csv, re, sqlite3, sys


def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"^void ", "", n)
    return n.split("(")[0][:64]


def rows_from(path):
    """-> [(name, calls, total ns, avg ns, min ns, fraction of the kernel's time spent beside a running persistent tower launch)]"""
    if path.endswith(".db"):
        c = sqlite3.connect(path).cursor()
        ks = list(c.execute("select name, start, end from kernels"))
        towers = sorted((s0, e0) for n, s0, e0 in ks if "tower_x3_kernel" in n or "tower_head_kernel" in n)
        import bisect
        starts = [t[0] for t in towers]
        agg = {}
        for n, s0, e0 in ks:
            ov = 0
            i = max(bisect.bisect_right(starts, s0) - 1, 0)
            while i < len(towers) and towers[i][0] < e0:
                ov += max(0, min(e0, towers[i][1]) - max(s0, towers[i][0]))
                i += 1
            a = agg.setdefault(n, [0, 0.0, 1e30, 0.0])
            a[0] += 1; a[1] += e0 - s0; a[2] = min(a[2], e0 - s0); a[3] += ov
        return [(n, a[0], a[1], a[1] / a[0], a[2], a[3] / a[1] if a[1] else 0.0) for n, a in agg.items()]
    return [(r["Name"], int(r["Calls"]), float(r["TotalDurationNs"]), float(r["AverageNs"]), float(r.get("MinNs", r["AverageNs"])), 0.0)
            for r in csv.DictReader(open(path))]


def main(path, title=""):
    rows = sorted(rows_from(path), key=lambda r: -r[2])
    tot = sum(r[2] for r in rows)
    print(f"# {title}\n\ntotal kernel time {tot/1e6:.1f} ms over {sum(r[1] for r in rows)} launches\n")
    print("A kernel's duration runs from its dispatch to its last wave's end: a microsecond job dispatched while the persistent "
          "tower launch (one workgroup per CU, 0.4 ms) holds every CU is billed the time it WAITED for a CU.  Rows that spend most of "
          "their time beside a running tower launch and whose average is far above their fastest launch are marked: their `total` "
          "is waiting, not work (the fastest launch is what the job costs).\n")
    print("| kernel | calls | total ms | avg us | min us | % | note |\n|---|---|---|---|---|---|---|")
    for n, calls, t, avg, mn, beside in rows[:45]:
        note = ("%.0f %% of it beside the persistent tower launch: waiting for a CU" % (100 * beside)
                if beside > 0.5 and avg > 5 * mn and "tower" not in n else "")
        print(f"| `{short(n)}` | {calls} | {t/1e6:.2f} | {avg/1e3:.1f} | {mn/1e3:.1f} | {100*t/tot:.1f} | {note} |")


if __name__ == "__main__":
    main(sys.argv[1], " ".join(sys.argv[2:]))
