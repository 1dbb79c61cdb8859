"""Summarise a rocprofv3 run (rocpd `*_results.db`, or a `*_kernel_stats.csv`) as the markdown table kept under profiles/.
Usage: python tools/prof_summary.py <results.db|kernel_stats.csv> [title ...]"""
import csv, re, sqlite3, sys


def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"^void ", "", n)
    return n.split("(")[0][:64]


def rows_from(path):
    if path.endswith(".db"):
        c = sqlite3.connect(path).cursor()
        return [(r[0], r[1], float(r[2]), float(r[3]), float(r[4])) for r in c.execute(
            "select name, count(*), sum(end-start), avg(end-start), min(end-start) from kernels group by name")]
    return [(r["Name"], int(r["Calls"]), float(r["TotalDurationNs"]), float(r["AverageNs"]), float(r.get("MinNs", r["AverageNs"])))
            for r in csv.DictReader(open(path))]


def main(path, title=""):
    rows = sorted(rows_from(path), key=lambda r: -r[2])
    tot = sum(r[2] for r in rows)
    print(f"# {title}\n\ntotal kernel time {tot/1e6:.1f} ms over {sum(r[1] for r in rows)} launches\n")
    print("A kernel's duration runs from its dispatch to its last wave's end: a microsecond job dispatched while the persistent "
          "tower launch (one workgroup per CU, 0.4 ms) holds every CU is billed the time it WAITED for a CU.  Rows whose average is "
          "more than 5x their fastest launch are marked: their `total` is mostly waiting, not work.\n")
    print("| kernel | calls | total ms | avg us | min us | % | note |\n|---|---|---|---|---|---|---|")
    for n, calls, t, avg, mn in rows[:45]:
        note = "waits for a CU behind the persistent launch (its own work: the min)" if avg > 5 * mn and mn < 50e3 else ""
        print(f"| `{short(n)}` | {calls} | {t/1e6:.2f} | {avg/1e3:.1f} | {mn/1e3:.1f} | {100*t/tot:.1f} | {note} |")


if __name__ == "__main__":
    main(sys.argv[1], " ".join(sys.argv[2:]))
