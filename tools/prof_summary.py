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
        return [(r[0], r[1], float(r[2]), float(r[3])) for r in c.execute(
            "select name, count(*), sum(end-start), avg(end-start) from kernels group by name")]
    return [(r["Name"], int(r["Calls"]), float(r["TotalDurationNs"]), float(r["AverageNs"])) for r in csv.DictReader(open(path))]


def main(path, title=""):
    rows = sorted(rows_from(path), key=lambda r: -r[2])
    tot = sum(r[2] for r in rows)
    print(f"# {title}\n\ntotal kernel time {tot/1e6:.1f} ms over {sum(r[1] for r in rows)} launches\n")
    print("| kernel | calls | total ms | avg us | % |\n|---|---|---|---|---|")
    for n, calls, t, avg in rows[:45]:
        print(f"| `{short(n)}` | {calls} | {t/1e6:.2f} | {avg/1e3:.1f} | {100*t/tot:.1f} |")


if __name__ == "__main__":
    main(sys.argv[1], " ".join(sys.argv[2:]))
