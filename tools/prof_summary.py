"""Turn a rocprofv3 `*_kernel_stats.csv` into the markdown table kept under profiles/."""
import csv, sys, re

def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"^void ", "", n)
    return n.split("(")[0][:70]

def main(path, title=""):
    rows = list(csv.DictReader(open(path)))
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    print(f"# {title}\n")
    print(f"total kernel time {tot/1e6:.1f} ms\n")
    print("| kernel | calls | total ms | avg us | % |\n|---|---|---|---|---|")
    for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:40]:
        t = float(r["TotalDurationNs"])
        print(f"| `{short(r['Name'])}` | {r['Calls']} | {t/1e6:.2f} | {float(r['AverageNs'])/1e3:.1f} | {100*t/tot:.1f} |")

if __name__ == "__main__":
    main(sys.argv[1], " ".join(sys.argv[2:]))
