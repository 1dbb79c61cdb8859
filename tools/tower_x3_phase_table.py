"""Per-conv table of the bf16x3 tower bodies from the phase stamps of tools/x3_lab (profiles/r03_tower_x3_phases.txt): algorithmic
FLOPs of each conv over the time of its WHOLE phase (fragment reads, MFMAs, GroupNorm statistics and the normalise / split pass),
as a fraction of one CU's dense bf16 MFMA peak (2.5 PFLOP/s / 256), and the same in MFMA-issue terms (compensated bf16 issues
three MFMAs per product).  Usage: python tools/tower_x3_phase_table.py <phases.txt> [--json]"""
import json, re, sys


def table(path):
    txt = open(path).read()
    ph = {m.group(1).strip(): float(m.group(2)) for m in re.finditer(r"^\s+(.+?)\s+(\d+) cycles$", txt, re.M)}
    us = float(re.search(r"([\d.]+) us per launch", txt).group(1))
    tot = [float(x) for x in re.findall(r"total (\d+) cycles per item", txt)]
    # 768 items on 256 workgroups = 3 rounds: one round of each body + half of each
    clk = 1.5 * (tot[0] + tot[1]) / us / 1e3                     # GHz under load (cycles per microsecond / 1000)
    peak = 2500e12 / 256                                          # FLOP/s per CU
    mf = lambda px, co, k: 2.0 * px * co * k
    rows = []

    def add(name, flops, keys):
        cyc = sum(ph[k] for k in keys)
        sec = cyc / (clk * 1e9)
        f = flops / sec / peak
        rows.append({"conv": name, "MFLOP": round(flops / 1e6, 1), "cycles": int(cyc), "us": round(sec * 1e6, 2),
                     "frac_mfma_algorithmic": round(f, 4), "frac_mfma_issue": round(3 * f, 4)})
    add("stem 7x7 (3 -> 16 @ 64x64; rgb)", mf(4096, 16, 147), ["stem fill", "stem mma", "stem stats", "stem apply"])
    for i in (1, 2, 3, 4):
        tail = "apply+res" if i in (2, 4) else "apply"
        add(f"layer1 conv {i} 3x3 (16 -> 16 @ 64x64)", mf(4096, 16, 144), [f"c{i} mma h0", f"c{i} load h1", f"c{i} mma h1", f"c{i} stats", f"c{i} {tail}"])
    add("layer2 entry: 3x3 s2 16 -> 32 + 1x1 s2 downsample", mf(1024, 32, 144) + mf(1024, 32, 16),
        ["entry2 load h0", "entry2 mma h0", "entry2 load h1", "entry2 mma h1", "entry2 stats+apply"])
    for i in (1, 2, 3):
        add(f"layer2 conv {i} 3x3 (32 -> 32 @ 32x32)", mf(1024, 32, 288), [f"conv32 #{i}"])
    add("layer3 entry: 3x3 s2 32 -> 64 + downsample", mf(256, 64, 288) + mf(256, 64, 32), ["layer3 entry"])
    for i in (1, 2, 3):
        add(f"layer3 conv {i} 3x3 (64 -> 64 @ 16x16)", mf(256, 64, 576), [f"conv64 #{i}"])
    add("layer4 entry: 3x3 s2 64 -> 128 + downsample", mf(64, 128, 576) + mf(64, 128, 64), ["layer4 entry"])
    for i in (1, 2, 3):
        add(f"layer4 conv {i} 3x3 (128 -> 128 @ 8x8)", mf(64, 128, 1152), [f"conv128 #{i}"])
    whole = sum(r["MFLOP"] for r in rows) * 1e6 / ((tot[0] + tot[1]) / (clk * 1e9)) / peak
    return {"source": path.split("/")[-1], "launch_us": us, "clock_GHz_under_load": round(clk, 2), "rows": rows,
            "whole_tower_frac_mfma_algorithmic": round(whole, 4), "whole_tower_frac_mfma_issue": round(3 * whole, 4)}


if __name__ == "__main__":
    t = table(sys.argv[1])
    if "--json" in sys.argv:
        print(json.dumps(t))
    else:
        print(f"bf16x3 tower bodies, per workgroup (one image on one CU), {t['launch_us']} us per 384-image launch, {t['clock_GHz_under_load']} GHz under load\n")
        print("| conv (whole phase incl. GroupNorm statistics + normalise/split) | MFLOP | cycles | us | of CU MFMA peak (algorithmic) | MFMA issue (x3) |\n|---|---|---|---|---|---|")
        for r in t["rows"]:
            print(f"| {r['conv']} | {r['MFLOP']} | {r['cycles']} | {r['us']} | {100 * r['frac_mfma_algorithmic']:.1f} % | {100 * r['frac_mfma_issue']:.1f} % |")
        print(f"\nwhole tower: {100 * t['whole_tower_frac_mfma_algorithmic']:.1f} % algorithmic, {100 * t['whole_tower_frac_mfma_issue']:.1f} % of the MFMA issue peak")
