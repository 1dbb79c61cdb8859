"""Per-conv table of the FUSED tower kernels from the phase stamps of tools/head_lab.hip / tools/tail_lab.hip
(profiles/*_tower_head_phases.txt, *_tower_tail_phases.txt): algorithmic FLOPs of each conv (2 M N K, K = kh kw cin of the
reference layer) over the time its phase takes in one workgroup -- conv + GroupNorm statistics + normalise/apply, i.e.
everything the launch-per-layer path spent two kernels on -- scaled to the 256 CUs.
Usage: python tools/tower_phase_table.py <head_phases.txt> <tail_phases.txt>"""
import re, sys

def phases(path):
    out, total = [], None
    for ln in open(path):
        m = re.match(r"\s+(.+?)\s+([\d.]+) %\s+\((\d+) ticks\)", ln)
        if m:
            out.append((m.group(1).strip(), int(m.group(3))))
        m = re.match(r"\s+workgroup total (\d+) ticks", ln)
        if m:
            total = int(m.group(1))
        m = re.match(r"B=\d+ G=\d+.*: ([\d.]+) us per launch \((\d+) workgroups\)", ln)
        if m:
            us, wgs = float(m.group(1)), int(m.group(2))
    rounds = -(-wgs // 256)
    ghz = total / (us / rounds * 1e3)               # ticks per ns: one workgroup runs us / rounds
    return dict(out), total, ghz

head, ht, hg = phases(sys.argv[1])
tail, tt, tg = phases(sys.argv[2])
conv = lambda px, cout, k: 2.0 * px * cout * k
rows = [
    ("stem 7x7 (3 -> 16, 64x64; rgb)", conv(4096, 16, 147), head, ("stem conv", "stem stats", "a0 write"), hg),
    ("layer1 conv 3x3 (16 -> 16) x4", 4 * conv(4096, 16, 144), head, ("conv1", "stats1", "apply1+conv2", "stats2", "apply2+conv3", "stats3", "apply3+conv4", "stats4", "apply4"), hg),
    ("layer2.0 downsample 1x1 s2 (16 -> 32)", conv(1024, 32, 16), head, ("l2 downsample",), hg),
    ("layer2.0.conv1 3x3 s2 (16 -> 32)", conv(1024, 32, 144), head, ("l2 conv s2",), hg),
    ("layer2 conv 3x3 (32 -> 32) x3", 3 * conv(1024, 32, 288), head, ("l2 conv2", "l2 conv3", "l2 conv4"), hg),
    ("layer3.0 downsample 1x1 s2 (32 -> 64)", conv(256, 64, 32), tail, ("l3 downsample",), tg),
    ("layer3.0.conv1 3x3 s2 (32 -> 64)", conv(256, 64, 288), tail, ("l3 conv s2",), tg),
    ("layer3 conv 3x3 (64 -> 64) x3", 3 * conv(256, 64, 576), tail, ("l3 conv2", "l3 conv3", "l3 conv4"), tg),
    ("layer4.0 downsample 1x1 s2 (64 -> 128)", conv(64, 128, 64), tail, ("l4 downsample",), tg),
    ("layer4.0.conv1 3x3 s2 (64 -> 128)", conv(64, 128, 576), tail, ("l4 conv s2",), tg),
    ("layer4 conv 3x3 (128 -> 128) x3", 3 * conv(64, 128, 1152), tail, ("l4 conv2", "l4 conv3", "l4 conv4"), tg),
]
print("| conv(s) of one tower-image | MFLOP | us in its workgroup (conv + GroupNorm + apply) | TFLOP/s x 256 CUs | of 2.5 PF |")
print("|---|---|---|---|---|")
tf_all = us_all = 0.0
for name, fl, tab, keys, ghz in rows:
    us = sum(tab[k] for k in keys) / ghz / 1e3
    tf = fl / (us * 1e-6) / 1e12 * 256
    tf_all += fl; us_all += us
    print(f"| {name} | {fl / 1e6:.1f} | {us:.2f} | {tf:.0f} | {tf / 2500:.1%} |")
other = (ht / hg + tt / tg) / 1e3 - us_all
print(f"| image load / preprocessing / output store | - | {other:.2f} | - | - |")
print(f"| whole tower (one workgroup per image per kernel) | {tf_all / 1e6:.1f} | {us_all + other:.2f} | {tf_all / ((us_all + other) * 1e-6) / 1e12 * 256:.0f} | {tf_all / ((us_all + other) * 1e-6) / 1e12 * 256 / 2500:.1%} |")
print(f"\n(clock from the stamps: {hg:.2f} / {tg:.2f} GHz; 384 workgroups = 1.5 rounds of 256 CUs at the rollout batch, so the launch-level rate is 3/4 of the last row)")
