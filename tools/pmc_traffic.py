"""Builds profiles/r01_pmc_traffic.json from the two rocprofv3 --pmc passes over tools/roofline_probe.py
(FETCH_SIZE and WRITE_SIZE, separate runs, csv output): per-launch averages for the GEMM and direct-conv kernels with the
gfx950 correction of MI355X_MICROARCH.md (FETCH_SIZE counts half of wide coalesced reads -> x2; WRITE_SIZE exact).
Usage: python tools/pmc_traffic.py <fetch_counter_collection.csv> <write_counter_collection.csv> <out.json>"""
import csv, json, sys


def per_kernel(path, counter):
    acc = {}
    for r in csv.DictReader(open(path)):
        if r.get("Counter_Name") != counter:
            continue
        name = r["Kernel_Name"]
        probe_gemm = "g2_kernel" in name and ("ILi64ELi128ELi2ELi4ELi2ELi512ELb0E" in name or "<64, 128, 2, 4, 2, 512, false" in name)
        # the text tower goes out as two launches (4-way and 2-way column split); the work list lets one of them run, the other
        # one's workgroups exit at once -- keyed apart, the busy one is reported as `clip_tower`
        clip = None
        if "clip_tower_kernel" in name:
            clip = "clip_tower_4way" if ("ELi4E" in name or ", 4>" in name) else "clip_tower_2way"
        key = ("gemm" if probe_gemm else "dconv" if "dconv3x3_kernel" in name else
               "tower_x3" if "tower_x3_kernel" in name else clip)
        if key is None:
            continue
        a = acc.setdefault(key, [0.0, 0, name])
        a[0] += float(r["Counter_Value"]); a[1] += 1
    return {k: (v[0] / v[1], v[2]) for k, v in acc.items()}


def mfma_util(path):
    """MfmaUtil = SQ_VALU_MFMA_BUSY_CYCLES / (cycles x 1024 SIMDs) with cycles = GRBM_GUI_ACTIVE / 8 (rocprofv3 reports the
    sum over the 8 XCDs, MI355X_MICROARCH.md), per kernel class, from the third pass."""
    busy, act = per_kernel(path, "SQ_VALU_MFMA_BUSY_CYCLES"), per_kernel(path, "GRBM_GUI_ACTIVE")
    out = {k: 100.0 * busy[k][0] / (act[k][0] / 8.0 * 1024.0) for k in busy if k in act and act[k][0] > 0}
    hot = max((k for k in busy if k.startswith("clip_tower_")), key=lambda k: busy[k][0], default=None)
    if hot in out:
        out["clip_tower"] = out[hot]
    return out


def main(fetch_csv, write_csv, out, mfma_csv=None):
    sys.path.insert(0, __file__.rsplit("/", 1)[0])
    f, w = per_kernel(fetch_csv, "FETCH_SIZE"), per_kernel(write_csv, "WRITE_SIZE")
    # tower_x3: the images read once (uint8 rgb, fp32 depth: 3 pairs x 64 samples) + the layer-4 planes written; clip_tower: the
    # 16-bit weights of 12 blocks read once (what a perfect cache hierarchy would fetch) + tokens / embedding rows / outputs
    alg = {"gemm": 2464 * 512 * 2 + 2048 * 512 * 2 + 2464 * 2048 * 2, "dconv": 384 * 64 * 64 * 16 * 2 * 2,
           "tower_x3": 3 * 64 * (128 * 128 * 3 + 128 * 128 * 4) + 384 * 8192 * 4, "clip_tower": 12 * 3145728 * 2 + 2501 * 512 * 4 * 2}
    res = {"command": "rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --kernel-trace --output-format csv -- python tools/roofline_probe.py "
                      "(two separate passes), summarised by tools/pmc_traffic.py",
           "gfx950_note": "MI355X_MICROARCH.md HBM section: FETCH_SIZE (KB) reports half the bytes of wide (16 B/lane) coalesced "
                          "reads incl. global_load...lds -> x2; WRITE_SIZE (KB) is exact for wide stores"}
    for side in (f, w):
        busy = max((k for k in side if k.startswith("clip_tower_")), key=lambda k: f.get(k, (0,))[0], default=None)
        if busy is not None:
            side["clip_tower"] = side[busy]
    for k in ("gemm", "dconv", "tower_x3", "clip_tower"):
        if k not in f or k not in w:
            continue
        fk, name = f[k]; wk, _ = w[k]
        res[k] = {"kernel": name[:120], "FETCH_SIZE_KB": fk, "WRITE_SIZE_KB": wk, "traffic_bytes": (2 * fk + wk) * 1024,
                  "algorithmic_bytes": alg[k]}
    if mfma_csv:
        for k, v in mfma_util(mfma_csv).items():
            if k in res:
                res[k]["MfmaUtil_percent"] = round(v, 2)
        res["mfma_note"] = ("MfmaUtil = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs), third rocprofv3 --pmc pass over "
                            "tools/roofline_probe.py")
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main(*sys.argv[1:5])
