"""Per-launch HBM/fabric traffic (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, two separate passes over `bench.py --config gru --steps 1
--warmup 0`) of the cfg2 update's new kernels against their algorithmic bytes (X and dY read once, fp32; R = 1200 rows per minibatch).
Usage: python tools/pmc_gru.py <fetch.csv> <write.csv> <out.json>"""
import csv, json, re, sys


def per_kernel(path, counter):
    acc = {}
    for r in csv.DictReader(open(path)):
        if r.get("Counter_Name") != counter:
            continue
        n = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"]); n = re.sub(r"^void ", "", n).split("(")[0]
        if not any(k in n for k in ("conv_dw_kernel", "conv_dx_kernel", "gru_seq", "conv_dw_reduce")):
            continue
        a = acc.setdefault(n, [0.0, 0]); a[0] += float(r["Counter_Value"]); a[1] += 1
    return {k: (v[0] / v[1], v[1]) for k, v in acc.items()}


R = 1200
MB = lambda *dims: R * 4.0 * __import__("math").prod(dims) / 1e6
ALG = {  # MB per launch: input + output gradient of the conv, read once (mean of the two layers that share an instance)
    "conv_dw_kernel<2, 1, 4>": MB(257, 101, 2) + MB(63, 24, 32),
    "conv_dw_kernel<2, 2, 4>": MB(128, 128, 4) + MB(31, 31, 32),
    "conv_dw_kernel<4, 4, 4>": 0.5 * (MB(63, 24, 32) + MB(30, 11, 64) + MB(31, 31, 32) + MB(14, 14, 64)),
    "conv_dw_kernel<4, 5, 4>": 0.5 * (MB(30, 11, 64) + MB(28, 9, 64) + MB(14, 14, 64) + MB(6, 6, 64)),
    # data gradient: dY read, mask read, dX written
    "conv_dx_kernel<2>": 0.5 * (MB(30, 11, 64) + 2 * MB(63, 24, 32) + MB(14, 14, 64) + 2 * MB(31, 31, 32)),
    "conv_dx_kernel<4>": 0.5 * (MB(28, 9, 64) + 2 * MB(30, 11, 64) + MB(6, 6, 64) + 2 * MB(14, 14, 64)),
}
f, w = per_kernel(sys.argv[1], "FETCH_SIZE"), per_kernel(sys.argv[2], "WRITE_SIZE")
out = {"command": "rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE --kernel-trace --output-format csv -- python bench.py --config gru --steps 1 "
                  "--warmup 0 --no-cpu-baseline --no-roofline (two passes)",
       "note": "FETCH_SIZE / WRITE_SIZE in KB as reported (MI355X_MICROARCH.md: FETCH_SIZE counts half of wide 16 B/lane reads; these kernels "
               "are corrected x2 like the others); algorithmic = every operand byte once"}
for k in sorted(f):
    fk, n = f[k]; wk = w.get(k, (0.0, 0))[0]
    rec = {"launches": n, "FETCH_SIZE_MB": round(fk / 1024, 1), "WRITE_SIZE_MB": round(wk / 1024, 1)}
    if k in ALG:
        rec["algorithmic_MB"] = round(ALG[k], 1)
        rec["traffic_over_algorithmic_with_x2_fetch_correction"] = round((2 * fk + wk) / 1024 / ALG[k], 2)     # MI355X_MICROARCH.md: FETCH_SIZE counts half
    out[k] = rec
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(json.dumps(out, indent=1))
