import sqlite3, sys, re
c = sqlite3.connect(sys.argv[1]).cursor()
cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
qcol = "queue_id" if "queue_id" in cols else "stream_id"
rows = list(c.execute(f"select start, end, name, {qcol} from kernels order by start"))
idx = [i for i, r in enumerate(rows) if "gru_seq_fwd_kernel<16>" in r[2] or "gru_seq_fwd_kernelILi16" in r[2]]
a, b = idx[200], idx[201]
t0 = rows[a][0]
for r in rows[a:b + 1]:
    n = re.sub(r"\(anonymous namespace\)::", "", r[2]); n = re.sub(r"^void ", "", n).split("(")[0][:44]
    print(f"{(r[0]-t0)/1e3:8.1f} {(r[1]-r[0])/1e3:7.1f}  q{r[3]}  {n}")
