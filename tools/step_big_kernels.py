import sqlite3, sys
c=sqlite3.connect(sys.argv[1])
tabs=[r[0] for r in c.execute("select name from sqlite_master where type='table'")]
kd=[t for t in tabs if 'kernel_dispatch' in t][0]; ks=[t for t in tabs if 'kernel_symbol' in t][0]
rows=c.execute(f"select s.kernel_name, d.start, d.end from {kd} d join {ks} s on d.kernel_id=s.id order by d.start").fetchall()
# last step: find last tower_x3 kernel start
idx=[i for i,r in enumerate(rows) if 'tower_x3_kernel' in r[0]]
i0=idx[-3]; t0=rows[i0][1]
for n,s,e in rows[i0:]:
    if s-t0 > 2.3e6: break
    if (e-s) > 15e3: print("%8.1f -> %8.1f us (%7.1f)  %s" % ((s-t0)/1e3,(e-t0)/1e3,(e-s)/1e3,n.split('(')[0][:80]))
