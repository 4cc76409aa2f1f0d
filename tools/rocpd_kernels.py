"""List kernel dispatches (name, duration) from a rocprofv3 rocpd sqlite file, in dispatch order or as a summary.
usage: python tools/rocpd_kernels.py <results.db> [--list]"""
import sqlite3, sys, re
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if t.startswith('rocpd_kernel_dispatch')][0]
ks = [t for t in tabs if t.startswith('rocpd_info_kernel_symbol')][0]
scol = [r[1] for r in cur.execute(f"pragma table_info({ks})")]
name = 'display_name' if 'display_name' in scol else 'kernel_name'
rows = list(cur.execute(f"select s.{name}, d.start, d.end, d.grid_size_x from {kd} d join {ks} s on d.kernel_id = s.id order by d.start"))
short = lambda n: re.sub(r"\(anonymous namespace\)::", "", n)[:90]
if '--list' in sys.argv:
    for n, s, e, g in rows: print(f"{(e-s)/1e3:10.1f} us  grid {g:8d}  {short(n)}")
else:
    agg = {}
    for n, s, e, g in rows:
        a = agg.setdefault(short(n), [0, 0.0]); a[0] += 1; a[1] += (e - s) / 1e3
    tot = sum(a[1] for a in agg.values())
    for n, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print(f"{t:12.1f} us {100*t/tot:5.1f}%  calls {c:5d}  avg {t/c:10.1f} us  {n}")
