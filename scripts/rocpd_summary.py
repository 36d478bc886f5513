"""Per-kernel totals from a rocprofv3 results database: python scripts/rocpd_summary.py <db> [steps]"""
import re, sqlite3, sys
db = sqlite3.connect(sys.argv[1])
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rows = list(db.execute("select name, count(*), sum(end-start) from kernels group by name order by 3 desc"))
tot = sum(r[2] for r in rows)
print(f"total {tot/1e6/steps:.2f} ms/step over {steps} steps")
for n, c, t in rows[:45]:
    n = re.sub(r"\(.*", "", n.replace("(anonymous namespace)::", "").replace("void ", ""))[:110]
    print(f"{t/1e6/steps:8.2f} ms {100*t/tot:5.1f}% {c/steps:8.1f}  {n}")
