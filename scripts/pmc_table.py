"""per-kernel PMC totals from a rocprofv3 rocpd database (one row per kernel name, counters summed over dispatches)"""
import re, sqlite3, sys
from collections import defaultdict
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
pat = sys.argv[2] if len(sys.argv) > 2 else "."
cols = [r[1] for r in cur.execute("pragma table_info(counters_collection)")]
q = "select kernel_name, counter_name, sum(value), count(distinct dispatch_id) from counters_collection group by kernel_name, counter_name"
agg = defaultdict(dict); calls = {}
for kn, cn, v, n in cur.execute(q):
    nm = re.sub(r"\(.*", "", kn.replace("(anonymous namespace)::", "")).replace("void ", "").replace("unsigned short", "bf16")
    agg[nm][cn] = v; calls[nm] = n
names = sorted({c for d in agg.values() for c in d})
print("kernel".ljust(60), "calls", *[n[-18:].rjust(19) for n in names])
for nm, d in sorted(agg.items(), key=lambda kv: -max(kv[1].values())):
    if re.search(pat, nm):
        print(nm[:60].ljust(60), str(calls[nm]).rjust(5), *[f"{d.get(n, 0):19.4g}" for n in names])
