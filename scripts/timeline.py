"""print the kernel timeline of the last train step in a rocpd database"""
import re, sqlite3, sys
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
pat = sys.argv[2] if len(sys.argv) > 2 else "igemm|wgrad"
rows = list(cur.execute("select name, start, end, grid_x, workgroup_x from kernels order by start"))
idx = [i for i, r in enumerate(rows) if 'pack_input_s2d' in r[0]]
last = rows[idx[-1]:]
t0 = last[0][1]
for n, s, e, g, w in last:
    nm = re.sub(r"\(.*", "", n.replace("(anonymous namespace)::", "")).replace("void ", "").replace("unsigned short", "bf16")
    if re.search(pat, nm):
        print(f"{(s - t0) / 1e6:9.3f} ms  {(e - s) / 1e3:9.1f} us  grid {g // max(w, 1):6d}  {nm[:80]}")
print("step span ms", (last[-1][2] - t0) / 1e6, "kernel busy ms", sum(e - s for _, s, e, _, _ in last) / 1e6)
