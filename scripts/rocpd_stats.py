"""Summarise a rocprofv3 rocpd sqlite database (kernel-trace) into a per-kernel stats table
(name, calls, total ms, avg us, % of GPU kernel time) -- what `--stats` prints in csv mode."""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"\(.*", "", name.replace("(anonymous namespace)::", ""))
    name = name.replace("void ", "")
    return name[:110]


def main(path, steps=None):
    db = sqlite3.connect(path)
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
    rows = list(cur.execute(f"select {name_col}, start, end from kernels"))
    agg = {}
    for n, s, e in rows:
        d = agg.setdefault(short(n), [0, 0])
        d[0] += 1
        d[1] += e - s
    total = sum(v[1] for v in agg.values())
    print(f"# {path}: {len(rows)} kernel dispatches, {total / 1e6:.3f} ms of kernel time" + (f" over {steps} timed+warmup steps" if steps else ""))
    print(f"{'kernel':112s} {'calls':>7s} {'total_ms':>10s} {'avg_us':>10s} {'pct':>6s}")
    for n, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print(f"{n:112s} {c:7d} {t / 1e6:10.3f} {t / 1e3 / c:10.2f} {100.0 * t / total:6.2f}")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None)
