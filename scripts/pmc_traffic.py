"""HBM traffic of the dominant kernel family from rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE collected in SEPARATE --pmc runs,
KiB units).  Correction per /opt/skills/guides/MI355X_MICROARCH.md section HBM: on gfx950 FETCH_SIZE reports exactly 1/2 of the bytes
of wide (16 B/lane) coalesced streaming reads -- every load of these kernels is such a load -- so the read side is doubled;
WRITE_SIZE is taken as is.

Two users: `python scripts/pmc_traffic.py <fetch.db> <write.db> <tag>` (scripts/gpu_pmc_traffic.sh) writes
profiles/<tag>_pmc_traffic.json + the per-kernel table; bench.py imports `summarise` for the passes it runs itself."""
import glob
import hashlib
import json
import os
import re
import sqlite3
import sys

FAMILIES = (("dpc_conv_igemm", r"igemm_kernel|igemm_ws_kernel|igemm_wsp_kernel|conv_halo_kernel|conv_halo_ws_kernel"),
            ("dpc_conv_wgrad", r"wgrad_kernel|wgrad2_kernel|wgrad_patch_kernel|wgrad_stem_kernel"))


def per_kernel(path, counter):
    """{kernel name: (sum of the counter over its dispatches, dispatches)} of one rocprofv3 database"""
    cur = sqlite3.connect(path).cursor()
    out = {}
    for kn, v, n in cur.execute("select kernel_name, sum(value), count(distinct dispatch_id) from counters_collection where counter_name=? group by kernel_name", (counter,)):
        out[re.sub(r"\(.*", "", kn.replace("(anonymous namespace)::", "")).replace("void ", "")] = (v, n)
    return out


def csrc_sha16(root="."):
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(root, "dpc_amd", "csrc", "*"))):
        if f.endswith((".hip", ".h")):
            h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def summarise(fetch, write):
    """per kernel family: launches, raw counters, corrected HBM bytes per launch"""
    res = {}
    for fam, pat in FAMILIES:
        f = sum(v for k, (v, n) in fetch.items() if re.search(pat, k))
        w = sum(v for k, (v, n) in write.items() if re.search(pat, k))
        n = sum(n for k, (v, n) in fetch.items() if re.search(pat, k))
        res[fam] = {"launches": n, "fetch_KiB_raw": f, "write_KiB": w,
                    "hbm_bytes_per_launch": (2.0 * f + w) * 1024.0 / max(n, 1),
                    "note": "FETCH_SIZE x2 (gfx950 wide-load correction), WRITE_SIZE as reported; separate --pmc passes"}
    return res


def main(argv):
    fetch = per_kernel(argv[1], "FETCH_SIZE")
    write = per_kernel(argv[2], "WRITE_SIZE")
    tag = argv[3]
    res = summarise(fetch, write)
    # stamp: which kernel sources the counters were collected on (bench.py prints it and flags a stale file)
    res["csrc_sha16"] = csrc_sha16()
    res["git_head"] = os.environ.get("DPC_GIT_HEAD") or (open(".git_head").read().strip() if os.path.exists(".git_head") else None)
    # per kernel: bytes per launch, read side corrected as above (where the HBM traffic of the step goes, kernel by kernel)
    rows = []
    for k in sorted(set(fetch) | set(write)):
        fv, fn = fetch.get(k, (0.0, 0))
        wv, wn = write.get(k, (0.0, 0))
        n = max(fn, wn, 1)
        rows.append((2.0 * fv * 1024.0 + wv * 1024.0, n, 2.0 * fv * 1024.0 / n, wv * 1024.0 / n, k))
    rows.sort(reverse=True)
    with open(f"profiles/{tag}_pmc_traffic_per_kernel.txt", "w") as f:
        tot = sum(r[0] for r in rows)
        f.write(f"# HBM traffic per kernel over the profiled steps (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes; reads x2: gfx950 "
                f"wide-load correction); total {tot / 1e9:.2f} GB; csrc {res['csrc_sha16']} head {res['git_head']}\n")
        f.write(f"{'kernel':92s} {'launches':>8s} {'read MB/launch':>15s} {'write MB/launch':>16s} {'total GB':>9s}\n")
        for t, n, r, w, k in rows[:60]:
            f.write(f"{k[:92]:92s} {n:8d} {r / 1e6:15.1f} {w / 1e6:16.1f} {t / 1e9:9.2f}\n")
    json.dump(res, open(f"profiles/{tag}_pmc_traffic.json", "w"), indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main(sys.argv)
