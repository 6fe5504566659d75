"""register / scratch / LDS / occupancy of every gfx950 kernel of the product library, from hipcc's own resource remarks
(-Rpass-analysis=kernel-resource-usage).  Runs on the build container, no GPU needed.
usage: python scripts/kernel_resources.py [tag]   ->   profiles/<tag>_kernel_resources.txt"""
import glob
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r04"
head = subprocess.run(["git", "rev-parse", "--short", "HEAD"], cwd=ROOT, capture_output=True, text=True).stdout.strip()
rows = []
for f in sorted(glob.glob(os.path.join(ROOT, "dpc_amd/csrc/*.hip"))):
    r = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-I" + os.path.join(ROOT, "include"), "-c", f,
                        "-o", "/dev/null", "-Rpass-analysis=kernel-resource-usage"], capture_output=True, text=True)
    name, rec = None, {}
    def flush():
        if name:
            rows.append((os.path.basename(f), name, dict(rec)))
    for line in r.stderr.splitlines():
        m = re.search(r"remark:\s+(.*?):\s+(\S+)\s+\[-Rpass", line)
        if not m:
            continue
        k, v = m.group(1).strip(), m.group(2)
        if k == "Function Name":
            flush()
            name, rec = v, {}
        else:
            rec[k] = v
    flush()
dem = subprocess.run(["c++filt"], input="\n".join(n for _, n, _ in rows), capture_output=True, text=True).stdout.splitlines()
out = os.path.join(ROOT, "profiles", f"{tag}_kernel_resources.txt")
with open(out, "w") as o:
    o.write(f"# hipcc --offload-arch=gfx950 -O3 -Rpass-analysis=kernel-resource-usage over dpc_amd/csrc/*.hip (commit {head}); "
            f"scratch in bytes per lane, LDS = static bytes per workgroup (dynamic LDS not included), occ = waves per SIMD\n")
    o.write("%-22s %-118s %5s %5s %8s %9s %7s %4s\n" % ("file", "kernel", "VGPR", "AGPR", "scratch", "sgprSpill", "LDS", "occ"))
    for (fn, _, rec), d in zip(rows, dem):
        d = re.sub(r"\(.*", "", d).replace("void ", "").replace("(anonymous namespace)::", "")
        o.write("%-22s %-118s %5s %5s %8s %9s %7s %4s\n" % (fn, d[:118], rec.get("VGPRs", "?"), rec.get("AGPRs", "?"), rec.get("ScratchSize [bytes/lane]", "?"),
                                                            rec.get("SGPRs Spill", "?"), rec.get("LDS Size [bytes/block]", "?"), rec.get("Occupancy [waves/SIMD]", "?")))
print(out, len(rows), "kernels")
