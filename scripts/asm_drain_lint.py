"""Does hipcc drain an LDS-DMA prefetch that is meant to stay in flight?  (round 6)

The loader / helper waves of the ring kernels keep several LDS-DMA pieces outstanding across loop iterations and retire them with
hand-counted ``s_waitcnt vmcnt(N)`` (inline asm).  Everything hipcc tracks itself -- ordinary global loads whose results are carried
over the loop edge, register copies of such results, loads behind divergent branches -- can make its wait-count pass emit
``s_waitcnt vmcnt(0)`` INSIDE such a loop, which silently drains the ring once per iteration: conv_halo_ws_kernel's residual /
fused-reduction variants ran 3.3 us per tile instead of 1.45 for five rounds because of one (profiles/r06_halo_epi_probe.txt).

This lint reads ``hipcc -S`` output and reports, per kernel, every compiler-inserted ``vmcnt(0)`` (outside ;;#ASMSTART / ;;#ASMEND
blocks) that sits in an INNERMOST loop which also issues LDS-DMA (``buffer_load ... lds`` / ``global_load_lds``).

    python scripts/asm_drain_lint.py file.s [...]            lint assembly files
    python scripts/asm_drain_lint.py --build [src.hip ...]   compile (default: the ring kernels) and lint
"""
import os
import re
import subprocess
import sys
import tempfile

RING_SOURCES = ["conv_halo.hip", "conv_igemm_ws.hip", "conv_wgrad_patch.hip", "conv_wgrad_stem.hip", "gemm_ws.hip", "score_fused.hip"]
# loops that MAY drain: (kernel substring, reason).  The general (first / last three intervals) form of conv_halo_ws's helper loop
# keeps its conditions -- it is the steady-state loop that must not drain; both live in the same kernel, so that kernel is judged
# by its FIRST DMA loop (the steady one precedes the tail in program order).
def demangle(names):
    if not names:   # (c++filt without arguments reads stdin)
        return {}
    try:
        out = subprocess.run(["c++filt"] + names, capture_output=True, text=True, stdin=subprocess.DEVNULL, timeout=60).stdout.split("\n")
        return dict(zip(names, out))
    except Exception:
        return {n: n for n in names}


def inner_loop_label(lines, i):
    """the label of line i if it heads an INNERMOST loop.  hipcc annotates `.LBBx_y: ; =>This Inner Loop Header` on the label line for a
    top-level loop, and on a comment-only continuation line (`;   Parent Loop ...` / `; =>  This Inner Loop Header: Depth=2`) for a
    nested one -- the K loops of the persistent kernels are nested in their tile loops (the first version of this lint missed them)"""
    m = re.match(r"^(\.LBB\d+_\d+):(.*)", lines[i])
    if not m:
        return None
    note = m.group(2)
    j = i + 1
    while j < len(lines) and re.match(r"^\s*;", lines[j]) and "ASMSTART" not in lines[j] and "ASMEND" not in lines[j]:
        note += lines[j]
        j += 1
    return m.group(1) if "Inner Loop Header" in note else None


def lint_file(path):
    """-> list of (kernel, first line of the loop, line of the wait, number of DMA instructions in the loop)"""
    lines = open(path).read().split("\n")
    hits = []
    kern, start = None, 0
    bounds = []
    for i, l in enumerate(lines):
        m = re.match(r"^(_Z\w+):", l)
        if m:
            if kern:
                bounds.append((kern, start, i))
            kern, start = m.group(1), i
    if kern:
        bounds.append((kern, start, len(lines)))
    for kern, a, b in bounds:
        # innermost loops as hipcc annotates them: a block label with "Inner Loop Header" up to the backward branch to it
        i = a
        while i < b:
            label = inner_loop_label(lines, i)
            if not label:
                i += 1
                continue
            end = None
            for j in range(i + 1, b):
                if re.search(r"s_cbranch\w*\s+" + re.escape(label) + r"\b|s_branch\s+" + re.escape(label) + r"\b", lines[j]):
                    end = j
            if end is None:
                i += 1
                continue
            body = lines[i:end + 1]
            dma = sum(1 for x in body if re.search(r"buffer_load_dword\w*.*\blds\b|global_load_lds", x))
            if dma:
                inasm = False
                for off, x in enumerate(body):
                    if "ASMSTART" in x:
                        inasm = True
                    elif "ASMEND" in x:
                        inasm = False
                    elif not inasm and re.search(r"s_waitcnt.*vmcnt\(0\)", x):
                        hits.append((kern, i + 1, i + off + 1, dma))
            i = end + 1
    return hits


def lint_file_reads(path, skip=("conv_halo_kernel", "score_fwd_kernel", "score_bwd_kernel")):
    """Second class (late round 6): a compiler-inserted ``lgkmcnt(0)`` in an innermost loop whose LDS fragment reads are hand-issued
    (inline asm) and retired with hand-counted ``lgkmcnt(N)`` -- it waits for ALL of them where the next MFMA needs the oldest two.
    conv_halo_ws_kernel had one in front of every tile's first MFMA: its register-resident weights were loaded through generic
    pointers (flat loads count on vmcnt AND lgkmcnt) and first used inside the tile loop.  A wait directly in front of a barrier is
    what the code asks for and is not reported; `skip`: kernels that are not pipelined across iterations (one-barrier-per-chunk
    generic forms).  -> list of (kernel, first line of the loop, line of the wait, number of hand-issued reads in the loop)"""
    lines = open(path).read().split("\n")
    hits = []
    kern, start, bounds = None, 0, []
    for i, l in enumerate(lines):
        m = re.match(r"^(_Z\w+):", l)
        if m:
            if kern:
                bounds.append((kern, start, i))
            kern, start = m.group(1), i
    if kern:
        bounds.append((kern, start, len(lines)))
    for kern, a, b in bounds:
        if any(k in kern for k in skip):
            continue
        i = a
        while i < b:
            label = inner_loop_label(lines, i)
            if not label:
                i += 1
                continue
            end = None
            for j in range(i + 1, b):
                if re.search(r"s_cbranch\w*\s+" + re.escape(label) + r"\b|s_branch\s+" + re.escape(label) + r"\b", lines[j]):
                    end = j
            if end is None:
                i += 1
                continue
            body = lines[i:end + 1]
            inasm, reads, waits = False, 0, []
            for off, x in enumerate(body):
                if "ASMSTART" in x:
                    inasm = True
                elif "ASMEND" in x:
                    inasm = False
                elif inasm and re.search(r"\bds_read", x):
                    reads += 1
                elif not inasm and re.search(r"s_waitcnt.*lgkmcnt\(0\)", x):
                    nxt = [y for y in body[off + 1:off + 5] if y.strip() and "ASMSTART" not in y and "ASMEND" not in y and not y.strip().startswith(";")]
                    if not (nxt and "s_barrier" in nxt[0]):
                        waits.append(off)
            if reads:
                hits += [(kern, i + 1, i + off + 1, reads) for off in waits]
            i = end + 1
    return hits


def build_and_lint(root, srcs=None, defines=()):
    srcs = srcs or RING_SOURCES
    out = []
    with tempfile.TemporaryDirectory() as tmp:
        for s in srcs:
            asm = os.path.join(tmp, s.replace(".hip", ".s"))
            cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", f"-I{os.path.join(root, 'include')}", "-S",
                   "--cuda-device-only"] + [f"-D{d}" for d in defines] + [os.path.join(root, "dpc_amd", "csrc", s), "-o", asm]
            subprocess.run(cmd, check=True, capture_output=True, stdin=subprocess.DEVNULL, timeout=900)
            out += [(s,) + h for h in lint_file(asm)]
            out += [(s, h[0], h[1], h[2], -h[3]) for h in lint_file_reads(asm)]   # (negative count: the fragment-read class)
    names = demangle(sorted({h[1] for h in out}))
    return [(h[0], names.get(h[1], h[1]), h[2], h[3], h[4]) for h in out]


if __name__ == "__main__":
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if len(sys.argv) > 1 and sys.argv[1] == "--build":
        hits = build_and_lint(root, sys.argv[2:] or None)
    else:
        hits = []
        for f in sys.argv[1:]:
            hits += [(f,) + h for h in lint_file(f)]
            hits += [(f, h[0], h[1], h[2], -h[3]) for h in lint_file_reads(f)]
    for h in hits:
        if h[4] < 0:
            print(f"{h[0]}: {h[1][:100]}: compiler-inserted lgkmcnt(0) at line {h[3]} in the loop at line {h[2]} ({-h[4]} hand-issued LDS reads)")
            continue
        print(f"{h[0]}: {h[1][:100]}: compiler-inserted vmcnt(0) at line {h[3]} in the DMA loop at line {h[2]} ({h[4]} LDS-DMA instructions)")
    print(f"{len(hits)} drain(s)")
    sys.exit(1 if hits else 0)
