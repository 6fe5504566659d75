#!/usr/bin/env python3
"""Static lint of hipcc's gfx950 assembly for the hazard class that hid behind round 3's "LDS claim":

    an inline-asm LDS read (`ds_read_*` between ;;#ASMSTART / ;;#ASMEND) is invisible to hipcc's wait-count
    bookkeeping, and its VGPR destination counts as written at ;;#ASMEND.  If the value is dead afterwards (the
    end-of-tile "junk" fragment reads of the loader / compute kernels were), the register allocator hands the
    register to the next live range while the read is still in flight; the LDS data then lands ON TOP of the new
    value (write-after-write) whenever the LDS round trip takes longer than the instructions in between -- i.e.
    only under LDS contention, e.g. with a foreign workgroup on the CU.

The lint walks every kernel of a `--save-temps` .s file in program order (loop bodies once more along each backward
branch), keeps the in-order queue of LDS operations a wave has outstanding, retires them at every
`s_waitcnt lgkmcnt(N)` (asm or compiler: the hardware counter does not care who wrote the wait) and reports

    WAW  a compiler instruction or another asm statement writes a VGPR an in-flight asm read still targets
    RAW  an instruction reads such a VGPR before a wait has retired the read

Scalar / FLAT loads share the counter but can only make a counted wait MORE conservative, so they are ignored.
Usage:  asm_hazard_lint.py file.s [...]      exit status 1 when a hazard is found
        asm_hazard_lint.py --build           compile the csrc files that carry asm reads with -save-temps and lint them
"""
import os
import re
import subprocess
import sys
import tempfile

VREG = re.compile(r"\bv\[(\d+):(\d+)\]|\bv(\d+)\b")
LGKM = re.compile(r"lgkmcnt\((\d+)\)")
LABEL = re.compile(r"^(\.LBB[0-9_]+):")
KERNEL = re.compile(r"^(_Z\w+|[A-Za-z_]\w*):\s*(;.*)?$")
STORE_LIKE = ("ds_write", "ds_store", "global_store", "flat_store", "buffer_store", "scratch_store", "buffer_atomic", "global_atomic",
              "ds_add", "ds_min", "ds_max", "ds_and", "ds_or", "ds_xor", "s_", "v_cmp", "v_cmpx", ";", "buffer_wbl2", "buffer_inv",
              "v_nop", "ds_nop")


def regs_of(tok):
    out = set()
    for m in VREG.finditer(tok):
        if m.group(3) is not None:
            out.add(int(m.group(3)))
        else:
            out.update(range(int(m.group(1)), int(m.group(2)) + 1))
    return out


def split_ops(line):
    line = line.split(";")[0].strip()
    if not line:
        return None, []
    parts = line.split(None, 1)
    op = parts[0]
    ops = [o.strip() for o in parts[1].split(",")] if len(parts) > 1 else []
    return op, ops


def defs_uses(op, ops):
    """(written VGPRs, read VGPRs) of one instruction -- first operand is the destination except for store-like forms."""
    if op.startswith(STORE_LIKE) or not ops:
        return set(), set().union(*[regs_of(o) for o in ops]) if ops else set()
    is_lds_dma = op.startswith("buffer_load") and any(o.split()[-1:] == ["lds"] or " lds" in o for o in ops)
    if is_lds_dma or (op.startswith("global_load_lds")):
        return set(), set().union(*[regs_of(o) for o in ops])
    d = regs_of(ops[0])
    u = set().union(*[regs_of(o) for o in ops[1:]]) if len(ops) > 1 else set()
    if op.startswith("v_swap"):
        d |= regs_of(ops[1])
        u |= regs_of(ops[0])
    if op.startswith(("v_mfma", "v_smfmac")) or op.startswith(("v_fmac", "v_mac", "v_dot2c", "v_pk_fmac")):
        u |= regs_of(ops[0]) if op.startswith(("v_fmac", "v_mac", "v_dot2c", "v_pk_fmac", "v_smfmac")) else set()
    if op.startswith(("v_readlane", "v_readfirstlane")):
        return set(), u
    return d, u


class Kernel:
    def __init__(self, name):
        self.name = name
        self.ins = []      # (lineno, in_asm, op, ops, raw)
        self.labels = {}


def parse(path):
    kernels, cur, in_asm = [], None, False
    with open(path) as f:
        for no, raw in enumerate(f, 1):
            s = raw.rstrip("\n")
            m = KERNEL.match(s)
            if m and not s.startswith(".") and "@" in s:
                cur = Kernel(m.group(1))
                kernels.append(cur)
                continue
            if cur is None:
                continue
            if s.strip().startswith(".Lfunc_end"):
                cur = None
                continue
            lm = LABEL.match(s)
            if lm:
                cur.labels[lm.group(1)] = len(cur.ins)
                continue
            t = s.strip()
            if t.startswith(";;#ASMSTART"):
                in_asm = True
                continue
            if t.startswith(";;#ASMEND"):
                in_asm = False
                continue
            if not t or t.startswith((";", ".")):
                continue
            op, ops = split_ops(t)
            if op:
                cur.ins.append((no, in_asm, op, ops, t))
    return kernels


def lint_kernel(k, report, max_states=3_000_000):
    """Control-flow exact: a worklist over (instruction, queue of outstanding LDS operations).  A queue entry is
    (line of the operation, VGPRs an ASM read targets -- empty for compiler-visible operations)."""
    n = len(k.ins)
    visited = set()
    work = [(0, ())]
    QCAP = 20   # a wave never has more LDS operations in flight than this in these kernels; bounds the state space
    while work:
        i, queue = work.pop()
        while i < n:
            key = (i, queue)
            if key in visited:
                break
            visited.add(key)
            if len(visited) > max_states:
                raise RuntimeError(f"{k.name}: state space exceeds {max_states}")
            no, in_asm, op, ops, raw = k.ins[i]
            if op == "s_waitcnt":
                m = LGKM.search(raw)
                if m:
                    keep = int(m.group(1))
                    if len(queue) > keep:
                        queue = queue[len(queue) - keep:] if keep else ()
            elif op == "s_endpgm":
                break
            else:
                d, u = defs_uses(op, ops)
                for qno, regs in queue:
                    if regs and regs & d:
                        report("WAW", k.name, no, raw, qno, sorted(regs & d))
                    if regs and regs & u:
                        report("RAW", k.name, no, raw, qno, sorted(regs & u))
                if op.startswith("ds_"):
                    is_read = op.startswith(("ds_read", "ds_load")) and not op.startswith(STORE_LIKE)
                    queue = (queue + ((no, frozenset(d) if (in_asm and is_read) else frozenset()),))[-QCAP:]
                elif op.startswith(("s_cbranch", "s_branch")) and ops:
                    tgt = k.labels.get(ops[0])
                    if tgt is not None:
                        if op.startswith("s_branch"):
                            i = tgt
                            continue
                        work.append((tgt, queue))
            i += 1


def lint_file(path, quiet=False):
    found = []
    seen = set()

    def report(kind, kern, no, raw, qno, regs):
        key = (kind, kern, no, qno)
        if key in seen:
            return
        seen.add(key)
        found.append((kind, kern, no, raw, qno, regs))

    for k in parse(path):
        lint_kernel(k, report)
    if not quiet:
        for kind, kern, no, raw, qno, regs in found:
            print(f"{os.path.basename(path)}:{no}: {kind} on v{regs[0]}..v{regs[-1]} (asm read at line {qno} still in flight) in {kern}\n      {raw}")
    return found


ASM_READ_SOURCES = ["conv_igemm.hip", "conv_igemm_ws.hip", "conv_halo.hip", "conv_wgrad.hip", "conv_wgrad_patch.hip", "conv_wgrad_stem.hip",
                    "score_fused.hip", "gemm_ws.hip"]


def build_and_lint(root, srcs=None, quiet=False, defines=()):
    csrc = os.path.join(root, "dpc_amd", "csrc")
    total = []
    with tempfile.TemporaryDirectory() as td:
        for f in srcs or ASM_READ_SOURCES:
            src = os.path.join(csrc, f)
            if not os.path.exists(src):
                continue
            out = os.path.join(td, f.replace(".hip", ".s"))
            cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I" + os.path.join(root, "include"),
                   "--cuda-device-only", "-S", src, "-o", out] + ["-D" + d for d in defines]
            subprocess.run(cmd, check=True, cwd=td)
            total += [(f,) + h for h in lint_file(out, quiet)]
    return total


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--build":
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        args = sys.argv[2:]
        hits = build_and_lint(root, [a for a in args if not a.startswith("-D")] or None, defines=[a[2:] for a in args if a.startswith("-D")])
    else:
        hits = []
        for p in sys.argv[1:]:
            hits += lint_file(p)
    print(f"{len(hits)} hazard(s)")
    sys.exit(1 if hits else 0)
