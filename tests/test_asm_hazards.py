"""The compiled gfx950 code must not hand a register to a new value while an inline-asm LDS read still targets it.

Round 3 shipped exactly that in igemm_ws_kernel / igemm_wsp_kernel (the end-of-tile fragment reads; conv_igemm_ws.hip
WS_RETIRE_TAIL_READS) and saw it as "one wave's 64 x 128 block wrong in 38 of 15 000 launches beside a foreign workgroup".  The
kernels' asm reads are invisible to hipcc's wait counts, so the property is checked on hipcc's own output
(scripts/asm_hazard_lint.py walks the control-flow graph of every kernel with the in-order queue of outstanding LDS operations).
The positive control compiles the same source with the fix left out (-DDPC_WS_NOFIX) and must be flagged."""
import os
import sys
from concurrent.futures import ThreadPoolExecutor

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "scripts"))
import asm_hazard_lint as lint  # noqa: E402

pytestmark = pytest.mark.skipif(not os.path.exists("/opt/rocm/bin/hipcc"), reason="needs hipcc (cross-compiles without a GPU)")


def test_no_inflight_asm_read_is_overwritten_or_consumed():
    srcs = [s for s in lint.ASM_READ_SOURCES if os.path.exists(os.path.join(ROOT, "dpc_amd", "csrc", s))]
    assert "conv_igemm_ws.hip" in srcs and len(srcs) >= 7
    with ThreadPoolExecutor(4) as ex:
        hits = sum(ex.map(lambda s: lint.build_and_lint(ROOT, [s], quiet=True), srcs), [])
    assert not hits, "\n".join(f"{h[0]}:{h[3]}: {h[1]} {h[4]} (asm read at line {h[5]}) in {h[2]}" for h in hits[:20])


def test_lint_sees_the_round3_hazard():
    hits = lint.build_and_lint(ROOT, ["conv_igemm_ws.hip"], quiet=True, defines=["DPC_WS_NOFIX"])
    kinds = {h[1] for h in hits}
    kernels = {h[2] for h in hits}
    assert "WAW" in kinds and any("igemm_ws_kernel" in k for k in kernels) and any("igemm_wsp_kernel" in k for k in kernels), hits[:5]


def test_no_compiler_wait_drains_an_lds_dma_ring():
    """round 6: hipcc must not put `s_waitcnt vmcnt(0)` into an innermost loop that issues LDS-DMA meant to stay in flight -- the
    residual / fused-reduction variants of conv_halo_ws_kernel carried one for five rounds (a register copy of in-flight loads at the
    end of every helper interval: 3.3 us per tile instead of 1.45; profiles/r06_halo_epi_probe.txt).  scripts/asm_drain_lint.py reads
    `hipcc -S` of every ring kernel.  Positive control: a minimal loop of the shape rounds 1-5 had (operands requested one iteration
    ahead, one of them behind a lane condition, copied over the loop edge behind the DMA piece) must be reported."""
    import asm_drain_lint as dl
    import subprocess
    import tempfile
    hits = dl.build_and_lint(ROOT)
    assert not hits, "\n".join(f"{h[0]}: {h[1]}: vmcnt(0) at line {h[3]} in the DMA loop at line {h[2]}" for h in hits[:10])
    ctl = r'''
#include "dpc_rt.h"
__global__ void ctl_kernel(const u32x4* src, const uint8_t* m, const void* dma_src, u32x4* out, int n, int lim) {
    __shared__ __attribute__((aligned(1024))) unsigned char lds[4096];
    const BufRsrc rs = make_buf_rsrc(dma_src, 1u << 20);
    const bool ok0 = (int)threadIdx.x < lim;
    u32x4 cur = src[threadIdx.x], acc = {0u, 0u, 0u, 0u};
    unsigned cb = ok0 ? (unsigned)m[threadIdx.x] : 0u;
    for (int j = 1; j < n; ++j) {
        wait_vmcnt<1>();
        barrier_lds_only();
        u32x4 nxt = cur;
        unsigned nb = cb;
        if (j + 1 < n) {
            const bool ok = (int)(threadIdx.x + j) < lim;
            nxt = *(const u32x4*)(ok ? (const char*)(src + j * 64 + threadIdx.x) : (const char*)dpc_zero16);
            nb = ok ? (unsigned)m[j * 64 + threadIdx.x] : 0u;      // a load behind a divergent branch
        }
        acc[0] += (cur[0] ^ cur[3]) + cb;
        acc[1] += ((const uint32_t*)lds)[threadIdx.x];
        out[j * 64 + threadIdx.x] = acc;
        glds16_buf(rs, (unsigned)(threadIdx.x * 16 + j * 1024), 0u, lds + (j & 3) * 1024, threadIdx.x & 63);
        barrier_lds_only();
        cur = nxt;                                   // copied over the loop edge
        cb = nb;
    }
    out[threadIdx.x] = acc;
}
'''
    with tempfile.TemporaryDirectory() as tmp:
        src, asm = os.path.join(tmp, "ctl.hip"), os.path.join(tmp, "ctl.s")
        open(src, "w").write(ctl)
        subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", f"-I{os.path.join(ROOT, 'include')}",
                        f"-I{os.path.join(ROOT, 'dpc_amd', 'csrc')}", "-S", "--cuda-device-only", src, "-o", asm], check=True, capture_output=True,
                       stdin=subprocess.DEVNULL, timeout=600)
        assert dl.lint_file(asm), "the lint did not see the drain of the control kernel"
        # second class (late round 6): operands that live in registers, loaded through GENERIC pointers (flat loads count on vmcnt and
        # lgkmcnt) and first used inside a loop whose LDS reads are hand-issued: hipcc waits lgkmcnt(0) behind the reads.  Waiting for
        # the operands in front of the loop (an empty asm that consumes them, conv_halo.hip) removes it.
        ctl2 = r'''
#include "dpc_rt.h"
template <bool FIX>
__global__ void ctl2_kernel(const void* wgt, int Co, float* out, int n) {
    __shared__ __attribute__((aligned(1024))) unsigned char lds[8192];
    const int lane = threadIdx.x & 63;
    const char* wp = lane < Co ? (const char*)wgt + lane * 64 : nullptr;
    u32x4 b[4];
    for (int k = 0; k < 4; ++k) b[k] = *(const u32x4*)(wp ? wp + k * 16 : (const char*)dpc_zero16);
    if (FIX)
        for (int k = 0; k < 4; ++k) asm volatile("" : "+v"(b[k]));
    f32x16 acc;
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    for (int j = 0; j < n; ++j) {
        barrier_lds_only();
        u32x4 a[4];
        for (int k = 0; k < 4; ++k) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(a[k]) : "v"((uint32_t)(lane * 16 + (j & 1) * 4096)), "n"(0) : "memory");
        for (int k = 0; k < 4; ++k) {
            if (k == 0) asm volatile("s_waitcnt lgkmcnt(3)" : "+v"(a[0]));
            if (k == 1) asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(a[1]));
            if (k == 2) asm volatile("s_waitcnt lgkmcnt(1)" : "+v"(a[2]));
            if (k == 3) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a[3]));
            acc = mfma_32x32x16_bf16(a[k], b[k], acc);
        }
    }
    for (int r = 0; r < 16; ++r) out[threadIdx.x * 16 + r] = acc[r];
}
template __global__ void ctl2_kernel<false>(const void*, int, float*, int);
template __global__ void ctl2_kernel<true>(const void*, int, float*, int);
'''
        src2, asm2 = os.path.join(tmp, "ctl2.hip"), os.path.join(tmp, "ctl2.s")
        open(src2, "w").write(ctl2)
        subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", f"-I{os.path.join(ROOT, 'include')}",
                        f"-I{os.path.join(ROOT, 'dpc_amd', 'csrc')}", "-S", "--cuda-device-only", src2, "-o", asm2], check=True, capture_output=True,
                       stdin=subprocess.DEVNULL, timeout=600)
        got = dl.lint_file_reads(asm2)
        assert any("Lb0E" in h[0] for h in got), "the lint did not see the fragment-read drain of the control kernel"
        assert not any("Lb1E" in h[0] for h in got), got
