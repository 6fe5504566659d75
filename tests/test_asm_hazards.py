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
