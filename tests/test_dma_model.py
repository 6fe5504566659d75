"""CPU tier: the simulator's LDS-DMA completion model (tests/simt_emu/simt_emu.h) gives the kernel tier teeth for the hand-counted
`s_waitcnt vmcnt(N)` of the LDS-DMA rings.  Default = every piece lands as LATE as its wait allows (the whole CPU tier runs like that);
here the same kernel cases also run with pieces landing at once (the other extreme: a piece may overwrite a stage somebody still
reads), and -- the positive control -- with every counted wait weakened by one piece, which the cases must notice."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# the cases whose kernels keep LDS-DMA pieces in flight across a counted wait: generic implicit GEMM (two chunks in flight), the
# loader / compute kernels, the layer1 / stem patch kernels, the score GEMMs, the split-K GEMM, the fused stem weight gradient
SUBSET = ("test_conv_fwd or test_conv_dgrad_ex or test_gemm_ws_splitk or test_gemm_nt_splitk or test_score_gemm or test_stem_s2d "
          "or test_stem_wgrad_fused")


def _run(env_extra):
    subprocess.run(["make", "-s", "-j8", "emu"], cwd=ROOT, check=True)
    env = dict(os.environ, **env_extra)
    return subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_kernels_emu.py"), "-q", "-p", "no:cacheprovider",
                           "-k", SUBSET], env=env, capture_output=True, text=True, timeout=1500, cwd=ROOT)


def test_pieces_that_land_at_once_are_fine_too():
    r = _run({"DPC_EMU_DMA": "eager"})
    assert r.returncode == 0, r.stdout[-3000:]


def test_a_counted_wait_that_is_one_piece_too_weak_is_caught():
    r = _run({"DPC_EMU_DMA_WEAK": "1"})
    m = re.search(r"(\d+) failed", r.stdout)
    assert r.returncode != 0 and m and int(m.group(1)) >= 10, r.stdout[-3000:]
    failed = set(re.findall(r"FAILED tests/test_kernels_emu.py::(\w+)", r.stdout))
    # every family with a counted wait notices
    assert {"test_conv_fwd", "test_conv_dgrad_ex", "test_gemm_ws_splitk", "test_gemm_nt_splitk", "test_score_gemm",
            "test_stem_wgrad_fused"} <= failed, failed
