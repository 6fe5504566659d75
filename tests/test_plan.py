"""Kernel selection is part of the contract: which gfx950 kernel serves each convolution of the train step
is queried through the C ABI (dpc_conv_plan: the library's own dispatch code with the launch skipped -- no GPU needed) and
pinned here for BASELINE.json's configurations.  A threshold / switch change that silently demotes a shape to the generic
implicit-GEMM kernel keeps every numerical test green; it turns these red.

tests/golden/plan_tables.json is the snapshot (regenerate with ``python tests/test_plan.py`` after an INTENDED change);
the rule test below states what must hold whatever the snapshot says."""
import json
import os
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from dpc_amd import _lib as L  # noqa: E402
from dpc_amd import plan as P  # noqa: E402

SNAP = os.path.join(ROOT, "tests", "golden", "plan_tables.json")


def tables():
    lib = L.Lib(L.HIP_LIB_PATH, "probe")  # dlopen + symbol binding only
    out = {}
    for cfg, (net, size, batch) in P.CONFIGS.items():
        for dt, tag in ((torch.bfloat16, "bf16"), (torch.float32, "f32")):
            out[f"{cfg}/{tag}"] = [list(r) for r in P.plan_table(lib, net, size, batch, dt)]
    return out


@pytest.fixture(scope="module")
def tabs():
    subprocess.run(["make", "-s", "-j8", "all"], cwd=ROOT, check=True)
    return tables()


def test_plan_snapshot(tabs):
    want = json.load(open(SNAP))
    assert sorted(tabs) == sorted(want)
    for key in want:
        got = {(u, op): kern for u, op, kern in tabs[key]}
        for u, op, kern in want[key]:
            assert got.get((u, op)) == kern, f"{key} {u} {op}: planned {got.get((u, op))}, pinned {kern}"
        assert len(got) == len(want[key])


@pytest.mark.parametrize("cfg", sorted(P.CONFIGS))
def test_throughput_mode_uses_the_specialised_kernels(tabs, cfg):
    """bf16 (the benchmarked mode): role-specialised / staged-patch kernels everywhere except the in-place input-gradients of the
    strided 1x1 downsamples (parity-class gather of the generic kernel, 0.2 ms/step at cfg2) -- nothing else may fall back."""
    img = P.CONFIGS[cfg][1]
    for unit, op, kern in tabs[f"{cfg}/bf16"]:
        strided = unit.endswith(".0.conv1") and not unit.startswith("layer1.") or "downsample" in unit
        if unit == "conv1":
            assert kern == {"fwd": "conv_halo_ws_kernel<false,2,256>", "wgrad": "wgrad_stem_kernel"}[op], (unit, op, kern)
        elif op == "fwd":
            want = "conv_halo_ws_kernel<false,8,128>" if unit.startswith("layer1.") else ("igemm_ws_kernel<false>", "igemm_wsp_kernel<false>")
            assert kern == want or kern in want, (unit, op, kern)
        elif op.startswith("dgrad"):
            add = "true" if op == "dgrad+addend" else "false"
            if strided and unit == "layer2.0.conv1" and img == 128:   # 64 output columns over 16 x 16 gradient planes: the shift-convolution kernel (round 6)
                assert kern == "igemm_wsd_kernel", (unit, op, kern)
            elif strided and ("downsample" in unit or unit.startswith("layer2.")):   # in-place 1x1s; layer2.0.conv1 (28 x 28 planes at 224 px) has 64 output columns
                assert kern.startswith("igemm_kernel<T,TO,BN,3>[T=bf16"), (unit, op, kern)  # in-place 1x1: parity classes of the generic kernel, never the dense gather
            elif strided:
                assert kern == "igemm_ws_kernel<false,true>", (unit, op, kern)               # parity classes on the loader / compute kernel
            elif unit.startswith("layer1."):
                assert kern == f"conv_halo_ws_kernel<{add},8,128>", (unit, op, kern)
            else:
                assert kern in (f"igemm_ws_kernel<{add}>", f"igemm_wsp_kernel<{add}>"), (unit, op, kern)
        else:  # weight gradients: staged patch for every 3x3 stride-1 conv whose plane is not mostly padding, wgrad2 otherwise
            assert not kern.startswith("wgrad_kernel"), (unit, op, kern)
            if strided:
                assert kern.startswith("wgrad2_kernel") and f"padded={int(img == 224 and not unit.startswith('layer4.'))}" in kern or \
                    kern.startswith("wgrad2_kernel"), (unit, op, kern)
            elif unit.startswith(("layer1.", "layer2.", "layer3.")):
                assert kern.startswith("wgrad_patch_kernel<"), (unit, op, kern)
    # the plane variant is what serves layer2's unit-stride convs at 128 px (16 x 16 planes)
    if img == 128:
        assert all(k == "igemm_wsp_kernel<false>" for u, op, k in tabs[f"{cfg}/bf16"] if u.startswith("layer2.") and op == "fwd"
                   and not u.endswith(".0.conv1") and "downsample" not in u)


def test_plan_query_matches_the_launch(tabs):
    """the query and a real call run the same dispatch code; on the simulator library (which really launches) the recorded
    last kernel of a call equals the plan for the same descriptor"""
    import ctypes as C
    subprocess.run(["make", "-s", "-j8", "emu"], cwd=ROOT, check=True)
    emu = L.load_emulator()
    f, d, w = P.unit_descs(64, 64, (1, 3, 3), (1, 1, 1), (0, 1, 1), (2, 1, 8, 32), torch.bfloat16)
    x = torch.randn(2, 1, 8, 32, 64).to(torch.bfloat16)
    wt = torch.randn(64, 9 * 64).to(torch.bfloat16)
    out = torch.empty(2, 1, 8, 32, 64, dtype=torch.bfloat16)
    planned = L.conv_plan(emu, f, L.PLAN_IGEMM)
    emu.call("dpc_conv_igemm", C.byref(f), x, wt, out, None, None, emu.stream())
    assert L.last_kernel(emu) == planned and planned.startswith("conv_halo")
    ns = C.c_int32(0)
    emu.call("dpc_conv_wgrad", C.byref(w), None, None, 64, None, C.byref(ns), emu.stream())
    part = torch.zeros(ns.value, 64, 9 * 64)
    planned = L.conv_plan(emu, w, L.PLAN_WGRAD, dy_ld=64)
    emu.call("dpc_conv_wgrad", C.byref(w), x, out, 64, part, C.byref(ns), emu.stream())
    assert L.last_kernel(emu) == planned and planned.startswith("wgrad_patch_kernel<32>")


if __name__ == "__main__":
    json.dump(tables(), open(SNAP, "w"), indent=0)
    print("wrote", SNAP)
