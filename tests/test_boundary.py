"""Host-side boundary checks that need no GPU: the C-ABI library loads and exports every
symbol include/dpc_hip.h declares; the DPC_RNN module mirrors the reference's constructor,
attributes and state_dict keys; the product path refuses to run without the HIP device."""
import os
import re
import subprocess

import pytest
import torch

from dpc_amd import _lib as L
from oracle import dpc_oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    src = open(os.path.join(ROOT, "include", "dpc_hip.h")).read()
    return sorted(set(re.findall(r"\bint\s+(dpc_[a-z0-9_]+)\s*\(", src)))


def test_header_and_binding_agree():
    assert header_symbols() == sorted(L._SIGS)


@pytest.mark.parametrize("path", [L.HIP_LIB_PATH, L.EMU_LIB_PATH])
def test_library_exports_every_declared_symbol(path):
    subprocess.run(["make", "-s", "-j8", "all", "emu"], cwd=ROOT, check=True)
    lib = L.Lib(path, "probe")  # dlopen only; no kernel is launched
    assert lib.missing_symbols() == []
    assert lib.call("dpc_abi_version") == L.ABI_VERSION


def test_module_mirrors_reference_boundary():
    from dpc_amd.model import DPC_RNN
    m = DPC_RNN(sample_size=128, num_seq=8, seq_len=5, pred_step=3, network="resnet18")
    ref = O.param_shapes("resnet18", with_alias=True)
    sd = m.state_dict()
    assert list(sd.keys()) == list(ref.keys())
    assert all(tuple(sd[k].shape) == ref[k] for k in ref)
    assert sd["agg.cell_list.0.out_gate.weight"].data_ptr() == sd["agg.ConvGRUCell_00.out_gate.weight"].data_ptr()
    assert sum(p.numel() for p in m.parameters()) == 14583104
    assert (m.last_size, m.last_duration, m.pred_step, m.param["feature_size"]) == (4, 2, 3, 256)
    assert m.mask is None
    m.load_state_dict(O.make_params_pcg("resnet18"), strict=True)
    with pytest.raises(IOError):
        DPC_RNN(128, network="resnet50")
    with pytest.raises(L.DpcError):
        m(torch.zeros(1, 8, 3, 5, 128, 128))  # CPU tensors: no fallback
    m34 = DPC_RNN(224, network="resnet34")
    assert sum(p.numel() for p in m34.parameters()) == 32947776


def test_kernel_timer_charges_the_median_per_launch_position(monkeypatch):
    """bench.py's roofline pass: an event pair also contains host time between the first event and the launch; with several
    identical steps recorded, every launch position is charged the median of its samples (engine.KernelTimer.summary)."""
    from dpc_amd import engine as E

    class Ev:
        def __init__(self, t):
            self.t = t

        def elapsed_time(self, other):
            return other.t - self.t

    monkeypatch.setattr(torch.cuda, "synchronize", lambda *a, **k: None)
    tm = E.KernelTimer(["a", "b"])
    for x, y in ((1.0, 2.0), (1.1, 2.1), (30.0, 2.0), (0.9, 1.9)):   # four steps; one hiccup on "a"
        tm.records.append(("a", None, Ev(0.0), Ev(x), 1.0, 8.0, 1.0))
        tm.records.append(("b", "score", Ev(0.0), Ev(y), 2.0, 4.0, 1.5))
    s = tm.summary(4)
    assert s["a"]["launches"] == 4 and abs(s["a"]["ms"] - 4 * 1.05) < 1e-9 and s["a"]["bytes"] == 32.0
    assert abs(s["b"]["ms"] - 4 * 2.0) < 1e-9 and abs(s["tag:score"]["ms"] - 8.0) < 1e-9
    assert abs(tm.summary(1)["a"]["ms"] - 33.0) < 1e-9          # plain totals when the steps are not declared
    assert abs(tm.summary(3)["a"]["ms"] - 33.0) < 1e-9          # ... or do not divide the record count


def test_executed_fraction_of_the_temporal_taps():
    """bench.py `roofline.executed`: igemm_ws runs 7 of the 9 temporal taps of a unit-stride 3x3x3 conv at T = 3 (layer3) and 4 of 6
    at T = 2 (layer4); every other kernel and shape executes what it is charged"""
    import ctypes as C
    from dpc_amd import engine as E, _lib as L
    from dpc_amd.plan import unit_descs

    def frac(T, k=(3, 3, 3), s=(1, 1, 1), kernel="igemm_ws_kernel<false, false>", name="dpc_conv_igemm"):
        f, d, w = unit_descs(256, 256, k, s, (1, 1, 1) if k[0] == 3 else (0, 1, 1), (4, T, 8, 8), torch.bfloat16, False)
        return [E.executed_fraction(name, (C.byref(x),), kernel) for x in (f, d)]

    assert frac(3) == [7 / 9, 7 / 9] and frac(2) == [4 / 6, 4 / 6]
    assert frac(3, kernel="igemm_kernel<bf16, bf16, 128, 3, false>") == [1.0, 1.0]      # the generic kernel skips nothing
    assert frac(5, k=(1, 3, 3)) == [1.0, 1.0] and frac(5, s=(2, 2, 2))[0] == 1.0        # 2D convs; strided convs
    assert frac(3, name="dpc_conv_wgrad") == [1.0, 1.0]
