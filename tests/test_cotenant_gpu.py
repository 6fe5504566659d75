"""GPU tier: the train step computes the same bits with a co-tenant on every CU as alone.

Round 3 found a rare divergence beside a small LDS-using workgroup and fenced it off by claiming the whole CU's LDS.  The
mechanism (conv_igemm_ws.hip WS_RETIRE_TAIL_READS: an in-flight asm LDS read landing on a re-used register when the LDS round trip
is stretched) is fixed in round 4 and the claim is gone, so the persistent kernels DO share CUs now -- with the side stream's
kernels and, on multi-GPU nodes, with RCCL's.  The squatter (include/dpc_hip.h: dpc_diag_squat) provides that company on one GPU:
small workgroups that fit beside the 144 KB kernels, with and without LDS traffic.  scripts/probes/squat_probe.py is the long form
with the positive control (profiles/r04_cotenant.txt)."""
import ctypes as C

import pytest
import torch

import kcases as kc
from dpc_amd import _lib as L
from dpc_amd.engine import DPCEngine
from oracle import dpc_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
# (workgroups, waves, LDS bytes, mode): one 4-wave workgroup per CU = a wave on every SIMD (48 VGPRs beside 2 x 216: fits; two 2-wave
# workgroups may share two SIMDs and then do NOT fit beside igemm_ws -- measured); LDS traffic / vector-memory traffic / idle with LDS
SQUATS = [(256, 4, 8192, 1), (256, 4, 0, 2), (256, 4, 8192, 0)]


def _squat(lib, stream, n, waves, lds, mode, usec, scratch, sink, where=None):
    lib.call("dpc_diag_squat", n, waves, lds, mode, usec, scratch, scratch.numel() * 4, where, sink, C.c_void_p(stream.cuda_stream))


@pytest.fixture(scope="module")
def company():
    scratch = torch.randint(0, 1 << 30, (16 << 20,), device=DEV, dtype=torch.int32)
    return scratch, torch.zeros(4, device=DEV, dtype=torch.int32), torch.cuda.Stream()


def test_squatter_runs_and_reports_placement(company):
    scratch, sink, side = company
    lib = L.load_hip()
    where = torch.zeros(256, device=DEV, dtype=torch.int32)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    _squat(lib, torch.cuda.current_stream(), 256, 4, 8192, 1, 300, scratch, sink, where)
    e1.record()
    torch.cuda.synchronize()
    assert 0.25 < e0.elapsed_time(e1) < 5.0   # lives for the time it was asked to (300 us), not forever
    w = where.cpu().numpy().astype("uint32")
    assert (w >> 31).all()
    cus = {(int(v >> 16) & 0xf, int(v >> 12) & 0xf, int(v >> 8) & 0xf) for v in w}
    assert len(cus) >= 200, len(cus)   # spread over the chip's 256 CUs
    with pytest.raises(L.DpcError):
        lib.call("dpc_diag_squat", 1, 5, 0, 0, 1, None, 0, None, None, None)


@pytest.mark.parametrize("which", ["l3", "l2"])
def test_loader_compute_kernels_beside_a_cotenant(company, which):
    """igemm_ws_kernel (layer3 input-gradient, 3x3x3) and igemm_wsp_kernel (layer2, plane variant): 150 launches per kind of company"""
    scratch, sink, side = company
    lib = L.load_hip()
    N, T, H, W, Cc, ks, pd = {"l3": (256, 3, 8, 8, 256, (3, 3, 3), (1, 1, 1)), "l2": (128, 5, 16, 16, 128, (1, 3, 3), (0, 1, 1))}[which]
    BF = torch.bfloat16
    taps = ks[0] * ks[1] * ks[2]
    g = torch.Generator(device=DEV).manual_seed(5)
    dy = torch.randn(N, T, H, W, Cc, device=DEV, generator=g).to(BF)
    wd = (torch.randn(Cc, taps * Cc, device=DEV, generator=g) * 0.05).to(BF)
    out = torch.empty(N, T, H, W, Cc, device=DEV, dtype=BF)
    dd = kc.conv_desc(BF, BF, 1, N, (T, H, W), (T, H, W), Cc, Cc, Cc, taps * Cc, Cc, ks, (1, 1, 1), pd)
    main = torch.cuda.current_stream()

    def dgrad():
        assert lib.call("dpc_conv_igemm", C.byref(dd), dy, wd, out, None, None, C.c_void_p(main.cuda_stream)) == 0

    dgrad()
    torch.cuda.synchronize()
    assert L.last_kernel(lib).startswith("igemm_ws_kernel<false>" if which == "l3" else "igemm_wsp_kernel<false>"), L.last_kernel(lib)
    ref = out.clone()
    for n, waves, lds, mode in SQUATS:
        for it in range(150):
            out.zero_()
            torch.cuda.synchronize()
            _squat(lib, side, n, waves, lds, mode, 400, scratch, sink)
            _squat(lib, main, 1, 1, 0, 0, 30, scratch, sink)   # head start: the squatters are resident when the kernel arrives
            dgrad()
            torch.cuda.synchronize()
            assert torch.equal(out, ref), f"launch {it} beside squatter {(n, waves, lds, mode)} differs from the solo result"


@pytest.mark.parametrize("graph", [False, True])
def test_train_step_beside_a_cotenant_is_bit_identical(company, graph):
    scratch, sink, side = company
    lib = L.load_hip()
    net, size, B, P = "resnet18", 128, 16, 3
    engs = []
    for _ in range(2):
        e = DPCEngine(net, size, 8, 5, P, B, DEV, torch.bfloat16)
        e.load_params(O.init_params_reference_style(net, seed=3))
        engs.append(e)
    solo, crowded = engs
    x = torch.randn(B, 8, 3, 5, size, size, device=DEV, generator=torch.Generator(DEV).manual_seed(9))
    fs, fc = (solo.capture_train_step(x), crowded.capture_train_step(x)) if graph else ((lambda: solo.train_step(x)), (lambda: crowded.train_step(x)))
    for step in range(6):
        rs = fs().clone()
        torch.cuda.synchronize()
        n, waves, lds, mode = SQUATS[step % len(SQUATS)]
        _squat(lib, side, n, waves, lds, mode, 12000, scratch, sink)   # outlives the step (~5 ms at this batch)
        _squat(lib, torch.cuda.current_stream(), 1, 1, 0, 0, 30, scratch, sink)
        rc = fc().clone()
        torch.cuda.synchronize()
        assert torch.equal(rs, rc), (step, rs, rc)
        assert torch.equal(solo.flat_g, crowded.flat_g), f"gradients differ at step {step}"
        assert torch.equal(solo.flat_p, crowded.flat_p), f"parameters differ at step {step}"
