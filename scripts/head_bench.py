"""isolated timing of the head kernels (fused score fwd/bwd, fused ConvGRU recurrence fwd/bwd) at BASELINE sizes"""
import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from dpc_amd import _lib as L
import kcases as kc

k = kc.K(L.Lib(os.environ["DPC_BENCH_LIB"], "hip") if os.environ.get("DPC_BENCH_LIB") else L.load_hip(), "cuda:0")   # A/B against another build
bf = torch.bfloat16
reps = int(os.environ.get("REPS", "10"))

def timeit(fn, name, flops=None):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record(); torch.cuda.synchronize()
    us = a.elapsed_time(b) * 1e3 / reps
    print(f"{name:40s} {us:9.1f} us" + (f"  {flops / us / 1e6:8.1f} TFLOP/s" if flops else ""), flush=True)

for R in (6144, 15680):
    D = 256
    g = torch.Generator(device="cuda").manual_seed(1)
    pred = (torch.randn(R, D, device="cuda", generator=g) * 0.1).to(bf)
    finf = (torch.randn(R, D, device="cuda", generator=g) * 0.1).to(bf)
    ld = (R + 7) // 8 * 8
    predT, finfT = torch.zeros(D, ld, dtype=bf, device="cuda"), torch.zeros(D, ld, dtype=bf, device="cuda")
    predT[:, :R] = pred.t(); finfT[:, :R] = finf.t()
    nf, nb = C.c_int64(0), C.c_int64(0)
    ns = k.lib.call("dpc_score_ws_floats", R, D, C.byref(nf), C.byref(nb))
    ws = k.empty(max(nf.value, nb.value))
    diag, lse2, row_ws = k.empty(R), k.empty(R), k.empty(R, 2)
    fl = 2.0 * R * R * D
    timeit(lambda: k.call("dpc_score_fwd", pred, finf, R, D, diag, lse2, row_ws, None, ws), f"score_fwd R={R} (splits {ns})", fl)
    timeit(lambda: k.lib.call("dpc_score_bwd", pred, finf, finfT, ld, R, D, lse2, 1, ws, k.lib.stream()), f"score_bwd d_pred R={R}", fl)
    timeit(lambda: k.lib.call("dpc_score_bwd", finf, pred, predT, ld, R, D, lse2, 0, ws, k.lib.stream()), f"score_bwd d_finf R={R}", fl)

# ---- materialised path: score GEMM (dpc_conv_igemm), CE / top-k + dS, d_pred (split-K NT GEMM), d_finf (transpose-read TN GEMM)
for R in (6144, 6468, 15680):
    D = 256
    g = torch.Generator(device="cuda").manual_seed(2)
    pred = (torch.randn(R, D, device="cuda", generator=g) * 0.1).to(bf)
    finf = (torch.randn(R, D, device="cuda", generator=g) * 0.1).to(bf)
    ld = (R + 7) // 8 * 8
    finfT = torch.zeros(D, ld, dtype=bf, device="cuda"); finfT[:, :R] = finf.t()
    score = k.empty(R, R)
    dS = k.empty(R, ld, dtype=bf)
    row_ws, res = k.empty(R, 2), k.empty(4)
    d = kc.conv_desc(bf, torch.float32, 0, R, (1, 1, 1), (1, 1, 1), D, D, R, D, R, (1, 1, 1), (1, 1, 1), (0, 0, 0))
    fl = 2.0 * R * R * D
    timeit(lambda: k.call("dpc_conv_igemm", C.byref(d), pred, finf, score, None, None), f"score GEMM R={R} [{L.conv_plan(k.lib, d)}]", fl)
    timeit(lambda: k.call("dpc_ce_topk", score, R, R, R, row_ws, res, dS, 1, ld), f"CE/top-k + dS R={R}")
    nsk = C.c_int32(0)
    k.call("dpc_gemm_nt_splitk", 1, R, D, ld, None, ld, None, ld, None, C.byref(nsk))
    part = k.empty(max(nsk.value, 1) * R * D)
    dp = k.empty(R, D)
    torch.cuda.synchronize()
    dS[:, R:].zero_()   # the CE kernel zero-fills the padding columns; make sure they are finite even before its first launch here
    def dpred():
        k.call("dpc_gemm_nt_splitk", 1, R, D, ld, dS, ld, finfT, ld, part, C.byref(nsk))
        k.call("dpc_reduce_unpack", part, nsk.value, dp, R, 1, D, D, 0, 1, 0)
    timeit(dpred, f"d_pred split-K x{nsk.value} + reduce R={R} [{L.last_kernel(k.lib) if dpred() is None else ''}]", fl)
    timeit(lambda: k.call("dpc_gemm_nt_splitk", 1, R, D, ld, dS, ld, finfT, ld, part, C.byref(nsk)), f"  d_pred GEMM alone R={R} [{L.last_kernel(k.lib)}]", fl)
    predT = torch.zeros(D, ld, dtype=bf, device="cuda"); predT[:, :R] = pred.t()
    nst = C.c_int32(0)
    if k.lib._fn("dpc_gemm_tn_splitk")(1, R, D, R, None, ld, None, ld, None, C.byref(nst), k.lib.stream()) == 0:
        part3 = k.empty(nst.value * R * D)
        df2 = k.empty(R, D)
        def dfinf_ws():
            k.call("dpc_gemm_tn_splitk", 1, R, D, R, dS, ld, predT, ld, part3, C.byref(nst))
            k.call("dpc_reduce_unpack", part3, nst.value, df2, R, 1, D, D, 0, 1, 0)
        timeit(dfinf_ws, f"d_finf K-major split-K x{nst.value} + reduce R={R}", fl)
        timeit(lambda: k.call("dpc_gemm_tn_splitk", 1, R, D, R, dS, ld, predT, ld, part3, C.byref(nst)), f"  d_finf GEMM alone R={R} [{L.last_kernel(k.lib)}]", fl)
        timeit(lambda: k.call("dpc_transpose2d", pred, 1, D, predT, 1, ld, R, D), f"  pred -> pred^T R={R}")
        del part3
    dw = kc.conv_desc(bf, torch.float32, 0, R, (1, 1, 1), (1, 1, 1), D, D, R, D, R, (1, 1, 1), (1, 1, 1), (0, 0, 0))
    nsw = C.c_int32(0)
    k.call("dpc_conv_wgrad", C.byref(dw), None, None, ld, None, C.byref(nsw))
    part2 = k.empty(max(nsw.value, 1) * R * D)
    df = k.empty(R, D)
    def dfinf():
        k.call("dpc_conv_wgrad", C.byref(dw), pred, dS, ld, part2, C.byref(nsw))
        k.call("dpc_reduce_unpack", part2, nsw.value, df, R, 1, D, D, 0, 1, 0)
    timeit(dfinf, f"d_finf wgrad2 x{nsw.value} + reduce R={R}", fl)
    del score, dS, part, part2

for (B, SQ, P) in ((128, 16, 3), (64, 49, 5)):
    d, dev, *_ = kc._chain_setup(k, bf, B, SQ, 256, P, 8 - P, seed=3)
    step = torch.tensor([1], dtype=torch.int32, device="cuda")
    d.step_dev = step.data_ptr()
    M, ns_ = B * SQ, 8 - 1
    fl_f = 2.0 * M * 256 * 256 * (6 * ns_ + 2 * P)
    timeit(lambda: k.call("dpc_gru_chain_fwd", C.byref(d)), f"gru_chain_fwd M={M} P={P}", fl_f)
    timeit(lambda: k.call("dpc_gru_chain_bwd", C.byref(d)), f"gru_chain_bwd M={M} P={P}", fl_f)
    d.step_dev = None
    timeit(lambda: k.call("dpc_gru_chain_fwd", C.byref(d)), f"gru_chain_fwd M={M} P={P} (eval, no dropout)", fl_f)
