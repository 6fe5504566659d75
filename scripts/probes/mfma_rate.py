"""MFMA-rate micro-benchmark: sustained v_mfma_f32_32x32x16_bf16 rate of the whole chip, zeros vs N(0,1)
operands, 1 / 2 / 4 waves per SIMD.  Prints one table; the output is tracked under profiles/.
    hipcc --offload-arch=gfx950 -O3 -shared -fPIC -o scripts/probes/libmfma_rate.so scripts/probes/mfma_rate.hip
    python scripts/probes/mfma_rate.py
Clock: the wave-cycle counter (clock64 = s_memtime, shader clock) of every workgroup over the loop divided by the loop's
wall time bounds the frequency the matrix pipe ran at; rocprofv3 --pmc GRBM_GUI_ACTIVE on the same command gives it directly."""
import ctypes
import os

import torch

lib = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "libmfma_rate.so"))
lib.run_mfma_stream.restype = ctypes.c_float
lib.run_mfma_stream.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int]
dev = "cuda"
sink = torch.zeros(4, device=dev)
FLOP = 2 * 32 * 32 * 16
print(f"{'operands':9s} {'waves/SIMD':>10s} {'acc/wave':>8s} {'ms':>8s} {'TFLOP/s':>9s} {'of 2500':>8s} {'cyc/MFMA/SIMD':>14s} {'clk64 GHz?':>10s}")
for name, ops in (("zeros", torch.zeros(1024 * 8, dtype=torch.bfloat16, device=dev)),
                  ("N(0,1)", torch.randn(1024 * 8, device=dev).to(torch.bfloat16)),
                  ("N(0,30)", (30 * torch.randn(1024 * 8, device=dev)).to(torch.bfloat16))):
    for wps in (1, 2, 4):
        for nacc in (4, 2) if wps == 1 else (4,):
            blocks, iters = 256 * wps, 20000 // wps
            clk = torch.zeros(blocks, dtype=torch.int64, device=dev)
            best = None
            for rep in range(4):  # first repetition warms the clock governor
                ms = lib.run_mfma_stream(ops.data_ptr(), sink.data_ptr(), clk.data_ptr(), blocks, iters, nacc)
                if rep and (best is None or ms < best[0]):
                    best = (ms, clk.float().mean().item())
            ms, cyc = best
            n_mfma = blocks * 4 * iters * 4 * nacc
            tf = n_mfma * FLOP / (ms * 1e-3) / 1e12
            per_simd = iters * 4 * nacc * wps  # MFMAs each SIMD issued
            print(f"{name:9s} {wps:10d} {nacc:8d} {ms:8.3f} {tf:9.1f} {tf / 2500:8.3f} {cyc / (iters * 4 * nacc) / 1.0:14.2f} {cyc / (ms * 1e6):10.3f}")
