// diag.hip -- a co-tenant on demand (include/dpc_hip.h: dpc_diag_squat).
//
// The train step runs beside other kernels: its own side stream (weight gradients, reductions) and, with more than one GPU,
// RCCL's channel kernels during the overlapped gradient all-reduce (dpc/main.py:65's DataParallel reduce).  What a co-tenant
// does to the one-workgroup-per-CU kernels -- to their results (round 3's "LDS race", conv_igemm_ws.hip WS_RETIRE_TAIL_READS)
// and to their time (a persistent grid that finds k CUs taken) -- has to be measurable on ONE GPU.  The squatter is that
// co-tenant with every property under control: n workgroups of `waves` waves holding `lds_bytes` of LDS for `usec`
// microseconds (s_memrealtime, 100 MHz), doing one of
//   mode 0  nothing (s_sleep between clock reads): occupies wave slots, registers and LDS only;
//   mode 1  LDS traffic: ds_write_b128 / ds_read_b128 sweeps over its own allocation (stretches the LDS latency of the CU);
//   mode 2  vector-memory traffic: 16-byte loads over `scratch` (texture path and L2, no LDS);
//   mode 3  VALU work (issue slots of the SIMDs it sits on).
// where[wg] receives the hardware placement of the workgroup: XCC_ID << 16 | (HW_ID & 0xffff) -- SE / SH / CU / SIMD / wave slot
// -- so a probe can tell which CUs were shared.  It is a diagnostic: nothing on the training path calls it; tests and
// scripts/probes/squat_probe.py do.
#include "dpc_rt.h"
#include "../../include/dpc_hip.h"

#ifndef DPC_SIMT_EMU
__global__ __launch_bounds__(256) void squat_kernel(int mode, long long ticks, const unsigned* scratch, unsigned scratch_units,
                                                    unsigned* where, unsigned* sink, int lds_bytes) {
    extern __shared__ __attribute__((aligned(16))) unsigned char sq_lds[];
    const int tid = threadIdx.x;
    if (tid == 0 && where) {
        unsigned hw, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        where[blockIdx.x] = ((xcc & 0xf) << 16) | (hw & 0xffffu) | 0x80000000u;
    }
    const long long t0 = (long long)wall_clock64();   // s_memrealtime: 100 MHz, independent of the core clock
    const int units = lds_bytes / 16;
    u32x4 acc = {(unsigned)tid, 1u, 2u, 3u};
    unsigned guard = 0;
    while ((long long)wall_clock64() - t0 < ticks) {
        if (mode == 0) {
            __builtin_amdgcn_s_sleep(32);
        } else if (mode == 1) {
            for (int r = 0; r < 8 && units > 0; ++r) {
                const int u = (tid + r * (int)blockDim.x) % units;
                *(u32x4*)(sq_lds + u * 16) = acc;
                const u32x4 v = *(const u32x4*)(sq_lds + ((u * 7 + 3) % units) * 16);
                acc[0] ^= v[0]; acc[1] += v[1]; acc[2] ^= v[2]; acc[3] += v[3];
            }
        } else if (mode == 2) {
            for (int r = 0; r < 8 && scratch_units > 0; ++r) {
                const unsigned u = ((unsigned)blockIdx.x * 977u + (unsigned)tid + guard * 4099u + (unsigned)r * 65537u) % scratch_units;
                const u32x4 v = *(const u32x4*)(scratch + (size_t)u * 4);
                acc[0] ^= v[0]; acc[1] += v[1]; acc[2] ^= v[2]; acc[3] += v[3];
            }
        } else {
            float f = (float)acc[0];
            for (int r = 0; r < 256; ++r) f = f * 1.0001f + 0.5f;
            acc[0] = (unsigned)f;
        }
        ++guard;
    }
    if (sink && acc[0] == 0x12345678u && acc[1] == 0x9abcdef0u) sink[0] = acc[2] + acc[3] + guard;   // keeps the loops alive
}
#endif

extern "C" int dpc_diag_squat(int32_t n_wg, int32_t waves, int32_t lds_bytes, int32_t mode, int32_t usec, const void* scratch,
                              int64_t scratch_bytes, uint32_t* where, uint32_t* sink, dpc_stream_t stream_) {
    if (n_wg <= 0 || waves < 1 || waves > 4 || lds_bytes < 0 || lds_bytes > 160 * 1024 || (lds_bytes & 15) || mode < 0 || mode > 3 || usec < 0)
        return DPC_ERR_ARG;
    if (mode == 2 && (!scratch || scratch_bytes < 16)) return DPC_ERR_ARG;
#ifdef DPC_SIMT_EMU
    (void)where; (void)sink; (void)stream_;
    return DPC_ERR_UNSUPPORTED;   // there is no co-residency to emulate on the host simulator
#else
    hipStream_t stream = (hipStream_t)stream_;
    if (lds_bytes > 64 * 1024 &&
        hipFuncSetAttribute((const void*)squat_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes) != hipSuccess)
        return DPC_ERR_LAUNCH;
    DPC_LAUNCH_DYN(squat_kernel, dim3((unsigned)n_wg), dim3((unsigned)(waves * 64)), (size_t)lds_bytes, stream, (int)mode,
                   (long long)usec * 100ll, (const unsigned*)scratch, (unsigned)(scratch_bytes / 16), where, sink, (int)lds_bytes);
    return dpc_launch_status();
#endif
}
