// probe: what does the matrix pipe of an MI355X sustain?  Nothing but independent v_mfma_f32_32x32x16_bf16 streams --
// no LDS, no global loads in the loop -- with W waves per SIMD and operands that are all zeros or N(0,1) bf16 values
// (switching power: random operands toggle the multiplier array, zeros do not; the clock governor reacts to it).
// FLOPs per MFMA = 2 * 32 * 32 * 16 = 32 768.  The clock under load is measured inside the kernel with s_memrealtime
// (100 MHz constant) against s_memtime-free wall time: cycles come from the shader clock counter (clock64()).
// Build: hipcc --offload-arch=gfx950 -O3 -shared -fPIC -o libmfma_rate.so mfma_rate.hip ; run: python mfma_rate.py
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC>
__global__ __launch_bounds__(256) void mfma_stream(const uint4* ops, float* sink, long long* clk, int iters) {
    // every lane loads its two 16-byte operands once; the loop is NACC independent accumulator chains
    const int lane = threadIdx.x & 63;
    const uint4 ua = ops[(blockIdx.x * 7 + lane) & 1023], ub = ops[(blockIdx.x * 13 + lane + 64) & 1023];
    const bf16x8_t a = __builtin_bit_cast(bf16x8_t, ua), b = __builtin_bit_cast(bf16x8_t, ub);
    f32x16 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    const long long c0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
    }
    const long long c1 = clock64();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][15];
    if (s == 12345.678f) sink[0] = s;  // keeps the chains alive
    if (threadIdx.x == 0) clk[blockIdx.x] = c1 - c0;
}

// returns the kernel time in ms (HIP events) for `blocks` workgroups of 256 threads (4 waves = one per SIMD),
// blocks = 256 * W puts W waves on every SIMD of the chip.  mfmas_per_wave = iters * 4 * nacc.
extern "C" float run_mfma_stream(const void* ops, float* sink, long long* clk, int blocks, int iters, int nacc) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipEventRecord(e0, 0);
    if (nacc == 4) hipLaunchKernelGGL(mfma_stream<4>, dim3(blocks), dim3(256), 0, 0, (const uint4*)ops, sink, clk, iters);
    else if (nacc == 2) hipLaunchKernelGGL(mfma_stream<2>, dim3(blocks), dim3(256), 0, 0, (const uint4*)ops, sink, clk, iters);
    else hipLaunchKernelGGL(mfma_stream<1>, dim3(blocks), dim3(256), 0, 0, (const uint4*)ops, sink, clk, iters);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    hipEventDestroy(e0);
    hipEventDestroy(e1);
    return hipGetLastError() == hipSuccess ? ms : -1.f;
}
