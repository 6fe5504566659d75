// probe: buffer_load_dwordx4 ... lds (LDS-DMA through a buffer resource): do out-of-range lanes write zeros?
#include <hip/hip_runtime.h>
#include <stdint.h>
extern "C" __global__ void probe(const uint32_t* src, uint32_t* out, int nbytes) {
    __shared__ __attribute__((aligned(1024))) uint32_t lds[256 * 2];
    const int lane = threadIdx.x;
    for (int i = lane; i < 512; i += 64) lds[i] = 0xdeadbeefu;
    __syncthreads();
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, nbytes, 0x00020000);
    // lanes 0..31 in range (lane*16), odd lanes of the upper half far out of range
    uint32_t voff = (lane < 32 || (lane & 1) == 0) ? lane * 16 : 0x80000000u;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)lds, 16, voff, /*soffset*/ 32, /*imm*/ 0, 0);
    __syncthreads();
    for (int i = lane; i < 512; i += 64) out[i] = lds[i];
}
extern "C" int run_probe(const uint32_t* src, uint32_t* out, int nbytes) {
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, src, out, nbytes);
    return (int)hipDeviceSynchronize();
}
