// probe: exact lane/element semantics of ds_read_b64_tr_b16 on gfx950 (LDS word i holds the value i)
#include <hip/hip_runtime.h>
typedef short v4i16 __attribute__((ext_vector_type(4)));
__global__ void probe(const int* addr_in, unsigned short* out) {
    __shared__ __attribute__((aligned(16))) unsigned short lds[64 * 64];
    for (int i = threadIdx.x; i < 64 * 64; i += 64) lds[i] = (unsigned short)i;
    __syncthreads();
    const int a = addr_in[threadIdx.x];  // byte offset per lane
    v4i16 r = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4i16*)((char*)lds + a));
    for (int j = 0; j < 4; ++j) out[threadIdx.x * 4 + j] = (unsigned short)r[j];
}
extern "C" int run_probe(const int* addr_in, unsigned short* out) {
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, addr_in, out);
    return (int)hipDeviceSynchronize();
}
