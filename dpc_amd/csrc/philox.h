// philox.h -- Philox4x32-10 counter-based generator (Salmon et al., SC'11; the generator torch's CUDA/HIP
// dropout uses) for the ConvGRU's Dropout(p=0.1) on the carried hidden state (backbone/convrnn.py:39,59,78).
// Stateless: word w of block `ctr` under key (seed, step) is a pure function, so the standalone mask kernel
// (dpc_dropout_mask, tests / mask injection) and the fused recurrence kernel generate the same bits.
#pragma once
#include "dpc_rt.h"

struct Philox4 { uint32_t v[4]; };

__device__ __host__ __forceinline__ Philox4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1) {
    DPC_UNROLL
    for (int r = 0; r < 10; ++r) {
        const unsigned long long p0 = (unsigned long long)0xD2511F53u * c0;
        const unsigned long long p1 = (unsigned long long)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        const uint32_t n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        const uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    Philox4 o;
    o.v[0] = c0; o.v[1] = c1; o.v[2] = c2; o.v[3] = c3;
    return o;
}

// pre-scaled keep masks of elements 4*blk .. 4*blk+3 (element index = position in [n_steps][M][D]) at optimizer step `step`:
// keep (value 1/(1-p)) iff the 24-bit uniform u = word >> 8 satisfies u >= p * 2^24  (P[keep] = 1 - p)
// `stream` (counter word c2) separates independent mask streams drawn under one (seed, step): 0 = the ConvGRU's carried state,
// 1 = the classifier head's Dropout(0.5) (eval/model_3d_lc.py:42) -- never seed + k, which would collide with rank k's stream.
#define DPC_PHILOX_STREAM_GRU 0u
#define DPC_PHILOX_STREAM_LC_FC 1u
__device__ __host__ __forceinline__ void dropout_keep4(uint64_t seed, uint32_t step, uint32_t blk, uint32_t thresh24, float inv_keep, float* out4,
                                                       uint32_t stream = DPC_PHILOX_STREAM_GRU) {
    const Philox4 r = philox4x32_10(blk, step, stream, 0u, (uint32_t)seed, (uint32_t)(seed >> 32));
    DPC_UNROLL
    for (int e = 0; e < 4; ++e) out4[e] = (r.v[e] >> 8) >= thresh24 ? inv_keep : 0.f;
}
static inline uint32_t dropout_thresh24(float p) { return (uint32_t)((double)p * 16777216.0 + 0.5); }
