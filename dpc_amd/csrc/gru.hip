// gru.hip -- small pieces around the fused ConvGRU recurrence (csrc/gru_chain.hip): column sums for the bias gradients
// of the ConvGRU / network_pred (backbone/convrnn.py:13-15, dpc/model_3d.py:36-40) and the standalone dropout-mask
// generator (the Philox stream the recurrence kernels use in train mode, backbone/convrnn.py:78).
#include "dpc_rt.h"
#include "../../include/dpc_hip.h"
#include "philox.h"

static inline unsigned grid_for(long long n, int block = 256, int cap = 8192) {
    long long g = (n + block - 1) / block;
    if (g < 1) g = 1;
    if (g > cap) g = cap;
    return (unsigned)g;
}

template <class T>
__global__ __launch_bounds__(256) void colsum_kernel(const T* x, int ld, int M, int D, float* out, int rows_per_y) {
    __shared__ float red[8][32];
    const int col = blockIdx.x * 32 + (threadIdx.x & 31);
    const int rg = threadIdx.x >> 5;
    const int m0 = blockIdx.y * rows_per_y;
    const int m1 = (m0 + rows_per_y < M) ? m0 + rows_per_y : M;
    float s = 0.f;
    if (col < D)
        for (int m = m0 + rg; m < m1; m += 8) s += Elt<T>::to_f32(x[(long long)m * ld + col]);
    red[rg][threadIdx.x & 31] = s;
    __syncthreads();
    if (threadIdx.x < 32 && col < D) {
        float t = 0.f;
        DPC_UNROLL
        for (int g = 0; g < 8; ++g) t += red[g][threadIdx.x];
        out[(long long)blockIdx.y * D + col] = t;
    }
}

__global__ void colsum_finish_kernel(const float* ws, int ny, int D, float* out, int accumulate) {
    const int d = blockIdx.x * blockDim.x + threadIdx.x;
    if (d >= D) return;
    float t = 0.f;
    for (int y = 0; y < ny; ++y) t += ws[(long long)y * D + d];
    out[d] = accumulate ? out[d] + t : t;
}

extern "C" int dpc_colsum(const void* x, int32_t dtype, int32_t ld, int32_t M, int32_t D, float* out, int32_t accumulate,
                          float* ws, int64_t ws_floats, dpc_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!x || !out || M <= 0 || D <= 0 || ld < D) return DPC_ERR_ARG;
    int ny = (M + 255) / 256;
    if (ny > 64) ny = 64;
    if (!ws || ws_floats < (long long)ny * D) return DPC_ERR_ARG;
    const int rpy = (M + ny - 1) / ny;
    dim3 grid((D + 31) / 32, ny), block(256);
    if (dtype == DPC_F32) {
        DPC_LAUNCH((colsum_kernel<float>), grid, block, stream, (const float*)x, ld, M, D, ws, rpy);
    } else if (dtype == DPC_BF16) {
        DPC_LAUNCH((colsum_kernel<bf16_t>), grid, block, stream, (const bf16_t*)x, ld, M, D, ws, rpy);
    } else {
        return DPC_ERR_ARG;
    }
    DPC_LAUNCH(colsum_finish_kernel, dim3((D + 255) / 256), dim3(256), stream, (const float*)ws, ny, D, out, accumulate);
    return dpc_launch_status();
}

// ---------------------------------------------------------------- dropout keep masks (convrnn.py:78), Philox4x32-10
// mask[i] = 1/(1-p) with probability 1-p, else 0, for i in [0, n) -- the masks of ALL recurrence steps of one optimizer
// step in one launch.  The optimizer-step counter is read from device memory so a captured hipGraph draws fresh masks on
// every replay (dpc_step_advance moves it).
__global__ void dropout_mask_kernel(float* mask, long long n4, long long n, unsigned long long seed, const int32_t* step_dev,
                                    uint32_t thresh24, float inv_keep) {
    const uint32_t step = (uint32_t)step_dev[0];
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
        float k[4];
        dropout_keep4(seed, step, (uint32_t)i, thresh24, inv_keep, k);
        if (i * 4 + 4 <= n) {
            ((f32x4*)mask)[i] = f32x4{k[0], k[1], k[2], k[3]};
        } else {
            for (long long e = i * 4; e < n; ++e) mask[e] = k[e - i * 4];
        }
    }
}

extern "C" int dpc_dropout_mask(float* mask, int64_t n, float p, uint64_t seed, const int32_t* step_dev, dpc_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!mask || n <= 0 || !step_dev || !(p >= 0.f) || !(p < 1.f) || n >= (1ll << 33)) return DPC_ERR_ARG;
    const long long n4 = (n + 3) / 4;
    DPC_LAUNCH(dropout_mask_kernel, dim3(grid_for(n4)), dim3(256), stream, mask, n4, (long long)n, (unsigned long long)seed, step_dev,
               dropout_thresh24(p), 1.f / (1.f - p));
    return dpc_launch_status();
}
