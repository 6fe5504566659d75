// gru.hip -- gate math of the kernel-size-1 ConvGRU (backbone/convrnn.py:24-34,76-79) and the
// small pointwise pieces of network_pred / the predict loop (dpc/model_3d.py:36-40,66-71).
// The 1x1 convolutions themselves are GEMMs on the matrix cores (conv_igemm.hip); these
// kernels consume their f32 outputs.  Rows are (b, s) positions, M = B*SQ, D channels.
#include "dpc_rt.h"
#include "../../include/dpc_hip.h"
#include "philox.h"

static inline unsigned grid_for(long long n, int block = 256, int cap = 8192) {
    long long g = (n + block - 1) / block;
    if (g < 1) g = 1;
    if (g > cap) g = cap;
    return (unsigned)g;
}

__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + expf(-x)); }

// u = sig(px_u+ph_u+bu), r = sig(px_r+ph_r+br), hr = h*r     px [M][3D] = [u|r|o], ph [M][2D] = [u|r]
template <class T>
__global__ void gru_gates1_kernel(const float* px, const float* ph, const float* bu, const float* br, const T* h, int M,
                                  int D, float* u, float* r, T* hr) {
    const long long n = (long long)M * D;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const int d = (int)(i % D);
        const long long m = i / D;
        const float uu = sigmoidf_(px[m * 3 * D + d] + ph[m * 2 * D + d] + bu[d]);
        const float rr = sigmoidf_(px[m * 3 * D + D + d] + ph[m * 2 * D + D + d] + br[d]);
        u[i] = uu;
        r[i] = rr;
        hr[i] = Elt<T>::from_f32(Elt<T>::to_f32(h[i]) * rr);
    }
}

extern "C" int dpc_gru_gates1(const float* px, const float* ph, const float* bias_u, const float* bias_r, const void* h,
                              int32_t dtype, int32_t M, int32_t D, float* u, float* r, void* hr, dpc_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!px || !ph || !bias_u || !bias_r || !h || !u || !r || !hr || M <= 0 || D <= 0) return DPC_ERR_ARG;
    const long long n = (long long)M * D;
    if (dtype == DPC_F32) {
        DPC_LAUNCH((gru_gates1_kernel<float>), dim3(grid_for(n)), dim3(256), stream, px, ph, bias_u, bias_r, (const float*)h, M, D, u, r, (float*)hr);
    } else if (dtype == DPC_BF16) {
        DPC_LAUNCH((gru_gates1_kernel<bf16_t>), dim3(grid_for(n)), dim3(256), stream, px, ph, bias_u, bias_r, (const bf16_t*)h, M, D, u, r, (bf16_t*)hr);
    } else {
        return DPC_ERR_ARG;
    }
    return dpc_launch_status();
}

// o = tanh(px_o+po+bo), hn = h*(1-u)+o*u, hout = hn*drop (drop: pre-scaled keep mask, Dropout p=0.1
// on the carried state every step, convrnn.py:78; NULL = eval)
template <class T>
__global__ void gru_gates2_kernel(const float* px, const float* po, const float* bo, const T* h, const float* u,
                                  const float* drop, int M, int D, float* o, T* hout) {
    const long long n = (long long)M * D;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const int d = (int)(i % D);
        const long long m = i / D;
        const float oo = tanhf(px[m * 3 * D + 2 * D + d] + po[i] + bo[d]);
        const float uu = u[i];
        float hn = Elt<T>::to_f32(h[i]) * (1.f - uu) + oo * uu;
        if (drop) hn *= drop[i];
        o[i] = oo;
        hout[i] = Elt<T>::from_f32(hn);
    }
}

extern "C" int dpc_gru_gates2(const float* px, const float* po, const float* bias_o, const void* h, const float* u,
                              const float* drop, int32_t dtype, int32_t M, int32_t D, float* o, void* hout,
                              dpc_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!px || !po || !bias_o || !h || !u || !o || !hout || M <= 0 || D <= 0) return DPC_ERR_ARG;
    const long long n = (long long)M * D;
    if (dtype == DPC_F32) {
        DPC_LAUNCH((gru_gates2_kernel<float>), dim3(grid_for(n)), dim3(256), stream, px, po, bias_o, (const float*)h, u, drop, M, D, o, (float*)hout);
    } else if (dtype == DPC_BF16) {
        DPC_LAUNCH((gru_gates2_kernel<bf16_t>), dim3(grid_for(n)), dim3(256), stream, px, po, bias_o, (const bf16_t*)h, u, drop, M, D, o, (bf16_t*)hout);
    } else {
        return DPC_ERR_ARG;
    }
    return dpc_launch_status();
}

// backward, first half.  dhn = dh*drop; do = dhn*u; du = dhn*(o-h); dhprev = dhn*(1-u)
// G[:,0:D] = du*u*(1-u) (pre-sigmoid update grad), G[:,2D:3D] = do*(1-o^2) (pre-tanh out grad)
template <class T>
__global__ void gru_bwd1_kernel(const float* dh, const float* drop, const float* u, const float* o, const T* h, int M, int D,
                                T* G, float* dhprev) {
    const long long n = (long long)M * D;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const int d = (int)(i % D);
        const long long m = i / D;
        float dhn = dh[i];
        if (drop) dhn *= drop[i];
        const float uu = u[i], oo = o[i], hh = Elt<T>::to_f32(h[i]);
        G[m * 3 * D + d] = Elt<T>::from_f32(dhn * (oo - hh) * uu * (1.f - uu));
        G[m * 3 * D + 2 * D + d] = Elt<T>::from_f32(dhn * uu * (1.f - oo * oo));
        dhprev[i] = dhn * (1.f - uu);
    }
}

extern "C" int dpc_gru_bwd1(const float* dh, const float* drop, const float* u, const float* o, const void* h, int32_t dtype,
                            int32_t M, int32_t D, void* G, float* dhprev, dpc_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!dh || !u || !o || !h || !G || !dhprev || M <= 0 || D <= 0) return DPC_ERR_ARG;
    const long long n = (long long)M * D;
    if (dtype == DPC_F32) {
        DPC_LAUNCH((gru_bwd1_kernel<float>), dim3(grid_for(n)), dim3(256), stream, dh, drop, u, o, (const float*)h, M, D, (float*)G, dhprev);
    } else if (dtype == DPC_BF16) {
        DPC_LAUNCH((gru_bwd1_kernel<bf16_t>), dim3(grid_for(n)), dim3(256), stream, dh, drop, u, o, (const bf16_t*)h, M, D, (bf16_t*)G, dhprev);
    } else {
        return DPC_ERR_ARG;
    }
    return dpc_launch_status();
}

// backward, second half.  dhr = dpo * Wo_h (GEMM);  dr = dhr*h; G[:,D:2D] = dr*r*(1-r); dhprev += dhr*r
template <class T>
__global__ void gru_bwd2_kernel(const float* dhr, const float* r, const T* h, int M, int D, T* G, float* dhprev) {
    const long long n = (long long)M * D;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const int d = (int)(i % D);
        const long long m = i / D;
        const float g = dhr[i], rr = r[i], hh = Elt<T>::to_f32(h[i]);
        G[m * 3 * D + D + d] = Elt<T>::from_f32(g * hh * rr * (1.f - rr));
        dhprev[i] += g * rr;
    }
}

extern "C" int dpc_gru_bwd2(const float* dhr, const float* r, const void* h, int32_t dtype, int32_t M, int32_t D, void* G,
                            float* dhprev, dpc_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!dhr || !r || !h || !G || !dhprev || M <= 0 || D <= 0) return DPC_ERR_ARG;
    const long long n = (long long)M * D;
    if (dtype == DPC_F32) {
        DPC_LAUNCH((gru_bwd2_kernel<float>), dim3(grid_for(n)), dim3(256), stream, dhr, r, (const float*)h, M, D, (float*)G, dhprev);
    } else if (dtype == DPC_BF16) {
        DPC_LAUNCH((gru_bwd2_kernel<bf16_t>), dim3(grid_for(n)), dim3(256), stream, dhr, r, (const bf16_t*)h, M, D, (bf16_t*)G, dhprev);
    } else {
        return DPC_ERR_ARG;
    }
    return dpc_launch_status();
}

// y[rowmap(m)][d] = act(x[m][d] + bias[d]);  y2[m][d] = relu(x[m][d]+bias[d]) (optional)
// rowmap(m) = (m/SQ*P + p)*SQ + m%SQ when P > 0 (step p of pred [B][P][SQ][D]), identity otherwise
template <class TY, class TY2>
__global__ void bias_act_kernel(const float* x, const float* bias, int M, int D, int relu, TY* y, int P, int p, int SQ, TY2* y2) {
    const long long n = (long long)M * D;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const int d = (int)(i % D);
        const int m = (int)(i / D);
        float v = x[i] + bias[d];
        const float vr = v > 0.f ? v : 0.f;
        if (relu) v = vr;
        long long row = m;
        if (P > 0) row = ((long long)(m / SQ) * P + p) * SQ + (m % SQ);
        y[row * D + d] = Elt<TY>::from_f32(v);
        if (y2) y2[i] = Elt<TY2>::from_f32(vr);
    }
}

extern "C" int dpc_bias_act(const float* x, const float* bias, int32_t M, int32_t D, int32_t relu, void* y, int32_t dtype_y,
                            int32_t P, int32_t p, int32_t SQ, void* y2, int32_t dtype_y2, dpc_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!x || !bias || !y || M <= 0 || D <= 0 || (P > 0 && (SQ <= 0 || p < 0 || p >= P || M % SQ))) return DPC_ERR_ARG;
    const long long n = (long long)M * D;
    dim3 grid(grid_for(n)), block(256);
    const int key = dtype_y * 2 + (y2 ? dtype_y2 : dtype_y);
    if (key == 0) {
        DPC_LAUNCH((bias_act_kernel<float, float>), grid, block, stream, x, bias, M, D, relu, (float*)y, P, p, SQ, (float*)y2);
    } else if (key == 1) {
        DPC_LAUNCH((bias_act_kernel<float, bf16_t>), grid, block, stream, x, bias, M, D, relu, (float*)y, P, p, SQ, (bf16_t*)y2);
    } else if (key == 2) {
        DPC_LAUNCH((bias_act_kernel<bf16_t, float>), grid, block, stream, x, bias, M, D, relu, (bf16_t*)y, P, p, SQ, (float*)y2);
    } else if (key == 3) {
        DPC_LAUNCH((bias_act_kernel<bf16_t, bf16_t>), grid, block, stream, x, bias, M, D, relu, (bf16_t*)y, P, p, SQ, (bf16_t*)y2);
    } else {
        return DPC_ERR_ARG;
    }
    return dpc_launch_status();
}

// out = dy * (y > 0) (+ add)
template <class TY, class TO>
__global__ void relu_bwd_kernel(const float* dy, const TY* y, const float* add, long long n, TO* out) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        float v = (Elt<TY>::to_f32(y[i]) > 0.f) ? dy[i] : 0.f;
        if (add) v += add[i];
        out[i] = Elt<TO>::from_f32(v);
    }
}

extern "C" int dpc_relu_bwd(const float* dy, const void* y, int32_t dtype_y, const float* add, int64_t n, void* out,
                            int32_t dtype_out, dpc_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!dy || !y || !out || n <= 0) return DPC_ERR_ARG;
    dim3 grid(grid_for(n)), block(256);
    const int key = dtype_y * 2 + dtype_out;
    if (key == 0) {
        DPC_LAUNCH((relu_bwd_kernel<float, float>), grid, block, stream, dy, (const float*)y, add, (long long)n, (float*)out);
    } else if (key == 1) {
        DPC_LAUNCH((relu_bwd_kernel<float, bf16_t>), grid, block, stream, dy, (const float*)y, add, (long long)n, (bf16_t*)out);
    } else if (key == 2) {
        DPC_LAUNCH((relu_bwd_kernel<bf16_t, float>), grid, block, stream, dy, (const bf16_t*)y, add, (long long)n, (float*)out);
    } else if (key == 3) {
        DPC_LAUNCH((relu_bwd_kernel<bf16_t, bf16_t>), grid, block, stream, dy, (const bf16_t*)y, add, (long long)n, (bf16_t*)out);
    } else {
        return DPC_ERR_ARG;
    }
    return dpc_launch_status();
}

// out[d] (+)= sum_m x[m*ld + d]   -- bias gradients.  Stage 1: grid (D/32, ny), 8 row groups x 32
// columns per workgroup over a slice of the rows -> ws[ny][D]; stage 2 sums the ny slices in fixed
// order (deterministic).  One workgroup per column group over all 14k rows was latency-bound.
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

// dst[m][d] = src[(m/SQ*P + p)*SQ + m%SQ][d] (+ add[m][d])  -- pull step p out of a [B][P][SQ][D] gradient
__global__ void gather_rows_kernel(const float* src, int B, int P, int p, int SQ, int D, float* dst, const float* add) {
    const long long n = (long long)B * SQ * D;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const int d = (int)(i % D);
        const int m = (int)(i / D);
        const long long row = ((long long)(m / SQ) * P + p) * SQ + (m % SQ);
        float v = src[row * D + d];
        if (add) v += add[i];
        dst[i] = v;
    }
}

extern "C" int dpc_gather_rows(const float* src, int32_t B, int32_t P, int32_t p, int32_t SQ, int32_t D, float* dst,
                               const float* add, dpc_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!src || !dst || B <= 0 || P <= 0 || p < 0 || p >= P || SQ <= 0 || D <= 0) return DPC_ERR_ARG;
    DPC_LAUNCH(gather_rows_kernel, dim3(grid_for((long long)B * SQ * D)), dim3(256), stream, src, B, P, p, SQ, D, dst, add);
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
