// lc_head.hip -- what the downstream classifier LC (eval/model_3d_lc.py:12-65) adds around the shared 2d3d-ResNet and
// ConvGRU kernels: ReLU -> temporal mean of the backbone features (the OTHER order than DPC_RNN, model_3d_lc.py:52-54),
// and the head: spatial mean of the last ConvGRU state, BatchNorm1d (running statistics), Dropout, Linear,
// CrossEntropyLoss + top-1 (eval/test.py:244-255), forward and backward.  Tiny tensors ([B][256] -> [B][num_class]):
// plain HBM/L2-resident kernels, one thread per channel / class; the heavy lifting stays in the backbone kernels.
#include "dpc_rt.h"
#include "../../include/dpc_hip.h"
#include "philox.h"

static inline unsigned grid_for(long long n, int block = 256, int cap = 8192) {
    long long g = (n + block - 1) / block;
    if (g < 1) g = 1;
    if (g > cap) g = cap;
    return (unsigned)g;
}

// ---------------------------------------------------------------- relu -> temporal mean  (model_3d_lc.py:52-54)
// x [B*N][T][SQ][D] -> feat [N][B*SQ][D] = mean_t relu(x)
template <class T>
__global__ void relu_tpool_fwd_kernel(const T* x, int B, int N, int Tt, int SQ, int D, T* feat) {
    constexpr int E = Elt<T>::PER16;
    const int upr = D / E;
    const long long units = (long long)B * N * SQ * upr;
    const float inv = 1.f / (float)Tt;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < units; i += (long long)gridDim.x * blockDim.x) {
        const int cu = (int)(i % upr);
        long long q = i / upr;
        const int s = (int)(q % SQ);
        q /= SQ;
        const int n = (int)(q % N);
        const int b = (int)(q / N);
        float m[E];
        DPC_UNROLL
        for (int e = 0; e < E; ++e) m[e] = 0.f;
        for (int t = 0; t < Tt; ++t) {
            const u32x4 v = ((const u32x4*)x)[((((long long)b * N + n) * Tt + t) * SQ + s) * upr + cu];
            DPC_UNROLL
            for (int e = 0; e < E; ++e) { const float f = unit_get<T>(v, e); m[e] += f > 0.f ? f : 0.f; }
        }
        DPC_UNROLL
        for (int e = 0; e < E; ++e) m[e] *= inv;
        ((u32x4*)feat)[(((long long)n * B + b) * SQ + s) * upr + cu] = unit_pack<T>(m);
    }
}

// dx[b*N+n][t][s][d] = (x > 0) * d_feat[n][b*SQ+s][d] / T
template <class T>
__global__ void relu_tpool_bwd_kernel(const T* x, const float* d_feat, int B, int N, int Tt, int SQ, int D, T* dx) {
    constexpr int E = Elt<T>::PER16;
    const int upr = D / E;
    const long long units = (long long)B * N * SQ * upr;
    const float inv = 1.f / (float)Tt;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < units; i += (long long)gridDim.x * blockDim.x) {
        const int cu = (int)(i % upr);
        long long q = i / upr;
        const int s = (int)(q % SQ);
        q /= SQ;
        const int n = (int)(q % N);
        const int b = (int)(q / N);
        const float* g = d_feat + (((long long)n * B + b) * SQ + s) * D + cu * E;
        for (int t = 0; t < Tt; ++t) {
            const long long o = ((((long long)b * N + n) * Tt + t) * SQ + s) * upr + cu;
            const u32x4 v = ((const u32x4*)x)[o];
            float r[E];
            DPC_UNROLL
            for (int e = 0; e < E; ++e) r[e] = unit_get<T>(v, e) > 0.f ? g[e] * inv : 0.f;
            ((u32x4*)dx)[o] = unit_pack<T>(r);
        }
    }
}

extern "C" int dpc_relu_tpool_fwd(const void* x, int32_t dtype, int32_t B, int32_t N, int32_t T, int32_t SQ, int32_t D, void* feat,
                                  dpc_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!x || !feat || B <= 0 || N <= 0 || T <= 0 || SQ <= 0 || D <= 0) return DPC_ERR_ARG;
    const int E = dtype == DPC_BF16 ? 8 : 4;
    if (D % E) return DPC_ERR_UNSUPPORTED;
    const long long units = (long long)B * N * SQ * (D / E);
    if (dtype == DPC_F32) {
        DPC_LAUNCH((relu_tpool_fwd_kernel<float>), dim3(grid_for(units)), dim3(256), stream, (const float*)x, B, N, T, SQ, D, (float*)feat);
    } else if (dtype == DPC_BF16) {
        DPC_LAUNCH((relu_tpool_fwd_kernel<bf16_t>), dim3(grid_for(units)), dim3(256), stream, (const bf16_t*)x, B, N, T, SQ, D, (bf16_t*)feat);
    } else {
        return DPC_ERR_ARG;
    }
    return dpc_launch_status();
}

extern "C" int dpc_relu_tpool_bwd(const void* x, const float* d_feat, int32_t dtype, int32_t B, int32_t N, int32_t T, int32_t SQ,
                                  int32_t D, void* dx, dpc_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!x || !d_feat || !dx || B <= 0 || N <= 0 || T <= 0 || SQ <= 0 || D <= 0) return DPC_ERR_ARG;
    const int E = dtype == DPC_BF16 ? 8 : 4;
    if (D % E) return DPC_ERR_UNSUPPORTED;
    const long long units = (long long)B * N * SQ * (D / E);
    if (dtype == DPC_F32) {
        DPC_LAUNCH((relu_tpool_bwd_kernel<float>), dim3(grid_for(units)), dim3(256), stream, (const float*)x, d_feat, B, N, T, SQ, D, (float*)dx);
    } else if (dtype == DPC_BF16) {
        DPC_LAUNCH((relu_tpool_bwd_kernel<bf16_t>), dim3(grid_for(units)), dim3(256), stream, (const bf16_t*)x, d_feat, B, N, T, SQ, D, (bf16_t*)dx);
    } else {
        return DPC_ERR_ARG;
    }
    return dpc_launch_status();
}

// ---------------------------------------------------------------- head forward
// 1. ctx[b][d] = mean_s h[(b,s)][d];  BatchNorm1d over the batch (train: biased batch variance, running buffers updated with
//    momentum and the unbiased variance; eval: running statistics);  y = bn * dropout mask        (model_3d_lc.py:58-64)
template <class T>
__global__ void lc_ctx_bn_kernel(const T* h, int B, int SQ, int D, const float* gamma, const float* beta, float* rmean, float* rvar,
                                 long long* nbt, float momentum, float eps, int train, const float* drop, const int32_t* step_dev,
                                 unsigned long long seed, uint32_t thresh24, float inv_keep, float* ctx, float* xhat, float* bnout,
                                 float* y, float* stat /* [2][D]: mean, invstd used */) {
    const int d = blockIdx.x * blockDim.x + threadIdx.x;
    if (d == 0 && train && nbt) nbt[0] += 1;
    if (d >= D) return;
    float s1 = 0.f, s2 = 0.f;
    for (int b = 0; b < B; ++b) {
        float c = 0.f;
        for (int s = 0; s < SQ; ++s) c += Elt<T>::to_f32(h[((long long)b * SQ + s) * D + d]);
        c *= 1.f / (float)SQ;
        ctx[(long long)b * D + d] = c;
        s1 += c;
        s2 += c * c;
    }
    float mean, var;
    if (train) {
        mean = s1 / (float)B;
        var = s2 / (float)B - mean * mean;
        if (var < 0.f) var = 0.f;
        const float unb = B > 1 ? var * (float)B / (float)(B - 1) : var;
        rmean[d] = (1.f - momentum) * rmean[d] + momentum * mean;
        rvar[d] = (1.f - momentum) * rvar[d] + momentum * unb;
    } else {
        mean = rmean[d];
        var = rvar[d];
    }
    const float is = 1.f / sqrtf(var + eps);
    stat[d] = mean;
    stat[D + d] = is;
    const uint32_t step = step_dev ? (uint32_t)step_dev[0] : 0u;
    for (int b = 0; b < B; ++b) {
        const long long o = (long long)b * D + d;
        const float xh = (ctx[o] - mean) * is;
        const float bn = xh * gamma[d] + beta[d];
        float k = 1.f;
        if (drop) k = drop[o];
        else if (step_dev) {
            float kk[4];
            dropout_keep4(seed, step, (uint32_t)(o >> 2), thresh24, inv_keep, kk, DPC_PHILOX_STREAM_LC_FC);
            k = kk[o & 3];
        }
        xhat[o] = xh;
        bnout[o] = bn;
        y[o] = bn * k;
    }
}

// 2. logits[b][c] = y[b] . W[c] + bias[c];  CrossEntropyLoss term, top-1 hit and d loss / d logits per clip
__global__ __launch_bounds__(256) void lc_fc_ce_kernel(const float* y, const float* W, const float* bias, const int64_t* target, int B, int D,
                                                        int NC, float* logits, float* dlogits, float* row_ws) {
    __shared__ float lg[1024];
    __shared__ float red[256];
    const int b = blockIdx.x, tid = threadIdx.x;
    for (int c = tid; c < NC; c += 256) {
        float a = bias[c];
        for (int d = 0; d < D; ++d) a += y[(long long)b * D + d] * W[(long long)c * D + d];
        lg[c] = a;
        logits[(long long)b * NC + c] = a;
    }
    __syncthreads();
    float mx = -3.0e38f;
    for (int c = tid; c < NC; c += 256) mx = lg[c] > mx ? lg[c] : mx;
    red[tid] = mx;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
        if (tid < st) red[tid] = red[tid + st] > red[tid] ? red[tid + st] : red[tid];
        __syncthreads();
    }
    mx = red[0];
    __syncthreads();
    float se = 0.f;
    for (int c = tid; c < NC; c += 256) se += expf(lg[c] - mx);
    red[tid] = se;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
        if (tid < st) red[tid] += red[tid + st];
        __syncthreads();
    }
    se = red[0];
    const int t = (int)target[b];
    const float invB = 1.f / (float)B;
    for (int c = tid; c < NC; c += 256) {
        float g = expf(lg[c] - mx) / se;
        if (c == t) g -= 1.f;
        dlogits[(long long)b * NC + c] = g * invB;
    }
    if (tid == 0) {
        int arg = 0;  // torch.max returns the first maximum
        for (int c = 1; c < NC; ++c)
            if (lg[c] > lg[arg]) arg = c;
        row_ws[2 * b + 0] = logf(se) + mx - lg[t];
        row_ws[2 * b + 1] = arg == t ? 1.f : 0.f;
    }
}

__global__ void lc_result_kernel(const float* row_ws, int B, float* result) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        float l = 0.f, a = 0.f;
        for (int b = 0; b < B; ++b) { l += row_ws[2 * b]; a += row_ws[2 * b + 1]; }
        result[0] = l / (float)B;
        result[1] = a / (float)B;
    }
}

extern "C" int dpc_lc_head_fwd(const dpc_lc_head_desc* c, dpc_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!c || c->B <= 0 || c->SQ <= 0 || c->D <= 0 || c->num_class <= 0 || c->num_class > 1024) return DPC_ERR_ARG;
    if (!c->h_last || !c->bn_weight || !c->bn_bias || !c->bn_running_mean || !c->bn_running_var || !c->fc_weight || !c->fc_bias ||
        !c->target || !c->ctx || !c->xhat || !c->bn_out || !c->y || !c->stat || !c->logits || !c->dlogits || !c->row_ws || !c->result)
        return DPC_ERR_ARG;
    if (!(c->p_drop >= 0.f) || !(c->p_drop < 1.f)) return DPC_ERR_ARG;
    const int32_t* step_dev = (c->train && !c->drop_mask && c->p_drop > 0.f) ? c->step_dev : nullptr;
    const float* drop = c->train ? c->drop_mask : nullptr;
    const dim3 g1((c->D + 255) / 256);
    if (c->dtype == DPC_F32) {
        DPC_LAUNCH((lc_ctx_bn_kernel<float>), g1, dim3(256), stream, (const float*)c->h_last, c->B, c->SQ, c->D, c->bn_weight, c->bn_bias,
                   c->bn_running_mean, c->bn_running_var, (long long*)c->bn_num_batches, c->momentum, c->eps, c->train, drop, step_dev,
                   (unsigned long long)c->seed, dropout_thresh24(c->p_drop), 1.f / (1.f - c->p_drop), c->ctx, c->xhat, c->bn_out, c->y, c->stat);
    } else if (c->dtype == DPC_BF16) {
        DPC_LAUNCH((lc_ctx_bn_kernel<bf16_t>), g1, dim3(256), stream, (const bf16_t*)c->h_last, c->B, c->SQ, c->D, c->bn_weight, c->bn_bias,
                   c->bn_running_mean, c->bn_running_var, (long long*)c->bn_num_batches, c->momentum, c->eps, c->train, drop, step_dev,
                   (unsigned long long)c->seed, dropout_thresh24(c->p_drop), 1.f / (1.f - c->p_drop), c->ctx, c->xhat, c->bn_out, c->y, c->stat);
    } else {
        return DPC_ERR_ARG;
    }
    DPC_LAUNCH(lc_fc_ce_kernel, dim3(c->B), dim3(256), stream, (const float*)c->y, c->fc_weight, c->fc_bias, c->target, c->B, c->D, c->num_class,
               c->logits, c->dlogits, c->row_ws);
    DPC_LAUNCH(lc_result_kernel, dim3(1), dim3(64), stream, (const float*)c->row_ws, c->B, c->result);
    return dpc_launch_status();
}

// ---------------------------------------------------------------- head backward (train mode)
// dW[c][d] = sum_b dlogits[b][c] y[b][d];  db[c] = sum_b dlogits[b][c]
__global__ void lc_fc_wgrad_kernel(const float* dlogits, const float* y, int B, int D, int NC, float* dW, float* db) {
    const int c = blockIdx.x;
    for (int d = threadIdx.x; d < D; d += blockDim.x) {
        float a = 0.f;
        for (int b = 0; b < B; ++b) a += dlogits[(long long)b * NC + c] * y[(long long)b * D + d];
        dW[(long long)c * D + d] = a;
    }
    if (threadIdx.x == 0) {
        float a = 0.f;
        for (int b = 0; b < B; ++b) a += dlogits[(long long)b * NC + c];
        db[c] = a;
    }
}

// per channel d: dy -> dropout -> BatchNorm1d backward (batch statistics) -> d ctx -> d h_last[(b,s)][d] = d ctx[b][d] / SQ
__global__ void lc_bn_bwd_kernel(const float* dlogits, const float* W, const float* y, const float* bnout, const float* xhat, const float* stat,
                                 const float* gamma, int B, int SQ, int D, int NC, float* dgamma, float* dbeta, float* dctx, float* d_hlast) {
    const int d = blockIdx.x * blockDim.x + threadIdx.x;
    if (d >= D) return;
    float s1 = 0.f, s2 = 0.f;
    for (int b = 0; b < B; ++b) {
        const long long o = (long long)b * D + d;
        float dy = 0.f;
        for (int c = 0; c < NC; ++c) dy += dlogits[(long long)b * NC + c] * W[(long long)c * D + d];
        const float bn = bnout[o];
        const float k = bn != 0.f ? y[o] / bn : 0.f;  // the keep mask value (0 or 1/(1-p)); bn == 0 contributes nothing anyway
        const float dz = dy * k;
        dctx[o] = dz;
        s1 += dz;
        s2 += dz * xhat[o];
    }
    dbeta[d] = s1;
    dgamma[d] = s2;
    const float is = stat[D + d], g = gamma[d];
    const float c1 = s1 / (float)B, c2 = s2 / (float)B;
    const float inv_sq = 1.f / (float)SQ;
    for (int b = 0; b < B; ++b) {
        const long long o = (long long)b * D + d;
        const float dc = g * is * (dctx[o] - c1 - xhat[o] * c2);
        dctx[o] = dc;
        for (int s = 0; s < SQ; ++s) d_hlast[((long long)b * SQ + s) * D + d] = dc * inv_sq;
    }
}

extern "C" int dpc_lc_head_bwd(const dpc_lc_head_desc* c, dpc_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!c || c->B <= 0 || c->SQ <= 0 || c->D <= 0 || c->num_class <= 0) return DPC_ERR_ARG;
    if (!c->train) return DPC_ERR_ARG;  // eval-mode BatchNorm1d has no batch-statistics backward here (the reference never needs it)
    if (!c->dlogits || !c->y || !c->bn_out || !c->xhat || !c->stat || !c->bn_weight || !c->fc_weight || !c->g_fc_weight || !c->g_fc_bias ||
        !c->g_bn_weight || !c->g_bn_bias || !c->dctx || !c->d_hlast)
        return DPC_ERR_ARG;
    DPC_LAUNCH(lc_fc_wgrad_kernel, dim3(c->num_class), dim3(256), stream, (const float*)c->dlogits, (const float*)c->y, c->B, c->D, c->num_class,
               c->g_fc_weight, c->g_fc_bias);
    DPC_LAUNCH(lc_bn_bwd_kernel, dim3((c->D + 255) / 256), dim3(256), stream, (const float*)c->dlogits, c->fc_weight, (const float*)c->y,
               (const float*)c->bn_out, (const float*)c->xhat, (const float*)c->stat, c->bn_weight, c->B, c->SQ, c->D, c->num_class, c->g_bn_weight,
               c->g_bn_bias, c->dctx, c->d_hlast);
    return dpc_launch_status();
}
