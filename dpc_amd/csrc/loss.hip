// loss.hip -- contrastive head and optimizer of the train step:
//   mask      dpc/model_3d.py:86-96 (closed form; the reference builds it with B + B*SQ
//             Python index assignments, every step under multi-GPU DataParallel)
//   ce_topk   nn.CrossEntropyLoss + calc_topk_accuracy(1,3,5) with target = arange
//             (dpc/main.py:178-185,213-218; utils/utils.py:38-55), plus d(loss)/d(score)
//   adam      torch.optim.Adam(lr, weight_decay as L2) on the flat parameter buffer (main.py:80-81)
// All HBM-bound; row reductions use wave64 shuffles then one LDS hop across the 4 waves.
#include "dpc_rt.h"
#include "../../include/dpc_hip.h"

static inline unsigned grid_for(long long n, int block = 256, int cap = 8192) {
    long long g = (n + block - 1) / block;
    if (g < 1) g = 1;
    if (g > cap) g = cap;
    return (unsigned)g;
}

// mask[b][p][s][b2][n][s2]: 1 pos (b=b2,s=s2,p=n) | -1 temporal neg (b=b2,s=s2,p!=n) | -3 spatial neg (b=b2,s!=s2) | 0
__global__ void mask_gen_kernel(int8_t* mask, int B, int P, int SQ) {
    const long long row_len = (long long)B * P * SQ;
    const long long n = row_len * row_len;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const long long r = i / row_len, c = i % row_len;
        const int s = (int)(r % SQ), p = (int)((r / SQ) % P), b = (int)(r / ((long long)SQ * P));
        const int s2 = (int)(c % SQ), n2 = (int)((c / SQ) % P), b2 = (int)(c / ((long long)SQ * P));
        int8_t v = 0;
        if (b == b2) v = (s != s2) ? -3 : ((p == n2) ? 1 : -1);
        mask[i] = v;
    }
}

extern "C" int dpc_mask_gen(int8_t* mask, int32_t B, int32_t P, int32_t SQ, dpc_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!mask || B <= 0 || P <= 0 || SQ <= 0) return DPC_ERR_ARG;
    const long long rl = (long long)B * P * SQ;
    DPC_LAUNCH(mask_gen_kernel, dim3(grid_for(rl * rl)), dim3(256), stream, mask, B, P, SQ);
    return dpc_launch_status();
}

__device__ __forceinline__ float wave_max(float v) {
    DPC_UNROLL
    for (int m = 32; m >= 1; m >>= 1) { const float o = __shfl_xor(v, m); v = o > v ? o : v; }
    return v;
}
__device__ __forceinline__ float wave_sum(float v) {
    DPC_UNROLL
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m);
    return v;
}

// one workgroup per score row; three L2-resident sweeps (max, sum-exp + rank, gradient)
template <class TD>
__global__ __launch_bounds__(256) void ce_row_kernel(const float* score, int rows, int cols, int ld, float* row_ws,
                                                     TD* dscore, int ld_d) {
    __shared__ float sh[8];
    const int row = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const float* s = score + (long long)row * ld;
    const float tgt = s[row];  // target column = row (closed form of process_output on one GPU)
    float mx = -3.0e38f;
    for (int j = tid; j < cols; j += 256) { const float v = s[j]; mx = v > mx ? v : mx; }
    mx = wave_max(mx);
    if (lane == 0) sh[wv] = mx;
    __syncthreads();
    mx = sh[0];
    DPC_UNROLL
    for (int w = 1; w < 4; ++w) mx = sh[w] > mx ? sh[w] : mx;
    float se = 0.f, rk = 0.f;
    for (int j = tid; j < cols; j += 256) {
        const float v = s[j];
        se += expf(v - mx);
        rk += (v > tgt) ? 1.f : 0.f;
    }
    se = wave_sum(se);
    rk = wave_sum(rk);
    __syncthreads();
    if (lane == 0) { sh[wv] = se; sh[4 + wv] = rk; }
    __syncthreads();
    se = sh[0] + sh[1] + sh[2] + sh[3];
    rk = sh[4] + sh[5] + sh[6] + sh[7];
    if (tid == 0) {
        row_ws[2 * row + 0] = logf(se) + mx - tgt;
        row_ws[2 * row + 1] = rk;
    }
    if (dscore) {
        const float inv = 1.f / se, invr = 1.f / (float)rows;
        TD* d = dscore + (long long)row * ld_d;
        for (int j = tid; j < cols; j += 256) {
            float g = expf(s[j] - mx) * inv;
            if (j == row) g -= 1.f;
            d[j] = Elt<TD>::from_f32(g * invr);
        }
        for (int j = cols + tid; j < ld_d; j += 256) d[j] = Elt<TD>::from_f32(0.f);
    }
}

// The same row kernel with the row held in REGISTERS (NV float4 per thread: one HBM read of the logits instead of three sweeps),
// 16-byte loads, one exponential per logit (kept for the gradient), 8/16-byte gradient stores.  Needs cols % 4 == 0, 16-byte
// aligned rows and cols <= 1024 NV; anything else takes ce_row_kernel.  FAST: v_exp_f32 (2^x, ~1 ulp) instead of libm expf --
// the throughput mode (bf16 gradient); the f32 parity mode keeps the exact form bit for bit.
// Measured at R = 6 144: ce_row_kernel 117 us for 151 MB of logits + 75 MB of gradient (1.9 TB/s).
template <class TD, int NV, bool FAST>
__global__ __launch_bounds__(256) void ce_row_vec_kernel(const float* score, int rows, int cols, int ld, float* row_ws, TD* dscore, int ld_d) {
    __shared__ float sh[8];
    const int row = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const float* s = score + (long long)row * ld;
    const float tgt = s[row];
    const int nq = cols >> 2;
    f32x4 v[NV];
    float mx = -3.0e38f;
    DPC_UNROLL
    for (int i = 0; i < NV; ++i) {
        const int q = tid + 256 * i;
        f32x4 t = {-3.0e38f, -3.0e38f, -3.0e38f, -3.0e38f};
        if (q < nq) t = ((const f32x4*)s)[q];
        v[i] = t;
        DPC_UNROLL
        for (int e = 0; e < 4; ++e) mx = t[e] > mx ? t[e] : mx;
    }
    mx = wave_max(mx);
    if (lane == 0) sh[wv] = mx;
    __syncthreads();
    mx = sh[0];
    DPC_UNROLL
    for (int w = 1; w < 4; ++w) mx = sh[w] > mx ? sh[w] : mx;
    float se = 0.f, rk = 0.f;
    const float nm = -mx * 1.4426950408889634f;
    DPC_UNROLL
    for (int i = 0; i < NV; ++i) {
        const bool ok = tid + 256 * i < nq;
        DPC_UNROLL
        for (int e = 0; e < 4; ++e) {
            const float x = v[i][e];
            rk += (ok && x > tgt) ? 1.f : 0.f;
            const float ex = FAST ? fast_exp2(fmaf(x, 1.4426950408889634f, nm)) : expf(x - mx);
            v[i][e] = ok ? ex : 0.f;
            se += v[i][e];
        }
    }
    se = wave_sum(se);
    rk = wave_sum(rk);
    __syncthreads();
    if (lane == 0) { sh[wv] = se; sh[4 + wv] = rk; }
    __syncthreads();
    se = sh[0] + sh[1] + sh[2] + sh[3];
    rk = sh[4] + sh[5] + sh[6] + sh[7];
    if (tid == 0) {
        row_ws[2 * row + 0] = logf(se) + mx - tgt;
        row_ws[2 * row + 1] = rk;
    }
    if (dscore) {
        const float inv = 1.f / se, invr = 1.f / (float)rows;
        TD* d = dscore + (long long)row * ld_d;
        DPC_UNROLL
        for (int i = 0; i < NV; ++i) {
            const int q = tid + 256 * i;
            if (q < nq) {
                float g[4];
                DPC_UNROLL
                for (int e = 0; e < 4; ++e) {
                    float t = v[i][e] * inv;
                    if (4 * q + e == row) t -= 1.f;
                    g[e] = t * invr;
                }
                if constexpr (sizeof(TD) == 2) {
                    const u32x2 o = {bf16x2_pack(g[0], g[1]), bf16x2_pack(g[2], g[3])};
                    *(u32x2*)(d + 4 * q) = o;
                } else {
                    const f32x4 o = {g[0], g[1], g[2], g[3]};
                    *(f32x4*)(d + 4 * q) = o;
                }
            }
        }
        for (int j = cols + tid; j < ld_d; j += 256) d[j] = Elt<TD>::from_f32(0.f);
    }
}


// bf16 LOGITS (round 6): the train step never returns its score (engine.train_step), so the materialised score of that path is
// written and read in the compute dtype -- 75 MB instead of 151 MB at R = 6 144 on each side.  Same one-read row kernel: a thread
// holds NV 16-byte units of 8 logits; loss, rank and d(loss)/d(score) are those of the ROUNDED logits (the target logit included,
// so "strictly greater than the target" is decided among values of one precision).
template <int NV>
__global__ __launch_bounds__(256) void ce_row_vec16_kernel(const bf16_t* score, int rows, int cols, int ld, float* row_ws, bf16_t* dscore, int ld_d) {
    __shared__ float sh[8];
    const int row = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const bf16_t* s = score + (long long)row * ld;
    const float tgt = bf16_to_f32(s[row]);
    const int nq = cols >> 3;
    float v[NV][8];
    float mx = -3.0e38f;
    DPC_UNROLL
    for (int i = 0; i < NV; ++i) {
        const int q = tid + 256 * i;
        if (q < nq) {
            const u32x4 t = ((const u32x4*)s)[q];
            DPC_UNROLL
            for (int e = 0; e < 8; ++e) { v[i][e] = unit_get<bf16_t>(t, e); mx = v[i][e] > mx ? v[i][e] : mx; }
        } else {
            DPC_UNROLL
            for (int e = 0; e < 8; ++e) v[i][e] = -3.0e38f;
        }
    }
    mx = wave_max(mx);
    if (lane == 0) sh[wv] = mx;
    __syncthreads();
    mx = sh[0];
    DPC_UNROLL
    for (int w = 1; w < 4; ++w) mx = sh[w] > mx ? sh[w] : mx;
    float se = 0.f, rk = 0.f;
    const float nm = -mx * 1.4426950408889634f;
    DPC_UNROLL
    for (int i = 0; i < NV; ++i) {
        const bool ok = tid + 256 * i < nq;
        DPC_UNROLL
        for (int e = 0; e < 8; ++e) {
            const float x = v[i][e];
            rk += (ok && x > tgt) ? 1.f : 0.f;
            const float ex = fast_exp2(fmaf(x, 1.4426950408889634f, nm));
            v[i][e] = ok ? ex : 0.f;
            se += v[i][e];
        }
    }
    se = wave_sum(se);
    rk = wave_sum(rk);
    __syncthreads();
    if (lane == 0) { sh[wv] = se; sh[4 + wv] = rk; }
    __syncthreads();
    se = sh[0] + sh[1] + sh[2] + sh[3];
    rk = sh[4] + sh[5] + sh[6] + sh[7];
    if (tid == 0) {
        row_ws[2 * row + 0] = logf(se) + mx - tgt;
        row_ws[2 * row + 1] = rk;
    }
    if (dscore) {
        const float inv = 1.f / se, invr = 1.f / (float)rows;
        bf16_t* d = dscore + (long long)row * ld_d;
        DPC_UNROLL
        for (int i = 0; i < NV; ++i) {
            const int q = tid + 256 * i;
            if (q < nq) {
                float g[8];
                DPC_UNROLL
                for (int e = 0; e < 8; ++e) {
                    float t = v[i][e] * inv;
                    if (8 * q + e == row) t -= 1.f;
                    g[e] = t * invr;
                }
                *(u32x4*)(d + 8 * q) = unit_pack<bf16_t>(g);
            }
        }
        for (int j = cols + tid; j < ld_d; j += 256) d[j] = Elt<bf16_t>::from_f32(0.f);
    }
}

__global__ __launch_bounds__(256) void ce_finalize_kernel(const float* row_ws, int rows, float* result) {
    __shared__ float sh[4][4];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    float a[4] = {0.f, 0.f, 0.f, 0.f};
    for (int r = tid; r < rows; r += 256) {
        const float rk = row_ws[2 * r + 1];
        a[0] += row_ws[2 * r];
        a[1] += rk < 1.f ? 1.f : 0.f;
        a[2] += rk < 3.f ? 1.f : 0.f;
        a[3] += rk < 5.f ? 1.f : 0.f;
    }
    DPC_UNROLL
    for (int k = 0; k < 4; ++k) {
        a[k] = wave_sum(a[k]);
        if (lane == 0) sh[wv][k] = a[k];
    }
    __syncthreads();
    if (tid < 4) result[tid] = (sh[0][tid] + sh[1][tid] + sh[2][tid] + sh[3][tid]) / (float)rows;
}

extern "C" int dpc_ce_topk(const float* score, int32_t rows, int32_t cols, int32_t ld, float* row_ws, float* result,
                           void* dscore, int32_t dtype_d, int32_t ld_d, dpc_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!score || !row_ws || !result || rows <= 0 || cols < rows || ld < cols) return DPC_ERR_ARG;
    if (dscore && ld_d < cols) return DPC_ERR_ARG;
    // register-resident row form: 16-byte aligned rows of the logits (and of the gradient), cols a multiple of 4, at most 16 K columns
    const bool vec = cols % 4 == 0 && ld % 4 == 0 && ((uintptr_t)score % 16) == 0 && cols <= 16384 &&
                     (!dscore || (ld_d % 4 == 0 && ((uintptr_t)dscore % 16) == 0));
    if (!dscore || dtype_d == DPC_F32) {
        if (vec && cols <= 8192) {
            DPC_LAUNCH((ce_row_vec_kernel<float, 8, false>), dim3(rows), dim3(256), stream, score, rows, cols, ld, row_ws, (float*)dscore, ld_d);
        } else if (vec) {
            DPC_LAUNCH((ce_row_vec_kernel<float, 16, false>), dim3(rows), dim3(256), stream, score, rows, cols, ld, row_ws, (float*)dscore, ld_d);
        } else {
            DPC_LAUNCH((ce_row_kernel<float>), dim3(rows), dim3(256), stream, score, rows, cols, ld, row_ws, (float*)dscore, ld_d);
        }
    } else if (dtype_d == DPC_BF16) {
        if (vec && cols <= 8192) {
            DPC_LAUNCH((ce_row_vec_kernel<bf16_t, 8, true>), dim3(rows), dim3(256), stream, score, rows, cols, ld, row_ws, (bf16_t*)dscore, ld_d);
        } else if (vec) {
            DPC_LAUNCH((ce_row_vec_kernel<bf16_t, 16, true>), dim3(rows), dim3(256), stream, score, rows, cols, ld, row_ws, (bf16_t*)dscore, ld_d);
        } else {
            DPC_LAUNCH((ce_row_kernel<bf16_t>), dim3(rows), dim3(256), stream, score, rows, cols, ld, row_ws, (bf16_t*)dscore, ld_d);
        }
    } else {
        return DPC_ERR_ARG;
    }
    DPC_LAUNCH(ce_finalize_kernel, dim3(1), dim3(256), stream, row_ws, rows, result);
    return dpc_launch_status();
}

// dpc_ce_topk on bf16 logits with a bf16 gradient (the train step's materialised score in the compute dtype): rows of 16-byte units,
// cols a multiple of 8, at most 16 384 columns -- anything else is DPC_ERR_UNSUPPORTED (the caller keeps f32 logits then).
extern "C" int dpc_ce_topk_bf16(const void* score, int32_t rows, int32_t cols, int32_t ld, float* row_ws, float* result,
                                void* dscore, int32_t ld_d, dpc_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!score || !row_ws || !result || rows <= 0 || cols < rows || ld < cols) return DPC_ERR_ARG;
    if (dscore && ld_d < cols) return DPC_ERR_ARG;
    if (cols % 8 || ld % 8 || ((uintptr_t)score % 16) || cols > 16384 || (dscore && (ld_d % 8 || ((uintptr_t)dscore % 16)))) return DPC_ERR_UNSUPPORTED;
    if (cols <= 2048 * 4) {
        DPC_LAUNCH((ce_row_vec16_kernel<4>), dim3(rows), dim3(256), stream, (const bf16_t*)score, rows, cols, ld, row_ws, (bf16_t*)dscore, ld_d);
    } else {
        DPC_LAUNCH((ce_row_vec16_kernel<8>), dim3(rows), dim3(256), stream, (const bf16_t*)score, rows, cols, ld, row_ws, (bf16_t*)dscore, ld_d);
    }
    DPC_LAUNCH(ce_finalize_kernel, dim3(1), dim3(256), stream, row_ws, rows, result);
    return dpc_launch_status();
}

// mean loss + top-1/3/5 from per-row (loss term, rank) pairs: the last stage of dpc_ce_topk, exposed for the fused score
// path (csrc/score_fused.hip), whose forward produces the pairs without a score matrix
extern "C" int dpc_ce_finalize(const float* row_ws, int32_t rows, float* result, dpc_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!row_ws || rows <= 0 || !result) return DPC_ERR_ARG;
    DPC_LAUNCH(ce_finalize_kernel, dim3(1), dim3(256), stream, row_ws, rows, result);
    return dpc_launch_status();
}

// Adam, weight decay folded into the gradient (torch.optim.Adam semantics), 16-byte accesses
__global__ void adam_kernel(float* p, const float* g, float* m, float* v, long long n4, long long n, float lr, float b1,
                            float b2, float eps, float wd, float bc1, float bc2, float gscale) {
    const float step = lr / bc1;
    const float isb2 = 1.f / sqrtf(bc2);
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
        const long long base = i * 4;
        if (base + 4 <= n) {
            f32x4 pv = ((f32x4*)p)[i], gv = ((const f32x4*)g)[i], mv = ((f32x4*)m)[i], vv = ((f32x4*)v)[i];
            DPC_UNROLL
            for (int e = 0; e < 4; ++e) {
                const float gg = gv[e] * gscale + wd * pv[e];
                mv[e] = b1 * mv[e] + (1.f - b1) * gg;
                vv[e] = b2 * vv[e] + (1.f - b2) * gg * gg;
                pv[e] -= step * mv[e] / (sqrtf(vv[e]) * isb2 + eps);
            }
            ((f32x4*)p)[i] = pv; ((f32x4*)m)[i] = mv; ((f32x4*)v)[i] = vv;
        } else {
            for (long long k = base; k < n; ++k) {
                const float gg = g[k] * gscale + wd * p[k];
                m[k] = b1 * m[k] + (1.f - b1) * gg;
                v[k] = b2 * v[k] + (1.f - b2) * gg * gg;
                p[k] -= step * m[k] / (sqrtf(v[k]) * isb2 + eps);
            }
        }
    }
}

// ---- graph-capturable optimizer step: the step counter and Adam's bias corrections live in device memory, so a captured
// hipGraph of the whole train step advances them on every replay (host scalars would be frozen at capture time)
__global__ void step_advance_kernel(int32_t* step, float* bc, double b1, double b2) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        const int t = step[0] + 1;
        step[0] = t;
        bc[0] = (float)(1.0 - pow(b1, (double)t));
        bc[1] = (float)(1.0 - pow(b2, (double)t));
    }
}

extern "C" int dpc_step_advance(int32_t* step_dev, float* bias_corr_dev, double beta1, double beta2, dpc_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!step_dev || !bias_corr_dev) return DPC_ERR_ARG;
    DPC_LAUNCH(step_advance_kernel, dim3(1), dim3(64), stream, step_dev, bias_corr_dev, beta1, beta2);
    return dpc_launch_status();
}

// draw counter of the dropout streams: advanced by every train-mode forward (not by the optimizer step), so a caller that never
// runs the engine's own Adam -- the nn.Module boundary with a torch optimizer -- still sees fresh masks each forward
__global__ void counter_advance_kernel(int32_t* ctr) {
    if (threadIdx.x == 0 && blockIdx.x == 0) ctr[0] += 1;
}

extern "C" int dpc_counter_advance(int32_t* counter_dev, dpc_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!counter_dev) return DPC_ERR_ARG;
    DPC_LAUNCH(counter_advance_kernel, dim3(1), dim3(64), stream, counter_dev);
    return dpc_launch_status();
}

__global__ void adam_dev_kernel(float* p, const float* g, float* m, float* v, long long n4, long long n, float lr, float b1,
                                float b2, float omb1, float omb2, float eps, float wd, const float* bc, float gscale) {
    const float step = lr / bc[0];
    const float isb2 = 1.f / sqrtf(bc[1]);
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
        const long long base = i * 4;
        if (base + 4 <= n) {
            f32x4 pv = ((f32x4*)p)[i], gv = ((const f32x4*)g)[i], mv = ((f32x4*)m)[i], vv = ((f32x4*)v)[i];
            DPC_UNROLL
            for (int e = 0; e < 4; ++e) {
                const float gg = gv[e] * gscale + wd * pv[e];
                mv[e] = b1 * mv[e] + omb1 * gg;
                vv[e] = b2 * vv[e] + omb2 * gg * gg;
                pv[e] -= step * mv[e] / (sqrtf(vv[e]) * isb2 + eps);
            }
            ((f32x4*)p)[i] = pv; ((f32x4*)m)[i] = mv; ((f32x4*)v)[i] = vv;
        } else {
            for (long long k = base; k < n; ++k) {
                const float gg = g[k] * gscale + wd * p[k];
                m[k] = b1 * m[k] + omb1 * gg;
                v[k] = b2 * v[k] + omb2 * gg * gg;
                p[k] -= step * m[k] / (sqrtf(v[k]) * isb2 + eps);
            }
        }
    }
}

// betas arrive as doubles: 1 - beta2 is formed in double like torch does (1 - 0.999f would be off by 1.3e-5 relative)
extern "C" int dpc_adam_dev(float* p, const float* g, float* m, float* v, int64_t n, float lr, double beta1, double beta2,
                            float eps, float wd, const float* bias_corr_dev, float grad_scale, dpc_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!p || !g || !m || !v || n <= 0 || !bias_corr_dev) return DPC_ERR_ARG;
    const long long n4 = (n + 3) / 4;
    DPC_LAUNCH(adam_dev_kernel, dim3(grid_for(n4)), dim3(256), stream, p, g, m, v, n4, (long long)n, lr, (float)beta1, (float)beta2, (float)(1.0 - beta1), (float)(1.0 - beta2), eps, wd, bias_corr_dev, grad_scale);
    return dpc_launch_status();
}

extern "C" int dpc_adam(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2,
                        float eps, float wd, float bias_corr1, float bias_corr2, float grad_scale, dpc_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!p || !g || !m || !v || n <= 0 || bias_corr1 <= 0.f || bias_corr2 <= 0.f) return DPC_ERR_ARG;
    const long long n4 = (n + 3) / 4;
    DPC_LAUNCH(adam_kernel, dim3(grid_for(n4)), dim3(256), stream, p, g, m, v, n4, (long long)n, lr, beta1, beta2, eps, wd, bias_corr1, bias_corr2, grad_scale);
    return dpc_launch_status();
}
