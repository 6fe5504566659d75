// pool.hip -- the stem tail BN+ReLU+MaxPool3d((1,3,3),s(1,2,2),p(0,1,1))
// (backbone/resnet_2d3d.py:212-214,260-263) and the temporal mean / ReLU split that feeds
// the ConvGRU and the score (dpc/model_3d.py:53-59).  Channels-last, 16-byte units, HBM-bound.
#include "dpc_rt.h"
#include "../../include/dpc_hip.h"

static inline unsigned grid_for(long long n, int block = 256, int cap = 16384) {
    long long g = (n + block - 1) / block;
    if (g < 1) g = 1;
    if (g > cap) g = cap;
    return (unsigned)g;
}

// y[nt][oh][ow][c] = max_{3x3} relu(x*scale+shift); argmax = kh*3+kw of the FIRST maximum in
// scan order (torch max_pool semantics), or 9 when the maximum is 0 (ReLU kills the gradient).
template <class T>
__global__ void bn_relu_maxpool_fwd_kernel(const T* x, int NT, int H, int W, int C, int Ho, int Wo,
                                           const float* scale, const float* shift, T* y, uint8_t* argmax, int rev) {
    constexpr int E = Elt<T>::PER16;
    const int upr = C / E;
    const long long units = (long long)NT * Ho * Wo * upr;
    // Vertically adjacent pooled rows share an input row (2*oh+1 == 2*(oh+1)-1) and, at 32 pooled positions per 256-thread chunk,
    // sit in neighbouring chunks.  Workgroup b runs on XCD b & 7: with chunk = workgroup index the two readers of every odd input
    // row had different L2s (the second read came over the fabric from the Infinity Cache: 4.2 TB/s where the other kernels of
    // the family run 5.5).  Here every XCD owns a contiguous eighth of each sweep of the grid.  (A contiguous span per WORKGROUP
    // -- 2 048 independent streams -- measured slower: 1 132 us against 878.)
    unsigned blk = blockIdx.x;
    if ((gridDim.x & 7u) == 0u) blk = (blk & 7u) * (gridDim.x >> 3) + (blk >> 3);
    // rev (round 6): the sweeps of the grid run from the END of the tensor to its start -- the stem convolution wrote x front to back
    // (its tail is what the Infinity Cache still holds) and layer1's first convolution reads y front to back next
    const long long stride = (long long)gridDim.x * blockDim.x, first = (long long)blk * blockDim.x + threadIdx.x;
    const long long nsweep = first < units ? (units - 1 - first) / stride + 1 : 0;
    for (long long k = 0; k < nsweep; ++k) {
        const long long i = first + (rev ? nsweep - 1 - k : k) * stride;
        const unsigned pidx = (unsigned)(i / upr);  // pooled position < 2^31: 32-bit index math from here on
        const int cu = (int)(i - (long long)pidx * upr);
        const unsigned q1 = pidx / (unsigned)Wo;
        const int ow = (int)(pidx - q1 * (unsigned)Wo);
        const unsigned nt_ = q1 / (unsigned)Ho;
        const int oh = (int)(q1 - nt_ * (unsigned)Ho);
        const int nt = (int)nt_;
        float sc[E], sh[E], best[E];
        int bi[E];
        DPC_UNROLL
        for (int e = 0; e < E; ++e) {
            sc[e] = scale[cu * E + e];
            sh[e] = shift[cu * E + e];
            best[e] = -1.f;
            bi[e] = 9;
        }
        // all nine window loads are issued before the first use: with a branch around a border tap hipcc waits vmcnt(0) right
        // behind every load (nine serialised round trips per thread: 880 us = 4.2 TB/s where the family's other kernels run 5+).
        // Border taps re-read the window's centre (always inside) and are kept out of the comparison instead.
        u32x4 v[9];
        bool ok[9];
        DPC_UNROLL
        for (int kh = 0; kh < 3; ++kh) {
            const int h = 2 * oh - 1 + kh;
            const bool okh = (unsigned)h < (unsigned)H;
            DPC_UNROLL
            for (int kw = 0; kw < 3; ++kw) {
                const int w = 2 * ow - 1 + kw;
                const bool o = okh && (unsigned)w < (unsigned)W;
                const int hc = o ? h : 2 * oh, wc = o ? w : 2 * ow;
                ok[kh * 3 + kw] = o;
                v[kh * 3 + kw] = ((const u32x4*)x)[(long long)((unsigned)((nt * H + hc) * W + wc)) * upr + cu];
            }
        }
        sched_fence();   // keep the nine loads together (hipcc otherwise sinks each next to its use: load, wait, use, load, ...)
        DPC_UNROLL
        for (int k = 0; k < 9; ++k) {
            DPC_UNROLL
            for (int e = 0; e < E; ++e) {
                float a = unit_get<T>(v[k], e) * sc[e] + sh[e];
                a = a > 0.f ? a : 0.f;
                const bool upd = ok[k] & (a > best[e]);   // selects, not branches
                best[e] = upd ? a : best[e];
                bi[e] = upd ? k : bi[e];
            }
        }
        uint32_t amw[2] = {0u, 0u};
        DPC_UNROLL
        for (int e = 0; e < E; ++e) amw[e >> 2] |= (uint32_t)(best[e] > 0.f ? bi[e] : 9) << (8 * (e & 3));
        if (E == 8) {
            u32x2 t = {amw[0], amw[1]};
            *(u32x2*)(argmax + i * E) = t;
        } else {
            *(uint32_t*)(argmax + i * E) = amw[0];
        }
        ((u32x4*)y)[i] = unit_pack<T>(best);
    }
}

extern "C" int dpc_bn_relu_maxpool_fwd(const void* x, int32_t dtype, int32_t NT, int32_t H, int32_t W, int32_t C,
                                       const float* scale, const float* shift, void* y, uint8_t* argmax,
                                       dpc_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!x || !y || !argmax || !scale || !shift || NT <= 0 || H <= 0 || W <= 0 || C <= 0) return DPC_ERR_ARG;
    const int E = dtype == DPC_BF16 ? 8 : 4;
    if (C % E) return DPC_ERR_UNSUPPORTED;
    static const int rev = getenv("DPC_POOL_FWD_REV") ? atoi(getenv("DPC_POOL_FWD_REV")) : 0;
    const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
    const long long units = (long long)NT * Ho * Wo * (C / E);
    if (dtype == DPC_F32) {
        DPC_LAUNCH((bn_relu_maxpool_fwd_kernel<float>), dim3(grid_for(units)), dim3(256), stream, (const float*)x, NT, H, W, C, Ho, Wo, scale, shift, (float*)y, argmax, rev);
    } else if (dtype == DPC_BF16) {
        DPC_LAUNCH((bn_relu_maxpool_fwd_kernel<bf16_t>), dim3(grid_for(units)), dim3(256), stream, (const bf16_t*)x, NT, H, W, C, Ho, Wo, scale, shift, (bf16_t*)y, argmax, rev);
    } else {
        return DPC_ERR_ARG;
    }
    return dpc_launch_status();
}

// gradient routed by the pooling to full-resolution position (nt,h,w): sum over the (<=4) windows that
// contain it and whose saved argmax is this position (ReLU mask already folded into the argmax byte)
template <class T>
__device__ __forceinline__ void pool_routed_grad(const T* dy, const uint8_t* argmax, int nt, int h, int w, int cu, int Ho, int Wo,
                                                 int upr, float (&acc)[Elt<T>::PER16]) {
    constexpr int E = Elt<T>::PER16;
    DPC_UNROLL
    for (int e = 0; e < E; ++e) acc[e] = 0.f;
    const int oh0 = h >> 1, ow0 = w >> 1;  // window with kh = (h odd ? 2 : 1); h odd also hits oh0+1 with kh = 0
    // branch-free: the (<= 4) windows are all loaded (clamped to an existing one) before the first use and the ones that do not
    // exist are kept out of the sum -- a branch around a load is a vmcnt(0) behind it
    u32x4 g[4];
    uint32_t am[4][2];
    int want[4];
    bool ok[4];
    DPC_UNROLL
    for (int a = 0; a < 2; ++a) {
        DPC_UNROLL
        for (int b = 0; b < 2; ++b) {
            const int oh = oh0 + a, ow = ow0 + b;
            const bool o = (a == 0 || (h & 1)) && (b == 0 || (w & 1)) && oh < Ho && ow < Wo;
            const int ohc = o ? oh : oh0, owc = o ? ow : ow0;
            const long long ui = (long long)((unsigned)((nt * Ho + ohc) * Wo + owc)) * upr + cu;
            ok[a * 2 + b] = o;
            want[a * 2 + b] = (h - (2 * oh - 1)) * 3 + (w - (2 * ow - 1));
            g[a * 2 + b] = ((const u32x4*)dy)[ui];
            if (E == 8) {
                const u32x2 t = *(const u32x2*)(argmax + ui * E);
                am[a * 2 + b][0] = t[0]; am[a * 2 + b][1] = t[1];
            } else {
                am[a * 2 + b][0] = *(const uint32_t*)(argmax + ui * E); am[a * 2 + b][1] = 0u;
            }
        }
    }
    sched_fence();
    DPC_UNROLL
    for (int k = 0; k < 4; ++k)
        DPC_UNROLL
        for (int e = 0; e < E; ++e)
            acc[e] += (ok[k] & ((int)((am[k][e >> 2] >> (8 * (e & 3))) & 0xffu) == want[k])) ? unit_get<T>(g[k], e) : 0.f;
}

template <class T>
__global__ void maxpool_bwd_kernel(const T* dy, const uint8_t* argmax, int NT, int H, int W, int C, int Ho, int Wo, T* dz) {
    constexpr int E = Elt<T>::PER16;
    const int upr = C / E;
    const long long units = (long long)NT * H * W * upr;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < units; i += (long long)gridDim.x * blockDim.x) {
        const int cu = (int)(i % upr);
        long long q = i / upr;
        const int w = (int)(q % W);
        q /= W;
        const int h = (int)(q % H);
        const int nt = (int)(q / H);
        float acc[E];
        pool_routed_grad<T>(dy, argmax, nt, h, w, cu, Ho, Wo, upr, acc);
        ((u32x4*)dz)[i] = unit_pack<T>(acc);
    }
}

// ---- stem backward without materialising dz: BN backward whose incoming gradient is routed through the
// pooling on the fly (reads dpool + argmax + raw instead of a 2.7 GB dz tensor, twice)
template <class T>
__global__ __launch_bounds__(256) void pool_bn_bwd_reduce_kernel(const T* dy, const uint8_t* argmax, const T* x, int NT, int H, int W,
                                                                 int C, int Ho, int Wo, const float* mean, const float* invstd,
                                                                 float* partials, long long rows_per_block) {
    constexpr int E = Elt<T>::PER16;
    __shared__ float red[2][256 * E];
    const int upr = C / E, rpi = 256 / upr;
    const int tid = threadIdx.x, cu = tid % upr, rr = tid / upr;
    const long long rows = (long long)NT * H * W;
    float a1[E], a2[E], mu[E], is[E];
    DPC_UNROLL
    for (int e = 0; e < E; ++e) { a1[e] = 0.f; a2[e] = 0.f; }
    if (rr < rpi) {
        DPC_UNROLL
        for (int e = 0; e < E; ++e) { mu[e] = mean[cu * E + e]; is[e] = invstd[cu * E + e]; }
        const long long r_begin = (long long)blockIdx.x * rows_per_block;
        long long r_end = r_begin + rows_per_block;
        if (r_end > rows) r_end = rows;
        for (unsigned r = (unsigned)(r_begin + rr); r < (unsigned)r_end; r += (unsigned)rpi) {  // rows < 2^31: 32-bit index math
            const unsigned q = r / (unsigned)W;
            const int w = (int)(r - q * (unsigned)W);
            const unsigned nt_ = q / (unsigned)H;
            const int h = (int)(q - nt_ * (unsigned)H), nt = (int)nt_;
            float dz[E];
            pool_routed_grad<T>(dy, argmax, nt, h, w, cu, Ho, Wo, upr, dz);
            const u32x4 xv = ((const u32x4*)x)[(long long)r * upr + cu];
            DPC_UNROLL
            for (int e = 0; e < E; ++e) {
                a1[e] += dz[e];
                a2[e] += dz[e] * ((unit_get<T>(xv, e) - mu[e]) * is[e]);
            }
        }
    }
    DPC_UNROLL
    for (int e = 0; e < E; ++e) { red[0][tid * E + e] = a1[e]; red[1][tid * E + e] = a2[e]; }
    __syncthreads();
    if (tid < C) {
        const int cu2 = tid / E, e2 = tid % E;
        float s1 = 0.f, s2 = 0.f;
        for (int g = 0; g < rpi; ++g) {
            s1 += red[0][(g * upr + cu2) * E + e2];
            s2 += red[1][(g * upr + cu2) * E + e2];
        }
        partials[((long long)blockIdx.x * 2 + 0) * C + tid] = s1;
        partials[((long long)blockIdx.x * 2 + 1) * C + tid] = s2;
    }
}

// One thread = one 2x2 quad of full-resolution positions x E channels.  The quad (2a..2a+1, 2b..2b+1)
// is covered by exactly the four pooling windows (a..a+1, b..b+1), so their gradient/argmax units are
// loaded once for four outputs (the per-position form re-read them 9 times per quad).
template <class T>
__global__ __launch_bounds__(256) void pool_bn_bwd_apply_kernel(const T* dy, const uint8_t* argmax, const T* x, int NT, int H, int W, int C,
                                                                int Ho, int Wo, const float* mean, const float* invstd, const float* gamma,
                                                                const float* coef, T* dx, long long quads_per_block) {
    constexpr int E = Elt<T>::PER16;
    const int upr = C / E, rpi = 256 / upr;
    const int tid = threadIdx.x, cu = tid % upr, rr = tid / upr;
    if (rr >= rpi) return;
    const int Qh = (H + 1) / 2, Qw = (W + 1) / 2;
    const long long quads = (long long)NT * Qh * Qw;
    float mu[E], is[E], ga[E], c1[E], c2[E];
    DPC_UNROLL
    for (int e = 0; e < E; ++e) {
        const int c = cu * E + e;
        mu[e] = mean[c]; is[e] = invstd[c]; ga[e] = gamma[c] * is[e]; c1[e] = coef[c]; c2[e] = coef[C + c];
    }
    const long long q_begin = (long long)blockIdx.x * quads_per_block;
    long long q_end = q_begin + quads_per_block;
    if (q_end > quads) q_end = quads;
    for (unsigned q = (unsigned)(q_begin + rr); q < (unsigned)q_end; q += (unsigned)rpi) {
        const unsigned q1 = q / (unsigned)Qw;
        const int b = (int)(q - q1 * (unsigned)Qw);
        const unsigned nt_ = q1 / (unsigned)Qh;
        const int a = (int)(q1 - nt_ * (unsigned)Qh), nt = (int)nt_;
        // the four windows (a+i, b+j): gradient unit + argmax bytes
        float g[2][2][E];
        uint32_t am[2][2][2];
        DPC_UNROLL
        for (int i = 0; i < 2; ++i)
            DPC_UNROLL
            for (int j = 0; j < 2; ++j) {
                const bool ok = a + i < Ho && b + j < Wo;
                const long long ui = (long long)((unsigned)((nt * Ho + a + i) * Wo + b + j)) * upr + cu;
                u32x4 gv = {0u, 0u, 0u, 0u};
                am[i][j][0] = 0x09090909u; am[i][j][1] = 0x09090909u;
                if (ok) {
                    gv = ((const u32x4*)dy)[ui];
                    if (E == 8) {
                        const u32x2 t = *(const u32x2*)(argmax + ui * E);
                        am[i][j][0] = t[0]; am[i][j][1] = t[1];
                    } else {
                        am[i][j][0] = *(const uint32_t*)(argmax + ui * E);
                    }
                }
                DPC_UNROLL
                for (int e = 0; e < E; ++e) g[i][j][e] = unit_get<T>(gv, e);
            }
        // position (2a+di, 2b+dj) is tap (kh,kw) of window (a+i, b+j) with kh = di + 1 - 2i, kw = dj + 1 - 2j
        DPC_UNROLL
        for (int di = 0; di < 2; ++di)
            DPC_UNROLL
            for (int dj = 0; dj < 2; ++dj) {
                const int h = 2 * a + di, w = 2 * b + dj;
                if (h >= H || w >= W) continue;
                float dz[E];
                DPC_UNROLL
                for (int e = 0; e < E; ++e) dz[e] = 0.f;
                DPC_UNROLL
                for (int i = 0; i < 2; ++i)
                    DPC_UNROLL
                    for (int j = 0; j < 2; ++j) {
                        const int kh = di + 1 - 2 * i, kw = dj + 1 - 2 * j;
                        if (kh < 0 || kh > 2 || kw < 0 || kw > 2) continue;  // compile-time after unrolling
                        const int want = kh * 3 + kw;
                        DPC_UNROLL
                        for (int e = 0; e < E; ++e)
                            if ((int)((am[i][j][e >> 2] >> (8 * (e & 3))) & 0xffu) == want) dz[e] += g[i][j][e];
                    }
                const long long ui = (long long)((unsigned)((nt * H + h) * W + w)) * upr + cu;
                const u32x4 xv = ((const u32x4*)x)[ui];
                float ov[E];
                DPC_UNROLL
                for (int e = 0; e < E; ++e) {
                    const float xh = (unit_get<T>(xv, e) - mu[e]) * is[e];
                    ov[e] = ga[e] * (dz[e] - c1[e] - xh * c2[e]);
                }
                ((u32x4*)dx)[ui] = unit_pack<T>(ov);
            }
    }
}

// Partial sums for the stem BN backward WITHOUT touching the full-resolution tensor: the gradient
// reaches the BN output only at pooling argmax positions, and there y = relu(gamma*xhat+beta) is the
// pooled value itself, so  sum dz = sum_p dp[p]  and  sum dz*xhat = sum_p dp[p]*(ypool[p]-beta)/gamma
// over pooled positions p whose argmax byte is valid (< 9).  Reads 1.7 GB instead of 3.7 GB at
// cfg2.  (gamma == 0 exactly would make xhat unrecoverable; the term is then dropped -- it only
// enters d(gamma) of a channel whose output is constant.)
template <class T>
__global__ __launch_bounds__(256) void pooled_bn_bwd_reduce_kernel(const T* dy, const uint8_t* argmax, const T* ypool, long long rows,
                                                                   int C, const float* gamma, const float* beta, float* partials,
                                                                   long long rows_per_block) {
    constexpr int E = Elt<T>::PER16;
    __shared__ float red[2][256 * E];
    const int upr = C / E, rpi = 256 / upr;
    const int tid = threadIdx.x, cu = tid % upr, rr = tid / upr;
    float a1[E], a2[E], be[E], ig[E];
    DPC_UNROLL
    for (int e = 0; e < E; ++e) { a1[e] = 0.f; a2[e] = 0.f; }
    if (rr < rpi) {
        DPC_UNROLL
        for (int e = 0; e < E; ++e) {
            const float gmm = gamma[cu * E + e];
            be[e] = beta[cu * E + e];
            ig[e] = gmm != 0.f ? 1.f / gmm : 0.f;
        }
        const long long r_begin = (long long)blockIdx.x * rows_per_block;
        long long r_end = r_begin + rows_per_block;
        if (r_end > rows) r_end = rows;
        for (long long r = r_begin + rr; r < r_end; r += rpi) {
            const long long ui = r * upr + cu;
            const u32x4 gv = ((const u32x4*)dy)[ui];
            const u32x4 yv = ((const u32x4*)ypool)[ui];
            uint32_t amw[2] = {0u, 0u};
            if (E == 8) {
                const u32x2 t = *(const u32x2*)(argmax + ui * E);
                amw[0] = t[0]; amw[1] = t[1];
            } else {
                amw[0] = *(const uint32_t*)(argmax + ui * E);
            }
            DPC_UNROLL
            for (int e = 0; e < E; ++e) {
                const bool valid = ((amw[e >> 2] >> (8 * (e & 3))) & 0xffu) < 9u;
                const float dz = valid ? unit_get<T>(gv, e) : 0.f;
                a1[e] += dz;
                a2[e] += dz * (unit_get<T>(yv, e) - be[e]) * ig[e];
            }
        }
    }
    DPC_UNROLL
    for (int e = 0; e < E; ++e) { red[0][tid * E + e] = a1[e]; red[1][tid * E + e] = a2[e]; }
    __syncthreads();
    if (tid < C) {
        const int cu2 = tid / E, e2 = tid % E;
        float s1 = 0.f, s2 = 0.f;
        for (int g = 0; g < rpi; ++g) {
            s1 += red[0][(g * upr + cu2) * E + e2];
            s2 += red[1][(g * upr + cu2) * E + e2];
        }
        partials[((long long)blockIdx.x * 2 + 0) * C + tid] = s1;
        partials[((long long)blockIdx.x * 2 + 1) * C + tid] = s2;
    }
}

static int pool_bwd_blocks(long long rows, int C, int E, long long* rows_per_block) {
    const int rpi = 256 / (C / E);
    long long per = (long long)rpi * 16;
    long long blocks = (rows + per - 1) / per;
    if (blocks > 1024) {
        blocks = 1024;
        per = (rows + blocks - 1) / blocks;
        per = (per + rpi - 1) / rpi * rpi;
        blocks = (rows + per - 1) / per;
    }
    *rows_per_block = per;
    return (int)blocks;
}

extern "C" int dpc_pool_bn_bwd_reduce(const void* dy, const uint8_t* argmax, const void* x, int32_t dtype, int32_t NT, int32_t H,
                                      int32_t W, int32_t C, const float* mean, const float* invstd, float* partials, int32_t* prow,
                                      dpc_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (NT <= 0 || H <= 0 || W <= 0 || C <= 0) return DPC_ERR_ARG;
    const int E = dtype == DPC_BF16 ? 8 : 4;
    if (C % E || C > 256) return DPC_ERR_UNSUPPORTED;
    long long rpb;
    const int blocks = pool_bwd_blocks((long long)NT * H * W, C, E, &rpb);
    if (prow) *prow = blocks;
    if (!partials) return DPC_OK;
    if (!dy || !argmax || !x || !mean || !invstd) return DPC_ERR_ARG;
    const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
    if (dtype == DPC_F32) {
        DPC_LAUNCH((pool_bn_bwd_reduce_kernel<float>), dim3(blocks), dim3(256), stream, (const float*)dy, argmax, (const float*)x, NT, H, W, C, Ho, Wo, mean, invstd, partials, rpb);
    } else if (dtype == DPC_BF16) {
        DPC_LAUNCH((pool_bn_bwd_reduce_kernel<bf16_t>), dim3(blocks), dim3(256), stream, (const bf16_t*)dy, argmax, (const bf16_t*)x, NT, H, W, C, Ho, Wo, mean, invstd, partials, rpb);
    } else {
        return DPC_ERR_ARG;
    }
    return dpc_launch_status();
}

extern "C" int dpc_pooled_bn_bwd_reduce(const void* dy, const uint8_t* argmax, const void* ypool, int32_t dtype, int64_t rows,
                                        int32_t C, const float* gamma, const float* beta, float* partials, int32_t* prow,
                                        dpc_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (rows <= 0 || C <= 0) return DPC_ERR_ARG;
    const int E = dtype == DPC_BF16 ? 8 : 4;
    if (C % E || C > 256) return DPC_ERR_UNSUPPORTED;
    long long rpb;
    const int blocks = pool_bwd_blocks(rows, C, E, &rpb);
    if (prow) *prow = blocks;
    if (!partials) return DPC_OK;
    if (!dy || !argmax || !ypool || !gamma || !beta) return DPC_ERR_ARG;
    if (dtype == DPC_F32) {
        DPC_LAUNCH((pooled_bn_bwd_reduce_kernel<float>), dim3(blocks), dim3(256), stream, (const float*)dy, argmax, (const float*)ypool, (long long)rows, C, gamma, beta, partials, rpb);
    } else if (dtype == DPC_BF16) {
        DPC_LAUNCH((pooled_bn_bwd_reduce_kernel<bf16_t>), dim3(blocks), dim3(256), stream, (const bf16_t*)dy, argmax, (const bf16_t*)ypool, (long long)rows, C, gamma, beta, partials, rpb);
    } else {
        return DPC_ERR_ARG;
    }
    return dpc_launch_status();
}

extern "C" int dpc_pool_bn_bwd_apply(const void* dy, const uint8_t* argmax, const void* x, int32_t dtype, int32_t NT, int32_t H,
                                     int32_t W, int32_t C, const float* mean, const float* invstd, const float* gamma,
                                     const float* coef, void* dx, dpc_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!dy || !argmax || !x || !dx || !mean || !invstd || !gamma || !coef || NT <= 0 || H <= 0 || W <= 0 || C <= 0) return DPC_ERR_ARG;
    const int E = dtype == DPC_BF16 ? 8 : 4;
    if (C % E) return DPC_ERR_UNSUPPORTED;
    if (C > 256) return DPC_ERR_UNSUPPORTED;
    const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
    const int rpi = 256 / (C / E);
    const long long rows = (long long)NT * ((H + 1) / 2) * ((W + 1) / 2);  // 2x2 quads
    long long blocks = (rows + (long long)rpi * 4 - 1) / ((long long)rpi * 4);
    if (blocks > 16384) blocks = 16384;
    long long rpb = (rows + blocks - 1) / blocks;
    rpb = (rpb + rpi - 1) / rpi * rpi;
    blocks = (rows + rpb - 1) / rpb;
    if (dtype == DPC_F32) {
        DPC_LAUNCH((pool_bn_bwd_apply_kernel<float>), dim3((unsigned)blocks), dim3(256), stream, (const float*)dy, argmax, (const float*)x, NT, H, W, C, Ho, Wo, mean, invstd, gamma, coef, (float*)dx, rpb);
    } else if (dtype == DPC_BF16) {
        DPC_LAUNCH((pool_bn_bwd_apply_kernel<bf16_t>), dim3((unsigned)blocks), dim3(256), stream, (const bf16_t*)dy, argmax, (const bf16_t*)x, NT, H, W, C, Ho, Wo, mean, invstd, gamma, coef, (bf16_t*)dx, rpb);
    } else {
        return DPC_ERR_ARG;
    }
    return dpc_launch_status();
}

extern "C" int dpc_maxpool_bwd(const void* dy, const uint8_t* argmax, int32_t dtype, int32_t NT, int32_t H, int32_t W,
                               int32_t C, void* dz, dpc_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!dy || !argmax || !dz || NT <= 0 || H <= 0 || W <= 0 || C <= 0) return DPC_ERR_ARG;
    const int E = dtype == DPC_BF16 ? 8 : 4;
    if (C % E) return DPC_ERR_UNSUPPORTED;
    const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
    const long long units = (long long)NT * H * W * (C / E);
    if (dtype == DPC_F32) {
        DPC_LAUNCH((maxpool_bwd_kernel<float>), dim3(grid_for(units)), dim3(256), stream, (const float*)dy, argmax, NT, H, W, C, Ho, Wo, (float*)dz);
    } else if (dtype == DPC_BF16) {
        DPC_LAUNCH((maxpool_bwd_kernel<bf16_t>), dim3(grid_for(units)), dim3(256), stream, (const bf16_t*)dy, argmax, NT, H, W, C, Ho, Wo, (bf16_t*)dz);
    } else {
        return DPC_ERR_ARG;
    }
    return dpc_launch_status();
}

// ---- temporal mean over the T frames left by layer4, ReLU split (dpc/model_3d.py:53-59)
// x [B*N][T][SQ][D] -> feat_relu [N][B*SQ][D] ; feat_inf [B][P][SQ][D] (blocks n >= N-P, pre-ReLU)
template <class T>
__global__ void tpool_split_fwd_kernel(const T* x, int B, int N, int Tt, int SQ, int D, int P, T* feat_relu, T* feat_inf) {
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
            for (int e = 0; e < E; ++e) m[e] += unit_get<T>(v, e);
        }
        u32x4 orl, oin;
        DPC_UNROLL
        for (int e = 0; e < E; ++e) {
            const float v = m[e] * inv;
            unit_set<T>(oin, e, v);
            unit_set<T>(orl, e, v > 0.f ? v : 0.f);
        }
        ((u32x4*)feat_relu)[(((long long)n * B + b) * SQ + s) * upr + cu] = orl;
        if (n >= N - P) ((u32x4*)feat_inf)[(((long long)b * P + (n - (N - P))) * SQ + s) * upr + cu] = oin;
    }
}

extern "C" int dpc_tpool_split_fwd(const void* x, int32_t dtype, int32_t B, int32_t N, int32_t T, int32_t SQ, int32_t D,
                                   int32_t P, void* feat_relu, void* feat_inf, dpc_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!x || !feat_relu || !feat_inf || B <= 0 || N <= 0 || T <= 0 || SQ <= 0 || D <= 0 || P <= 0 || P > N) return DPC_ERR_ARG;
    const int E = dtype == DPC_BF16 ? 8 : 4;
    if (D % E) return DPC_ERR_UNSUPPORTED;
    const long long units = (long long)B * N * SQ * (D / E);
    if (dtype == DPC_F32) {
        DPC_LAUNCH((tpool_split_fwd_kernel<float>), dim3(grid_for(units)), dim3(256), stream, (const float*)x, B, N, T, SQ, D, P, (float*)feat_relu, (float*)feat_inf);
    } else if (dtype == DPC_BF16) {
        DPC_LAUNCH((tpool_split_fwd_kernel<bf16_t>), dim3(grid_for(units)), dim3(256), stream, (const bf16_t*)x, B, N, T, SQ, D, P, (bf16_t*)feat_relu, (bf16_t*)feat_inf);
    } else {
        return DPC_ERR_ARG;
    }
    return dpc_launch_status();
}

// dx[b*N+n][t][s][d] = (1/T) * ( d_relu[n][b*SQ+s][d]*(mean>0)  [n < N-P]  +  d_inf[b][n-(N-P)][s][d]  [n >= N-P] )
template <class T>
__global__ void tpool_split_bwd_kernel(const T* x, const float* d_relu, const float* d_inf, int B, int N, int Tt, int SQ,
                                       int D, int P, T* dx) {
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
        float g[E];
        if (n < N - P) {
            float m[E];
            DPC_UNROLL
            for (int e = 0; e < E; ++e) m[e] = 0.f;
            for (int t = 0; t < Tt; ++t) {
                const u32x4 v = ((const u32x4*)x)[((((long long)b * N + n) * Tt + t) * SQ + s) * upr + cu];
                DPC_UNROLL
                for (int e = 0; e < E; ++e) m[e] += unit_get<T>(v, e);
            }
            const float* gr = d_relu + (((long long)n * B + b) * SQ + s) * D + cu * E;
            DPC_UNROLL
            for (int e = 0; e < E; ++e) g[e] = (m[e] > 0.f) ? gr[e] * inv : 0.f;
        } else {
            const float* gi = d_inf + (((long long)b * P + (n - (N - P))) * SQ + s) * D + cu * E;
            DPC_UNROLL
            for (int e = 0; e < E; ++e) g[e] = gi[e] * inv;
        }
        u32x4 o;
        DPC_UNROLL
        for (int e = 0; e < E; ++e) unit_set<T>(o, e, g[e]);
        for (int t = 0; t < Tt; ++t) ((u32x4*)dx)[((((long long)b * N + n) * Tt + t) * SQ + s) * upr + cu] = o;
    }
}

extern "C" int dpc_tpool_split_bwd(const void* x, const void* d_relu, const float* d_inf, int32_t dtype, int32_t B,
                                   int32_t N, int32_t T, int32_t SQ, int32_t D, int32_t P, void* dx, dpc_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!x || !d_relu || !d_inf || !dx || B <= 0 || N <= 0 || T <= 0 || SQ <= 0 || D <= 0 || P <= 0 || P > N) return DPC_ERR_ARG;
    const int E = dtype == DPC_BF16 ? 8 : 4;
    if (D % E) return DPC_ERR_UNSUPPORTED;
    const long long units = (long long)B * N * SQ * (D / E);
    if (dtype == DPC_F32) {
        DPC_LAUNCH((tpool_split_bwd_kernel<float>), dim3(grid_for(units)), dim3(256), stream, (const float*)x, (const float*)d_relu, d_inf, B, N, T, SQ, D, P, (float*)dx);
    } else if (dtype == DPC_BF16) {
        DPC_LAUNCH((tpool_split_bwd_kernel<bf16_t>), dim3(grid_for(units)), dim3(256), stream, (const bf16_t*)x, (const float*)d_relu, d_inf, B, N, T, SQ, D, P, (bf16_t*)dx);
    } else {
        return DPC_ERR_ARG;
    }
    return dpc_launch_status();
}
