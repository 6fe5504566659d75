// bn.hip -- BatchNorm3d with batch statistics (track_running_stats=False, dpc/model_3d.py:28;
// nn.BatchNorm3d at backbone/resnet_2d3d.py:55,59,93,97,212,243) on channels-last tensors,
// fused with the residual add + ReLU of the BasicBlock tail (resnet_2d3d.py:67-80,105-116).
// All kernels are HBM-bound: every access is a 16-byte unit (4 f32 / 8 bf16 channels),
// statistics are accumulated in f32 per thread, f64 across partial rows (deterministic).
#include "dpc_rt.h"
#include "../../include/dpc_hip.h"
#include <stdlib.h>

static inline unsigned grid_for(long long n, int block = 256, int cap = 8192) {
    long long g = (n + block - 1) / block;
    if (g < 1) g = 1;
    if (g > cap) g = cap;
    return (unsigned)g;
}

// ------------------------------------------------------------------ forward finalize
// One workgroup = 32 channels, 256 threads = four waves, one per SIMD: thread (q = tid & 7, rl = tid >> 3) sums channels 4q..4q+3 of
// rows rl, rl + 32, ... with 16-byte loads (f64), the eight row lanes of a wave meet through lane shuffles, the four waves through
// 2 KB of LDS.  Small on purpose (rounds 1-5: 1 024 threads, 16 KB): these kernels run on the main stream while a weight gradient
// of the side stream holds every CU -- two wgrad_patch workgroups leave 112 VGPRs per SIMD lane and 66 KB of LDS, room for a
// 60-register wave per SIMD but not for four -- and the big workgroup waited for a whole CU to drain: 230 us on the critical path
// of layer2.0 in every step (profiles/r06_r18_128_graph_timeline.txt: bn_bwd_finalize at 17.235 ms; 0.44 ms per step in these
// kernels against 0.11 ms of work).  A serial per-channel loop over ~1-2k rows was latency-bound (0.4 ms per call, round 1).
// After the call threads 0..31 hold the sums of channel blockIdx.x * 32 + tid.  The summation order is fixed (deterministic).
constexpr int FIN_THREADS = 256;
__device__ __forceinline__ void partial_rows_sum(const float* partials, int rows, int C, double& s1, double& s2, double (*red)[4][32]) {
    const int tid = threadIdx.x, q = tid & 7, rl = tid >> 3;
    const int c0 = blockIdx.x * 32 + 4 * q;
    double a[4] = {0.0, 0.0, 0.0, 0.0}, b[4] = {0.0, 0.0, 0.0, 0.0};
    if ((C & 3) == 0 && ((uintptr_t)partials & 15) == 0) {
        if (c0 < C) {
            // four rows per trip, loaded before any is added: the loop is a chain of L2 round trips otherwise
            int r = rl;
            for (; r + 96 < rows; r += 128) {
                f32x4 x[4], y[4];
                DPC_UNROLL
                for (int k = 0; k < 4; ++k) {
                    x[k] = *(const f32x4*)(partials + ((long long)(r + 32 * k) * 2 + 0) * C + c0);
                    y[k] = *(const f32x4*)(partials + ((long long)(r + 32 * k) * 2 + 1) * C + c0);
                }
                DPC_UNROLL
                for (int e = 0; e < 4; ++e) {
                    a[e] += ((double)x[0][e] + (double)x[1][e]) + ((double)x[2][e] + (double)x[3][e]);
                    b[e] += ((double)y[0][e] + (double)y[1][e]) + ((double)y[2][e] + (double)y[3][e]);
                }
            }
            for (; r < rows; r += 32) {
                const f32x4 x = *(const f32x4*)(partials + ((long long)r * 2 + 0) * C + c0);
                const f32x4 y = *(const f32x4*)(partials + ((long long)r * 2 + 1) * C + c0);
                DPC_UNROLL
                for (int e = 0; e < 4; ++e) { a[e] += (double)x[e]; b[e] += (double)y[e]; }
            }
        }
    } else {   // channel counts that are no multiple of four (or an unaligned table): the same walk, element by element
        for (int r = rl; r < rows; r += 32)
            DPC_UNROLL
            for (int e = 0; e < 4; ++e)
                if (c0 + e < C) {
                    a[e] += (double)partials[((long long)r * 2 + 0) * C + c0 + e];
                    b[e] += (double)partials[((long long)r * 2 + 1) * C + c0 + e];
                }
    }
    // lanes q, q + 8, ..., q + 56 of a wave hold the same channels
    DPC_UNROLL
    for (int m = 8; m < 64; m <<= 1)
        DPC_UNROLL
        for (int e = 0; e < 4; ++e) { a[e] += __shfl_xor(a[e], m); b[e] += __shfl_xor(b[e], m); }
    const int wave = tid >> 6;
    if ((tid & 63) < 8)
        DPC_UNROLL
        for (int e = 0; e < 4; ++e) { red[0][wave][4 * q + e] = a[e]; red[1][wave][4 * q + e] = b[e]; }
    __syncthreads();
    s1 = 0.0; s2 = 0.0;
    if (tid < 32)
        for (int w = 0; w < 4; ++w) { s1 += red[0][w][tid]; s2 += red[1][w][tid]; }
}

__global__ __launch_bounds__(FIN_THREADS) void bn_finalize_kernel(const float* partials, int rows, int C, double count, const float* gamma,
                                   const float* beta, float eps, float* mean, float* invstd, float* scale, float* shift) {
    __shared__ double red[2][4][32];
    const int c = blockIdx.x * 32 + (int)threadIdx.x;
    double s1, s2;
    partial_rows_sum(partials, rows, C, s1, s2, red);
    if (threadIdx.x >= 32 || c >= C) return;
    const double m = s1 / count;
    double var = s2 / count - m * m;
    if (var < 0.0) var = 0.0;
    const double is = 1.0 / sqrt(var + (double)eps);
    mean[c] = (float)m;
    invstd[c] = (float)is;
    const float sc = gamma[c] * (float)is;
    scale[c] = sc;
    shift[c] = beta[c] - (float)m * sc;
}

extern "C" int dpc_bn_finalize(const float* partials, int32_t rows, int32_t C, double count, const float* gamma,
                               const float* beta, float eps, float* mean, float* invstd, float* scale, float* shift,
                               dpc_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!partials || rows <= 0 || C <= 0 || count <= 0 || !gamma || !beta || !mean || !invstd || !scale || !shift) return DPC_ERR_ARG;
    DPC_LAUNCH(bn_finalize_kernel, dim3((C + 31) / 32), dim3(FIN_THREADS), stream, partials, rows, C, count, gamma, beta, eps, mean, invstd, scale, shift);
    return dpc_launch_status();
}

// ---- BatchNorm3d with track_running_stats=True (the LC classifier's backbone, eval/model_3d_lc.py:27-29) --------------------
// train: batch statistics as above + running_mean/var <- (1-m) running + m batch (variance unbiased, torch semantics);
// eval: coefficients from the running buffers (dpc_bn_eval_coeffs), no batch statistics at all.
__global__ __launch_bounds__(FIN_THREADS) void bn_finalize_running_kernel(const float* partials, int rows, int C, double count, const float* gamma,
                                   const float* beta, float eps, float* mean, float* invstd, float* scale, float* shift,
                                   float* rmean, float* rvar, long long* nbt, float momentum) {
    __shared__ double red[2][4][32];
    const int c = blockIdx.x * 32 + (int)threadIdx.x;
    double s1, s2;
    partial_rows_sum(partials, rows, C, s1, s2, red);
    if (blockIdx.x == 0 && threadIdx.x == 0 && nbt) nbt[0] += 1;
    if (threadIdx.x >= 32 || c >= C) return;
    const double m = s1 / count;
    double var = s2 / count - m * m;
    if (var < 0.0) var = 0.0;
    const double is = 1.0 / sqrt(var + (double)eps);
    mean[c] = (float)m;
    invstd[c] = (float)is;
    const float sc = gamma[c] * (float)is;
    scale[c] = sc;
    shift[c] = beta[c] - (float)m * sc;
    const double unb = count > 1.0 ? var * count / (count - 1.0) : var;
    rmean[c] = (1.f - momentum) * rmean[c] + momentum * (float)m;
    rvar[c] = (1.f - momentum) * rvar[c] + momentum * (float)unb;
}

extern "C" int dpc_bn_finalize_running(const float* partials, int32_t rows, int32_t C, double count, const float* gamma, const float* beta,
                                       float eps, float* mean, float* invstd, float* scale, float* shift, float* running_mean,
                                       float* running_var, int64_t* num_batches_tracked, float momentum, dpc_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!partials || rows <= 0 || C <= 0 || count <= 0 || !gamma || !beta || !mean || !invstd || !scale || !shift || !running_mean || !running_var)
        return DPC_ERR_ARG;
    DPC_LAUNCH(bn_finalize_running_kernel, dim3((C + 31) / 32), dim3(FIN_THREADS), stream, partials, rows, C, count, gamma, beta, eps, mean, invstd, scale,
               shift, running_mean, running_var, (long long*)num_batches_tracked, momentum);
    return dpc_launch_status();
}

__global__ void bn_eval_coeffs_kernel(const float* gamma, const float* beta, const float* rmean, const float* rvar, float eps, int C, float* mean,
                                      float* invstd, float* scale, float* shift) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const float is = 1.f / sqrtf(rvar[c] + eps);
    mean[c] = rmean[c];
    invstd[c] = is;
    const float sc = gamma[c] * is;
    scale[c] = sc;
    shift[c] = beta[c] - rmean[c] * sc;
}

extern "C" int dpc_bn_eval_coeffs(const float* gamma, const float* beta, const float* running_mean, const float* running_var, float eps,
                                  int32_t C, float* mean, float* invstd, float* scale, float* shift, dpc_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!gamma || !beta || !running_mean || !running_var || C <= 0 || !mean || !invstd || !scale || !shift) return DPC_ERR_ARG;
    DPC_LAUNCH(bn_eval_coeffs_kernel, dim3((C + 255) / 256), dim3(256), stream, gamma, beta, running_mean, running_var, eps, C, mean, invstd, scale, shift);
    return dpc_launch_status();
}

// ------------------------------------------------------------------ forward apply (+res)(+relu)
// FIXED: 256*E is a multiple of C, so a thread's channel group never changes across the grid-stride
// loop and the per-channel coefficients live in registers (no 64-bit modulo, no per-element loads).
// Streaming accesses: these kernels touch every byte once and the tensors of the big layers (0.3-2.7 GB) do not
// fit any cache level, so above a size threshold loads and stores carry the non-temporal hint (measured on a
// 671 MB tensor: apply 4.85 -> 5.34 TB/s, apply + residual 4.33 -> 4.94).  Small tensors keep the default policy:
// the next kernel finds them in L2 / Infinity Cache.
#ifdef DPC_SIMT_EMU
template <bool NT> static inline u32x4 ld16(const void* p, long long i) { return ((const u32x4*)p)[i]; }
template <bool NT> static inline void st16(void* p, long long i, const u32x4& v) { ((u32x4*)p)[i] = v; }
#else
template <bool NT> __device__ __forceinline__ u32x4 ld16(const void* p, long long i) {
    const u32x4* q = (const u32x4*)p + i;
    if constexpr (NT) return __builtin_nontemporal_load(q);
    else return *q;
}
template <bool NT> __device__ __forceinline__ void st16(void* p, long long i, const u32x4& v) {
    u32x4* q = (u32x4*)p + i;
    if constexpr (NT) __builtin_nontemporal_store(v, q);
    else *q = v;
}
#endif
static inline bool bn_streaming(long long units) {
    const long long thr = getenv("DPC_BN_NT_MB") ? atoll(getenv("DPC_BN_NT_MB")) : 192;  // tensor size in MB; 0 = never (read per call: the test tiers lower it)
    return thr > 0 && units * 16 >= thr * (1ll << 20);
}
// units per thread in flight in the streaming bf16 instantiations of bn_apply / bn_bwd_apply (1: round 3's rolled loops) and the
// workgroup cap of the forward apply (per-call reads: A/B knobs, DPC_BN_UNROLL / DPC_BN_APPLY_GRID)
static inline int bn_unroll() { const char* e = getenv("DPC_BN_UNROLL"); const int v = e ? atoi(e) : 4; return v == 2 || v == 4 ? v : 1; }
static inline int bn_unroll_grid() { const char* e = getenv("DPC_BN_APPLY_GRID"); return e ? atoi(e) : 8192; }
static inline int bn_bwd_grid() { const char* e = getenv("DPC_BN_BWD_GRID"); return e ? atoi(e) : 8192; }
// tensors below the streaming threshold (layer3 / layer4 of cfg2: 100 / 17 MB): the same unrolled forms with the default cache policy
static inline int bn_small_unroll() { const char* e = getenv("DPC_BN_SMALL_UNROLL"); const int v = e ? atoi(e) : 4; return v == 2 || v == 4 ? v : 1; }
static inline int bn_small_grid(int dflt) { const char* e = getenv("DPC_BN_SMALL_GRID"); return e ? atoi(e) : dflt; }
// bit e = element e of the unit is > 0 (the ReLU pass-through mask)
template <class T> __device__ __forceinline__ unsigned sign_bits(const u32x4& v) {
    unsigned b = 0;
    DPC_UNROLL
    for (int e = 0; e < Elt<T>::PER16; ++e) b |= (unit_get<T>(v, e) > 0.f ? 1u : 0u) << e;
    return b;
}

template <class T, bool FIXED, bool NT, int U = 1>
__global__ void bn_apply_kernel(const T* x, T* y, long long units, int C, const float* scale, const float* shift,
                                const T* res, const float* rscale, const float* rshift, int relu, uint8_t* mask, int rev) {
    constexpr int E = Elt<T>::PER16;
    float sc[E], sh[E], rs[E], rb[E];
    if (FIXED) {
        const int c0 = (int)((threadIdx.x * E) % C);
        DPC_UNROLL
        for (int e = 0; e < E; ++e) {
            sc[e] = scale[c0 + e]; sh[e] = shift[c0 + e];
            rs[e] = rscale ? rscale[c0 + e] : 1.f; rb[e] = rscale ? rshift[c0 + e] : 0.f;
        }
    }
    // one contiguous span per workgroup (a multiple of 256 units, so a thread keeps its channel group): streams
    // 4 KB per step from one DRAM region instead of striding the whole tensor (+8 % bandwidth measured)
    // rev (round 6): the spans are handed out from the END of the tensor.  The convolution that produced x wrote it front to back, so
    // its tail is what the Infinity Cache (256 MB) / L2 still hold when this kernel starts; and y, written back to front, leaves its
    // HEAD in the cache for the convolution that reads it front to back next (bn_bwd_apply_kernel has walked its spans this way since round 3).
    const long long span = (units + (long long)gridDim.x * 256 - 1) / ((long long)gridDim.x * 256) * 256;
    const long long sb = (long long)(rev ? gridDim.x - 1 - blockIdx.x : blockIdx.x) * span;
    const long long se = sb + span < units ? sb + span : units;
    long long i = sb + threadIdx.x;
    // Round 4: U units per thread in flight.  The rolled loop has ONE 16-byte load (two with a residual) outstanding per thread --
    // 32 KB per CU at full occupancy, 8 MB on the chip: by Little's law about 4 TB/s of reads at the ~2 us a loaded HBM round trip
    // takes -- which is where the family sat (5.1-5.3 TB/s with its writes).  The channel group of a thread does not change over
    // i += 256, so the coefficients stay in registers (FIXED only).
    if constexpr (FIXED && U > 1) {
        for (; i + (U - 1) * 256 < se; i += 256 * U) {
            u32x4 xv[U], rv[U];
            DPC_UNROLL
            for (int u = 0; u < U; ++u) xv[u] = ld16<NT>(x, i + u * 256);
            if (res) {
                DPC_UNROLL
                for (int u = 0; u < U; ++u) rv[u] = ld16<NT>(res, i + u * 256);
            }
            DPC_UNROLL
            for (int u = 0; u < U; ++u) {
                float ov[E];
                unsigned bits = 0;
                DPC_UNROLL
                for (int e = 0; e < E; ++e) {
                    float v = unit_get<T>(xv[u], e) * sc[e] + sh[e];
                    if (res) v += unit_get<T>(rv[u], e) * rs[e] + rb[e];
                    if (relu) v = v > 0.f ? v : 0.f;
                    ov[e] = v;
                }
                const u32x4 o = unit_pack<T>(ov);
                st16<NT>(y, i + u * 256, o);
                if (mask) {
                    DPC_UNROLL
                    for (int e = 0; e < E; ++e) bits |= (unit_get<T>(o, e) > 0.f ? 1u : 0u) << e;
                    mask[i + u * 256] = (uint8_t)bits;
                }
            }
        }
    }
    for (; i < se; i += 256) {
        if (!FIXED) {
            const int c0 = (int)((i * E) % C);
            DPC_UNROLL
            for (int e = 0; e < E; ++e) {
                sc[e] = scale[c0 + e]; sh[e] = shift[c0 + e];
                rs[e] = rscale ? rscale[c0 + e] : 1.f; rb[e] = rscale ? rshift[c0 + e] : 0.f;
            }
        }
        const u32x4 xv = ld16<NT>(x, i);
        u32x4 rv = {0u, 0u, 0u, 0u};
        if (res) rv = ld16<NT>(res, i);
        float ov[E];
        unsigned bits = 0;
        DPC_UNROLL
        for (int e = 0; e < E; ++e) {
            float v = unit_get<T>(xv, e) * sc[e] + sh[e];
            if (res) v += unit_get<T>(rv, e) * rs[e] + rb[e];
            if (relu) v = v > 0.f ? v : 0.f;
            ov[e] = v;
        }
        const u32x4 o = unit_pack<T>(ov);
        st16<NT>(y, i, o);
        if (mask) {  // one byte per 16-byte unit: bit e = "stored y[e] > 0"; the backward reads this instead of y
            DPC_UNROLL
            for (int e = 0; e < E; ++e) bits |= (unit_get<T>(o, e) > 0.f ? 1u : 0u) << e;
            mask[i] = (uint8_t)bits;
        }
    }
}

extern "C" int dpc_bn_apply(const void* x, void* y, int32_t dtype, int64_t rows, int32_t C, const float* scale,
                            const float* shift, const void* res, const float* rscale, const float* rshift, int32_t relu,
                            uint8_t* mask, dpc_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!x || !y || rows <= 0 || C <= 0 || !scale || !shift) return DPC_ERR_ARG;
    const int E = dtype == DPC_BF16 ? 8 : 4;
    if (C % E) return DPC_ERR_UNSUPPORTED;
    const long long units = rows * C / E;
    const bool fixed = (256 * E) % C == 0;
    static const int rev_on = getenv("DPC_BN_APPLY_REV") ? atoi(getenv("DPC_BN_APPLY_REV")) : 0;
    const int rev = rev_on && bn_streaming(units) ? 1 : 0;   // small tensors fit the caches whole: the order does not matter
    if (dtype == DPC_F32) {
        if (fixed) {
            if (bn_streaming(units)) { DPC_LAUNCH((bn_apply_kernel<float, true, true>), dim3(grid_for(units)), dim3(256), stream, (const float*)x, (float*)y, units, C, scale, shift, (const float*)res, rscale, rshift, relu, mask, rev); } else { DPC_LAUNCH((bn_apply_kernel<float, true, false>), dim3(grid_for(units)), dim3(256), stream, (const float*)x, (float*)y, units, C, scale, shift, (const float*)res, rscale, rshift, relu, mask, rev); }
        } else {
            if (bn_streaming(units)) { DPC_LAUNCH((bn_apply_kernel<float, false, true>), dim3(grid_for(units)), dim3(256), stream, (const float*)x, (float*)y, units, C, scale, shift, (const float*)res, rscale, rshift, relu, mask, rev); } else { DPC_LAUNCH((bn_apply_kernel<float, false, false>), dim3(grid_for(units)), dim3(256), stream, (const float*)x, (float*)y, units, C, scale, shift, (const float*)res, rscale, rshift, relu, mask, rev); }
        }
    } else if (dtype == DPC_BF16) {
        if (fixed) {
            // (NT, U) instantiation and grid: U units per thread per iteration, so at most units / (256 U) workgroups
            const bool nt = bn_streaming(units);
            const int un = nt ? bn_unroll() : bn_small_unroll();
            const unsigned grid = grid_for((units + un - 1) / un, 256, nt ? (un > 1 ? bn_unroll_grid() : 8192) : bn_small_grid(512));   // small tensors: workgroup dispatch (~5 ns each) is what 8 192 short workgroups cost
#define DPC_BN_GO(NTV, UV) DPC_LAUNCH((bn_apply_kernel<bf16_t, true, NTV, UV>), dim3(grid), dim3(256), stream, (const bf16_t*)x, (bf16_t*)y, units, C, scale, shift, (const bf16_t*)res, rscale, rshift, relu, mask, rev)
            if (nt) { if (un == 4) DPC_BN_GO(true, 4); else if (un == 2) DPC_BN_GO(true, 2); else DPC_BN_GO(true, 1); }
            else { if (un == 4) DPC_BN_GO(false, 4); else if (un == 2) DPC_BN_GO(false, 2); else DPC_BN_GO(false, 1); }
#undef DPC_BN_GO
        } else {
            if (bn_streaming(units)) { DPC_LAUNCH((bn_apply_kernel<bf16_t, false, true>), dim3(grid_for(units)), dim3(256), stream, (const bf16_t*)x, (bf16_t*)y, units, C, scale, shift, (const bf16_t*)res, rscale, rshift, relu, mask, rev); } else { DPC_LAUNCH((bn_apply_kernel<bf16_t, false, false>), dim3(grid_for(units)), dim3(256), stream, (const bf16_t*)x, (bf16_t*)y, units, C, scale, shift, (const bf16_t*)res, rscale, rshift, relu, mask, rev); }
        }
    } else {
        return DPC_ERR_ARG;
    }
    return dpc_launch_status();
}

// ------------------------------------------------------------------ backward reduce
// dz = dy * (y > 0 if relu);  partial[b][0][c] = sum dz, partial[b][1][c] = sum dz * xhat
template <class T, bool NT, int U = 1>
__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(const T* dy, const T* y, const uint8_t* mask, const T* x, long long rows, int C,
                                                            const float* mean, const float* invstd, int relu,
                                                            float* partials, long long rows_per_block) {
    constexpr int E = Elt<T>::PER16;
    __shared__ float red[2][256 * E];
    const int upr = C / E;            // units per row
    const int rpi = 256 / upr;        // rows per iteration
    const int tid = threadIdx.x;
    const int cu = tid % upr, rr = tid / upr;
    float a1[E], a2[E], mu[E], is[E];
    DPC_UNROLL
    for (int e = 0; e < E; ++e) { a1[e] = 0.f; a2[e] = 0.f; }
    const bool active = rr < rpi;
    if (active) {
        DPC_UNROLL
        for (int e = 0; e < E; ++e) { mu[e] = mean[cu * E + e]; is[e] = invstd[cu * E + e]; }
        const long long r_begin = (long long)blockIdx.x * rows_per_block;
        long long r_end = r_begin + rows_per_block;
        if (r_end > rows) r_end = rows;
        long long r = r_begin + rr;
        if constexpr (U > 1) {   // U rows per thread in flight (see bn_apply_kernel); byte-mask form of the gate only
            if (!relu || mask) {
                for (; r + (long long)(U - 1) * rpi < r_end; r += (long long)U * rpi) {
                    u32x4 dv[U], xv[U];
                    unsigned bits[U];
                    DPC_UNROLL
                    for (int u = 0; u < U; ++u) dv[u] = ld16<NT>(dy, (r + (long long)u * rpi) * upr + cu);
                    DPC_UNROLL
                    for (int u = 0; u < U; ++u) xv[u] = ld16<NT>(x, (r + (long long)u * rpi) * upr + cu);
                    DPC_UNROLL
                    for (int u = 0; u < U; ++u) bits[u] = relu ? (unsigned)mask[(r + (long long)u * rpi) * upr + cu] : ~0u;
                    DPC_UNROLL
                    for (int u = 0; u < U; ++u)
                        DPC_UNROLL
                        for (int e = 0; e < E; ++e) {
                            float dz = unit_get<T>(dv[u], e);
                            if (!((bits[u] >> e) & 1u)) dz = 0.f;
                            const float xh = (unit_get<T>(xv[u], e) - mu[e]) * is[e];
                            a1[e] += dz;
                            a2[e] += dz * xh;
                        }
                }
            }
        }
        for (; r < r_end; r += rpi) {
            const long long ui = r * upr + cu;
            const u32x4 dv = ld16<NT>(dy, ui);
            const u32x4 xv = ld16<NT>(x, ui);
            unsigned bits = ~0u;
            if (relu) bits = mask ? (unsigned)mask[ui] : sign_bits<T>(((const u32x4*)y)[ui]);
            DPC_UNROLL
            for (int e = 0; e < E; ++e) {
                float dz = unit_get<T>(dv, e);
                if (!((bits >> e) & 1u)) dz = 0.f;
                const float xh = (unit_get<T>(xv, e) - mu[e]) * is[e];
                a1[e] += dz;
                a2[e] += dz * xh;
            }
        }
    }
    DPC_UNROLL
    for (int e = 0; e < E; ++e) {
        red[0][tid * E + e] = a1[e];
        red[1][tid * E + e] = a2[e];
    }
    __syncthreads();
    if (tid < C) {
        // channel tid lives in unit cu2 = tid / E, element e2 = tid % E of every row-group
        const int cu2 = tid / E, e2 = tid % E;
        float s1 = 0.f, s2 = 0.f;
        for (int g = 0; g < rpi; ++g) {
            const int t2 = g * upr + cu2;
            s1 += red[0][t2 * E + e2];
            s2 += red[1][t2 * E + e2];
        }
        partials[((long long)blockIdx.x * 2 + 0) * C + tid] = s1;
        partials[((long long)blockIdx.x * 2 + 1) * C + tid] = s2;
    }
}

static int bn_bwd_blocks(long long rows, int C, int E, long long* rows_per_block) {
    const int rpi = 256 / (C / E);
    long long per = (long long)rpi * 16;          // >=16 iterations per block
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

extern "C" int dpc_bn_bwd_reduce(const void* dy, const void* y, const uint8_t* mask, const void* x, int32_t dtype, int64_t rows, int32_t C,
                                 const float* mean, const float* invstd, int32_t relu, float* partials, int32_t* prow,
                                 dpc_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (rows <= 0 || C <= 0) return DPC_ERR_ARG;
    const int E = dtype == DPC_BF16 ? 8 : 4;
    if (C % E || C / E > 256 || C > 256) return DPC_ERR_UNSUPPORTED;
    long long rpb;
    const int blocks = bn_bwd_blocks(rows, C, E, &rpb);
    if (prow) *prow = blocks;
    if (!partials) return DPC_OK;  // size query
    if (!dy || !x || !mean || !invstd || (relu && !y && !mask)) return DPC_ERR_ARG;
    if (dtype == DPC_F32) {
        if (bn_streaming((long long)rows * C / E)) { DPC_LAUNCH((bn_bwd_reduce_kernel<float, true>), dim3(blocks), dim3(256), stream, (const float*)dy, (const float*)y, mask, (const float*)x, (long long)rows, C, mean, invstd, relu, partials, rpb); } else { DPC_LAUNCH((bn_bwd_reduce_kernel<float, false>), dim3(blocks), dim3(256), stream, (const float*)dy, (const float*)y, mask, (const float*)x, (long long)rows, C, mean, invstd, relu, partials, rpb); }
    } else if (dtype == DPC_BF16) {
        const bool nt = bn_streaming((long long)rows * C / E);
        const int un = nt ? bn_unroll() : bn_small_unroll();
#define DPC_BN_GO(NTV, UV) DPC_LAUNCH((bn_bwd_reduce_kernel<bf16_t, NTV, UV>), dim3(blocks), dim3(256), stream, (const bf16_t*)dy, (const bf16_t*)y, mask, (const bf16_t*)x, (long long)rows, C, mean, invstd, relu, partials, rpb)
        if (nt) { if (un > 1) DPC_BN_GO(true, 4); else DPC_BN_GO(true, 1); }
        else { if (un > 1) DPC_BN_GO(false, 4); else DPC_BN_GO(false, 1); }
#undef DPC_BN_GO
    } else {
        return DPC_ERR_ARG;
    }
    return dpc_launch_status();
}

__global__ __launch_bounds__(FIN_THREADS) void bn_bwd_finalize_kernel(const float* partials, int prow, int C, double count, float* dgamma, float* dbeta, float* coef) {
    __shared__ double red[2][4][32];
    const int c = blockIdx.x * 32 + (int)threadIdx.x;
    double s1, s2;
    partial_rows_sum(partials, prow, C, s1, s2, red);
    if (threadIdx.x >= 32 || c >= C) return;
    dbeta[c] = (float)s1;
    dgamma[c] = (float)s2;
    coef[c] = (float)(s1 / count);
    coef[C + c] = (float)(s2 / count);
}

extern "C" int dpc_bn_bwd_finalize(const float* partials, int32_t prow, int32_t C, double count, float* dgamma,
                                   float* dbeta, float* coef, dpc_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!partials || prow <= 0 || C <= 0 || count <= 0 || !dgamma || !dbeta || !coef) return DPC_ERR_ARG;
    DPC_LAUNCH(bn_bwd_finalize_kernel, dim3((C + 31) / 32), dim3(FIN_THREADS), stream, partials, prow, C, count, dgamma, dbeta, coef);
    return dpc_launch_status();
}

// dx = gamma*invstd*(dz - c1 - xhat*c2)
template <class T, bool FIXED, bool NT, int U = 1>
__global__ void bn_bwd_apply_kernel(const T* dy, const T* y, const uint8_t* mask, const T* x, long long units, int C, const float* mean,
                                    const float* invstd, const float* gamma, const float* coef, int relu, T* dx, T* dzout) {
    constexpr int E = Elt<T>::PER16;
    float mu[E], is[E], ga[E], c1[E], c2[E];
    if (FIXED) {
        const int c0 = (int)((threadIdx.x * E) % C);
        DPC_UNROLL
        for (int e = 0; e < E; ++e) {
            mu[e] = mean[c0 + e]; is[e] = invstd[c0 + e]; ga[e] = gamma[c0 + e] * is[e];
            c1[e] = coef[c0 + e]; c2[e] = coef[C + c0 + e];
        }
    }
    // Each workgroup owns one contiguous span (a multiple of 256 units, so a thread keeps its channel group) and
    // the spans are walked from the END of the tensor: bn_bwd_reduce has just streamed dy and x front to back, so
    // their tails are what the Infinity Cache / L2 still hold.
    const long long span = (units + (long long)gridDim.x * 256 - 1) / ((long long)gridDim.x * 256) * 256;
    const long long sb = (long long)(gridDim.x - 1 - blockIdx.x) * span;
    const long long se = sb + span < units ? sb + span : units;
    long long i = sb + threadIdx.x;
    if constexpr (FIXED && U > 1) {   // U units per thread in flight (see bn_apply_kernel); only the byte-mask form of the ReLU gate
        if (!relu || mask) {
            for (; i + (U - 1) * 256 < se; i += 256 * U) {
                u32x4 dv[U], xv[U];
                unsigned bits[U];
                DPC_UNROLL
                for (int u = 0; u < U; ++u) dv[u] = ld16<NT>(dy, i + u * 256);
                DPC_UNROLL
                for (int u = 0; u < U; ++u) xv[u] = ld16<NT>(x, i + u * 256);
                DPC_UNROLL
                for (int u = 0; u < U; ++u) bits[u] = ~0u;
                if (relu) {
                    DPC_UNROLL
                    for (int u = 0; u < U; ++u) bits[u] = (unsigned)mask[i + u * 256];
                }
                DPC_UNROLL
                for (int u = 0; u < U; ++u) {
                    float ov[E], oz[E];
                    DPC_UNROLL
                    for (int e = 0; e < E; ++e) {
                        float dz = unit_get<T>(dv[u], e);
                        if (!((bits[u] >> e) & 1u)) dz = 0.f;
                        const float xh = (unit_get<T>(xv[u], e) - mu[e]) * is[e];
                        ov[e] = ga[e] * (dz - c1[e] - xh * c2[e]);
                        oz[e] = dz;
                    }
                    st16<NT>(dx, i + u * 256, unit_pack<T>(ov));
                    if (dzout) st16<NT>(dzout, i + u * 256, unit_pack<T>(oz));
                }
            }
        }
    }
    for (; i < se; i += 256) {
        if (!FIXED) {
            const int c0 = (int)((i * E) % C);
            DPC_UNROLL
            for (int e = 0; e < E; ++e) {
                mu[e] = mean[c0 + e]; is[e] = invstd[c0 + e]; ga[e] = gamma[c0 + e] * is[e];
                c1[e] = coef[c0 + e]; c2[e] = coef[C + c0 + e];
            }
        }
        const u32x4 dv = ld16<NT>(dy, i);
        const u32x4 xv = ld16<NT>(x, i);
        unsigned bits = ~0u;
        if (relu) bits = mask ? (unsigned)mask[i] : sign_bits<T>(((const u32x4*)y)[i]);
        float ov[E], oz[E];
        DPC_UNROLL
        for (int e = 0; e < E; ++e) {
            float dz = unit_get<T>(dv, e);
            if (!((bits >> e) & 1u)) dz = 0.f;
            const float xh = (unit_get<T>(xv, e) - mu[e]) * is[e];
            ov[e] = ga[e] * (dz - c1[e] - xh * c2[e]);
            oz[e] = dz;
        }
        st16<NT>(dx, i, unit_pack<T>(ov));
        if (dzout) st16<NT>(dzout, i, unit_pack<T>(oz));
    }
}

extern "C" int dpc_bn_bwd_apply(const void* dy, const void* y, const uint8_t* mask, const void* x, int32_t dtype, int64_t rows, int32_t C,
                                const float* mean, const float* invstd, const float* gamma, const float* coef, int32_t relu,
                                void* dx, void* dz, dpc_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!dy || !x || !dx || rows <= 0 || C <= 0 || !mean || !invstd || !gamma || !coef || (relu && !y && !mask)) return DPC_ERR_ARG;
    const int E = dtype == DPC_BF16 ? 8 : 4;
    if (C % E) return DPC_ERR_UNSUPPORTED;
    const long long units = rows * C / E;
    // 2048 workgroups = the 8 resident per CU: each streams one long contiguous span (5.5 -> 5.8 TB/s, with dz 5.35 -> 6.0)
    const bool fixed = (256 * E) % C == 0;
    if (dtype == DPC_F32) {
        if (fixed) {
            if (bn_streaming(units)) { DPC_LAUNCH((bn_bwd_apply_kernel<float, true, true>), dim3(grid_for(units, 256, 2048)), dim3(256), stream, (const float*)dy, (const float*)y, mask, (const float*)x, units, C, mean, invstd, gamma, coef, relu, (float*)dx, (float*)dz); } else { DPC_LAUNCH((bn_bwd_apply_kernel<float, true, false>), dim3(grid_for(units, 256, 2048)), dim3(256), stream, (const float*)dy, (const float*)y, mask, (const float*)x, units, C, mean, invstd, gamma, coef, relu, (float*)dx, (float*)dz); }
        } else {
            if (bn_streaming(units)) { DPC_LAUNCH((bn_bwd_apply_kernel<float, false, true>), dim3(grid_for(units, 256, 2048)), dim3(256), stream, (const float*)dy, (const float*)y, mask, (const float*)x, units, C, mean, invstd, gamma, coef, relu, (float*)dx, (float*)dz); } else { DPC_LAUNCH((bn_bwd_apply_kernel<float, false, false>), dim3(grid_for(units, 256, 2048)), dim3(256), stream, (const float*)dy, (const float*)y, mask, (const float*)x, units, C, mean, invstd, gamma, coef, relu, (float*)dx, (float*)dz); }
        }
    } else if (dtype == DPC_BF16) {
        if (fixed) {
            const bool nt = bn_streaming(units);
            const int un = nt ? bn_unroll() : bn_small_unroll();
            const unsigned grid = grid_for((units + un - 1) / un, 256, nt ? (un > 1 ? bn_bwd_grid() : 2048) : bn_small_grid(2048));
#define DPC_BN_GO(NTV, UV) DPC_LAUNCH((bn_bwd_apply_kernel<bf16_t, true, NTV, UV>), dim3(grid), dim3(256), stream, (const bf16_t*)dy, (const bf16_t*)y, mask, (const bf16_t*)x, units, C, mean, invstd, gamma, coef, relu, (bf16_t*)dx, (bf16_t*)dz)
            if (nt) { if (un == 4) DPC_BN_GO(true, 4); else if (un == 2) DPC_BN_GO(true, 2); else DPC_BN_GO(true, 1); }
            else { if (un == 4) DPC_BN_GO(false, 4); else if (un == 2) DPC_BN_GO(false, 2); else DPC_BN_GO(false, 1); }
#undef DPC_BN_GO
        } else {
            if (bn_streaming(units)) { DPC_LAUNCH((bn_bwd_apply_kernel<bf16_t, false, true>), dim3(grid_for(units, 256, 2048)), dim3(256), stream, (const bf16_t*)dy, (const bf16_t*)y, mask, (const bf16_t*)x, units, C, mean, invstd, gamma, coef, relu, (bf16_t*)dx, (bf16_t*)dz); } else { DPC_LAUNCH((bn_bwd_apply_kernel<bf16_t, false, false>), dim3(grid_for(units, 256, 2048)), dim3(256), stream, (const bf16_t*)dy, (const bf16_t*)y, mask, (const bf16_t*)x, units, C, mean, invstd, gamma, coef, relu, (bf16_t*)dx, (bf16_t*)dz); }
        }
    } else {
        return DPC_ERR_ARG;
    }
    return dpc_launch_status();
}
