// gru_chain.hip -- the whole ConvGRU aggregate / predict recurrence of DPC_RNN.forward in ONE launch, and its
// backward in one more (dpc/model_3d.py:62-72, backbone/convrnn.py:24-34,62-88, network_pred dpc/model_3d.py:36-40).
//
// Every matrix product of the recurrence is a 1x1 convolution, so the rows m = (b, s) of the [B*SQ][D] state are
// independent sequences: a workgroup owns 32 rows and runs ALL n_agg + P - 1 GRU steps and the P predictions for
// them without ever meeting another workgroup.  Round 1 ran this as ~100 dependent launches of 11-17 us on grids of
// 32-96 workgroups (2 ms of pure launch latency per train step at cfg2); here the state lives in LDS, the gate
// pre-activations never leave the accumulators and only what the backward needs is written (u, r, o, h, h*r).
//
// Matrix cores: 32x32 MFMA tiles (bf16 32x32x16 / exact f32 32x32x2 through mfma_unit<T>), wave w owns the output
// columns of tiles w, w+4 (D <= 256), for every gate -- so u, r, o, h of one (row, column) meet in one lane and the
// gate math is lane-local.  The A operand (x | h | h*r rows) is read from swizzled 128-byte LDS rows; the B operand
// (weights) is streamed from L2 in a FRAGMENT-MAJOR packing made once per optimizer step (dpc_gru_pack): the 64
// lanes of a wave read 1 KB contiguous per MFMA K-step, four steps ahead of use.
//
// Dropout on the carried state (convrnn.py:78): explicit pre-scaled masks (tests inject the reference's), or
// Philox4x32-10 keyed on (seed, device-side optimizer step) generated in the kernel -- the same bits
// dpc_dropout_mask writes (philox.h), so forward, backward and a mask dump agree by construction.
#include "dpc_rt.h"
#include "../../include/dpc_hip.h"
#include "philox.h"

namespace {

enum { M_WU = 0, M_WR, M_WOX, M_WOH, M_W1, M_W2, M_WOHT, M_WUXT, M_WRXT, M_WOXT, M_WUHT, M_WRHT, M_W2T, M_W1T, NMAT };

__device__ __host__ __forceinline__ long long mat_off(int i, int D) {  // element offset of packed matrix i
    const long long DD = (long long)D * D;
    return i == 0 ? 0 : (i == 1 ? 2 * DD : 4 * DD + (long long)(i - 2) * DD);
}

__device__ __forceinline__ float sigm(float x) { return fast_rcp(1.f + fast_exp(-x)); }
__device__ __forceinline__ float tanh_fast(float x) { return 1.f - 2.f * fast_rcp(1.f + fast_exp(2.f * x)); }  // +-1 at +-inf

// ---------------------------------------------------------------- weight packing (fragment-major)
// packed[mat][tile n/32][ks][lane = kg*32 + j] = 16 bytes = B[n = tile*32 + j][k = (ks*2 + kg)*E .. +E)
struct PackSrc {
    const float* w[5];  // update, reset, out gate [D][2D]; network_pred.0, .2 [D][D]
};

template <class T>
__global__ void gru_pack_kernel(PackSrc src, int D, T* packed) {
    constexpr int E = Elt<T>::PER16;
    const int mat = blockIdx.y;
    // source table: which parameter, element offset, stride of n, stride of k, K
    int wi, base, sn, sk, K = D;
    switch (mat) {
    case M_WU: wi = 0; base = 0; sn = 2 * D; sk = 1; K = 2 * D; break;
    case M_WR: wi = 1; base = 0; sn = 2 * D; sk = 1; K = 2 * D; break;
    case M_WOX: wi = 2; base = 0; sn = 2 * D; sk = 1; break;
    case M_WOH: wi = 2; base = D; sn = 2 * D; sk = 1; break;
    case M_W1: wi = 3; base = 0; sn = D; sk = 1; break;
    case M_W2: wi = 4; base = 0; sn = D; sk = 1; break;
    case M_WOHT: wi = 2; base = D; sn = 1; sk = 2 * D; break;
    case M_WUXT: wi = 0; base = 0; sn = 1; sk = 2 * D; break;
    case M_WRXT: wi = 1; base = 0; sn = 1; sk = 2 * D; break;
    case M_WOXT: wi = 2; base = 0; sn = 1; sk = 2 * D; break;
    case M_WUHT: wi = 0; base = D; sn = 1; sk = 2 * D; break;
    case M_WRHT: wi = 1; base = D; sn = 1; sk = 2 * D; break;
    case M_W2T: wi = 4; base = 0; sn = 1; sk = D; break;
    default: wi = 3; base = 0; sn = 1; sk = D; break;  // M_W1T
    }
    if (!src.w[wi]) return;  // P = 0 (LC classifier): no network_pred matrices
    const float* w = src.w[wi] + base;
    const int KS = K / (2 * E);
    const long long nunits = (long long)D * K / E;
    T* out = packed + mat_off(mat, D);
    for (long long q = (long long)blockIdx.x * blockDim.x + threadIdx.x; q < nunits; q += (long long)gridDim.x * blockDim.x) {
        const int lane = (int)(q & 63);
        const int ks = (int)((q >> 6) % KS);
        const int tile = (int)((q >> 6) / KS);
        const int n = tile * 32 + (lane & 31);
        const int k0 = (ks * 2 + (lane >> 5)) * E;
        DPC_UNROLL
        for (int e = 0; e < E; ++e) out[q * E + e] = Elt<T>::from_f32(w[(long long)n * sn + (long long)(k0 + e) * sk]);
    }
}

// ---------------------------------------------------------------- shared pieces of the two chain kernels
struct ChainP {
    int M, D, SQ, P, n_agg, n_steps;
    const void* packed;
    const float *bu, *br, *bo, *b1, *b2;
    // dropout: explicit masks [n_steps][M][D] (pre-scaled), else Philox when step_dev != NULL, else none
    const float* drop;
    const int32_t* step_dev;
    unsigned long long seed;
    uint32_t thresh24;
    float inv_keep;
    // state (T = compute dtype unless noted)
    void* X_all;      // [n_steps][M][D]   in: relu'd features of the aggregation steps; out: relu(pred_i) for the rest
    void* H_all;      // [n_steps+1][M][D] [0] = h_0 (zeros), [s+1] = state after step s (dropout applied)
    void* HR_all;     // [n_steps][M][D]   h * r
    float *U_all, *R_all, *O_all;  // [n_steps][M][D] f32 gates
    void* P1_all;     // [P][M][D]         relu(W1 h + b1)
    void* pred;       // [B][P][SQ][D]
    // backward
    const float* d_pred;  // [B][P][SQ][D] f32
    void* G_all;          // [n_steps][M][3D]  pre-activation gradients [u | r | o]
    void *dP1, *dP2;      // [P][M][D]
    float* d_x;           // [n_agg][M][D] f32: gradient w.r.t. the aggregation inputs
    float* ws;            // [2][M][D] f32 scratch: dh, dx of a prediction-feeding step
    const float* d_hlast; // optional [M][D] f32: gradient w.r.t. the last state (the LC classifier reads it; NULL = 0)
};

constexpr int TM = 32;          // rows per workgroup
#define NWV ((int)(blockDim.x >> 6))   // waves of the workgroup = stride between the column tiles a wave owns (4, or 8 for D = 256: one tile per wave)
constexpr int CHUNK = TM * 128; // bytes of one 128-byte-column chunk of an LDS operand tile

template <class T> struct TileIO {
    static constexpr int E = Elt<T>::PER16;
    static constexpr int ESZ = 16 / E;
    // byte address of 16-byte unit `unit` of row `row` inside an operand tile
    __device__ static __forceinline__ int unit_addr(int row, int unit) { return (unit >> 3) * CHUNK + lds_unit_off(row, unit & 7); }
    __device__ static __forceinline__ int elem_addr(int row, int col) { return unit_addr(row, col / E) + (col % E) * ESZ; }
    __device__ static __forceinline__ float get(const unsigned char* tile, int row, int col) {
        return Elt<T>::to_f32(*(const T*)(tile + elem_addr(row, col)));
    }
    __device__ static __forceinline__ void put(unsigned char* tile, int row, int col, float v) {
        *(T*)(tile + elem_addr(row, col)) = Elt<T>::from_f32(v);
    }
};

// rows [m0, m0+32) of a [M][D] matrix -> LDS operand tile (zero rows beyond M)
template <class T>
__device__ __forceinline__ void load_tile(unsigned char* tile, const T* src, int m0, int M, int D) {
    constexpr int E = Elt<T>::PER16;
    const int KU = D / E;
    for (int q = threadIdx.x; q < TM * KU; q += blockDim.x) {
        const int row = q / KU, unit = q - row * KU;
        u32x4 v = {0u, 0u, 0u, 0u};
        if (m0 + row < M) v = *(const u32x4*)(src + (long long)(m0 + row) * D + unit * E);
        *(u32x4*)(tile + TileIO<T>::unit_addr(row, unit)) = v;
    }
}
template <class T> __device__ __forceinline__ void zero_tile(unsigned char* tile, int D) {
    constexpr int E = Elt<T>::PER16;
    const int KU = D / E;
    const u32x4 z = {0u, 0u, 0u, 0u};
    for (int q = threadIdx.x; q < TM * KU; q += blockDim.x) *(u32x4*)(tile + TileIO<T>::unit_addr(q / KU, q % KU)) = z;
}

// acc[b][t] += A(tile rows, K = ks_n unit steps starting at the tile's unit 0) x B_b(tile t of the wave)^T
// Bp[b][t]: fragment base of packed matrix b, tile t, ALREADY advanced to the first K step of this call.
// The B fragments are fetched four K steps ahead of their MFMAs (register ring with compile-time slots).
// NT = number of 32-column tiles this wave owns (compile time: branches around MFMAs cost registers).
template <class T, int NB, int NT>
__device__ __forceinline__ void gemm_acc(f32x16 (&acc)[NB][2], const unsigned char* tile, int ks_n, const u32x4* const (&Bp)[NB][2], int lane) {
    if constexpr (NT == 0) return;
    const int row = lane & 31, kg = lane >> 5;
    constexpr int NTT = NT > 0 ? NT : 1;
    auto a_frag = [&](int ks) -> u32x4 {
        const int unit = ks * 2 + kg;
        return *(const u32x4*)(tile + (unit >> 3) * CHUNK + lds_unit_off(row, unit & 7));
    };
    if ((ks_n & 3) == 0 && ks_n >= 4) {
        // software pipeline without a single conditional MFMA (a branch around an MFMA doubles its accumulator registers)
        u32x4 ring[4][NB][NTT];
        DPC_UNROLL
        for (int j = 0; j < 4; ++j)
            DPC_UNROLL
            for (int b = 0; b < NB; ++b)
                DPC_UNROLL
                for (int t = 0; t < NT; ++t) ring[j][b][t] = Bp[b][t][j * 64 + lane];
        for (int ks0 = 0; ks0 < ks_n - 4; ks0 += 4) {
            DPC_UNROLL
            for (int j = 0; j < 4; ++j) {
                const u32x4 a = a_frag(ks0 + j);
                DPC_UNROLL
                for (int b = 0; b < NB; ++b)
                    DPC_UNROLL
                    for (int t = 0; t < NT; ++t) {
                        acc[b][t] = mfma_unit<T>(a, ring[j][b][t], acc[b][t]);
                        ring[j][b][t] = Bp[b][t][(ks0 + j + 4) * 64 + lane];
                    }
            }
        }
        DPC_UNROLL
        for (int j = 0; j < 4; ++j) {
            const u32x4 a = a_frag(ks_n - 4 + j);
            DPC_UNROLL
            for (int b = 0; b < NB; ++b)
                DPC_UNROLL
                for (int t = 0; t < NT; ++t) acc[b][t] = mfma_unit<T>(a, ring[j][b][t], acc[b][t]);
        }
    } else {  // narrow test networks (D = 32 in bf16: two K steps)
        for (int ks = 0; ks < ks_n; ++ks) {
            const u32x4 a = a_frag(ks);
            DPC_UNROLL
            for (int b = 0; b < NB; ++b)
                DPC_UNROLL
                for (int t = 0; t < NT; ++t) acc[b][t] = mfma_unit<T>(a, Bp[b][t][ks * 64 + lane], acc[b][t]);
        }
    }
}

template <int NB> __device__ __forceinline__ void zero_acc(f32x16 (&acc)[NB][2]) {
    DPC_UNROLL
    for (int b = 0; b < NB; ++b)
        DPC_UNROLL
        for (int t = 0; t < 2; ++t)
            DPC_UNROLL
            for (int r = 0; r < 16; ++r) acc[b][t][r] = 0.f;
}

// fragment base (in 16-byte units) of packed matrix `mat`, the wave's t-th tile, K step `ks_begin`; KS = K steps of the matrix
template <class T>
__device__ __forceinline__ const u32x4* frag(const ChainP& p, int mat, int tile, int KS, int ks_begin) {
    return (const u32x4*)((const T*)p.packed + mat_off(mat, p.D)) + ((long long)tile * KS + ks_begin) * 64;
}

// ---- accumulators (MFMA C layout: lane = column, 16 rows) <-> row-major f32 staging tile in LDS.
// Everything element-wise (gates, dropout, ReLU masks, the global loads / stores of the saved tensors) runs on the
// row-major side: a thread owns 4 consecutive columns of a row (one 16-byte access, one Philox block), the loops stay
// rolled-up small and no 64-bit per-element address lives in a register next to the accumulators.
constexpr int SPAD = 8;  // floats of row padding: rows 4 apart (the two half-waves of a C-layout store) land 32 banks apart
template <int NT>
__device__ __forceinline__ void stage_put(float* stage, int D, const f32x16 (&acc)[2], int wave, int lane) {
    const int ld = D + SPAD;
    DPC_UNROLL
    for (int t = 0; t < NT; ++t) {
        float* base = stage + (4 * (lane >> 5)) * ld + (wave + NWV * t) * 32 + (lane & 31);
        DPC_UNROLL
        for (int r = 0; r < 16; ++r) base[((r & 3) + 8 * (r >> 2)) * ld] = acc[t][r];
    }
}
// row-major iteration over the thread's 4-column groups (rolled: the state a thread carries from one pass to the next --
// u, dh, dx -- lives in memory the same thread wrote, never in per-iteration registers next to the accumulators)
#define RM_FOR(D_)                                                                      \
    DPC_NOUNROLL                                                                        \
    for (int q_ = (int)threadIdx.x; q_ < TM * ((D_) / 4); q_ += (int)blockDim.x)                    \
        if (const int row = q_ / ((D_) / 4), col = (q_ % ((D_) / 4)) * 4; true)

__device__ __forceinline__ f32x4 ld4(const float* p) { return *(const f32x4*)p; }
__device__ __forceinline__ void st4(float* p, const f32x4& v) { *(f32x4*)p = v; }
template <class T> __device__ __forceinline__ f32x4 ldT4(const T* p);
template <> __device__ __forceinline__ f32x4 ldT4<float>(const float* p) { return *(const f32x4*)p; }
template <> __device__ __forceinline__ f32x4 ldT4<bf16_t>(const bf16_t* p) {
    const u32x2 w = *(const u32x2*)p;
    f32x4 v = {bf16_to_f32((bf16_t)(w[0] & 0xffffu)), bf16_to_f32((bf16_t)(w[0] >> 16)), bf16_to_f32((bf16_t)(w[1] & 0xffffu)),
               bf16_to_f32((bf16_t)(w[1] >> 16))};
    return v;
}
template <class T> __device__ __forceinline__ void stT4(T* p, const f32x4& v);
template <> __device__ __forceinline__ void stT4<float>(float* p, const f32x4& v) { *(f32x4*)p = v; }
template <> __device__ __forceinline__ void stT4<bf16_t>(bf16_t* p, const f32x4& v) {
    u32x2 w = {bf16x2_pack(v[0], v[1]), bf16x2_pack(v[2], v[3])};
    *(u32x2*)p = w;
}
// round a value the way it is stored (so that what the next GEMM reads and what the math keeps agree)
template <class T> __device__ __forceinline__ f32x4 roundT4(const f32x4& v);
template <> __device__ __forceinline__ f32x4 roundT4<float>(const f32x4& v) { return v; }
template <> __device__ __forceinline__ f32x4 roundT4<bf16_t>(const f32x4& v) {
    f32x4 o;
    DPC_UNROLL
    for (int e = 0; e < 4; ++e) o[e] = bf16_to_f32(f32_to_bf16(v[e]));
    return o;
}
template <class T> __device__ __forceinline__ T* tile_ptr(unsigned char* tile, int row, int col) { return (T*)(tile + TileIO<T>::elem_addr(row, col)); }

// pre-scaled keep mask of elements (s, grow, col..col+3)
__device__ __forceinline__ f32x4 mask4(const ChainP& p, int s, int grow, int col) {
    const long long idx = ((long long)s * p.M + grow) * p.D + col;
    if (p.drop) return ld4(p.drop + idx);
    f32x4 k = {1.f, 1.f, 1.f, 1.f};
    if (p.step_dev) {
        float kk[4];
        dropout_keep4(p.seed, (uint32_t)p.step_dev[0], (uint32_t)(idx >> 2), p.thresh24, p.inv_keep, kk);
        k = f32x4{kk[0], kk[1], kk[2], kk[3]};
    }
    return k;
}

// ---------------------------------------------------------------- forward chain
template <class T, int NT, int DC>
__device__ __forceinline__ void gru_chain_fwd_body(const ChainP& p, unsigned char* smem) {
    constexpr int E = Elt<T>::PER16;
    const int D = DC > 0 ? DC : p.D, M = p.M;  // DC: compile-time feature size (256) -> shifts instead of divisions in the row-major passes
    const int tile_bytes = (D * (16 / E) + 127) / 128 * CHUNK;
    unsigned char* tx = smem;
    unsigned char* th = smem + tile_bytes;
    unsigned char* thr = smem + 2 * tile_bytes;
    float* stage = (float*)(smem + 3 * tile_bytes);
    const int ld = D + SPAD;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int m0 = blockIdx.x * TM;
    const int KSD = D / (2 * E);  // K steps over D input channels
    const long long MD = (long long)M * D;
    T* X_all = (T*)p.X_all;
    T* H_all = (T*)p.H_all;
    T* HR_all = (T*)p.HR_all;
    T* P1_all = (T*)p.P1_all;
    T* pred = (T*)p.pred;

    load_tile<T>(th, H_all, m0, M, D);  // h_0
    for (int s = 0; s < p.n_steps; ++s) {
        if (s < p.n_agg) load_tile<T>(tx, X_all + (long long)s * MD, m0, M, D);
        __syncthreads();
        // ---- [u | r] = [x | h] @ [Wu ; Wr]^T, o_x = x @ Wo_x^T
        f32x16 a3[3][2];
        zero_acc<3>(a3);
        {
            const u32x4* const Bx[3][2] = {{frag<T>(p, M_WU, wave, 2 * KSD, 0), frag<T>(p, M_WU, wave + NWV, 2 * KSD, 0)},
                                           {frag<T>(p, M_WR, wave, 2 * KSD, 0), frag<T>(p, M_WR, wave + NWV, 2 * KSD, 0)},
                                           {frag<T>(p, M_WOX, wave, KSD, 0), frag<T>(p, M_WOX, wave + NWV, KSD, 0)}};
            gemm_acc<T, 3, NT>(a3, tx, KSD, Bx, lane);
        }
        f32x16 aur[2][2];
        DPC_UNROLL
        for (int t = 0; t < 2; ++t) { aur[0][t] = a3[0][t]; aur[1][t] = a3[1][t]; }
        {
            const u32x4* const Bh[2][2] = {{frag<T>(p, M_WU, wave, 2 * KSD, KSD), frag<T>(p, M_WU, wave + NWV, 2 * KSD, KSD)},
                                           {frag<T>(p, M_WR, wave, 2 * KSD, KSD), frag<T>(p, M_WR, wave + NWV, 2 * KSD, KSD)}};
            gemm_acc<T, 2, NT>(aur, th, KSD, Bh, lane);
        }
        // ---- r = sigmoid(.), h*r (the operand of the next product)
        stage_put<NT>(stage, D, aur[1], wave, lane);
        __syncthreads();
        RM_FOR(D) {
            const int grow = m0 + row;
            const f32x4 pre = ld4(stage + row * ld + col), b = ld4(p.br + col);
            const f32x4 h = ldT4<T>(tile_ptr<T>(th, row, col));
            f32x4 rr, hr;
            DPC_UNROLL
            for (int e = 0; e < 4; ++e) { rr[e] = sigm(pre[e] + b[e]); hr[e] = h[e] * rr[e]; }
            stT4<T>(tile_ptr<T>(thr, row, col), hr);
            if (grow < M) {
                const long long o = (long long)s * MD + (long long)grow * D + col;
                st4(p.R_all + o, rr);
                stT4<T>(HR_all + o, hr);
            }
        }
        __syncthreads();
        // ---- u = sigmoid(.) (kept in registers for the state update)
        stage_put<NT>(stage, D, aur[0], wave, lane);
        __syncthreads();
        RM_FOR(D) {
            const f32x4 pre = ld4(stage + row * ld + col), b = ld4(p.bu + col);
            f32x4 u;
            DPC_UNROLL
            for (int e = 0; e < 4; ++e) u[e] = sigm(pre[e] + b[e]);
            if (m0 + row < M) st4(p.U_all + (long long)s * MD + (long long)(m0 + row) * D + col, u);  // re-read by this thread below
        }
        // ---- o = tanh(o_x + (h*r) @ Wo_h^T + bo); h' = (h (1-u) + o u) * drop
        f32x16 ao[1][2];
        DPC_UNROLL
        for (int t = 0; t < 2; ++t) ao[0][t] = a3[2][t];
        {
            const u32x4* const Bo[1][2] = {{frag<T>(p, M_WOH, wave, KSD, 0), frag<T>(p, M_WOH, wave + NWV, KSD, 0)}};
            gemm_acc<T, 1, NT>(ao, thr, KSD, Bo, lane);
        }
        __syncthreads();  // every thread has read its u pre-activations
        stage_put<NT>(stage, D, ao[0], wave, lane);
        __syncthreads();
        RM_FOR(D) {
            const int grow = m0 + row;
            const f32x4 pre = ld4(stage + row * ld + col), b = ld4(p.bo + col);
            const f32x4 h = ldT4<T>(tile_ptr<T>(th, row, col));
            f32x4 o, hn = {0.f, 0.f, 0.f, 0.f};
            DPC_UNROLL
            for (int e = 0; e < 4; ++e) o[e] = tanh_fast(pre[e] + b[e]);
            if (grow < M) {
                const f32x4 u = ld4(p.U_all + (long long)s * MD + (long long)grow * D + col), k = mask4(p, s, grow, col);
                DPC_UNROLL
                for (int e = 0; e < 4; ++e) hn[e] = (h[e] * (1.f - u[e]) + o[e] * u[e]) * k[e];
            }
            stT4<T>(tile_ptr<T>(th, row, col), hn);
            if (grow < M) {
                const long long oidx = (long long)s * MD + (long long)grow * D + col;
                st4(p.O_all + oidx, o);
                stT4<T>(H_all + oidx + MD, hn);
            }
        }
        __syncthreads();
        // ---- prediction i from the state after step s (dpc/model_3d.py:66-69), feeding the next step
        const int i = s - (p.n_agg - 1);
        if (i >= 0 && i < p.P) {
            f32x16 a1[1][2];
            zero_acc<1>(a1);
            {
                const u32x4* const B1[1][2] = {{frag<T>(p, M_W1, wave, KSD, 0), frag<T>(p, M_W1, wave + NWV, KSD, 0)}};
                gemm_acc<T, 1, NT>(a1, th, KSD, B1, lane);
            }
            stage_put<NT>(stage, D, a1[0], wave, lane);
            __syncthreads();
            RM_FOR(D) {
                const f32x4 pre = ld4(stage + row * ld + col), b = ld4(p.b1 + col);
                f32x4 v;
                DPC_UNROLL
                for (int e = 0; e < 4; ++e) { v[e] = pre[e] + b[e]; v[e] = v[e] > 0.f ? v[e] : 0.f; }
                stT4<T>(tile_ptr<T>(thr, row, col), v);
                if (m0 + row < M) stT4<T>(P1_all + (long long)i * MD + (long long)(m0 + row) * D + col, v);
            }
            __syncthreads();
            zero_acc<1>(a1);
            {
                const u32x4* const B2[1][2] = {{frag<T>(p, M_W2, wave, KSD, 0), frag<T>(p, M_W2, wave + NWV, KSD, 0)}};
                gemm_acc<T, 1, NT>(a1, thr, KSD, B2, lane);
            }
            stage_put<NT>(stage, D, a1[0], wave, lane);
            __syncthreads();
            RM_FOR(D) {
                const int grow = m0 + row;
                const f32x4 pre = ld4(stage + row * ld + col), b = ld4(p.b2 + col);
                f32x4 v, xr;
                DPC_UNROLL
                for (int e = 0; e < 4; ++e) { v[e] = pre[e] + b[e]; xr[e] = v[e] > 0.f ? v[e] : 0.f; }
                if (i < p.P - 1) stT4<T>(tile_ptr<T>(tx, row, col), xr);
                if (grow < M) {
                    const int b_ = grow / p.SQ, sq = grow - b_ * p.SQ;
                    stT4<T>(pred + ((long long)(b_ * p.P + i) * p.SQ + sq) * D + col, v);
                    if (i < p.P - 1) stT4<T>(X_all + (long long)(s + 1) * MD + (long long)grow * D + col, xr);
                }
            }
            __syncthreads();
        }
    }
}

// ---------------------------------------------------------------- backward chain
// One GRU step s, reversed (the arithmetic of round 1's dpc_gru_bwd1/2 kernels and their three GEMMs):
//   dhn = dh*drop; G_u = dhn (o-h) u (1-u); G_o = dhn u (1-o^2); dhprev = dhn (1-u)
//   dhr = G_o @ Wo_h;  G_r = dhr h r (1-r);  dhprev += dhr r
//   dx = [G_u|G_r|G_o] @ [Wu_x;Wr_x;Wo_x];  dh <- dhprev + [G_u|G_r] @ [Wu_h;Wr_h]
// dh lives in p.ws[0] (f32 [M][D]), the dx of a step that feeds a prediction in p.ws[1]: thread-private state (every element
// is read and written by the same thread in every pass), kept in L2-resident memory instead of registers.
template <class T, int NT, int DC>
__device__ __forceinline__ void gru_step_bwd(const ChainP& p, int s, int m0, int wave, int lane, unsigned char* tu, unsigned char* tr,
                                             unsigned char* to, float* stage, float* dx_out) {
    constexpr int E = Elt<T>::PER16;
    const int D = DC > 0 ? DC : p.D, M = p.M;
    const int ld = D + SPAD;
    const int KSD = D / (2 * E);
    const long long MD = (long long)M * D;
    const T* H_all = (const T*)p.H_all;
    T* G_all = (T*)p.G_all;
    float* dh_ws = p.ws;
    RM_FOR(D) {
        const int grow = m0 + row;
        f32x4 gu = {0.f, 0.f, 0.f, 0.f}, go = gu;
        if (grow < M) {
            const long long o2 = (long long)grow * D + col, o = (long long)s * MD + o2;
            const f32x4 k = mask4(p, s, grow, col), dh = ld4(dh_ws + o2);
            const f32x4 u = ld4(p.U_all + o), oo = ld4(p.O_all + o), h = ldT4<T>(H_all + o);
            f32x4 dp;
            DPC_UNROLL
            for (int e = 0; e < 4; ++e) {
                const float dhn = dh[e] * k[e];
                gu[e] = dhn * (oo[e] - h[e]) * u[e] * (1.f - u[e]);
                go[e] = dhn * u[e] * (1.f - oo[e] * oo[e]);
                dp[e] = dhn * (1.f - u[e]);
            }
            st4(dh_ws + o2, dp);
            T* g = G_all + ((long long)s * M + grow) * 3 * D + col;
            stT4<T>(g, gu);
            stT4<T>(g + 2 * D, go);
        }
        stT4<T>(tile_ptr<T>(tu, row, col), gu);
        stT4<T>(tile_ptr<T>(to, row, col), go);
    }
    __syncthreads();
    f32x16 acc[1][2];
    zero_acc<1>(acc);
    {
        const u32x4* const B[1][2] = {{frag<T>(p, M_WOHT, wave, KSD, 0), frag<T>(p, M_WOHT, wave + NWV, KSD, 0)}};
        gemm_acc<T, 1, NT>(acc, to, KSD, B, lane);
    }
    stage_put<NT>(stage, D, acc[0], wave, lane);
    __syncthreads();
    RM_FOR(D) {
        const int grow = m0 + row;
        f32x4 gr = {0.f, 0.f, 0.f, 0.f};
        if (grow < M) {
            const long long o2 = (long long)grow * D + col, o = (long long)s * MD + o2;
            const f32x4 g = ld4(stage + row * ld + col), rr = ld4(p.R_all + o), h = ldT4<T>(H_all + o);
            f32x4 dp = ld4(dh_ws + o2);
            DPC_UNROLL
            for (int e = 0; e < 4; ++e) {
                gr[e] = g[e] * h[e] * rr[e] * (1.f - rr[e]);
                dp[e] += g[e] * rr[e];
            }
            st4(dh_ws + o2, dp);
            stT4<T>(G_all + ((long long)s * M + grow) * 3 * D + D + col, gr);
        }
        stT4<T>(tile_ptr<T>(tr, row, col), gr);
    }
    __syncthreads();
    // dx
    zero_acc<1>(acc);
    {
        const u32x4* const B0[1][2] = {{frag<T>(p, M_WUXT, wave, KSD, 0), frag<T>(p, M_WUXT, wave + NWV, KSD, 0)}};
        const u32x4* const B1[1][2] = {{frag<T>(p, M_WRXT, wave, KSD, 0), frag<T>(p, M_WRXT, wave + NWV, KSD, 0)}};
        const u32x4* const B2[1][2] = {{frag<T>(p, M_WOXT, wave, KSD, 0), frag<T>(p, M_WOXT, wave + NWV, KSD, 0)}};
        gemm_acc<T, 1, NT>(acc, tu, KSD, B0, lane);
        gemm_acc<T, 1, NT>(acc, tr, KSD, B1, lane);
        gemm_acc<T, 1, NT>(acc, to, KSD, B2, lane);
    }
    stage_put<NT>(stage, D, acc[0], wave, lane);
    __syncthreads();
    RM_FOR(D) {
        if (m0 + row < M) st4(dx_out + (long long)(m0 + row) * D + col, ld4(stage + row * ld + col));
    }
    __syncthreads();
    // dh
    zero_acc<1>(acc);
    {
        const u32x4* const B0[1][2] = {{frag<T>(p, M_WUHT, wave, KSD, 0), frag<T>(p, M_WUHT, wave + NWV, KSD, 0)}};
        const u32x4* const B1[1][2] = {{frag<T>(p, M_WRHT, wave, KSD, 0), frag<T>(p, M_WRHT, wave + NWV, KSD, 0)}};
        gemm_acc<T, 1, NT>(acc, tu, KSD, B0, lane);
        gemm_acc<T, 1, NT>(acc, tr, KSD, B1, lane);
    }
    stage_put<NT>(stage, D, acc[0], wave, lane);
    __syncthreads();
    RM_FOR(D) {
        if (m0 + row < M) {
            const long long o2 = (long long)(m0 + row) * D + col;
            const f32x4 g = ld4(stage + row * ld + col);
            f32x4 dp = ld4(dh_ws + o2);
            DPC_UNROLL
            for (int e = 0; e < 4; ++e) dp[e] += g[e];
            st4(dh_ws + o2, dp);
        }
    }
    __syncthreads();  // tiles and staging are free again
}

template <class T, int NT, int DC>
__device__ __forceinline__ void gru_chain_bwd_body(const ChainP& p, unsigned char* smem) {
    constexpr int E = Elt<T>::PER16;
    const int D = DC > 0 ? DC : p.D, M = p.M;
    const int tile_bytes = (D * (16 / E) + 127) / 128 * CHUNK;
    unsigned char* tu = smem;
    unsigned char* tr = smem + tile_bytes;
    unsigned char* to = smem + 2 * tile_bytes;
    float* stage = (float*)(smem + 3 * tile_bytes);
    const int ld = D + SPAD;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int m0 = blockIdx.x * TM;
    const int KSD = D / (2 * E);
    const long long MD = (long long)M * D;
    const T* X_all = (const T*)p.X_all;
    const T* P1_all = (const T*)p.P1_all;
    T* dP1 = (T*)p.dP1;
    T* dP2 = (T*)p.dP2;
    float* dh_ws = p.ws;
    float* dxn_ws = p.ws + MD;
    RM_FOR(D) {
        if (m0 + row < M) {
            const long long o2 = (long long)(m0 + row) * D + col;
            st4(dh_ws + o2, p.d_hlast ? ld4(p.d_hlast + o2) : f32x4{0.f, 0.f, 0.f, 0.f});
        }
    }
    int step = p.n_steps;
    for (int i = p.P - 1; i >= 0; --i) {
        // p_i = W2 relu(W1 h + b1) + b2 feeds the score (row-mapped) and, for i < P-1, GRU step `step` through relu
        const bool feeds = i < p.P - 1;
        if (feeds) {
            --step;
            gru_step_bwd<T, NT, DC>(p, step, m0, wave, lane, tu, tr, to, stage, dxn_ws);
        }
        RM_FOR(D) {
            const int grow = m0 + row;
            f32x4 g = {0.f, 0.f, 0.f, 0.f};
            if (grow < M) {
                const int b_ = grow / p.SQ, sq = grow - b_ * p.SQ;
                g = ld4(p.d_pred + ((long long)(b_ * p.P + i) * p.SQ + sq) * D + col);
                if (feeds) {
                    const f32x4 xr = ldT4<T>(X_all + (long long)step * MD + (long long)grow * D + col);
                    const f32x4 dx = ld4(dxn_ws + (long long)grow * D + col);
                    DPC_UNROLL
                    for (int e = 0; e < 4; ++e) g[e] += xr[e] > 0.f ? dx[e] : 0.f;
                }
                stT4<T>(dP2 + (long long)i * MD + (long long)grow * D + col, g);
            }
            stT4<T>(tile_ptr<T>(tu, row, col), g);
        }
        __syncthreads();
        f32x16 acc[1][2];
        zero_acc<1>(acc);
        {
            const u32x4* const B[1][2] = {{frag<T>(p, M_W2T, wave, KSD, 0), frag<T>(p, M_W2T, wave + NWV, KSD, 0)}};
            gemm_acc<T, 1, NT>(acc, tu, KSD, B, lane);
        }
        stage_put<NT>(stage, D, acc[0], wave, lane);
        __syncthreads();
        RM_FOR(D) {
            const int grow = m0 + row;
            f32x4 g = {0.f, 0.f, 0.f, 0.f};
            if (grow < M) {
                const long long o = (long long)i * MD + (long long)grow * D + col;
                const f32x4 g1 = ld4(stage + row * ld + col), p1 = ldT4<T>(P1_all + o);
                DPC_UNROLL
                for (int e = 0; e < 4; ++e) g[e] = p1[e] > 0.f ? g1[e] : 0.f;
                stT4<T>(dP1 + o, g);
            }
            stT4<T>(tile_ptr<T>(tr, row, col), g);
        }
        __syncthreads();
        zero_acc<1>(acc);
        {
            const u32x4* const B[1][2] = {{frag<T>(p, M_W1T, wave, KSD, 0), frag<T>(p, M_W1T, wave + NWV, KSD, 0)}};
            gemm_acc<T, 1, NT>(acc, tr, KSD, B, lane);  // dP1 @ W1
        }
        stage_put<NT>(stage, D, acc[0], wave, lane);
        __syncthreads();
        RM_FOR(D) {
            if (m0 + row < M) {
                const long long o2 = (long long)(m0 + row) * D + col;
                const f32x4 g = ld4(stage + row * ld + col);
                f32x4 dh = ld4(dh_ws + o2);
                DPC_UNROLL
                for (int e = 0; e < 4; ++e) dh[e] += g[e];
                st4(dh_ws + o2, dh);
            }
        }
        __syncthreads();
    }
    for (int t = p.n_agg - 1; t >= 0; --t) {
        --step;
        gru_step_bwd<T, NT, DC>(p, step, m0, wave, lane, tu, tr, to, stage, p.d_x + (long long)t * MD);
    }
}

// The number of tiles a wave owns is wave-uniform; one instantiation per count keeps every MFMA unconditional.
// DC > 0 would fix the feature size at compile time (shifts instead of divisions in the row-major passes); measured dead end:
// with constant trip counts hipcc unrolls the K loops completely and spills 350-840 VGPRs (372 us instead of 174 us), so
// only DC = 0 is instantiated.
template <class T, int DC>
__global__ __launch_bounds__(256) void gru_chain_fwd_kernel(ChainP p) {
    DPC_DYN_SMEM(smem);
    if constexpr (DC == 256) {
        gru_chain_fwd_body<T, 2, 256>(p, smem);
    } else {
        const int wave = threadIdx.x >> 6, ntiles = p.D / 32;
        const int ntw = wave < ntiles ? (wave + NWV < ntiles ? 2 : 1) : 0;
        if (ntw == 2) gru_chain_fwd_body<T, 2, 0>(p, smem);
        else if (ntw == 1) gru_chain_fwd_body<T, 1, 0>(p, smem);
        else gru_chain_fwd_body<T, 0, 0>(p, smem);
    }
}
template <class T, int DC>
__global__ __launch_bounds__(256) void gru_chain_bwd_kernel(ChainP p) {
    DPC_DYN_SMEM(smem);
    if constexpr (DC == 256) {
        gru_chain_bwd_body<T, 2, 256>(p, smem);
    } else {
        const int wave = threadIdx.x >> 6, ntiles = p.D / 32;
        const int ntw = wave < ntiles ? (wave + NWV < ntiles ? 2 : 1) : 0;
        if (ntw == 2) gru_chain_bwd_body<T, 2, 0>(p, smem);
        else if (ntw == 1) gru_chain_bwd_body<T, 1, 0>(p, smem);
        else gru_chain_bwd_body<T, 0, 0>(p, smem);
    }
}

// D = 256 on EIGHT waves (round 4): a wave owns one column tile instead of two, so the dependent phases of a step (products, staging,
// gate math, barrier) are half as long per wave and each SIMD has two waves to interleave; 64 workgroups stay 64 (the rows of a
// sequence are all the parallelism there is), but each CU now runs 8 waves instead of 4.
template <class T>
__global__ __launch_bounds__(512) void gru_chain_fwd8_kernel(ChainP p) {
    DPC_DYN_SMEM(smem);
    gru_chain_fwd_body<T, 1, 0>(p, smem);
}
template <class T>
__global__ __launch_bounds__(512) void gru_chain_bwd8_kernel(ChainP p) {
    DPC_DYN_SMEM(smem);
    gru_chain_bwd_body<T, 1, 0>(p, smem);
}
static inline bool chain_eight_waves(const dpc_gru_chain_desc* c) {
    const char* e = getenv("DPC_GRU_WAVES");   // read per call (A/B knob)
    return c->D == 256 && (!e || atoi(e) == 8);
}

int chain_params(const dpc_gru_chain_desc* c, ChainP* p, bool backward) {
    if (!c || c->M <= 0 || c->D <= 0 || c->SQ <= 0 || c->P < 0 || c->n_agg <= 0 || c->n_steps != c->n_agg + (c->P > 0 ? c->P - 1 : 0)) return DPC_ERR_ARG;
    if (c->D % 32 || c->D > 256 || c->M % c->SQ) return DPC_ERR_UNSUPPORTED;
    if (!c->packed || !c->bias_u || !c->bias_r || !c->bias_o || !c->bias_1 || !c->bias_2 || !c->X_all || !c->H_all || !c->HR_all ||
        !c->U_all || !c->R_all || !c->O_all || (c->P > 0 && (!c->P1_all || !c->pred)))
        return DPC_ERR_ARG;
    if (backward && (!c->G_all || !c->d_x || !c->ws || (c->P > 0 && (!c->d_pred || !c->dP1 || !c->dP2)))) return DPC_ERR_ARG;
    if (backward && c->P == 0 && !c->d_hlast) return DPC_ERR_ARG;  // nothing would flow into the recurrence
    if (!(c->p_drop >= 0.f) || !(c->p_drop < 1.f)) return DPC_ERR_ARG;
    p->M = c->M; p->D = c->D; p->SQ = c->SQ; p->P = c->P; p->n_agg = c->n_agg; p->n_steps = c->n_steps;
    p->packed = c->packed;
    p->bu = c->bias_u; p->br = c->bias_r; p->bo = c->bias_o; p->b1 = c->bias_1; p->b2 = c->bias_2;
    p->drop = c->drop_masks;
    p->step_dev = c->drop_masks ? nullptr : c->step_dev;
    p->seed = c->seed;
    p->thresh24 = dropout_thresh24(c->p_drop);
    p->inv_keep = 1.f / (1.f - c->p_drop);
    p->X_all = c->X_all; p->H_all = c->H_all; p->HR_all = c->HR_all;
    p->U_all = c->U_all; p->R_all = c->R_all; p->O_all = c->O_all;
    p->P1_all = c->P1_all; p->pred = c->pred;
    p->d_pred = c->d_pred; p->G_all = c->G_all; p->dP1 = c->dP1; p->dP2 = c->dP2; p->d_x = c->d_x; p->ws = c->ws; p->d_hlast = c->d_hlast;
    return DPC_OK;
}

// more than 64 KB of dynamic LDS has to be granted per kernel function (once)
template <class K> int allow_lds(K kernel, size_t bytes) {
#ifndef DPC_SIMT_EMU
    if (bytes > 160 * 1024) return DPC_ERR_UNSUPPORTED;
    if (hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) != hipSuccess) return DPC_ERR_LAUNCH;
#else
    (void)kernel; (void)bytes;
#endif
    return DPC_OK;
}

size_t chain_lds(const dpc_gru_chain_desc* c) {
    const int esz = c->dtype == DPC_BF16 ? 2 : 4;
    const size_t tile = (size_t)((c->D * esz + 127) / 128) * CHUNK;
    return 3 * tile + (size_t)TM * (c->D + SPAD) * sizeof(float);
}

}  // namespace

extern "C" int dpc_gru_pack(const float* w_update, const float* w_reset, const float* w_out, const float* w_pred0, const float* w_pred2,
                            int32_t D, int32_t dtype, void* packed, dpc_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!w_update || !w_reset || !w_out || !packed || D <= 0 || (!w_pred0) != (!w_pred2)) return DPC_ERR_ARG;  // w_pred*: NULL for P = 0
    if (D % 32 || D > 256) return DPC_ERR_UNSUPPORTED;
    PackSrc src;
    src.w[0] = w_update; src.w[1] = w_reset; src.w[2] = w_out; src.w[3] = w_pred0; src.w[4] = w_pred2;
    const int E = dtype == DPC_BF16 ? 8 : 4;
    long long blocks = ((long long)D * 2 * D / E + 255) / 256;
    if (blocks > 64) blocks = 64;
    if (dtype == DPC_F32) {
        DPC_LAUNCH((gru_pack_kernel<float>), dim3((unsigned)blocks, NMAT), dim3(256), stream, src, D, (float*)packed);
    } else if (dtype == DPC_BF16) {
        DPC_LAUNCH((gru_pack_kernel<bf16_t>), dim3((unsigned)blocks, NMAT), dim3(256), stream, src, D, (bf16_t*)packed);
    } else {
        return DPC_ERR_ARG;
    }
    return dpc_launch_status();
}

extern "C" int dpc_gru_chain_fwd(const dpc_gru_chain_desc* c, dpc_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    ChainP p;
    const int rc = chain_params(c, &p, false);
    if (rc) return rc;
    const unsigned grid = (unsigned)((c->M + TM - 1) / TM);
    const size_t lds = chain_lds(c);
#define DPC_CHAIN_GO(T_, DC_)                                                                  \
    do {                                                                                       \
        if (int e = allow_lds(gru_chain_fwd_kernel<T_, DC_>, lds)) return e;                     \
        DPC_LAUNCH_DYN((gru_chain_fwd_kernel<T_, DC_>), dim3(grid), dim3(256), lds, stream, p); \
    } while (0)
    if (chain_eight_waves(c) && (c->dtype == DPC_F32 || c->dtype == DPC_BF16)) {
        if (c->dtype == DPC_F32) {
            if (int e = allow_lds(gru_chain_fwd8_kernel<float>, lds)) return e;
            DPC_LAUNCH_DYN((gru_chain_fwd8_kernel<float>), dim3(grid), dim3(512), lds, stream, p);
        } else {
            if (int e = allow_lds(gru_chain_fwd8_kernel<bf16_t>, lds)) return e;
            DPC_LAUNCH_DYN((gru_chain_fwd8_kernel<bf16_t>), dim3(grid), dim3(512), lds, stream, p);
        }
    } else if (c->dtype == DPC_F32) {
        DPC_CHAIN_GO(float, 0);
    } else if (c->dtype == DPC_BF16) {
        DPC_CHAIN_GO(bf16_t, 0);
#undef DPC_CHAIN_GO
    } else {
        return DPC_ERR_ARG;
    }
    return dpc_launch_status();
}

extern "C" int dpc_gru_chain_bwd(const dpc_gru_chain_desc* c, dpc_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    ChainP p;
    const int rc = chain_params(c, &p, true);
    if (rc) return rc;
    const unsigned grid = (unsigned)((c->M + TM - 1) / TM);
    const size_t lds = chain_lds(c);
#define DPC_CHAIN_GO(T_, DC_)                                                                  \
    do {                                                                                       \
        if (int e = allow_lds(gru_chain_bwd_kernel<T_, DC_>, lds)) return e;                     \
        DPC_LAUNCH_DYN((gru_chain_bwd_kernel<T_, DC_>), dim3(grid), dim3(256), lds, stream, p); \
    } while (0)
    if (chain_eight_waves(c) && (c->dtype == DPC_F32 || c->dtype == DPC_BF16)) {
        if (c->dtype == DPC_F32) {
            if (int e = allow_lds(gru_chain_bwd8_kernel<float>, lds)) return e;
            DPC_LAUNCH_DYN((gru_chain_bwd8_kernel<float>), dim3(grid), dim3(512), lds, stream, p);
        } else {
            if (int e = allow_lds(gru_chain_bwd8_kernel<bf16_t>, lds)) return e;
            DPC_LAUNCH_DYN((gru_chain_bwd8_kernel<bf16_t>), dim3(grid), dim3(512), lds, stream, p);
        }
    } else if (c->dtype == DPC_F32) {
        DPC_CHAIN_GO(float, 0);
    } else if (c->dtype == DPC_BF16) {
        DPC_CHAIN_GO(bf16_t, 0);
#undef DPC_CHAIN_GO
    } else {
        return DPC_ERR_ARG;
    }
    return dpc_launch_status();
}
