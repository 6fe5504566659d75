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

__device__ __forceinline__ float sigm(float x) { return 1.f / (1.f + expf(-x)); }

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
};

constexpr int TM = 32;          // rows per workgroup
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
template <class T, int NB>
__device__ __forceinline__ void gemm_acc(f32x16 (&acc)[NB][2], int ntw, const unsigned char* tile, int ks_n, const u32x4* const (&Bp)[NB][2],
                                         int lane) {
    const int row = lane & 31, kg = lane >> 5;
    u32x4 ring[4][NB][2];
    DPC_UNROLL
    for (int j = 0; j < 4; ++j)
        if (j < ks_n) {
            DPC_UNROLL
            for (int b = 0; b < NB; ++b) {
                if (ntw > 0) ring[j][b][0] = Bp[b][0][j * 64 + lane];
                if (ntw > 1) ring[j][b][1] = Bp[b][1][j * 64 + lane];
            }
        }
    for (int ks0 = 0; ks0 < ks_n; ks0 += 4) {
        DPC_UNROLL
        for (int j = 0; j < 4; ++j) {
            const int ks = ks0 + j;
            if (ks < ks_n) {
                const int unit = ks * 2 + kg;
                const u32x4 a = *(const u32x4*)(tile + (unit >> 3) * CHUNK + lds_unit_off(row, unit & 7));
                DPC_UNROLL
                for (int b = 0; b < NB; ++b) {
                    if (ntw > 0) acc[b][0] = mfma_unit<T>(a, ring[j][b][0], acc[b][0]);
                    if (ntw > 1) acc[b][1] = mfma_unit<T>(a, ring[j][b][1], acc[b][1]);
                }
                if (ks + 4 < ks_n) {
                    DPC_UNROLL
                    for (int b = 0; b < NB; ++b) {
                        if (ntw > 0) ring[j][b][0] = Bp[b][0][(ks + 4) * 64 + lane];
                        if (ntw > 1) ring[j][b][1] = Bp[b][1][(ks + 4) * 64 + lane];
                    }
                }
            }
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
    constexpr int E = Elt<T>::PER16;
    return (const u32x4*)((const T*)p.packed + mat_off(mat, p.D)) + ((long long)tile * KS + ks_begin) * 64;
    (void)E;
}

// keep-mask bytes of step s for the workgroup's rows -> LDS (Philox mode); element (row, col) at [row * D + col]
__device__ __forceinline__ void gen_mask_tile(uint8_t* mt, const ChainP& p, int s, int m0) {
    const uint32_t step = (uint32_t)p.step_dev[0];
    const int D4 = p.D / 4;
    for (int q = threadIdx.x; q < TM * D4; q += blockDim.x) {
        const int row = q / D4, c4 = q - row * D4;
        float k[4] = {0.f, 0.f, 0.f, 0.f};
        if (m0 + row < p.M) {
            const long long idx = ((long long)s * p.M + (m0 + row)) * p.D + c4 * 4;  // element index in [n_steps][M][D], multiple of 4
            dropout_keep4(p.seed, step, (uint32_t)(idx >> 2), p.thresh24, 1.f, k);
        }
        DPC_UNROLL
        for (int e = 0; e < 4; ++e) mt[row * p.D + c4 * 4 + e] = k[e] != 0.f ? 1 : 0;
    }
}
__device__ __forceinline__ float mask_at(const ChainP& p, const uint8_t* mt, int s, int grow, int row, int col) {
    if (p.drop) return p.drop[((long long)s * p.M + grow) * p.D + col];
    if (p.step_dev) return mt[row * p.D + col] ? p.inv_keep : 0.f;
    return 1.f;
}

// ---------------------------------------------------------------- forward chain
template <class T>
__global__ __launch_bounds__(256) void gru_chain_fwd_kernel(ChainP p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int E = Elt<T>::PER16;
    const int D = p.D, M = p.M;
    const int tile_bytes = (D * (16 / E) + 127) / 128 * CHUNK;
    unsigned char* tx = smem;
    unsigned char* th = smem + tile_bytes;
    unsigned char* thr = smem + 2 * tile_bytes;
    uint8_t* mt = smem + 3 * tile_bytes;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int m0 = blockIdx.x * TM;
    const int ntiles = D / 32;
    const int ntw = wave < ntiles ? (wave + 4 < ntiles ? 2 : 1) : 0;
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
        if (!p.drop && p.step_dev) gen_mask_tile(mt, p, s, m0);
        __syncthreads();
        // ---- [u | r] = [x | h] @ [Wu ; Wr]^T, o_x = x @ Wo_x^T
        f32x16 aur[2][2], ao[1][2];
        zero_acc<2>(aur);
        zero_acc<1>(ao);
        {
            const u32x4* const Bx[3][2] = {{frag<T>(p, M_WU, wave, 2 * KSD, 0), frag<T>(p, M_WU, wave + 4, 2 * KSD, 0)},
                                           {frag<T>(p, M_WR, wave, 2 * KSD, 0), frag<T>(p, M_WR, wave + 4, 2 * KSD, 0)},
                                           {frag<T>(p, M_WOX, wave, KSD, 0), frag<T>(p, M_WOX, wave + 4, KSD, 0)}};
            f32x16 a3[3][2];
            zero_acc<3>(a3);
            gemm_acc<T, 3>(a3, ntw, tx, KSD, Bx, lane);
            DPC_UNROLL
            for (int t = 0; t < 2; ++t) { aur[0][t] = a3[0][t]; aur[1][t] = a3[1][t]; ao[0][t] = a3[2][t]; }
            const u32x4* const Bh[2][2] = {{frag<T>(p, M_WU, wave, 2 * KSD, KSD), frag<T>(p, M_WU, wave + 4, 2 * KSD, KSD)},
                                           {frag<T>(p, M_WR, wave, 2 * KSD, KSD), frag<T>(p, M_WR, wave + 4, 2 * KSD, KSD)}};
            gemm_acc<T, 2>(aur, ntw, th, KSD, Bh, lane);
        }
        // ---- gates 1: u, r, h*r
        float uu[2][16];
        for (int t = 0; t < ntw; ++t) {
            const int col = (wave + 4 * t) * 32 + (lane & 31);
            const float bu = p.bu[col], br = p.br[col];
            DPC_UNROLL
            for (int r = 0; r < 16; ++r) {
                const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                const float u = sigm(aur[0][t][r] + bu), rr = sigm(aur[1][t][r] + br);
                const float h = TileIO<T>::get(th, row, col);
                const T hr = Elt<T>::from_f32(h * rr);
                uu[t][r] = u;
                *(T*)(thr + TileIO<T>::elem_addr(row, col)) = hr;
                if (m0 + row < M) {
                    const long long o = (long long)s * MD + (long long)(m0 + row) * D + col;
                    p.U_all[o] = u;
                    p.R_all[o] = rr;
                    HR_all[o] = hr;
                }
            }
        }
        __syncthreads();
        // ---- o = tanh(o_x + (h*r) @ Wo_h^T + bo); h' = (h (1-u) + o u) * drop
        {
            const u32x4* const Bo[1][2] = {{frag<T>(p, M_WOH, wave, KSD, 0), frag<T>(p, M_WOH, wave + 4, KSD, 0)}};
            gemm_acc<T, 1>(ao, ntw, thr, KSD, Bo, lane);
        }
        for (int t = 0; t < ntw; ++t) {
            const int col = (wave + 4 * t) * 32 + (lane & 31);
            const float bo = p.bo[col];
            DPC_UNROLL
            for (int r = 0; r < 16; ++r) {
                const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                const int grow = m0 + row;
                const float o = tanhf(ao[0][t][r] + bo);
                const float h = TileIO<T>::get(th, row, col);
                const float u = uu[t][r];
                float hn = h * (1.f - u) + o * u;
                if (grow < M) hn *= mask_at(p, mt, s, grow, row, col);
                const T hq = Elt<T>::from_f32(hn);
                *(T*)(th + TileIO<T>::elem_addr(row, col)) = hq;
                if (grow < M) {
                    const long long oidx = (long long)s * MD + (long long)grow * D + col;
                    p.O_all[oidx] = o;
                    H_all[oidx + MD] = hq;
                }
            }
        }
        __syncthreads();
        // ---- prediction i from the state after step s (dpc/model_3d.py:66-69), feeding the next step
        const int i = s - (p.n_agg - 1);
        if (i >= 0 && i < p.P) {
            f32x16 a1[1][2];
            zero_acc<1>(a1);
            const u32x4* const B1[1][2] = {{frag<T>(p, M_W1, wave, KSD, 0), frag<T>(p, M_W1, wave + 4, KSD, 0)}};
            gemm_acc<T, 1>(a1, ntw, th, KSD, B1, lane);
            for (int t = 0; t < ntw; ++t) {
                const int col = (wave + 4 * t) * 32 + (lane & 31);
                const float b1 = p.b1[col];
                DPC_UNROLL
                for (int r = 0; r < 16; ++r) {
                    const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                    float v = a1[0][t][r] + b1;
                    v = v > 0.f ? v : 0.f;
                    const T q = Elt<T>::from_f32(v);
                    *(T*)(thr + TileIO<T>::elem_addr(row, col)) = q;
                    if (m0 + row < M) P1_all[(long long)i * MD + (long long)(m0 + row) * D + col] = q;
                }
            }
            __syncthreads();
            zero_acc<1>(a1);
            const u32x4* const B2[1][2] = {{frag<T>(p, M_W2, wave, KSD, 0), frag<T>(p, M_W2, wave + 4, KSD, 0)}};
            gemm_acc<T, 1>(a1, ntw, thr, KSD, B2, lane);
            for (int t = 0; t < ntw; ++t) {
                const int col = (wave + 4 * t) * 32 + (lane & 31);
                const float b2 = p.b2[col];
                DPC_UNROLL
                for (int r = 0; r < 16; ++r) {
                    const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                    const int grow = m0 + row;
                    const float v = a1[0][t][r] + b2;
                    const T xq = Elt<T>::from_f32(v > 0.f ? v : 0.f);
                    if (i < p.P - 1) *(T*)(tx + TileIO<T>::elem_addr(row, col)) = xq;
                    if (grow < M) {
                        const int b = grow / p.SQ, sq = grow - b * p.SQ;
                        pred[((long long)(b * p.P + i) * p.SQ + sq) * D + col] = Elt<T>::from_f32(v);
                        if (i < p.P - 1) X_all[(long long)(s + 1) * MD + (long long)grow * D + col] = xq;
                    }
                }
            }
            // the next iteration's barrier orders these LDS writes before the reads
        }
    }
}

// ---------------------------------------------------------------- backward chain
// One GRU step s, reversed (the arithmetic of the former dpc_gru_bwd1/2 kernels + their three GEMMs):
//   dhn = dh*drop; G_u = dhn (o-h) u (1-u); G_o = dhn u (1-o^2); dhprev = dhn (1-u)
//   dhr = G_o @ Wo_h;  G_r = dhr h r (1-r);  dhprev += dhr r
//   dx = [G_u|G_r|G_o] @ [Wu_x;Wr_x;Wo_x];  dh <- dhprev + [G_u|G_r] @ [Wu_h;Wr_h]
template <class T>
__device__ __forceinline__ void gru_step_bwd(const ChainP& p, int s, int m0, int ntw, int wave, int lane, unsigned char* tu, unsigned char* tr,
                                             unsigned char* to, uint8_t* mt, f32x16 (&dh)[1][2], f32x16 (&dx)[1][2]) {
    constexpr int E = Elt<T>::PER16;
    const int D = p.D, M = p.M;
    const int KSD = D / (2 * E);
    const long long MD = (long long)M * D;
    const T* H_all = (const T*)p.H_all;
    T* G_all = (T*)p.G_all;
    if (!p.drop && p.step_dev) {
        gen_mask_tile(mt, p, s, m0);
        __syncthreads();
    }
    float dhprev[2][16];
    for (int t = 0; t < ntw; ++t) {
        const int col = (wave + 4 * t) * 32 + (lane & 31);
        DPC_UNROLL
        for (int r = 0; r < 16; ++r) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            const int grow = m0 + row;
            float gu = 0.f, go = 0.f, dp = 0.f;
            if (grow < M) {
                const long long o = (long long)s * MD + (long long)grow * D + col;
                const float dhn = dh[0][t][r] * mask_at(p, mt, s, grow, row, col);
                const float u = p.U_all[o], oo = p.O_all[o], h = Elt<T>::to_f32(H_all[o]);
                gu = dhn * (oo - h) * u * (1.f - u);
                go = dhn * u * (1.f - oo * oo);
                dp = dhn * (1.f - u);
                G_all[((long long)s * M + grow) * 3 * D + col] = Elt<T>::from_f32(gu);
                G_all[((long long)s * M + grow) * 3 * D + 2 * D + col] = Elt<T>::from_f32(go);
            }
            dhprev[t][r] = dp;
            TileIO<T>::put(tu, row, col, gu);
            TileIO<T>::put(to, row, col, go);
        }
    }
    __syncthreads();
    f32x16 dhr[1][2];
    zero_acc<1>(dhr);
    {
        const u32x4* const B[1][2] = {{frag<T>(p, M_WOHT, wave, KSD, 0), frag<T>(p, M_WOHT, wave + 4, KSD, 0)}};
        gemm_acc<T, 1>(dhr, ntw, to, KSD, B, lane);
    }
    for (int t = 0; t < ntw; ++t) {
        const int col = (wave + 4 * t) * 32 + (lane & 31);
        DPC_UNROLL
        for (int r = 0; r < 16; ++r) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            const int grow = m0 + row;
            float gr = 0.f;
            if (grow < M) {
                const long long o = (long long)s * MD + (long long)grow * D + col;
                const float g = dhr[0][t][r], rr = p.R_all[o], h = Elt<T>::to_f32(H_all[o]);
                gr = g * h * rr * (1.f - rr);
                dhprev[t][r] += g * rr;
                G_all[((long long)s * M + grow) * 3 * D + D + col] = Elt<T>::from_f32(gr);
            }
            TileIO<T>::put(tr, row, col, gr);
        }
    }
    __syncthreads();
    zero_acc<1>(dx);
    DPC_UNROLL
    for (int t = 0; t < 2; ++t)
        DPC_UNROLL
        for (int r = 0; r < 16; ++r) dh[0][t][r] = dhprev[t][r];
    {
        const u32x4* const Bx0[1][2] = {{frag<T>(p, M_WUXT, wave, KSD, 0), frag<T>(p, M_WUXT, wave + 4, KSD, 0)}};
        const u32x4* const Bx1[1][2] = {{frag<T>(p, M_WRXT, wave, KSD, 0), frag<T>(p, M_WRXT, wave + 4, KSD, 0)}};
        const u32x4* const Bx2[1][2] = {{frag<T>(p, M_WOXT, wave, KSD, 0), frag<T>(p, M_WOXT, wave + 4, KSD, 0)}};
        const u32x4* const Bh0[1][2] = {{frag<T>(p, M_WUHT, wave, KSD, 0), frag<T>(p, M_WUHT, wave + 4, KSD, 0)}};
        const u32x4* const Bh1[1][2] = {{frag<T>(p, M_WRHT, wave, KSD, 0), frag<T>(p, M_WRHT, wave + 4, KSD, 0)}};
        gemm_acc<T, 1>(dx, ntw, tu, KSD, Bx0, lane);
        gemm_acc<T, 1>(dh, ntw, tu, KSD, Bh0, lane);
        gemm_acc<T, 1>(dx, ntw, tr, KSD, Bx1, lane);
        gemm_acc<T, 1>(dh, ntw, tr, KSD, Bh1, lane);
        gemm_acc<T, 1>(dx, ntw, to, KSD, Bx2, lane);
    }
    __syncthreads();  // the three tiles are free again
}

template <class T>
__global__ __launch_bounds__(256) void gru_chain_bwd_kernel(ChainP p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int E = Elt<T>::PER16;
    const int D = p.D, M = p.M;
    const int tile_bytes = (D * (16 / E) + 127) / 128 * CHUNK;
    unsigned char* tu = smem;
    unsigned char* tr = smem + tile_bytes;
    unsigned char* to = smem + 2 * tile_bytes;
    uint8_t* mt = smem + 3 * tile_bytes;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int m0 = blockIdx.x * TM;
    const int ntiles = D / 32;
    const int ntw = wave < ntiles ? (wave + 4 < ntiles ? 2 : 1) : 0;
    const int KSD = D / (2 * E);
    const long long MD = (long long)M * D;
    const T* X_all = (const T*)p.X_all;
    const T* P1_all = (const T*)p.P1_all;
    T* dP1 = (T*)p.dP1;
    T* dP2 = (T*)p.dP2;
    f32x16 dh[1][2], dx[1][2];
    zero_acc<1>(dh);
    zero_acc<1>(dx);
    int step = p.n_steps;
    for (int i = p.P - 1; i >= 0; --i) {
        // p_i = W2 relu(W1 h + b1) + b2 feeds the score (row-mapped) and, for i < P-1, GRU step `step` through relu
        const bool feeds = i < p.P - 1;
        if (feeds) {
            --step;
            gru_step_bwd<T>(p, step, m0, ntw, wave, lane, tu, tr, to, mt, dh, dx);
        }
        for (int t = 0; t < ntw; ++t) {
            const int col = (wave + 4 * t) * 32 + (lane & 31);
            DPC_UNROLL
            for (int r = 0; r < 16; ++r) {
                const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                const int grow = m0 + row;
                float g = 0.f;
                if (grow < M) {
                    const int b = grow / p.SQ, sq = grow - b * p.SQ;
                    g = p.d_pred[((long long)(b * p.P + i) * p.SQ + sq) * D + col];
                    if (feeds && Elt<T>::to_f32(X_all[(long long)step * MD + (long long)grow * D + col]) > 0.f) g += dx[0][t][r];
                    dP2[(long long)i * MD + (long long)grow * D + col] = Elt<T>::from_f32(g);
                }
                TileIO<T>::put(tu, row, col, g);
            }
        }
        __syncthreads();
        f32x16 g1[1][2];
        zero_acc<1>(g1);
        {
            const u32x4* const B[1][2] = {{frag<T>(p, M_W2T, wave, KSD, 0), frag<T>(p, M_W2T, wave + 4, KSD, 0)}};
            gemm_acc<T, 1>(g1, ntw, tu, KSD, B, lane);
        }
        for (int t = 0; t < ntw; ++t) {
            const int col = (wave + 4 * t) * 32 + (lane & 31);
            DPC_UNROLL
            for (int r = 0; r < 16; ++r) {
                const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                const int grow = m0 + row;
                float g = 0.f;
                if (grow < M) {
                    const long long o = (long long)i * MD + (long long)grow * D + col;
                    g = Elt<T>::to_f32(P1_all[o]) > 0.f ? g1[0][t][r] : 0.f;
                    dP1[o] = Elt<T>::from_f32(g);
                }
                TileIO<T>::put(tr, row, col, g);
            }
        }
        __syncthreads();
        {
            const u32x4* const B[1][2] = {{frag<T>(p, M_W1T, wave, KSD, 0), frag<T>(p, M_W1T, wave + 4, KSD, 0)}};
            gemm_acc<T, 1>(dh, ntw, tr, KSD, B, lane);  // dh += dP1 @ W1
        }
        __syncthreads();
    }
    for (int t = p.n_agg - 1; t >= 0; --t) {
        --step;
        gru_step_bwd<T>(p, step, m0, ntw, wave, lane, tu, tr, to, mt, dh, dx);
        for (int tt = 0; tt < ntw; ++tt) {
            const int col = (wave + 4 * tt) * 32 + (lane & 31);
            DPC_UNROLL
            for (int r = 0; r < 16; ++r) {
                const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (m0 + row < M) p.d_x[(long long)t * MD + (long long)(m0 + row) * D + col] = dx[0][tt][r];
            }
        }
    }
}

int chain_params(const dpc_gru_chain_desc* c, ChainP* p, bool backward) {
    if (!c || c->M <= 0 || c->D <= 0 || c->SQ <= 0 || c->P <= 0 || c->n_agg <= 0 || c->n_steps != c->n_agg + c->P - 1) return DPC_ERR_ARG;
    if (c->D % 32 || c->D > 256 || c->M % c->SQ) return DPC_ERR_UNSUPPORTED;
    if (!c->packed || !c->bias_u || !c->bias_r || !c->bias_o || !c->bias_1 || !c->bias_2 || !c->X_all || !c->H_all || !c->HR_all ||
        !c->U_all || !c->R_all || !c->O_all || !c->P1_all || !c->pred)
        return DPC_ERR_ARG;
    if (backward && (!c->d_pred || !c->G_all || !c->dP1 || !c->dP2 || !c->d_x)) return DPC_ERR_ARG;
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
    p->d_pred = c->d_pred; p->G_all = c->G_all; p->dP1 = c->dP1; p->dP2 = c->dP2; p->d_x = c->d_x;
    return DPC_OK;
}

size_t chain_lds(const dpc_gru_chain_desc* c) {
    const int esz = c->dtype == DPC_BF16 ? 2 : 4;
    const size_t tile = (size_t)((c->D * esz + 127) / 128) * CHUNK;
    return 3 * tile + (size_t)TM * c->D;
}

}  // namespace

#ifdef DPC_SIMT_EMU
#define DPC_LAUNCH_DYN(kernel, grid, block, lds, stream, ...) simt::launch_dyn((grid), (block), (lds), [=]() { (kernel)(__VA_ARGS__); })
#else
#define DPC_LAUNCH_DYN(kernel, grid, block, lds, stream, ...) hipLaunchKernelGGL(kernel, (grid), (block), (lds), (stream), __VA_ARGS__)
#endif

extern "C" int64_t dpc_gru_packed_elems(int32_t D) { return 16ll * D * D; }

extern "C" int dpc_gru_pack(const float* w_update, const float* w_reset, const float* w_out, const float* w_pred0, const float* w_pred2,
                            int32_t D, int32_t dtype, void* packed, dpc_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!w_update || !w_reset || !w_out || !w_pred0 || !w_pred2 || !packed || D <= 0) return DPC_ERR_ARG;
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
    if (c->dtype == DPC_F32) {
        DPC_LAUNCH_DYN((gru_chain_fwd_kernel<float>), dim3(grid), dim3(256), lds, stream, p);
    } else if (c->dtype == DPC_BF16) {
        DPC_LAUNCH_DYN((gru_chain_fwd_kernel<bf16_t>), dim3(grid), dim3(256), lds, stream, p);
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
    if (c->dtype == DPC_F32) {
        DPC_LAUNCH_DYN((gru_chain_bwd_kernel<float>), dim3(grid), dim3(256), lds, stream, p);
    } else if (c->dtype == DPC_BF16) {
        DPC_LAUNCH_DYN((gru_chain_bwd_kernel<bf16_t>), dim3(grid), dim3(256), lds, stream, p);
    } else {
        return DPC_ERR_ARG;
    }
    return dpc_launch_status();
}
