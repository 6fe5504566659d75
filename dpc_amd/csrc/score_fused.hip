// score_fused.hip -- the dense contrastive score of DPC_RNN.forward (dpc/model_3d.py:79-84) fused with the
// CrossEntropyLoss / top-1/3/5 of the train loop (dpc/main.py:178-185,213-218, utils/utils.py:38-55) and with its
// own backward, flash-attention style: the [R x R] score (R = B*P*SQ: 6 144 at cfg2, 15 680 at cfg5 = 983 MB in
// f32) and its gradient are NEVER written to HBM in the train step.
//
//   forward   S = pred @ finf^T tile by tile on the matrix cores; per row the running (max, sum-exp, #{s > s_target})
//             stay in registers; a finalize pass turns them into the log-sum-exp, the loss and the three accuracies.
//   backward  d_pred = dS @ finf and d_finf = dS^T @ pred with dS = (softmax(S) - onehot)/R: every S tile is recomputed
//             from its operands (K = 256: cheap), turned into dS in registers, rounded to bf16 through a wave-private
//             LDS tile and multiplied straight into the output accumulators.  One kernel serves both products with the
//             roles of the operands swapped (the owner rows accumulate, the other side is streamed).
//
// Why: with an f32 score in HBM the forward contraction is output-write bound (AI = 128 FLOP/B: 250 TFLOP/s in round 1,
// 10 % of the bf16 MFMA peak) and the loss re-reads it three times.  Throughput (bf16) mode
// only: the f32 parity mode and the module boundary (which RETURNS the score) keep the materialised path.
//
// Tiling: a workgroup = 4 waves owns 128 rows (32 per wave, operand fragments held in registers for all of K) and a
// contiguous range of 64-column tiles of the other operand, which arrive in LDS by LDS-DMA, double-buffered; the grid is
// (row blocks) x (column splits) so that >= 240 workgroups are resident for 48 row blocks, and the per-split partial
// statistics / f32 output slabs are merged by small follow-up kernels.
#include "dpc_rt.h"
#include "../../include/dpc_hip.h"
#include <stdlib.h>

namespace {

constexpr float L2E = 1.4426950408889634f;
constexpr float NEG_BIG = -3.0e38f;
constexpr int BM = 128, BN = 64;

struct ScoreP {
    const bf16_t* own;    // [R][D] rows that stay in registers
    const bf16_t* oth;    // [R][D] rows streamed through LDS
    const bf16_t* othT;   // [D][ldT] the same operand transposed (backward only)
    int R, D, ldT;
    int ntiles, tiles_per_split, nsplit;
    // forward
    float* diag;          // [R]     target logit s[i][i]
    float* partial;       // [nsplit][R][4]  (max, sum-exp, rank, -)
    float* score;         // optional [R][R] f32 (NULL in the train step)
    // backward
    const float* lse2;    // [R] (lse + ln R) * log2(e)
    int lse_by_owner;     // 1: indexed by the owner row (d_pred), 0: by the streamed column (d_finf)
    float inv_rows;
    float* out_part;      // [nsplit][R][D] f32
};

__device__ __forceinline__ int crow(int r, int lane) { return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }

// rows [row0, row0 + nrows) x K of a row-major [R][ld] bf16 matrix -> LDS as 128-byte rows, chunk by chunk:
// tile[chunk c][row][8 swizzled units].  One LDS-DMA instruction of a wave fills 8 rows of one chunk; rows / columns
// outside the matrix read 16 zero bytes.
__device__ __forceinline__ void dma_rows(unsigned char* tile, const bf16_t* src, long long ld, int row0, int nrows, int row_limit, int col0,
                                         int ncols, int col_limit, int wave, int lane) {
    const char* const zero = (const char*)dpc_zero16;
    const int chunks = (ncols * 2 + 127) / 128;
    const int groups = nrows / 8;
    for (int g = wave; g < chunks * groups; g += 4) {
        const int c = g / groups, rg = g - c * groups;
        const int row = rg * 8 + (lane >> 3);
        const int u = (lane & 7) ^ lds_swz1(row);
        const int col = col0 + (c * 8 + u) * 8;
        const bool ok = (row0 + row < row_limit) && (col < col_limit) && ((c * 8 + u) * 8 < ncols);
        const char* a = (const char*)(src + (long long)(row0 + row) * ld + col);
        glds16(ok ? a : zero, tile + c * (nrows * 128) + rg * 8 * 128, lane);
    }
}

template <int KS>
__device__ __forceinline__ void load_own_ld(u32x4 (&own)[KS], const bf16_t* base, long long ld, int nrows, int r0, int lane) {
    const int row = r0 + (lane & 31), kg = lane >> 5;
    DPC_UNROLL
    for (int ks = 0; ks < KS; ++ks) {
        u32x4 v = {0u, 0u, 0u, 0u};
        if (row < nrows) v = *(const u32x4*)(base + (long long)row * ld + (ks * 2 + kg) * 8);
        own[ks] = v;
    }
}
template <int KS>
__device__ __forceinline__ void load_own(u32x4 (&own)[KS], const ScoreP& p, int r0, int lane) {
    load_own_ld<KS>(own, p.own, p.D, p.R, r0, lane);
}

// S[32 rows of the wave][64 columns of the tile] from the register-resident own fragments and the LDS tile.
// The B fragments of K step ks + PF are requested before the MFMAs of step ks are issued: hipcc otherwise emits
// ds_read -> s_waitcnt lgkmcnt(0) -> v_mfma per step and every MFMA eats a full LDS round trip (measured: 37 % of
// the wave's cycles parked in s_waitcnt).
// SWAP: the streamed operand goes first, so a lane holds ONE row of the wave (lane & 31) and quads of four consecutive streamed
// columns (8 (r >> 2) + 4 (lane >> 5) + (r & 3)): what a 16-byte row store wants (score_gemm_kernel).
template <int KS, bool SWAP = false>
__device__ __forceinline__ void s_tile(f32x16 (&s)[2], const u32x4 (&own)[KS], const unsigned char* tile, int lane) {
    constexpr int PF = KS >= 4 ? 3 : (KS - 1 > 0 ? KS - 1 : 1);  // K steps in flight ahead of the MFMAs
    const int j = lane & 31, kg = lane >> 5;
    DPC_UNROLL
    for (int t = 0; t < 2; ++t)
        DPC_UNROLL
        for (int r = 0; r < 16; ++r) s[t][r] = 0.f;
    u32x4 b[KS][2];
    // hand-issued reads (invisible to hipcc's wait-count pass, which would re-serialise them); in order, counted below
    auto req = [&](auto ks_) {
        constexpr int ks = decltype(ks_)::value;
        const int unit = ks * 2 + kg;
        const unsigned char* a0 = tile + (unit >> 3) * (BN * 128) + lds_unit_off(j, unit & 7);
        lds_read_b128_async(b[ks][0], a0);
        lds_read_b128_async_off<32 * 128>(b[ks][1], a0);  // row j + 32: same swizzle (swz1(r + 32) == swz1(r))
    };
    static_for<(PF < KS ? PF : KS)>([&](auto ks_) { req(ks_); });
    static_for<KS>([&](auto ks_) {
        constexpr int ks = decltype(ks_)::value;
        if constexpr (ks + PF < KS) req(std::integral_constant<int, ks + PF>{});
        constexpr int later = (KS - 1 - ks) < PF ? (KS - 1 - ks) : PF;  // K steps requested after this one
        lds_wait_tie_n(2 * later, b[ks][0], b[ks][1]);
        DPC_UNROLL
        for (int t = 0; t < 2; ++t) s[t] = SWAP ? mfma_32x32x16_bf16(b[ks][t], own[ks], s[t]) : mfma_32x32x16_bf16(own[ks], b[ks][t], s[t]);
    });
}

// ---------------------------------------------------------------- forward
// statistics of one 32 x 64 tile for the wave's rows.  EDGE: the tile reaches past column R or contains target columns
// (wave-uniform; the common interior tile takes the branch-free form: per row 2 logits -> one max3, one rescale, two
// fma + v_exp_f32 + add, two compare-and-count -- everything else would make the kernel VALU-bound, the matrix cores
// need 1 024 cycles per tile and wave)
template <bool EDGE>
__device__ __forceinline__ void tile_stats(const f32x16 (&s)[2], float (&m)[16], float (&l)[16], float (&rk)[16], const float (&dg)[16], int col0,
                                           int r0, int R, int lane) {
    const int c0 = col0 + (lane & 31), c1 = c0 + 32;
    DPC_UNROLL
    for (int r = 0; r < 16; ++r) {
        float v0 = s[0][r], v1 = s[1][r];
        bool a0 = v0 > dg[r], a1 = v1 > dg[r];
        if (EDGE) {
            const int grow = r0 + crow(r, lane);
            if (c0 >= R) { v0 = NEG_BIG; a0 = false; }
            if (c1 >= R) { v1 = NEG_BIG; a1 = false; }
            a0 = a0 && c0 != grow;  // the target column never counts (its recomputed value may differ from dg in the last bit)
            a1 = a1 && c1 != grow;
        }
        rk[r] += (a0 ? 1.f : 0.f) + (a1 ? 1.f : 0.f);
        const float mx = fmaxf(fmaxf(v0, v1), m[r]);
        const float nm = -mx * L2E;
        // (m - mx) first: with the -1e30 start value an fma against the ROUNDED product nm leaves a residual of ~1e22 -> 0 * inf
        l[r] = l[r] * fast_exp2((m[r] - mx) * L2E) + fast_exp2(fmaf(v0, L2E, nm)) + fast_exp2(fmaf(v1, L2E, nm));
        m[r] = mx;
    }
}

template <int KS, bool SCORE>
__global__ __launch_bounds__(256, 2) void score_fwd_kernel(ScoreP p) {  // 2 workgroups per CU: one's MFMAs cover the other's exp/compare VALU work
    DPC_DYN_SMEM(smem);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int rb = blockIdx.x, split = blockIdx.y;
    const int r0 = rb * BM + wave * 32;
    const int tile_bytes = ((KS * 32 + 127) / 128) * BN * 128;  // 64 rows x D*2 bytes in 128-byte chunks
    u32x4 own[KS];
    load_own<KS>(own, p, r0, lane);
    // target logit of the wave's rows: dot(pred[i], finf[i]); lane (i, kg) holds half of row i's products
    float dg[16];
    {
        const int row = r0 + (lane & 31), kg = lane >> 5;
        float acc = 0.f;
        if (row < p.R) {
            DPC_UNROLL
            for (int ks = 0; ks < KS; ++ks) {
                const u32x4 f = *(const u32x4*)(p.oth + (long long)row * p.D + (ks * 2 + kg) * 8);
                DPC_UNROLL
                for (int e = 0; e < 8; ++e) acc += unit_get<bf16_t>(own[ks], e) * unit_get<bf16_t>(f, e);
            }
        }
        acc += __shfl_xor(acc, 32);
        if (split == 0 && lane < 32 && row < p.R) p.diag[row] = acc;
        DPC_UNROLL
        for (int r = 0; r < 16; ++r) dg[r] = __shfl(acc, crow(r, lane));
    }
    float m[16], l[16], rk[16];
    DPC_UNROLL
    for (int r = 0; r < 16; ++r) { m[r] = -1.0e30f; l[r] = 0.f; rk[r] = 0.f; }
    const int jt0 = split * p.tiles_per_split;
    int jt1 = jt0 + p.tiles_per_split;
    if (jt1 > p.ntiles) jt1 = p.ntiles;
    if (jt0 < jt1) dma_rows(smem, p.oth, p.D, jt0 * BN, BN, p.R, 0, p.D, p.D, wave, lane);
    for (int jt = jt0; jt < jt1; ++jt) {
        const int buf = (jt - jt0) & 1;
        wait_vmcnt<0>();
        __syncthreads();  // tile jt has landed; every wave is done with the buffer the next DMA overwrites
        if (jt + 1 < jt1) dma_rows(smem + (buf ^ 1) * tile_bytes, p.oth, p.D, (jt + 1) * BN, BN, p.R, 0, p.D, p.D, wave, lane);
        f32x16 s[2];
        s_tile<KS>(s, own, smem + buf * tile_bytes, lane);
        const int col0 = jt * BN;
        if (SCORE) {
            DPC_UNROLL
            for (int t = 0; t < 2; ++t) {
                const int c = col0 + t * 32 + (lane & 31);
                DPC_UNROLL
                for (int r = 0; r < 16; ++r) {
                    const int grow = r0 + crow(r, lane);
                    if (c < p.R && grow < p.R) p.score[(long long)grow * p.R + c] = s[t][r];
                }
            }
        }
        const bool edge = col0 + BN > p.R || (col0 < r0 + 32 && col0 + BN > r0);  // wave-uniform
        if (edge) tile_stats<true>(s, m, l, rk, dg, col0, r0, p.R, lane);
        else tile_stats<false>(s, m, l, rk, dg, col0, r0, p.R, lane);
    }
    // merge the 32 lanes that share each row (xor butterflies stay inside a half-wave)
    DPC_UNROLL
    for (int r = 0; r < 16; ++r) {
        DPC_UNROLL
        for (int msk = 1; msk <= 16; msk <<= 1) {
            const float om = __shfl_xor(m[r], msk), ol = __shfl_xor(l[r], msk), ork = __shfl_xor(rk[r], msk);
            const float mx = om > m[r] ? om : m[r];
            l[r] = l[r] * fast_exp2((m[r] - mx) * L2E) + ol * fast_exp2((om - mx) * L2E);
            m[r] = mx;
            rk[r] += ork;
        }
        const int grow = r0 + crow(r, lane);
        if ((lane & 31) == 0 && grow < p.R) {
            float* o = p.partial + ((long long)split * p.R + grow) * 4;
            o[0] = m[r]; o[1] = l[r]; o[2] = rk[r]; o[3] = 0.f;
        }
    }
}

// per row: merge the column splits -> lse, loss term, rank; lse2 for the backward
__global__ void score_finalize_kernel(const float* partial, const float* diag, int R, int nsplit, float* row_ws, float* lse2) {
    const int row = blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= R) return;
    float M = NEG_BIG;
    for (int s = 0; s < nsplit; ++s) {
        const float v = partial[((long long)s * R + row) * 4];
        M = v > M ? v : M;
    }
    float L = 0.f, rk = 0.f;
    for (int s = 0; s < nsplit; ++s) {
        const float* q = partial + ((long long)s * R + row) * 4;
        L += q[1] * exp2f((q[0] - M) * L2E);  // (q[0] = -1e30 for a split that saw no column: term is 0)
        rk += q[2];
    }
    const float lse = M + logf(L);
    row_ws[2 * row + 0] = lse - diag[row];
    row_ws[2 * row + 1] = rk;
    lse2[row] = (lse + logf((float)R)) * L2E;
}

// ---------------------------------------------------------------- backward
// out[owner row][:] = sum over streamed columns c of dS(owner, c) * oth[c][:]      (dS in bf16, f32 accumulate)
template <int KS>
__global__ __launch_bounds__(256) void score_bwd_kernel(ScoreP p) {
    DPC_DYN_SMEM(smem);
    constexpr int NTD = KS / 2;  // 32-column tiles of D
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int rb = blockIdx.x, split = blockIdx.y;
    const int r0 = rb * BM + wave * 32;
    const int tile_bytes = ((KS * 32 + 127) / 128) * BN * 128;   // oth tile: 64 rows x D*2 bytes in 128-byte chunks
    const int tileT_bytes = KS * 16 * 128; // othT tile: D rows x 128 bytes
    const int stage_bytes = tile_bytes + tileT_bytes;
    unsigned char* ptile = smem + 2 * stage_bytes + wave * (32 * 128);  // wave-private dS tile: 32 rows x 64 bf16
    u32x4 own[KS];
    load_own<KS>(own, p, r0, lane);
    f32x16 out[NTD];
    DPC_UNROLL
    for (int n = 0; n < NTD; ++n)
        DPC_UNROLL
        for (int r = 0; r < 16; ++r) out[n][r] = 0.f;
    float lo[16];
    DPC_UNROLL
    for (int r = 0; r < 16; ++r) {
        const int grow = r0 + crow(r, lane);
        lo[r] = (p.lse_by_owner && grow < p.R) ? p.lse2[grow] : 0.f;
    }
    const int jt0 = split * p.tiles_per_split;
    int jt1 = jt0 + p.tiles_per_split;
    if (jt1 > p.ntiles) jt1 = p.ntiles;
    if (jt0 < jt1) {
        dma_rows(smem, p.oth, p.D, jt0 * BN, BN, p.R, 0, p.D, p.D, wave, lane);
        dma_rows(smem + tile_bytes, p.othT, p.ldT, 0, KS * 16, KS * 16, jt0 * BN, BN, p.ldT, wave, lane);
    }
    for (int jt = jt0; jt < jt1; ++jt) {
        const int buf = (jt - jt0) & 1;
        wait_vmcnt<0>();
        __syncthreads();
        if (jt + 1 < jt1) {
            unsigned char* nb = smem + (buf ^ 1) * stage_bytes;
            dma_rows(nb, p.oth, p.D, (jt + 1) * BN, BN, p.R, 0, p.D, p.D, wave, lane);
            dma_rows(nb + tile_bytes, p.othT, p.ldT, 0, KS * 16, KS * 16, (jt + 1) * BN, BN, p.ldT, wave, lane);
        }
        const unsigned char* tile = smem + buf * stage_bytes;
        const unsigned char* tileT = tile + tile_bytes;
        f32x16 s[2];
        s_tile<KS>(s, own, tile, lane);
        const int col0 = jt * BN;
        const bool edge = col0 + BN > p.R || (col0 < r0 + 32 && col0 + BN > r0) || r0 + 32 > p.R;  // wave-uniform
        DPC_UNROLL
        for (int t = 0; t < 2; ++t) {
            const int cl = t * 32 + (lane & 31), c = col0 + cl;
            const bool cv = c < p.R;
            const float lc = (!p.lse_by_owner && cv) ? p.lse2[c] : 0.f;
            unsigned char* pcol = ptile + (cl & 7) * 2;
            if (!edge) {  // interior tile: fma + v_exp_f32 + convert + store per logit
                DPC_UNROLL
                for (int r = 0; r < 16; ++r) {
                    const float g = fast_exp2(fmaf(s[t][r], L2E, -(p.lse_by_owner ? lo[r] : lc)));
                    *(bf16_t*)(pcol + lds_unit_off(crow(r, lane), cl >> 3)) = f32_to_bf16(g);
                }
            } else {
                DPC_UNROLL
                for (int r = 0; r < 16; ++r) {
                    const int row = crow(r, lane);
                    float g = cv ? fast_exp2(fmaf(s[t][r], L2E, -(p.lse_by_owner ? lo[r] : lc))) : 0.f;
                    if (c == r0 + row) g -= p.inv_rows;
                    if (r0 + row >= p.R) g = 0.f;
                    *(bf16_t*)(pcol + lds_unit_off(row, cl >> 3)) = f32_to_bf16(g);
                }
            }
        }
        wave_lds_fence();
        {   // out[n] += dS[32 x 64] @ othT[n][64]^T : 4 K steps x (1 A + NTD B) fragments; the reads of step k2 + 1 are in
            // flight while the MFMAs of step k2 run (hand-issued, counted: see s_tile)
            const int i = lane & 31, kg = lane >> 5;
            u32x4 a[4], b[4][NTD];
            auto req = [&](auto k2_) {
                constexpr int k2 = decltype(k2_)::value;
                const int unit = k2 * 2 + kg;
                lds_read_b128_async(a[k2], ptile + lds_unit_off(i, unit));
                const unsigned char* b0 = tileT + lds_unit_off(i, unit);
                static_for<NTD>([&](auto n_) {
                    constexpr int n = decltype(n_)::value;
                    lds_read_b128_async_off<n * 32 * 128>(b[k2][n], b0);  // rows n*32 + i: same swizzle every 32 rows
                });
            };
            req(std::integral_constant<int, 0>{});
            static_for<4>([&](auto k2_) {
                constexpr int k2 = decltype(k2_)::value;
                if constexpr (k2 + 1 < 4) req(std::integral_constant<int, k2 + 1>{});
                constexpr int later = k2 + 1 < 4 ? 1 + NTD : 0;
                static_for<NTD>([&](auto n_) {
                    constexpr int n = decltype(n_)::value;
                    lds_wait_tie<later>(a[k2], b[k2][n]);
                    out[n] = mfma_32x32x16_bf16(a[k2], b[k2][n], out[n]);
                });
            });
        }
        wave_lds_fence();  // the wave's next dS stores must not pass these reads
    }
    float* o = p.out_part + (long long)split * p.R * p.D;
    DPC_UNROLL
    for (int n = 0; n < NTD; ++n)
        DPC_UNROLL
        for (int r = 0; r < 16; ++r) {
            const int grow = r0 + crow(r, lane);
            if (grow < p.R) o[(long long)grow * p.D + n * 32 + (lane & 31)] = out[n][r];
        }
}


// ---------------------------------------------------------------- plain NT GEMM with a short reduction (K = D <= 256)
// out[M][N] (f32) = a[M][D] @ b[N][D]^T and nothing else: the MATERIALISED score (dpc/model_3d.py:83 -- what the module boundary
// returns, the f32 parity mode's loss reads, and the bf16 train step below R = 8 192 uses).  The generic implicit-GEMM kernel ran
// this 6 144 x 6 144 x 256 product in 89 us (217 TFLOP/s): four 64-wide K chunks with a barrier each, the f32 tile staged through
// LDS, 151 MB written at 1.7 TB/s.  Same structure as score_fwd_kernel instead: a wave's 32 rows stay in registers for all of K,
// 64-column tiles of b stream through LDS (double-buffered LDS-DMA), two workgroups per CU so that one's stores overlap the
// other's MFMAs; the MFMAs run with the operands swapped, so a lane owns one output row and quads of consecutive columns and the
// tile leaves as 16-byte row stores straight from the accumulators (no LDS staging, a quarter of the store instructions).
struct GemmKP {
    const bf16_t* a;
    const bf16_t* b;
    float* out;
    int M, N, D, lda, ldb, ldo;
    int ntiles, tiles_per_split, nsplit;
    int vec;  // 16-byte stores: ldo % 4 == 0 and out 16-byte aligned
};

// (The transposed-accumulator store form of this kernel -- 8 sixteen-byte stores per tile and wave instead of 32 four-byte ones, each
// touching 64 cache lines with 32 bytes -- and round 3's loop without the counted wait measured slower in rounds 3 / 4 and are gone;
// score_gemm2_kernel below is where the transposed accumulators pay: staged through LDS they leave as whole rows.)
template <int KS>
__global__ __launch_bounds__(256, 2) void score_gemm_kernel(GemmKP p) {
    DPC_DYN_SMEM(smem);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int rb = blockIdx.x, split = blockIdx.y;
    const int r0 = rb * BM + wave * 32;
    const int tile_bytes = ((KS * 32 + 127) / 128) * BN * 128;
    u32x4 own[KS];
    load_own_ld<KS>(own, p.a, p.lda, p.M, r0, lane);
    const int jt0 = split * p.tiles_per_split;
    int jt1 = jt0 + p.tiles_per_split;
    if (jt1 > p.ntiles) jt1 = p.ntiles;
    if (jt0 < jt1) dma_rows(smem, p.b, p.ldb, jt0 * BN, BN, p.N, 0, p.D, p.D, wave, lane);
    // counted wait: a wave whose 32 rows and whose column tile lie inside the matrix issues exactly 32 stores per tile, all younger
    // than the tile's DMA pieces -- let those stay in flight instead of draining them every tile (vmcnt counts stores too); edge
    // waves / edge tiles keep the full drain
    const bool rows_inside = r0 + 32 <= p.M;
    for (int jt = jt0; jt < jt1; ++jt) {
        const int buf = (jt - jt0) & 1;
        const bool prev_full = rows_inside && jt > jt0 && jt * BN <= p.N;   // the previous tile (jt - 1) ended at or before column N
        if (prev_full) wait_vmcnt<32>(); else wait_vmcnt<0>();
        barrier_lds_only();  // tile jt has landed; every wave is done with the buffer the next DMA overwrites
        if (jt + 1 < jt1) dma_rows(smem + (buf ^ 1) * tile_bytes, p.b, p.ldb, (jt + 1) * BN, BN, p.N, 0, p.D, p.D, wave, lane);
        f32x16 s[2];
        s_tile<KS, false>(s, own, smem + buf * tile_bytes, lane);
        // lanes = 32 consecutive columns of one row (x 2 rows): every store instruction writes two full 128-byte lines
        DPC_UNROLL
        for (int t = 0; t < 2; ++t) {
            const int c = jt * BN + t * 32 + (lane & 31);
            DPC_UNROLL
            for (int r = 0; r < 16; ++r) {
                const int grow = r0 + crow(r, lane);
                if (c < p.N && grow < p.M) p.out[(long long)grow * p.ldo + c] = s[t][r];
            }
        }
        vm_note_stores(32);   // what the counted wait of the next tile leaves in flight (simulator bookkeeping; nothing on the device)
    }
}

// Round 4 form of the same product (default when the output rows are 16-byte addressable): the 151 MB / 983 MB of f32 score are
// what bounds it, so everything is arranged around the stores.
//   * a workgroup is 8 waves = 256 rows sharing the streamed tile (half the LDS-DMA traffic per output of the 4-wave form);
//   * a wave's 32 x 64 tile (SWAP layout: a lane owns one row and quads of consecutive columns) is staged through 8.5 KB of its own
//     LDS and leaves as eight 16-byte-per-lane NON-TEMPORAL stores of four whole 256-byte rows each -- the 4-wave form issued 32
//     four-byte stores per tile (two 128-byte lines each), 3.6 TB/s;
//   * no wait ever drains the stores: the loop's one wait is a COUNTED vmcnt that retires the tile's LDS-DMA pieces (older than the
//     previous tile's eight stores, which stay in flight: "CDNA4 vmcnt counts stores too", cdna_hip_programming.md), and the
//     barrier is the LDS-only one -- __syncthreads() is vmcnt(0) while a DMA is pending.
constexpr int SG2_ROWB = BN * 4 + 16;          // staging row: 256 B + 16 (a ds_write_b128 lane group then covers all 64 banks)
constexpr int SG2_WAVE = 32 * SG2_ROWB;        // 8 704 B per wave
__device__ __forceinline__ void dma_rows8(unsigned char* tile, const bf16_t* src, long long ld, int row0, int row_limit, int ncols, int wave, int lane) {
    // dma_rows for BN rows and 8 waves: chunks x (BN / 8) pieces, pieces wave, wave + 8, ... (the same count for every wave)
    const char* const zero = (const char*)dpc_zero16;
    const int chunks = (ncols * 2 + 127) / 128;
    constexpr int groups = BN / 8;
    for (int g = wave; g < chunks * groups; g += 8) {
        const int c = g / groups, rg = g - c * groups;
        const int row = rg * 8 + (lane >> 3);
        const int u = (lane & 7) ^ lds_swz1(row);
        const int col = (c * 8 + u) * 8;
        const bool ok = (row0 + row < row_limit) && (col < ncols);
        const char* a = (const char*)(src + (long long)(row0 + row) * ld + col);
        glds16(ok ? a : zero, tile + c * (BN * 128) + rg * 8 * 128, lane);
    }
}

// OUT16 (round 6): the same tile leaves as bf16 -- the logits of a train step that never returns its score (75 MB instead of 151 MB
// at R = 6 144): a staged row is 64 columns x 2 bytes = one 128-byte line, eight rows per store instruction, four instructions per tile.
template <int KS, bool OUT16 = false>
__global__ __launch_bounds__(512, 2) void score_gemm2_kernel(GemmKP p) {
    DPC_DYN_SMEM(smem);
    const int lane = threadIdx.x & 63;
#ifdef DPC_SIMT_EMU
    const int wave = threadIdx.x >> 6;
#else
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
#endif
    constexpr int CH = (KS * 32 + 127) / 128;             // 128-byte K chunks of a streamed row
    constexpr int TILE = CH * BN * 128;
    constexpr int PIECES = (CH * (BN / 8) + 7) / 8;       // LDS-DMA instructions per wave and tile (4 at K = 256, 1 at K = 32)
    static_assert(CH * (BN / 8) % 8 == 0 || CH * (BN / 8) < 8, "every wave issues the same number of pieces (or at most one)");
    const int rb = blockIdx.x, split = blockIdx.y;
    const int r0 = rb * 256 + wave * 32;
    unsigned char* const stg = smem + 2 * TILE + wave * SG2_WAVE;
    u32x4 own[KS];
    load_own_ld<KS>(own, p.a, p.lda, p.M, r0, lane);
    const int jt0 = split * p.tiles_per_split;
    int jt1 = jt0 + p.tiles_per_split;
    if (jt1 > p.ntiles) jt1 = p.ntiles;
    if (jt0 < jt1) dma_rows8(smem, p.b, p.ldb, jt0 * BN, p.N, p.D, wave, lane);
    const int l31 = lane & 31, lhi = lane >> 5;
    constexpr int UPRW = OUT16 ? 8 : 16, RPI = 64 / UPRW, NIT = 32 / RPI;   // 16-byte units per staged row, rows per store instruction, instructions per tile
    constexpr int ROWB = OUT16 ? BN * 2 + 16 : SG2_ROWB;
    const int su = lane & (UPRW - 1), sr = lane / UPRW;   // store phase: 16-byte unit of the row, row inside a group of RPI
    // store instructions this wave issues per tile: a group of rows whose first row is outside the matrix is skipped as a
    // whole (exec = 0), so the count the wait below leaves in flight is per wave (NIT except in the last row block)
    int nst = 0;
    DPC_UNROLL
    for (int it = 0; it < NIT; ++it) nst += (r0 + RPI * it < p.M) ? 1 : 0;
    for (int jt = jt0; jt < jt1; ++jt) {
        const int buf = (jt - jt0) & 1;
        // the pieces of tile jt are older than the nst stores of tile jt - 1: let at most those stay in flight (if the compiler issues
        // a fully masked store anyway, the wait is merely stricter)
        if (jt == jt0) wait_vmcnt<0>(); else wait_vmcnt_upto(nst);
        barrier_lds_only();  // tile jt has landed; every wave is done with the buffer the next DMA overwrites
        if (jt + 1 < jt1) dma_rows8(smem + (buf ^ 1) * TILE, p.b, p.ldb, (jt + 1) * BN, p.N, p.D, wave, lane);
        f32x16 s[2];
        s_tile<KS, true>(s, own, smem + buf * TILE, lane);
        DPC_UNROLL
        for (int t = 0; t < 2; ++t)
            DPC_UNROLL
            for (int k = 0; k < 4; ++k) {
                if constexpr (OUT16) {
                    const u32x2 v = {bf16x2_pack(s[t][4 * k], s[t][4 * k + 1]), bf16x2_pack(s[t][4 * k + 2], s[t][4 * k + 3])};
                    *(u32x2*)(stg + l31 * ROWB + (t * 32 + 8 * k + 4 * lhi) * 2) = v;
                } else {
                    const f32x4 v = {s[t][4 * k], s[t][4 * k + 1], s[t][4 * k + 2], s[t][4 * k + 3]};
                    *(f32x4*)(stg + l31 * ROWB + (t * 32 + 8 * k + 4 * lhi) * 4) = v;
                }
            }
        wave_lds_fence();
        f32x4 ov[NIT];
        DPC_UNROLL
        for (int it = 0; it < NIT; ++it) ov[it] = *(const f32x4*)(stg + (RPI * it + sr) * ROWB + su * 16);
        wave_lds_fence();  // the next tile's staging writes must not pass these reads
        const int c = jt * BN + su * (OUT16 ? 8 : 4);
        DPC_UNROLL
        for (int it = 0; it < NIT; ++it) {
            const int row = r0 + RPI * it + sr;
            if (row < p.M && c < p.N) {   // N % 4 == 0 (vec); OUT16: N % 8 == 0
                if constexpr (OUT16) __builtin_nontemporal_store(ov[it], (f32x4*)((bf16_t*)p.out + (long long)row * p.ldo + c));
                else __builtin_nontemporal_store(ov[it], (f32x4*)(p.out + (long long)row * p.ldo + c));
            }
        }
        vm_note_stores(nst);   // simulator bookkeeping for wait_vmcnt_upto(nst) above; nothing on the device
    }
}

template <class K> int allow_lds(K kernel, size_t bytes) {
#ifndef DPC_SIMT_EMU
    if (bytes > 160 * 1024) return DPC_ERR_UNSUPPORTED;
    if (dpc_tls_plan_only) return DPC_OK;  // the plan query launches nothing (and must work without a device)
    if (hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) != hipSuccess) return DPC_ERR_LAUNCH;
#else
    (void)kernel; (void)bytes;
#endif
    return DPC_OK;
}

// column splits so that (row blocks x splits) fills the chip, each split keeping at least 4 tiles
void plan_splits(int R, int target_wgs, int* ntiles, int* tps, int* nsplit) {
    const int nrb = (R + BM - 1) / BM;
    *ntiles = (R + BN - 1) / BN;
    int s = (target_wgs + nrb - 1) / nrb;
    if (s < 1) s = 1;
    const int max_s = *ntiles / 4 > 0 ? *ntiles / 4 : 1;
    if (s > max_s) s = max_s;
    if (s > 16) s = 16;
    *tps = (*ntiles + s - 1) / s;
    *nsplit = (*ntiles + *tps - 1) / *tps;
}

}  // namespace

// dpc_conv_igemm hands plain NT GEMMs with bf16 operands, f32 output and K = 256 (or 32: the width-reduced test networks) to
// score_gemm_kernel when the output is large; returns 1 when the shape is not served (the generic kernel runs it)
int dpc_score_gemm_try(const dpc_conv_desc* d, const void* src, const void* wgt, void* out, hipStream_t stream) {
    if (d->dtype_in != DPC_BF16 || (d->dtype_out != DPC_F32 && d->dtype_out != DPC_BF16) || d->mode != 0) return 1;
    const bool out16 = d->dtype_out == DPC_BF16;   // bf16 logits: the staged 8-wave form only, whole 128-byte lines (N, ldo multiples of 64)
    if (out16 && (d->Co % 64 || d->ldo % 64 || ((uintptr_t)out % 16))) return 1;
    if (d->KT * d->KH * d->KW != 1 || d->RT * d->RH * d->RW != 1 || d->ST * d->SH * d->SW != 1) return 1;
    if (d->st != 1 || d->sh != 1 || d->sw != 1 || d->pt || d->ph || d->pw) return 1;
    if (d->Ci != 256 && d->Ci != 32) return 1;
    static const int on = getenv("DPC_SCORE_GEMM") ? atoi(getenv("DPC_SCORE_GEMM")) : 1;
    if (!on || (long long)d->N * d->Co < (1ll << 20) || d->N < 512 || d->Co < 512) return 1;  // small products: launch-bound either way
    if (d->src_ld % 8 || d->ldw % 8 || ((uintptr_t)src % 16) || ((uintptr_t)wgt % 16)) return 1;
    GemmKP p = {};
    p.a = (const bf16_t*)src; p.b = (const bf16_t*)wgt; p.out = (float*)out;
    p.M = d->N; p.N = d->Co; p.D = d->Ci; p.lda = d->src_ld; p.ldb = d->ldw; p.ldo = d->ldo;
    p.vec = (d->ldo % 4 == 0 && ((uintptr_t)out % 16) == 0) ? 1 : 0;
    p.ntiles = (p.N + BN - 1) / BN;
    const int v2 = getenv("DPC_SCORE_GEMM2") ? atoi(getenv("DPC_SCORE_GEMM2")) : 1;   // 0: never, 1: by size (default), 2: whenever the shape allows (tests)
    // the 8-wave form (staged whole-row non-temporal stores, counted waits) wins where the stores dominate -- R = 15 680: 319 -> 169 us
    // = 5.8 TB/s of score writes -- and loses its fixed cost on ten tiles per workgroup (R = 6 144: 44.8 against 42.1 us); rows that
    // are not cache-line multiples (R = 6 468) make a non-temporal store a partial-line write: 85 against 45 us.  Hence: large
    // outputs with 128-byte-aligned rows only (profiles/r04_head_kernels.txt).
    if (out16 || (v2 && p.vec && p.N % 4 == 0 && (v2 > 1 || ((long long)p.M * p.N >= (1ll << 26) && p.ldo % 32 == 0)))) {
        const int nrb2 = (p.M + 255) / 256;
        int sp2 = dpc_persistent_grid(256) / nrb2;   // one workgroup per CU (132 KB of LDS) and ONE wave of workgroups: rounding up put
                                                      // 264 of them on 256 CUs at R = 6 144 -- 50.8 us against 42 for the 4-wave form
        if (sp2 > p.ntiles / 4) sp2 = p.ntiles / 4 > 0 ? p.ntiles / 4 : 1;
        if (sp2 < 1) sp2 = 1;
        p.tiles_per_split = (p.ntiles + sp2 - 1) / sp2;
        p.nsplit = (p.ntiles + p.tiles_per_split - 1) / p.tiles_per_split;
        const dim3 grid2(nrb2, p.nsplit);
        const size_t lds2 = 2 * (size_t)((p.D * 2 + 127) / 128) * BN * 128 + 8 * (size_t)SG2_WAVE;
        if (out16) {
            if (p.D == 256) {
                if (int e = allow_lds(score_gemm2_kernel<16, true>, lds2)) return e;
                DPC_LAUNCH_DYN((score_gemm2_kernel<16, true>), grid2, dim3(512), lds2, stream, p);
            } else {
                if (int e = allow_lds(score_gemm2_kernel<2, true>, lds2)) return e;
                DPC_LAUNCH_DYN((score_gemm2_kernel<2, true>), grid2, dim3(512), lds2, stream, p);
            }
        } else if (p.D == 256) {
            if (int e = allow_lds(score_gemm2_kernel<16>, lds2)) return e;
            DPC_LAUNCH_DYN((score_gemm2_kernel<16>), grid2, dim3(512), lds2, stream, p);
        } else {
            if (int e = allow_lds(score_gemm2_kernel<2>, lds2)) return e;
            DPC_LAUNCH_DYN((score_gemm2_kernel<2>), grid2, dim3(512), lds2, stream, p);
        }
        return dpc_launch_status();
    }
    const int nrb = (p.M + BM - 1) / BM;
    // two workgroups per CU = 512 slots, and ONE wave of workgroups: rounded DOWN.  Rounds 3-4 rounded up -- 528 workgroups at
    // R = 6 144, 561 at R = 6 468 -- and the last 16 / 49 ran after everybody else (DPC_SCORE_GEMM_WGS: the slot count to fill, A/B)
    static const int slots = getenv("DPC_SCORE_GEMM_WGS") ? atoi(getenv("DPC_SCORE_GEMM_WGS")) : 512;
    int sp = slots / nrb;
    if (sp > p.ntiles / 4) sp = p.ntiles / 4 > 0 ? p.ntiles / 4 : 1;
    if (sp < 1) sp = 1;
    p.tiles_per_split = (p.ntiles + sp - 1) / sp;
    p.nsplit = (p.ntiles + p.tiles_per_split - 1) / p.tiles_per_split;
    const dim3 grid(nrb, p.nsplit);
    const size_t lds = 2 * (size_t)((p.D * 2 + 127) / 128) * BN * 128;
    if (p.D == 256) {
        if (int e = allow_lds(score_gemm_kernel<16>, lds)) return e;
        DPC_LAUNCH_DYN((score_gemm_kernel<16>), grid, dim3(256), lds, stream, p);
    } else {
        if (int e = allow_lds(score_gemm_kernel<2>, lds)) return e;
        DPC_LAUNCH_DYN((score_gemm_kernel<2>), grid, dim3(256), lds, stream, p);
    }
    return dpc_launch_status();
}

// workspace query: floats needed for `ws` of dpc_score_fwd / dpc_score_bwd (max of both)
extern "C" int dpc_score_ws_floats(int32_t R, int32_t D, int64_t* fwd_floats, int64_t* bwd_floats) {
    if (R <= 0 || D <= 0 || !fwd_floats || !bwd_floats) return DPC_ERR_ARG;
    if (D != 256 && D != 32) return DPC_ERR_UNSUPPORTED;
    int ntiles, tps, nsf, nsb;
    plan_splits(R, 512, &ntiles, &tps, &nsf);  // forward: two workgroups per CU
    plan_splits(R, 256, &ntiles, &tps, &nsb);
    *fwd_floats = (int64_t)nsf * R * 4;
    *bwd_floats = (int64_t)nsb * R * D;
    return nsb;
}

extern "C" int dpc_score_fwd(const void* pred, const void* finf, int32_t R, int32_t D, float* diag, float* lse2, float* row_ws,
                             float* score, float* ws, dpc_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!pred || !finf || R <= 0 || !diag || !lse2 || !row_ws || !ws) return DPC_ERR_ARG;
    if (D != 256 && D != 32) return DPC_ERR_UNSUPPORTED;
    ScoreP p = {};
    p.own = (const bf16_t*)pred; p.oth = (const bf16_t*)finf; p.R = R; p.D = D;
    plan_splits(R, 512, &p.ntiles, &p.tiles_per_split, &p.nsplit);
    p.diag = diag; p.partial = ws; p.score = score;
    const dim3 grid((R + BM - 1) / BM, p.nsplit);
    const size_t chunk_tile = (size_t)((D * 2 + 127) / 128) * BN * 128;
    const size_t lds = 2 * chunk_tile;
    if (D == 256) {
        if (int e = allow_lds(score_fwd_kernel<16, false>, lds)) return e;
        if (int e = allow_lds(score_fwd_kernel<16, true>, lds)) return e;
        if (score) { DPC_LAUNCH_DYN((score_fwd_kernel<16, true>), grid, dim3(256), lds, stream, p); } else { DPC_LAUNCH_DYN((score_fwd_kernel<16, false>), grid, dim3(256), lds, stream, p); }
    } else {
        if (int e = allow_lds(score_fwd_kernel<2, false>, lds)) return e;
        if (int e = allow_lds(score_fwd_kernel<2, true>, lds)) return e;
        if (score) { DPC_LAUNCH_DYN((score_fwd_kernel<2, true>), grid, dim3(256), lds, stream, p); } else { DPC_LAUNCH_DYN((score_fwd_kernel<2, false>), grid, dim3(256), lds, stream, p); }
    }
    DPC_LAUNCH(score_finalize_kernel, dim3((R + 255) / 256), dim3(256), stream, (const float*)ws, (const float*)diag, R, p.nsplit, row_ws, lse2);
    return dpc_launch_status();
}

// One of the two backward products: out_part[split][r][:] partial sums of  sum_c dS(r, c) oth[c][:]   (by_owner = 1: rows of
// `own` index the softmax rows -> d_pred with own = pred, oth = finf;  by_owner = 0: transposed roles -> d_finf with
// own = finf, oth = pred).  othT = oth transposed [D][ldT] (zero beyond column R).  Returns the number of slabs to sum.
extern "C" int dpc_score_bwd(const void* own, const void* oth, const void* othT, int32_t ldT, int32_t R, int32_t D, const float* lse2,
                             int32_t by_owner, float* out_part, dpc_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!own || !oth || !othT || R <= 0 || ldT < R || ldT % 8 || !lse2 || !out_part) return DPC_ERR_ARG;
    if (D != 256 && D != 32) return DPC_ERR_UNSUPPORTED;
    ScoreP p = {};
    p.own = (const bf16_t*)own; p.oth = (const bf16_t*)oth; p.othT = (const bf16_t*)othT; p.R = R; p.D = D; p.ldT = ldT;
    plan_splits(R, 256, &p.ntiles, &p.tiles_per_split, &p.nsplit);
    p.lse2 = lse2; p.lse_by_owner = by_owner; p.inv_rows = 1.f / (float)R; p.out_part = out_part;
    const dim3 grid((R + BM - 1) / BM, p.nsplit);
    const size_t chunk_tile = (size_t)((D * 2 + 127) / 128) * BN * 128;
    const size_t lds = 2 * (chunk_tile + (size_t)D * 128) + 4 * 32 * 128;
    if (D == 256) {
        if (int e = allow_lds(score_bwd_kernel<16>, lds)) return e;
        DPC_LAUNCH_DYN((score_bwd_kernel<16>), grid, dim3(256), lds, stream, p);
    } else {
        if (int e = allow_lds(score_bwd_kernel<2>, lds)) return e;
        DPC_LAUNCH_DYN((score_bwd_kernel<2>), grid, dim3(256), lds, stream, p);
    }
    const int rc = dpc_launch_status();
    return rc ? rc : p.nsplit;
}
