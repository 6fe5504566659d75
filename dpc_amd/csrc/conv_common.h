// conv_common.h -- geometry shared by the implicit-GEMM forward/dgrad kernel and the
// weight-gradient kernel: how GEMM row m and GEMM k' map onto the channels-last source.
#pragma once
#include "dpc_rt.h"
#include <stdlib.h>
#include "../../include/dpc_hip.h"

struct GatherGeom {
    int M;                 // rows = N*RT*RH*RW
    int Kp;                // taps*Ci
    int mode;              // 0 forward gather, 1 input-gradient gather
    int RT, RH, RW;
    int ST, SH, SW;
    int Ci, src_ld, log2C, taps;
    int KT, KH, KW;
    int st, sh, sw, lst, lsh, lsw;  // strides and their log2
    int pt, ph, pw;
    FastDiv dRW, dRH, dRT, dKW, dKH;
};

struct RowPos {  // decoded GEMM row: source-batch base and the tap-independent coordinates
    int nbase, t0, h0, w0;
};

static inline int ilog2_exact(int v) {
    int l = 0;
    while ((1 << l) < v) ++l;
    return ((1 << l) == v) ? l : -1;
}

static inline int make_gather_geom(const dpc_conv_desc* d, GatherGeom* g) {
    if (!d || d->N <= 0 || d->Ci <= 0 || d->Co <= 0) return DPC_ERR_ARG;
    const int per16 = d->dtype_in == DPC_BF16 ? 8 : 4;
    if (d->dtype_in != DPC_F32 && d->dtype_in != DPC_BF16) return DPC_ERR_ARG;
    g->taps = d->KT * d->KH * d->KW;
    if (g->taps <= 0) return DPC_ERR_ARG;
    if (d->Ci % per16 || d->src_ld % per16) return DPC_ERR_UNSUPPORTED;
    g->log2C = ilog2_exact(d->Ci);
    if (g->taps > 1 && g->log2C < 0) return DPC_ERR_UNSUPPORTED;
    g->lst = ilog2_exact(d->st);
    g->lsh = ilog2_exact(d->sh);
    g->lsw = ilog2_exact(d->sw);
    if (g->lst < 0 || g->lsh < 0 || g->lsw < 0) return DPC_ERR_UNSUPPORTED;
    long long M = (long long)d->N * d->RT * d->RH * d->RW;
    if (M <= 0 || M >= (1ll << 31)) return DPC_ERR_ARG;
    if ((long long)d->N * d->ST * d->SH * d->SW >= (1ll << 31)) return DPC_ERR_ARG;
    g->M = (int)M;
    g->Kp = g->taps * d->Ci;
    g->mode = d->mode;
    g->RT = d->RT; g->RH = d->RH; g->RW = d->RW;
    g->ST = d->ST; g->SH = d->SH; g->SW = d->SW;
    g->Ci = d->Ci; g->src_ld = d->src_ld;
    g->KT = d->KT; g->KH = d->KH; g->KW = d->KW;
    g->st = d->st; g->sh = d->sh; g->sw = d->sw;
    g->pt = d->pt; g->ph = d->ph; g->pw = d->pw;
    g->dRW = make_fastdiv(d->RW);
    g->dRH = make_fastdiv(d->RH);
    g->dRT = make_fastdiv(d->RT);
    g->dKW = make_fastdiv(d->KW);
    g->dKH = make_fastdiv(d->KH);
    return DPC_OK;
}

__device__ __forceinline__ RowPos decode_row(const GatherGeom& g, int m) {
    RowPos r;
    if (m >= g.M) {
        r.nbase = 0; r.t0 = -(1 << 28); r.h0 = 0; r.w0 = 0;
        return r;
    }
    unsigned q1 = fdiv((unsigned)m, g.dRW);
    int rw = m - (int)q1 * g.RW;
    unsigned q2 = fdiv(q1, g.dRH);
    int rh = (int)q1 - (int)q2 * g.RH;
    unsigned n = fdiv(q2, g.dRT);
    int rt = (int)q2 - (int)n * g.RT;
    r.nbase = (int)n * g.ST;
    if (g.mode == 0) {
        r.t0 = rt * g.st - g.pt; r.h0 = rh * g.sh - g.ph; r.w0 = rw * g.sw - g.pw;
    } else {
        r.t0 = rt + g.pt; r.h0 = rh + g.ph; r.w0 = rw + g.pw;
    }
    return r;
}

struct TapPos {
    int kt, kh, kw, ci;
    bool ok;  // k' inside the reduction extent
};
__device__ __forceinline__ TapPos decode_k(const GatherGeom& g, int k) {
    TapPos t;
    t.ok = k < g.Kp;
    if (g.taps == 1) {
        t.kt = t.kh = t.kw = 0; t.ci = k;
        return t;
    }
    int tap = k >> g.log2C;
    t.ci = k - (tap << g.log2C);
    unsigned q = fdiv((unsigned)tap, g.dKW);
    t.kw = tap - (int)q * g.KW;
    unsigned kt = fdiv(q, g.dKH);
    t.kh = (int)q - (int)kt * g.KH;
    t.kt = (int)kt;
    return t;
}

// element offset of (row, tap, ci) in the source, or -1 when the tap falls into padding
__device__ __forceinline__ long long gather_off(const GatherGeom& g, const RowPos& r, const TapPos& t) {
    int ti, hi, wi;
    bool ok = t.ok;
    if (g.mode == 0) {
        ti = r.t0 + t.kt; hi = r.h0 + t.kh; wi = r.w0 + t.kw;
        ok = ok && ti >= 0 && ti < g.ST && hi >= 0 && hi < g.SH && wi >= 0 && wi < g.SW;
    } else {
        int a = r.t0 - t.kt, b = r.h0 - t.kh, c = r.w0 - t.kw;
        ok = ok && a >= 0 && b >= 0 && c >= 0 && ((a & (g.st - 1)) | (b & (g.sh - 1)) | (c & (g.sw - 1))) == 0;
        ti = a >> g.lst; hi = b >> g.lsh; wi = c >> g.lsw;
        ok = ok && ti < g.ST && hi < g.SH && wi < g.SW;
    }
    if (!ok) return -1;
    long long pos = ((long long)(r.nbase + ti) * g.SH + hi) * g.SW + wi;
    return pos * g.src_ld + t.ci;
}

// Branch-free on purpose: a conditional load makes hipcc branch around every load and wait
// vmcnt(0) per element (cdna_hip_programming.md §5 trap (c)), serialising the HBM round trips.
// Padding/out-of-range units read the (always valid) first 16 bytes and are masked to zero.
// The mask is applied by the CONSUMER (the LDS store after the MFMA block): masking here would
// put the s_waitcnt right behind the load and drain the prefetch before the matrix work.
__device__ __forceinline__ u32x4 load_unit_raw(const void* base, long long elem_off, int esize) {
    const long long o = elem_off >= 0 ? elem_off : 0;
    return *(const u32x4*)((const char*)base + o * esize);
}
__device__ __forceinline__ u32x4 mask_unit(u32x4 v, bool ok) {
    const uint32_t m = ok ? 0xffffffffu : 0u;
    v[0] &= m; v[1] &= m; v[2] &= m; v[3] &= m;
    return v;
}

// ---- strided input-gradient by parity classes (conv_igemm.hip GATHER 3, conv_igemm_ws.hip PAR) -------------------------------------
// Output positions with the same residues (t % st, h % sh, w % sw) see the same subset of taps, and inside a class the gather is
// affine again (source = class coordinate + dl(tap)).  M-tiles are laid out class by class; taps that never hit a class are skipped
// wholesale (7/8 of them for 3x3x3 stride 2) instead of being multiplied by zeros.
struct ParityInfo {
    int ncls, N;
    int tile_start[9];  // first tile of class c = (ct * sh + ch) * sw + cw; [ncls..8] = number of tiles
    int dimc[3][2];     // class extent per dim (t,h,w) and residue
    FastDiv div[3][2];
    int cnt[3][2];      // taps that hit the class, per dim and residue
    int kl[3][2][4];    // their tap indices
    int dl[3][2][4];    // their source offsets (rt + pt - kt) / st
    // tile order inside one temporal class: the nsp = sh * sw spatial classes cover the same (clip, tq, hq, wq) ranges tile by tile
    // and re-read the same source rows.  Class after class those re-reads come from HBM nsp times; interleaved in groups of 8
    // tiles (tile l of the temporal class -> spatial class (l / 8) % nsp, tile-in-class (l / (8 nsp)) * 8 + l % 8) the nsp tiles of a
    // region run at the same time on the same XCD (workgroup b sits on XCD b % 8) and share its L2.
    // A program walks tiles mt, mt + gm, ...: with gm a multiple of 8 nsp it would meet the same spatial class every round and the
    // programs of the 4-tap class would run 4x longer than those of the 1-tap class; the class is therefore rotated by the round
    // (block index >> rot_sh, 2^rot_sh = blocks per round), so that every program sees every class.
    int ilv, nsp, lnsp, rot_sh;
    int tpc[2];         // tiles per spatial class, per temporal residue
    int tstart_t[2];    // first tile of the temporal class
    // cnt / kl / dl of one (dim, residue) in one word: n | kl_j << (3 + 6 j) | (dl_j + 4) << (6 + 6 j) -- tap j of a class is then
    // shifts and masks on a scalar register instead of a table lookup; valid when pk_ok (k < 8, -4 <= dl < 4)
    unsigned pk[3][2];
    int pk_ok;
};
__host__ __device__ __forceinline__ int par_pk_n(unsigned pk) { return (int)(pk & 7u); }
__host__ __device__ __forceinline__ int par_pk_kl(unsigned pk, int j) { return (int)((pk >> (3 + 6 * j)) & 7u); }
__host__ __device__ __forceinline__ int par_pk_dl(unsigned pk, int j) { return (int)((pk >> (6 + 6 * j)) & 7u) - 4; }

static inline int dpc_floordiv(int a, int b) { return (a >= 0) ? a / b : -((-a + b - 1) / b); }

// fills `par` for tiles of bm rows; false when the shape is outside the scheme.  skip_empty: classes no tap reaches get no tiles
// (in-place accumulation: their positions keep their value).
static inline void dpc_parity_rounds(ParityInfo& par, int gm) {   // gm: programs that walk the m-tiles (tile mt -> program mt % gm)
    int bpr = gm >> (3 + par.lnsp), sh = 0;
    while ((2 << sh) <= bpr) ++sh;
    par.rot_sh = bpr >= 1 ? sh : 30;
}
static inline bool dpc_plan_parity(const GatherGeom& g, ParityInfo& par, int bke, int bm, bool skip_empty, int* ntm) {
    if (g.log2C < 0 || g.Ci < bke || g.Ci % bke) return false;
    const int K[3] = {g.KT, g.KH, g.KW}, S[3] = {g.st, g.sh, g.sw}, P[3] = {g.pt, g.ph, g.pw}, R[3] = {g.RT, g.RH, g.RW};
    for (int d = 0; d < 3; ++d) {
        if (S[d] > 2) return false;
        for (int r = 0; r < 2; ++r) {
            par.cnt[d][r] = 0;
            par.dimc[d][r] = r < S[d] ? (R[d] - r + S[d] - 1) / S[d] : 0;
            par.div[d][r] = make_fastdiv(par.dimc[d][r] > 0 ? par.dimc[d][r] : 1);
            for (int j = 0; j < 4; ++j) { par.kl[d][r][j] = 0; par.dl[d][r][j] = 0; }
            if (r >= S[d]) continue;
            for (int k = 0; k < K[d]; ++k) {
                const int x = r + P[d] - k;
                if (((x % S[d]) + S[d]) % S[d] != 0) continue;
                if (par.cnt[d][r] >= 4) return false;
                par.kl[d][r][par.cnt[d][r]] = k;
                par.dl[d][r][par.cnt[d][r]] = dpc_floordiv(x, S[d]);
                ++par.cnt[d][r];
            }
        }
    }
    par.pk_ok = 1;
    for (int d = 0; d < 3; ++d)
        for (int r = 0; r < 2; ++r) {
            unsigned w = (unsigned)par.cnt[d][r];
            for (int j = 0; j < par.cnt[d][r]; ++j) {
                if (par.kl[d][r][j] > 7 || par.dl[d][r][j] < -4 || par.dl[d][r][j] > 3) par.pk_ok = 0;
                w |= ((unsigned)par.kl[d][r][j] & 7u) << (3 + 6 * j);
                w |= ((unsigned)(par.dl[d][r][j] + 4) & 7u) << (6 + 6 * j);
            }
            par.pk[d][r] = w;
        }
    par.ncls = g.st * g.sh * g.sw;
    par.N = g.M / (g.RT * g.RH * g.RW);
    par.nsp = g.sh * g.sw;
    par.lnsp = g.lsh + g.lsw;
    par.rot_sh = 30;
    par.ilv = par.nsp > 1 ? 1 : 0;
    par.tpc[0] = par.tpc[1] = 0;
    int tiles = 0;
    for (int c = 0; c < par.ncls; ++c) {
        const int cw = c % g.sw, ch = (c / g.sw) % g.sh, ct = c / (g.sw * g.sh);
        par.tile_start[c] = tiles;
        if (par.cnt[0][ct] * par.cnt[1][ch] * par.cnt[2][cw] > 64) return false;
        long long rows = (long long)par.N * par.dimc[0][ct] * par.dimc[1][ch] * par.dimc[2][cw];
        if (skip_empty && par.cnt[0][ct] * par.cnt[1][ch] * par.cnt[2][cw] == 0) rows = 0;
        const int tc = (int)((rows + bm - 1) / bm);
        if (c % par.nsp == 0) par.tpc[ct] = tc;
        else if (tc != par.tpc[ct]) par.ilv = 0;   // unequal spatial classes (odd extents, skipped classes): class after class
        if (tc == 0) par.ilv = 0;
        tiles += tc;
    }
    for (int c = par.ncls; c < 9; ++c) par.tile_start[c] = tiles;
    par.tstart_t[0] = 0;
    par.tstart_t[1] = par.tile_start[par.nsp];
    *ntm = tiles;
    return true;
}

// tile index -> (class, tile inside the class).  Every table access below has a compile-time index: a dynamic index into the
// by-value kernel argument makes hipcc copy the whole struct to scratch in some kernels (712 bytes per lane in igemm_ws_kernel,
// and a scratch load waits for every LDS-DMA piece in flight).
__device__ __forceinline__ void parity_tile(const ParityInfo& par, int mt, int& c, int& j) {
    if (!par.ilv) {
        int start = 0;
        c = 0;
        DPC_UNROLL
        for (int i = 1; i < 8; ++i) {
            const bool in = i < par.ncls && mt >= par.tile_start[i];
            c += in ? 1 : 0;
            start = in ? par.tile_start[i] : start;
        }
        j = mt - start;
        return;
    }
    const int ct = (par.ncls > par.nsp && mt >= par.tstart_t[1]) ? 1 : 0;
    const int l = mt - (ct ? par.tstart_t[1] : 0);
    const int tpc = ct ? par.tpc[1] : par.tpc[0];
    const int full = tpc & ~7;
    int cs;
    if (l < full * par.nsp) {
        const int blk = l >> (3 + par.lnsp), r = l & ((8 << par.lnsp) - 1);
        cs = ((r >> 3) + (blk >> par.rot_sh)) & (par.nsp - 1);
        j = blk * 8 + (r & 7);
    } else {
        const int l2 = l - full * par.nsp, rem = tpc - full;
        cs = l2 / rem;
        j = full + (l2 - cs * rem);
    }
    c = ct * par.nsp + cs;
}
// ---- fused backward pieces of an input-gradient epilogue (dpc_conv_igemm_ex, include/dpc_hip.h) -----------------------------
// addend_mask: the residual addend is the BLOCK's incoming gradient gated by the ReLU sign mask of the block output
//   (out = conv + (bit ? addend : 0)): the masked gradient dz is never written as a tensor.
// bn_raw / bn_mask / bn_mean / bn_invstd: the BatchNorm-backward reduction of the unit whose output gradient this launch produces,
//   taken from the stored (rounded) outputs: dz = out (bit-gated when bn_mask), stats rows = (sum dz, sum dz * xhat).
// Masks are dpc_bn_apply's: one byte per 16-byte unit of a DENSE [rows][Co] tensor (ldo == Co is required).
struct EpiExtra {
    const uint8_t* addend_mask;
    const void* bn_raw;
    const uint8_t* bn_mask;
    const float* bn_mean;
    const float* bn_invstd;
};
static inline EpiExtra epi_none() {
    EpiExtra e = {nullptr, nullptr, nullptr, nullptr, nullptr};
    return e;
}
static inline bool epi_any(const EpiExtra& e) { return e.addend_mask || e.bn_raw; }

// one 16-byte output unit: o = conv (+ gated addend), then either the forward statistics (sum, sum of squares of the stored
// values) or the BatchNorm-backward partial sums.  mu / is: the unit's EPO channel statistics (registers of the caller).
template <class TO, int EPO>
__device__ __forceinline__ void epi_unit(u32x4& o, bool has_add, const u32x4& add, unsigned addbits, bool bnred, const u32x4& raw,
                                         unsigned bnbits, const float (&mu)[EPO], const float (&is)[EPO], float (&s1)[EPO], float (&s2)[EPO]) {
    if (has_add) {
        float sv[EPO];
        DPC_UNROLL
        for (int e = 0; e < EPO; ++e) sv[e] = unit_get<TO>(o, e) + (((addbits >> e) & 1u) ? unit_get<TO>(add, e) : 0.f);
        o = unit_pack<TO>(sv);
    }
    if (bnred) {
        DPC_UNROLL
        for (int e = 0; e < EPO; ++e) {
            const float dz = ((bnbits >> e) & 1u) ? unit_get<TO>(o, e) : 0.f;
            s1[e] += dz;
            s2[e] += dz * ((unit_get<TO>(raw, e) - mu[e]) * is[e]);
        }
    } else {
        DPC_UNROLL
        for (int e = 0; e < EPO; ++e) {
            const float v = unit_get<TO>(o, e);
            s1[e] += v;
            s2[e] += v * v;
        }
    }
}

// conv_halo.hip: LDS-staged-patch kernel for 1xKHxKW stride-1 same-size convs.  _try returns 1 when
// the shape is not served (the caller then runs the generic implicit-GEMM path); _rows returns the
// number of batch-norm partial rows that kernel would write (0 = not served).
int dpc_conv_halo_try(const dpc_conv_desc* d, const void* src, const void* wgt, void* out, const void* addend, float* stats,
                      const EpiExtra& epi, hipStream_t stream);
int dpc_conv_halo_rows(const dpc_conv_desc* d);

// conv_igemm_ws.hip: loader/compute wave-specialised implicit GEMM for Co >= 128 (bf16); same contract.
int dpc_conv_ws_try(const dpc_conv_desc* d, const void* src, const void* wgt, void* out, const void* addend, float* stats,
                    const EpiExtra& epi, hipStream_t stream);
int dpc_conv_ws_rows(const dpc_conv_desc* d);

// score_fused.hip: plain NT GEMM with bf16 operands, f32 output and a short reduction (the materialised contrastive score).
int dpc_score_gemm_try(const dpc_conv_desc* d, const void* src, const void* wgt, void* out, hipStream_t stream);
// gemm_ws.hip: split-K NT (trans_a = 0) / TN (trans_a = 1: A given K-major) GEMM on the loader / compute machinery; 1 = shape not served
int dpc_gemm_ws_try(int trans_a, int M, int N, int K, const void* A, int lda, const void* B, int ldb, float* part, int32_t* nsplit,
                    hipStream_t stream);

// conv_wgrad_patch.hip: weight gradient of 1x3x3 stride-1 convs from one staged source patch (bf16).
int dpc_wgrad_patch_try(const dpc_conv_desc* d, const void* src, const void* dy, int dy_ld, float* part, int32_t* nsplit,
                        hipStream_t stream);

// conv_wgrad_stem.hip: weight gradient of the space-to-depth stem conv (1x4x4 taps, 16 channels) from one staged patch.
int dpc_wgrad_stem_try(const dpc_conv_desc* d, const void* src, const void* dy, int dy_ld, float* part, int32_t* nsplit,
                       hipStream_t stream);
