// conv_wgrad.hip -- weight gradients of every conv / 1x1 conv / matmul on the path, on the
// gfx950 matrix cores (autograd of backbone/resnet_2d3d.py:14-32,211,241-244,
// backbone/convrnn.py:13-15, dpc/model_3d.py:36-40,83).
//
//   part[ks][co][k'] = sum_{m in split ks} dy[m][co] * src[gather(m, tap(k'))][ci(k')]
//
// GEMM view: rows = co, cols = k' = tap*Ci+ci, reduction = output positions m (millions):
// split-K over workgroups, f32 partial slabs, reduced in fixed order by dpc_reduce_unpack
// (deterministic; no atomics).  Both operands arrive position-major ([m][channel], channels
// contiguous) while MFMA wants the reduction index contiguous per lane, so each thread loads
// an E x E (positions x channels, E = elements per 16 B) block with 16-byte loads, transposes
// it in registers (free for f32, one pack per dword for bf16) and writes E 16-byte rows into
// the swizzled LDS tile [channel][position].  The MFMA loop is the same as conv_igemm.hip.
#include "conv_common.h"
#include <stdlib.h>

struct WgradParams {
    GatherGeom g;
    const void* src;
    const void* dy;
    float* part;
    int Co, dy_ld;
    int nks, kcps;  // K-splits, position-chunks per split
    int ntm, ntn;
    int x6;         // f32 operands: contractions as bf16x6 (dpc_rt.h)
};

template <class T> struct Transposer;
template <> struct Transposer<float> {
    // v[e] = 4 channels of position e  ->  o[c] = 4 positions of channel c
    static __device__ __forceinline__ void run(const u32x4 (&v)[4], u32x4 (&o)[4]) {
        DPC_UNROLL
        for (int c = 0; c < 4; ++c) {
            u32x4 t = {v[0][c], v[1][c], v[2][c], v[3][c]};
            o[c] = t;
        }
    }
};
template <> struct Transposer<bf16_t> {
    // v[e] = 8 channels (4 dwords) of position e -> o[c] = 8 positions (4 dwords) of channel c
    static __device__ __forceinline__ void run(const u32x4 (&v)[8], u32x4 (&o)[8]) {
        DPC_UNROLL
        for (int c = 0; c < 8; ++c) {
            u32x4 t;
            DPC_UNROLL
            for (int q = 0; q < 4; ++q) {
                const uint32_t a = v[2 * q][c >> 1], b = v[2 * q + 1][c >> 1];
                t[q] = (c & 1) ? ((b & 0xffff0000u) | (a >> 16)) : ((b << 16) | (a & 0xffffu));
            }
            o[c] = t;
        }
    }
};

struct RowIt {
    int n, rt, rh, rw;
};
__device__ __forceinline__ RowIt rowit_decode(const GatherGeom& g, int m) {
    RowIt r;
    unsigned q1 = fdiv((unsigned)m, g.dRW);
    r.rw = m - (int)q1 * g.RW;
    unsigned q2 = fdiv(q1, g.dRH);
    r.rh = (int)q1 - (int)q2 * g.RH;
    unsigned n = fdiv(q2, g.dRT);
    r.rt = (int)q2 - (int)n * g.RT;
    r.n = (int)n;
    return r;
}
__device__ __forceinline__ void rowit_next(const GatherGeom& g, RowIt& r) {
    if (++r.rw == g.RW) {
        r.rw = 0;
        if (++r.rh == g.RH) {
            r.rh = 0;
            if (++r.rt == g.RT) { r.rt = 0; ++r.n; }
        }
    }
}
__device__ __forceinline__ RowPos rowit_pos(const GatherGeom& g, const RowIt& r) {
    RowPos p;
    p.nbase = r.n * g.ST;
    p.t0 = r.rt * g.st - g.pt; p.h0 = r.rh * g.sh - g.ph; p.w0 = r.rw * g.sw - g.pw;
    return p;
}

// ROWFAST: RW % E == 0, so the E consecutive positions of a unit share (n, t, h) and differ only in w:
// one row decode + one affine offset per unit instead of E coordinate->offset recomputations (the
// first version spent 22 VALU instructions per MFMA, rocprofv3 SQ_INSTS_VALU / SQ_INSTS_MFMA).
template <class T, int TM, int TN, bool ROWFAST>
__global__ __launch_bounds__(256) void wgrad_kernel(WgradParams p) {
    constexpr int E = Elt<T>::PER16;
    constexpr int BKP = 8 * E;            // positions per chunk (128-byte LDS rows)
    constexpr int MI = TM / 64, NT = TN / 64;
    constexpr int UA = (TM / E) * 8, UB = (TN / E) * 8;  // E x E units in the A / B chunk
    constexpr int NU = (UA + UB + 255) / 256;
    __shared__ __attribute__((aligned(16))) unsigned char lds[2 * (TM + TN) * 128];

    const GatherGeom& g = p.g;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wv = tid >> 6;
    const int wm = wv >> 1, wn = wv & 1;
    const int l31 = lane & 31, lhi = lane >> 5;
    int id = blockIdx.x;
    const int tile_n = id % p.ntn; id /= p.ntn;
    const int tile_m = id % p.ntm;
    const int ks = id / p.ntm;
    const int esz = (int)sizeof(T);

    const int nchunks_total = (g.M + BKP - 1) / BKP;
    const int c_begin = ks * p.kcps;
    const int c_end = (c_begin + p.kcps < nchunks_total) ? c_begin + p.kcps : nchunks_total;

    // per-unit constants: which operand, channel group, position group, decoded tap
    int u_cu[NU], u_pg[NU];
    bool u_isA[NU], u_on[NU];
    TapPos u_tp[NU];
    DPC_UNROLL
    for (int i = 0; i < NU; ++i) {
        const int uid = tid + 256 * i;
        u_on[i] = uid < UA + UB;
        u_isA[i] = uid < UA;
        const int loc = u_isA[i] ? uid : uid - UA;
        const int per_row = u_isA[i] ? (TM / E) : (TN / E);
        u_cu[i] = loc % per_row;
        u_pg[i] = loc / per_row;
        u_tp[i] = decode_k(g, tile_n * TN + u_cu[i] * E);
    }

    f32x16 acc[MI][NT];
    DPC_UNROLL
    for (int i = 0; i < MI; ++i)
        DPC_UNROLL
        for (int j = 0; j < NT; ++j)
            DPC_UNROLL
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const char* const zero = (const char*)dpc_zero16;
    u32x4 rv[NU][E];
    auto load_chunk = [&](int chunk) {
        DPC_UNROLL
        for (int i = 0; i < NU; ++i) {
            const int m_first = chunk * BKP + u_pg[i] * E;
            if (!u_on[i]) continue;
            if (u_isA[i]) {
                const int co = tile_m * TM + u_cu[i] * E;
                const bool cok = co + E <= p.dy_ld && co < p.Co;
                const char* a = (const char*)p.dy + ((long long)m_first * p.dy_ld + co) * esz;
                DPC_UNROLL
                for (int e = 0; e < E; ++e) {
                    const bool ok = cok && m_first + e < g.M;
                    rv[i][e] = *(const u32x4*)(ok ? a + (long long)e * p.dy_ld * esz : zero);
                }
            } else if (ROWFAST) {
                // positions m_first..m_first+E-1 = (n, rt, rh, rw0..rw0+E-1)
                const RowIt it = rowit_decode(g, m_first < g.M ? m_first : 0);
                const TapPos& tp = u_tp[i];
                const int ti = it.rt * g.st - g.pt + tp.kt, hi = it.rh * g.sh - g.ph + tp.kh;
                const int wi0 = it.rw * g.sw - g.pw + tp.kw;
                const bool rok = tp.ok && m_first < g.M && (unsigned)ti < (unsigned)g.ST && (unsigned)hi < (unsigned)g.SH;
                const long long off0 = ((((long long)(it.n * g.ST + ti) * g.SH + hi) * g.SW) + wi0) * g.src_ld + tp.ci;
                const char* a = (const char*)p.src + off0 * esz;
                const long long step = (long long)g.sw * g.src_ld * esz;
                DPC_UNROLL
                for (int e = 0; e < E; ++e) {
                    const bool ok = rok && (unsigned)(wi0 + e * g.sw) < (unsigned)g.SW;
                    rv[i][e] = *(const u32x4*)(ok ? a + e * step : zero);
                }
            } else {
                RowIt it = rowit_decode(g, m_first < g.M ? m_first : 0);
                DPC_UNROLL
                for (int e = 0; e < E; ++e) {
                    const int m = m_first + e;
                    long long off = gather_off(g, rowit_pos(g, it), u_tp[i]);
                    if (m >= g.M) off = -1;
                    rv[i][e] = *(const u32x4*)(off >= 0 ? (const char*)p.src + off * esz : zero);
                    rowit_next(g, it);
                }
            }
        }
    };
    auto store_chunk = [&](int buf) {
        unsigned char* As = lds + buf * (TM + TN) * 128;
        unsigned char* Bs = As + TM * 128;
        DPC_UNROLL
        for (int i = 0; i < NU; ++i) {
            if (!u_on[i]) continue;
            u32x4 o[E];
            Transposer<T>::run(rv[i], o);
            unsigned char* base = u_isA[i] ? As : Bs;
            DPC_UNROLL
            for (int c = 0; c < E; ++c) *(u32x4*)(base + lds_unit_off3(u_cu[i] * E + c, u_pg[i])) = o[c];
        }
    };

    if (c_begin < c_end) {
        load_chunk(c_begin);
        store_chunk(0);
    }
    __syncthreads();
    for (int ch = c_begin; ch < c_end; ++ch) {
        const int buf = (ch - c_begin) & 1;
        if (ch + 1 < c_end) load_chunk(ch + 1);
        const unsigned char* As = lds + buf * (TM + TN) * 128;
        const unsigned char* Bs = As + TM * 128;
        bool done = false;
        if constexpr (sizeof(T) == 4) {
            if (p.x6) {   // bf16x6: unit pairs (2 kp, 2 kp + 1) per lane half = 8 f32 of K, split once per fragment
                DPC_UNROLL
                for (int kp = 0; kp < 2; ++kp) {
                    const int u0 = 4 * kp + lhi, u1 = u0 + 2;
                    Split3 sa[MI], sb[NT];
                    DPC_UNROLL
                    for (int i = 0; i < MI; ++i)
                        sa[i] = split3_f32x8(*(const u32x4*)(As + lds_unit_off3(wm * (TM / 2) + i * 32 + l31, u0)),
                                             *(const u32x4*)(As + lds_unit_off3(wm * (TM / 2) + i * 32 + l31, u1)));
                    DPC_UNROLL
                    for (int j = 0; j < NT; ++j)
                        sb[j] = split3_f32x8(*(const u32x4*)(Bs + lds_unit_off3(wn * (TN / 2) + j * 32 + l31, u0)),
                                             *(const u32x4*)(Bs + lds_unit_off3(wn * (TN / 2) + j * 32 + l31, u1)));
                    DPC_UNROLL
                    for (int i = 0; i < MI; ++i)
                        DPC_UNROLL
                        for (int j = 0; j < NT; ++j) acc[i][j] = mfma_f32x6(sa[i], sb[j], acc[i][j]);
                }
                done = true;
            }
        }
        if (!done) {
        DPC_UNROLL
        for (int kk = 0; kk < 4; ++kk) {
            const int unit = 2 * kk + lhi;
            u32x4 fa[MI], fb[NT];
            DPC_UNROLL
            for (int i = 0; i < MI; ++i) fa[i] = *(const u32x4*)(As + lds_unit_off3(wm * (TM / 2) + i * 32 + l31, unit));
            DPC_UNROLL
            for (int j = 0; j < NT; ++j) fb[j] = *(const u32x4*)(Bs + lds_unit_off3(wn * (TN / 2) + j * 32 + l31, unit));
            DPC_UNROLL
            for (int i = 0; i < MI; ++i)
                DPC_UNROLL
                for (int j = 0; j < NT; ++j) acc[i][j] = mfma_unit<T>(fa[i], fb[j], acc[i][j]);
        }
        }
        if (ch + 1 < c_end) store_chunk(buf ^ 1);
        __syncthreads();
    }

    DPC_UNROLL
    for (int j = 0; j < NT; ++j) {
        const int col = tile_n * TN + wn * (TN / 2) + j * 32 + l31;
        if (col < g.Kp) {
            DPC_UNROLL
            for (int i = 0; i < MI; ++i)
                DPC_UNROLL
                for (int r = 0; r < 16; ++r) {
                    const int row = tile_m * TM + wm * (TM / 2) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
                    if (row < p.Co) p.part[((long long)ks * p.Co + row) * g.Kp + col] = acc[i][j][r];
                }
        }
    }
}


// ---------------------------------------------------------------------------------------------------
// Second-generation kernel (power-of-two RH, RW -- every conv of the 128-pixel configurations).
//
// The tiles stay in their memory orientation, [position][channel], so both operands are filled by
// LDS-DMA (global_load_lds_dwordx4: no VGPR staging, no register transposes, no ds_write), and the
// matrix-core fragments -- reduction index contiguous per lane -- are produced by the LDS transpose
// read ds_read_b64_tr_b16 (bf16; f32 reads single dwords).  Every operand is cut into sub-tiles of
// 64 channels x BKP positions = 8 KB; each wave owns one dy sub-tile and one source sub-tile and
// computes 64 x 64 outputs (4 MFMA per 8 fragment reads: LDS read bytes per MFMA are 2/3 of the
// 32 x 64 per-wave shape of wgrad_kernel, and there are no LDS writes from the SIMDs at all).
// bf16 rows are 128 B; the 16-byte slot of row r is XORed with 2*(r&3), which makes the transpose
// reads bank-conflict free (brute-forced against the ds_read_b64 lane groups); the DMA lands
// lane-linearly, so the swizzle is applied to the *source* channel each lane fetches.
struct Wgrad2Params {
    GatherGeom g;
    const void* src;
    const void* dy;
    float* part;
    int Co, dy_ld;
    int nks, kcps;
    int ntm, ntn;
    int lRW, lRH;  // log2 of the (padded) image width / height the chunk rows are decomposed with
    int RWm, RHm;  // padded width / height - 1
    int Mv;        // rows of the padded position grid (== g.M when RW, RH are powers of two)
    int xcd_remap;
    int x6;        // f32 operands: contractions as bf16x6 (dpc_rt.h)
};

template <class T, int NWM, int NWN, bool PU>
__global__ __launch_bounds__(64 * NWM * NWN, 2) void wgrad2_kernel(Wgrad2Params p) {
    constexpr int E = Elt<T>::PER16;
    constexpr int ESZ = (int)sizeof(T);
    constexpr int BKP = 8 * E;              // positions per chunk: 64 (bf16) / 32 (f32)
    constexpr int RB = 64 * ESZ;            // sub-tile row bytes: 128 / 256
    constexpr int SPR = RB / 16;            // 16-byte slots per row: 8 / 16
    constexpr int RPI = 64 / SPR;           // rows per DMA instruction: 8 / 4
    constexpr int SUB = BKP * RB;           // 8 KB
    constexpr int NW = NWM * NWN, NS = NWM + NWN;
    constexpr int RR = (8 + NW - 1) / NW;   // DMA instructions per wave per sub-tile
    constexpr int STAGE = NS * SUB;
    constexpr int TM = 64 * NWM, TN = 64 * NWN;
    __shared__ __attribute__((aligned(1024))) unsigned char lds[2 * STAGE];

    const GatherGeom& g = p.g;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
#ifdef DPC_SIMT_EMU
    const int wv = tid >> 6;
#else
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
#endif
    const int wm = wv / NWN, wn = wv % NWN;
    const int l31 = lane & 31, lhi = lane >> 5;
    // Workgroups are dealt to the 8 XCDs round-robin; remap so that an XCD owns a contiguous range of
    // logical ids: the ntm*ntn tiles of one K-split (same dy rows, same source rows, shifted taps) then run
    // next to each other under ONE L2 instead of being fetched by all eight.
    int id = blockIdx.x;
    if (p.xcd_remap) {
        const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, x = id & 7;
        id = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (id >> 3);
    }
    const int tile_n = id % p.ntn; id /= p.ntn;
    const int tile_m = id % p.ntm;
    const int ks = id / p.ntm;

    // Images whose width / height are not powers of two (the 224-pixel family: 28, 14, 7) are walked on a grid padded to the next
    // power of two: a chunk row decomposes as (row & RWm, row >> lRW) as before, rows that fall into the padding are out-of-image
    // lanes (zeros: they add nothing), and the REAL row pitch is used for every address.  12-23 % of the rows are padding there.
    const int nchunks_total = (p.Mv + BKP - 1) / BKP;
    const int c_begin = ks * p.kcps;
    const int c_end = (c_begin + p.kcps < nchunks_total) ? c_begin + p.kcps : nchunks_total;

    // ---- DMA lane constants: which channel group this lane fetches, which rows
    const int rowlane = lane / SPR, pslot = lane % SPR;
    const int lslot = (ESZ == 2) ? (pslot ^ (2 * (rowlane & 3))) : pslot;
    const int cch = lslot * E;  // channel inside the 64-channel sub-tile
    int r_row[RR], r_w[RR], r_h[RR], r_q[RR], r_dh[RR], r_real[RR];
    bool r_wok[RR];
    DPC_UNROLL
    for (int rr = 0; rr < RR; ++rr) {
        const int r = (wv + rr * NW) * RPI + rowlane;
        r_row[rr] = r;
        r_w[rr] = (r & p.RWm) * g.sw - g.pw;
        r_h[rr] = ((r >> p.lRW) & p.RHm) * g.sh - g.ph;
        r_q[rr] = r >> (p.lRW + p.lRH);
        r_dh[rr] = r >> p.lRW;                               // image rows below the chunk's first one
        r_wok[rr] = (r & p.RWm) < g.RW;                      // not a padding column
        r_real[rr] = r_dh[rr] * g.RW + (r & p.RWm);          // row offset in the real tensor (== r without padding)
    }
    // Thread-constant parts of every DMA source address.  In a chunk that stays inside one (n, t) plane
    // (PU) the position of row r is (n, t, h0 + r/RW, r%RW) with only (n, t, h0) -- block-uniform --
    // changing from chunk to chunk, so
    //   byte offset = U(chunk) * ld + b_off[rr][sb],  valid <=> t-tap inside && (unsigned)(h0*sh + b_h[rr][sb]) < SH
    // where b_h also carries the (constant) verdicts "w inside" and "k' < Kp" as an out-of-range value.
    int a_off[RR][NWM];
    DPC_UNROLL
    for (int sa = 0; sa < NWM; ++sa) {
        const int co = tile_m * TM + sa * 64 + cch;
        const bool ok = co + E <= p.dy_ld && co < p.Co;
        DPC_UNROLL
        for (int rr = 0; rr < RR; ++rr) a_off[rr][sa] = ok && r_wok[rr] ? (r_real[rr] * p.dy_ld + co) * ESZ : -1;
    }
    TapPos tp[NWN];
    int b_off[RR][NWN], b_h[RR][NWN];
    const int ldb = g.src_ld * ESZ;
    DPC_UNROLL
    for (int sb = 0; sb < NWN; ++sb) {
        tp[sb] = decode_k(g, tile_n * TN + sb * 64 + cch);
        DPC_UNROLL
        for (int rr = 0; rr < RR; ++rr) {
            const int wi = r_w[rr] + tp[sb].kw;
            const bool ok = tp[sb].ok && (unsigned)wi < (unsigned)g.SW;
            b_h[rr][sb] = ok ? r_h[rr] + tp[sb].kh : (1 << 29);
            b_off[rr][sb] = ((tp[sb].kt * g.SH + r_h[rr] + tp[sb].kh) * g.SW + wi) * ldb + tp[sb].ci * ESZ;
        }
    }

    f32x16 acc[2][2];
    DPC_UNROLL
    for (int i = 0; i < 2; ++i)
        DPC_UNROLL
        for (int j = 0; j < 2; ++j)
            DPC_UNROLL
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const char* const zero = (const char*)dpc_zero16;
    auto issue = [&](int chunk, int buf) {
        const int m0 = chunk * BKP;  // block-uniform
        unsigned char* stage = lds + buf * STAGE;
        const unsigned q0 = (unsigned)m0 >> (p.lRW + p.lRH);
        const int h0v = PU ? (int)(((unsigned)m0 >> p.lRW) & (unsigned)p.RHm) : 0;   // first image row of the chunk
        const int h0 = h0v * g.sh;
        // first real row of the chunk: (plane q0, image row h0v) on the real pitch; without padding this is m0 itself
        const long long m0r = PU ? ((long long)q0 * g.RH + h0v) * g.RW : (long long)m0;
        const char* abase = (const char*)p.dy + m0r * p.dy_ld * ESZ;
        const char* bbase = (const char*)p.src;
        bool tv[NWN];
        if (PU) {
            const unsigned n = fdiv(q0, g.dRT);
            const int t00 = ((int)q0 - (int)n * g.RT) * g.st - g.pt;
            bbase += (long long)(((int)n * g.ST + t00) * g.SH + h0) * g.SW * ldb;
            DPC_UNROLL
            for (int sb = 0; sb < NWN; ++sb) tv[sb] = (unsigned)(t00 + tp[sb].kt) < (unsigned)g.ST;
        }
        DPC_UNROLL
        for (int rr = 0; rr < RR; ++rr) {
            const int rg = wv + rr * NW;
            if (rg < 8) {
                const bool mok = m0 + r_row[rr] < p.Mv && r_wok[rr] && (!PU || h0v + r_dh[rr] < g.RH);
                DPC_UNROLL
                for (int sa = 0; sa < NWM; ++sa)
                    glds16((mok & (a_off[rr][sa] >= 0)) ? abase + a_off[rr][sa] : zero, stage + sa * SUB + rg * 1024, lane);
                if (PU) {
                    DPC_UNROLL
                    for (int sb = 0; sb < NWN; ++sb) {
                        const bool ok = mok & tv[sb] & ((unsigned)(h0 + b_h[rr][sb]) < (unsigned)g.SH);
                        glds16(ok ? bbase + b_off[rr][sb] : zero, stage + (NWM + sb) * SUB + rg * 1024, lane);
                    }
                } else {
                    // the chunk spans several (n, t) planes (RH*RW < BKP): decode the plane per row
                    const unsigned q = q0 + (unsigned)r_q[rr];
                    const unsigned n = fdiv(q, g.dRT);
                    const int t0 = ((int)q - (int)n * g.RT) * g.st - g.pt;
                    const char* pb = bbase + (long long)(((int)n * g.ST + t0) * g.SH) * g.SW * ldb;
                    DPC_UNROLL
                    for (int sb = 0; sb < NWN; ++sb) {
                        const bool ok = mok & ((unsigned)(t0 + tp[sb].kt) < (unsigned)g.ST) & ((unsigned)b_h[rr][sb] < (unsigned)g.SH);
                        glds16(ok ? pb + b_off[rr][sb] : zero, stage + (NWM + sb) * SUB + rg * 1024, lane);
                    }
                }
            }
        }
    };

    // ---- fragment lane offsets (same for both operands: all sub-tiles share one layout)
    int fo[2];
    if (ESZ == 2) {
        const int gq = lane >> 4, s16 = lane & 15;
        DPC_UNROLL
        for (int i = 0; i < 2; ++i) {
            const int colb = (i * 32 + (gq & 1) * 16 + 4 * (s16 & 3)) * 2;
            fo[i] = ((gq >> 1) * 8 + (s16 >> 2)) * RB + ((((colb >> 4) ^ (2 * (s16 >> 2))) & 7) << 4) + (colb & 15);
        }
    } else {
        DPC_UNROLL
        for (int i = 0; i < 2; ++i) fo[i] = lhi * RB + (i * 32 + l31) * 4;
    }

    auto compute = [&](int buf) {
        const unsigned char* As = lds + buf * STAGE + wm * SUB;
        const unsigned char* Bs = lds + buf * STAGE + (NWM + wn) * SUB;
#ifndef DPC_SIMT_EMU
        if constexpr (ESZ == 2) {
            // 16 MFMAs per chunk with their operand reads software-pipelined by hand (scheme of conv_wgrad_patch.hip).  Operand loads
            // in program order per K step: A0 B0 B1 A1 (two ds_read_b64_tr_b16 each); MFMAs (0,0) (0,1) (1,0) (1,1) first need
            // B0, B1, A1, nothing.  Before MFMA m the loads up to LOOKAHEAD operands beyond its own are issued; the counted wait lets
            // exactly those stay in flight.  Register sets alternate by K-step parity.
            constexpr int LOOKAHEAD = 4;
            const uint32_t as = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const unsigned char*)As;
            const uint32_t bs = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const unsigned char*)Bs;
            uint32_t ad[2][2];  // [operand A/B][fragment]
            DPC_UNROLL
            for (int i = 0; i < 2; ++i) { ad[0][i] = as + fo[i]; ad[1][i] = bs + fo[i]; }
            u32x2 lo[2][2][2], hi[2][2][2];  // [K-step parity][operand][fragment]
            auto load = [&](auto Ic) {
                constexpr int I = decltype(Ic)::value;
                constexpr int kk = I / 4, r4 = I % 4;
                constexpr int op = (r4 == 1 || r4 == 2) ? 1 : 0, fr = (r4 >= 2) ? 1 : 0;   // A0 B0 B1 A1
                asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(lo[kk & 1][op][fr]) : "v"(ad[op][fr]), "n"((kk * 16) * RB) : "memory");
                __builtin_amdgcn_sched_barrier(0);
                asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(hi[kk & 1][op][fr]) : "v"(ad[op][fr]), "n"((kk * 16 + 4) * RB) : "memory");
                __builtin_amdgcn_sched_barrier(0);
            };
            __builtin_amdgcn_sched_barrier(0);
            static_for<16>([&](auto Mc) {
                constexpr int m = decltype(Mc)::value;
                constexpr int kk = m / 4, q = m % 4;
                constexpr int idx = 4 * kk + (q == 0 ? 1 : q == 1 ? 2 : 3);
                constexpr int idx_prev = m == 0 ? -1 : (q == 0 ? 4 * (kk - 1) + 3 : 4 * kk + (q == 1 ? 1 : q == 2 ? 2 : 3));
                constexpr int f_prev = m == 0 ? 0 : (idx_prev + 1 + LOOKAHEAD < 16 ? idx_prev + 1 + LOOKAHEAD : 16);
                constexpr int f_now = idx + 1 + LOOKAHEAD < 16 ? idx + 1 + LOOKAHEAD : 16;
                static_for<f_now - f_prev>([&](auto Dc) { load(std::integral_constant<int, f_prev + decltype(Dc)::value>{}); });
                asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(2 * (f_now - idx - 1)) : "memory");
                __builtin_amdgcn_sched_barrier(0);
                constexpr int i = q >> 1, j = q & 1, s = kk & 1;
                const u32x4 av = {lo[s][0][i][0], lo[s][0][i][1], hi[s][0][i][0], hi[s][0][i][1]};
                const u32x4 bv = {lo[s][1][j][0], lo[s][1][j][1], hi[s][1][j][0], hi[s][1][j][1]};
                acc[i][j] = mfma_32x32x16_bf16(av, bv, acc[i][j]);
                __builtin_amdgcn_sched_barrier(0);
            });
            return;
        }
#endif
        if (ESZ == 2) {
            DPC_UNROLL
            for (int kk = 0; kk < 4; ++kk) {
                u32x4 fa[2], fb[2];
                DPC_UNROLL
                for (int i = 0; i < 2; ++i) {
                    const u32x2 a0 = lds_read_tr16(As + fo[i] + (kk * 16) * RB);
                    const u32x2 a1 = lds_read_tr16(As + fo[i] + (kk * 16 + 4) * RB);
                    const u32x2 b0 = lds_read_tr16(Bs + fo[i] + (kk * 16) * RB);
                    const u32x2 b1 = lds_read_tr16(Bs + fo[i] + (kk * 16 + 4) * RB);
                    u32x4 ta = {a0[0], a0[1], a1[0], a1[1]};
                    u32x4 tb = {b0[0], b0[1], b1[0], b1[1]};
                    fa[i] = ta; fb[i] = tb;
                }
                DPC_UNROLL
                for (int i = 0; i < 2; ++i)
                    DPC_UNROLL
                    for (int j = 0; j < 2; ++j) acc[i][j] = mfma_32x32x16_bf16(fa[i], fb[j], acc[i][j]);
            }
        } else if (p.x6) {
            // bf16x6 (dpc_rt.h): 16 positions per step, a lane half takes the eight positions 16 s + 2 e + lhi (the rows the f32 chain
            // reads in eight steps), split once per fragment, six bf16 MFMAs per block pair
            DPC_UNROLL
            for (int s16 = 0; s16 < BKP / 16; ++s16) {
                Split3 sa[2], sb[2];
                DPC_UNROLL
                for (int i = 0; i < 2; ++i) {
                    float xa[8], xb[8];
                    DPC_UNROLL
                    for (int e = 0; e < 8; ++e) {
                        xa[e] = *(const float*)(As + fo[i] + 2 * (8 * s16 + e) * RB);
                        xb[e] = *(const float*)(Bs + fo[i] + 2 * (8 * s16 + e) * RB);
                    }
                    sa[i] = split3_f32(xa);
                    sb[i] = split3_f32(xb);
                }
                DPC_UNROLL
                for (int i = 0; i < 2; ++i)
                    DPC_UNROLL
                    for (int j = 0; j < 2; ++j) acc[i][j] = mfma_f32x6(sa[i], sb[j], acc[i][j]);
            }
        } else {
            DPC_UNROLL
            for (int s2 = 0; s2 < BKP / 2; ++s2) {
                float fa[2], fb[2];
                DPC_UNROLL
                for (int i = 0; i < 2; ++i) {
                    fa[i] = *(const float*)(As + fo[i] + 2 * s2 * RB);
                    fb[i] = *(const float*)(Bs + fo[i] + 2 * s2 * RB);
                }
                DPC_UNROLL
                for (int i = 0; i < 2; ++i)
                    DPC_UNROLL
                    for (int j = 0; j < 2; ++j) acc[i][j] = mfma_32x32x2_f32(fa[i], fb[j], acc[i][j]);
            }
        }
    };

    if (c_begin < c_end) issue(c_begin, 0);
    __syncthreads();
    for (int ch = c_begin; ch < c_end; ++ch) {
        const int buf = (ch - c_begin) & 1;
        if (ch + 1 < c_end) issue(ch + 1, buf ^ 1);
        compute(buf);
        __syncthreads();
    }

    DPC_UNROLL
    for (int j = 0; j < 2; ++j) {
        const int col = tile_n * TN + wn * 64 + j * 32 + l31;
        if (col < g.Kp) {
            DPC_UNROLL
            for (int i = 0; i < 2; ++i)
                DPC_UNROLL
                for (int r = 0; r < 16; ++r) {
                    const int row = tile_m * TM + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
                    if (row < p.Co) p.part[((long long)ks * p.Co + row) * g.Kp + col] = acc[i][j][r];
                }
        }
    }
}

template <class T, int NWM, int NWN>
static int launch_wgrad2_t(const Wgrad2Params& p, bool pu, hipStream_t stream) {
    dim3 grid((unsigned)(p.ntm * p.ntn * p.nks)), block(64 * NWM * NWN);
    dpc_plan_detail("T=%s nwm=%d nwn=%d padded=%d", sizeof(T) == 2 ? "bf16" : "f32", NWM, NWN,
                    (int)(((p.RWm + 1) != p.g.RW) || ((p.RHm + 1) != p.g.RH)));
    if (pu) {
        DPC_LAUNCH((wgrad2_kernel<T, NWM, NWN, true>), grid, block, stream, p);
    } else {
        DPC_LAUNCH((wgrad2_kernel<T, NWM, NWN, false>), grid, block, stream, p);
    }
    return dpc_launch_status();
}

template <class T>
static int launch_wgrad2(const Wgrad2Params& p, int nwm, int nwn, bool pu, hipStream_t stream) {
    if (nwm == 1 && nwn == 3) return launch_wgrad2_t<T, 1, 3>(p, pu, stream);
    if (nwm == 1 && nwn == 4) return launch_wgrad2_t<T, 1, 4>(p, pu, stream);
    if (nwm == 2 && nwn == 2) return launch_wgrad2_t<T, 2, 2>(p, pu, stream);
    return launch_wgrad2_t<T, 2, 3>(p, pu, stream);
}

template <class T, bool RF>
static int launch_wgrad_rf(const WgradParams& p, int tm, int tn, hipStream_t stream) {
    dim3 grid((unsigned)(p.ntm * p.ntn * p.nks)), block(256);
    dpc_plan_detail("T=%s", sizeof(T) == 2 ? "bf16" : "f32");
    if (tm == 64 && tn == 64) {
        DPC_LAUNCH((wgrad_kernel<T, 64, 64, RF>), grid, block, stream, p);
    } else if (tm == 64 && tn == 128) {
        DPC_LAUNCH((wgrad_kernel<T, 64, 128, RF>), grid, block, stream, p);
    } else {
        DPC_LAUNCH((wgrad_kernel<T, 128, 128, RF>), grid, block, stream, p);
    }
    return dpc_launch_status();
}

template <class T>
static int launch_wgrad(const WgradParams& p, int tm, int tn, hipStream_t stream) {
    if (p.g.RW % Elt<T>::PER16 == 0) return launch_wgrad_rf<T, true>(p, tm, tn, stream);
    return launch_wgrad_rf<T, false>(p, tm, tn, stream);
}

extern "C" int dpc_conv_wgrad(const dpc_conv_desc* d, const void* src, const void* dy, int32_t dy_ld,
                              float* part, int32_t* nsplit, dpc_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    WgradParams p;
    p.x6 = dpc_f32_matmul_mode();
    if (!d || d->mode != 0) return DPC_ERR_ARG;
    int rc = make_gather_geom(d, &p.g);
    if (rc) return rc;
    const int per16 = d->dtype_in == DPC_BF16 ? 8 : 4;
    if (dy_ld % per16 || dy_ld < d->Co) return DPC_ERR_UNSUPPORTED;
    rc = dpc_wgrad_patch_try(d, src, dy, dy_ld, part, nsplit, stream);  // 3x3 stride-1: one staged patch serves all nine taps
    if (rc != 1) return rc;
    rc = dpc_wgrad_stem_try(d, src, dy, dy_ld, part, nsplit, stream);   // space-to-depth stem: same idea, 4x4 taps of 32-byte positions
    if (rc != 1) return rc;
    const int bkp = 8 * per16;
    // wgrad2 decomposes a chunk row index as (row % RWp, row / RWp) with RWp, RHp = width / height rounded up to powers of two
    // (padding rows are masked lanes): the chunk must start at a row boundary (RWp divides the chunk length); with padding it
    // must also stay inside one (n, t) plane and at least 60 % of the padded grid must be real (else the generic kernel).
    int rwp = 1, rhp = 1;
    while (rwp < d->RW) rwp <<= 1;
    while (rhp < d->RH) rhp <<= 1;
    const bool padded = rwp != d->RW || rhp != d->RH;
    const long long mv = (long long)d->N * d->RT * rhp * rwp;
    static const int pad_on = getenv("DPC_WGRAD2_PAD") ? atoi(getenv("DPC_WGRAD2_PAD")) : 1;
    const bool v2 = bkp % rwp == 0 && mv < (1ll << 31) &&
                    (!padded || (pad_on && (rwp * rhp) % bkp == 0 && (long long)d->RW * d->RH * 10 >= (long long)rwp * rhp * 6));
    const int lrw = ilog2_exact(rwp), lrh = ilog2_exact(rhp);
    const int nchunks = (int)(((v2 ? mv : (long long)p.g.M) + bkp - 1) / bkp);
    int tm, tn, nwm = 0, nwn = 0;
    if (v2) {
        // 64 x 64 per wave; the block is NWM x NWN waves.  NWN = 3 fits every 3x3 / 3x3x3 reduction
        // (Kp = 9*Ci or 27*Ci with 64 | Ci) exactly; otherwise take the width with the least padding.
        nwm = d->Co > 64 ? 2 : 1;
        const int na = nwm == 1 ? 4 : 2, nb = 3;
        const int pa = (p.g.Kp + 64 * na - 1) / (64 * na) * (64 * na), pb = (p.g.Kp + 64 * nb - 1) / (64 * nb) * (64 * nb);
        nwn = (pb < pa) ? nb : na;
        tm = 64 * nwm; tn = 64 * nwn;
    } else {
        tm = d->Co > 64 ? 128 : 64;
        tn = (p.g.Kp >= 128 || tm == 128) ? 128 : 64;
    }
    p.ntm = (d->Co + tm - 1) / tm;
    p.ntn = (p.g.Kp + tn - 1) / tn;
    // K-splits: enough workgroups to fill the chip, but every split must amortise its f32 partial slab (a tile
    // of tm x tn floats written, then re-read by the reduction) over at least `min_chunks` chunks -- the small
    // ConvGRU gradients ran 4 chunks per 64 KB slab
    static const int min_chunks = getenv("DPC_WGRAD_MINCHUNKS") ? atoi(getenv("DPC_WGRAD_MINCHUNKS")) : 16;
    static const int target_blocks = getenv("DPC_WGRAD_BLOCKS") ? atoi(getenv("DPC_WGRAD_BLOCKS")) : 512;  // fewer f32 slabs to write and re-read; 512 workgroups still fill the chip twice (sweep: profiles/r02_sweeps.txt)
    int want = target_blocks / (p.ntm * p.ntn);
    if (want > nchunks / min_chunks) want = nchunks / min_chunks;
    if (want < 1) want = 1;
    if (want > nchunks) want = nchunks;
    p.kcps = (nchunks + want - 1) / want;
    p.nks = (nchunks + p.kcps - 1) / p.kcps;
    if (nsplit) *nsplit = p.nks;
    if (!part) return DPC_OK;  // size query
    if (!src || !dy) return DPC_ERR_ARG;
    p.src = src; p.dy = dy; p.part = part; p.Co = d->Co; p.dy_ld = dy_ld;
    if (v2) {
        Wgrad2Params q;
        q.g = p.g; q.src = src; q.dy = dy; q.part = part; q.Co = d->Co; q.dy_ld = dy_ld;
        q.nks = p.nks; q.kcps = p.kcps; q.ntm = p.ntm; q.ntn = p.ntn; q.lRW = lrw; q.lRH = lrh;
        q.RWm = rwp - 1; q.RHm = rhp - 1; q.Mv = (int)mv; q.x6 = p.x6;
        q.xcd_remap = 1;   // the tiles of one K-split share an L2 (round-robin placement measured slower in round 2; the switch is gone)
        const bool pu = (rwp * rhp) % bkp == 0;  // a chunk never leaves its (n, t) plane
        if (d->dtype_in == DPC_F32) return launch_wgrad2<float>(q, nwm, nwn, pu, stream);
        return launch_wgrad2<bf16_t>(q, nwm, nwn, pu, stream);
    }
    if (d->dtype_in == DPC_F32) return launch_wgrad<float>(p, tm, tn, stream);
    return launch_wgrad<bf16_t>(p, tm, tn, stream);
}
