// conv_igemm_ws.hip -- implicit-GEMM convolution for the wide layers (Co >= 128: layer2..4 of the
// 2d3d-ResNet, backbone/resnet_2d3d.py:14-32,241-244, forward and unit-stride input-gradient; the
// 1x1 ConvGRU / prediction GEMMs, backbone/convrnn.py:13-15, dpc/model_3d.py:36-40), bf16 in / bf16 out.
//
// Same math, same LDS image and same fragment reads as conv_igemm.hip; what changes is WHO moves the
// operands.  Measured on conv_igemm's 128 x 128 tile (rocprofv3 PMC, profiles/): the matrix cores are
// busy 47 % of the main loop and the rest is the waves sitting in their own LDS-DMA issue -- a
// 1 KB global_load_lds piece holds its wave for 60..180 cycles (the CU's texture addresser moves
// 64 B/clk) and that wave cannot issue MFMAs meanwhile -- plus ~8 us per tile of exposed prologue
// (first chunk from HBM) and epilogue.  Here a workgroup is 8 waves with two roles:
//   * waves 0..3 (one per SIMD) only compute: 64 rows x 128 columns each of a 256 x 128 tile,
//     24 ds_read_b128 + 32 MFMA per 128-byte K chunk, nothing else in the loop;
//   * waves 4..7 (one per SIMD) only load: 12 LDS-DMA pieces per chunk each, three chunks in flight
//     in a ring of three 48 KB stages, counted s_waitcnt vmcnt(12) (never 0) before the one barrier
//     per chunk that publishes the oldest stage.  Their issue stalls no longer hold MFMA issue (the DMA is
//     not free for all that: round-2 probes, profiles/r02_phase_probes.txt, still see 12-28 % of a launch go
//     when it is left out -- LDS write traffic and power, see the note at igemm_wsp_kernel),
//     and they run ahead across tile boundaries: the next tile's first two chunks land while the
//     compute waves are in the epilogue, so there is no prologue after the first tile.
//   * 256-row tiles halve the weight bytes per MFMA: 48 KB per 32 MFMA-per-SIMD = 47 B/clk/CU of
//     the 64 the addresser can do.
// Ordering rules (cdna_hip_programming.md, "Read a staged buffer one phase AFTER the wait that retires
// it"): a stage is read only after the loaders' vmcnt wait AND the barrier that follows it; a stage is
// refilled only after a barrier that every reader passed with its ds_reads consumed by MFMAs.
// The epilogue stages each wave's 64 x 128 block through the LDS stage that was read last (free until
// the next chunk barrier) in two wave-private 32-row passes -- no workgroup barrier inside.
#include "conv_common.h"
#include <stdlib.h>

struct WsParams {
    GatherGeom g;
    const void* src;
    const void* wgt;
    void* out;
    const void* addend;
    float* stats;
    int Ncol, ldw, ldo;
    int gm, ntn, ntm;
    unsigned src_bytes, wgt_bytes;
    // temporal grouping (3x3x3 stride-1 convs): a tile holds 256 consecutive pixels of ONE output frame index t, taken
    // from consecutive clips (whole planes when the plane size divides 256, else a tile straddles clips -- 14 x 14 and
    // 7 x 7 planes of the 224-pixel configurations), so the temporal taps that fall into the zero padding are the same
    // for the whole tile and are skipped as a K sub-range (22 % of the chunks at T = 3, 33 % at T = 2).
    int tgroup, hw, nclip, tpt, cpkt;  // on/off, RH*RW, clips, tiles per t, chunks per temporal tap
    FastDiv d_hw;
    int plane;                        // 1: served by igemm_wsp_kernel (a tile is one 16 x 16 plane, staged as a patch)
    // strided input-gradient (igemm_ws_kernel<., true>): tiles are 256 rows of ONE parity class (conv_common.h), whose K range is
    // the class's own taps -- the generic kernel's scheme on this kernel's loaders.  There a tile of 4..32 chunks pays a decode, a
    // first-chunk round trip and an epilogue with one chunk in flight (layer3.0: 390 us for 62 us of MFMA work); here the loaders
    // run ahead across tile boundaries.
    int parity;
    ParityInfo par;
    int dbg;                          // DPC_WS_PROBE builds only (scripts/probes/ws_probe.py): phases to leave out, for timing
};
#ifdef DPC_WS_PROBE
#define WS_DBG(bit) (p.dbg & (bit))
#else
#define WS_DBG(bit) 0
#endif

#include "ws_frag.h"

template <bool HAS_ADD, bool PAR = false>
__global__ __launch_bounds__(512, 2) void igemm_ws_kernel(WsParams p) {
    typedef bf16_t T;
    typedef bf16_t TO;
    constexpr int BM = 256, BN = 128, BKE = 64;
    constexpr int STAGE = (BM + BN) * 128;  // 48 KB
    constexpr int NST = 3;
    constexpr int EPO = 8, UPR = BN / EPO;  // 16 output units per row
    // Round 3 claimed all 160 KB here "so that no LDS-using workgroup can sit beside this one": beside an 8 KB-LDS kernel of another
    // stream 38 of 15 000 launches came back with one wave's 64 x 128 block wrong.  Round 4 found the mechanism (WS_RETIRE_TAIL_READS
    // below, scripts/asm_hazard_lint.py): it was a register hazard of the end-of-tile fragment reads, not the ring's ordering, and a
    // co-tenant only stretched the LDS latency far enough to expose it.  With the reads retired the kernel takes what it needs
    // (144 KB) and shares the CU with whatever fits beside it (profiles/r04_cotenant.txt).
    __shared__ __attribute__((aligned(1024))) unsigned char lds[NST * STAGE];

    const GatherGeom& g = p.g;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
#ifdef DPC_SIMT_EMU
    const int wv = tid >> 6;
#else
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
#endif
    const int n_tile = blockIdx.x / p.gm;
    const int m_prog = blockIdx.x % p.gm;
    const int nkc = g.Kp / BKE;
    const int my_tiles = (p.ntm - m_prog + p.gm - 1) / p.gm;
    // parity class of a tile (wave-uniform): residues, extents, row count, tile inside the class
    struct ParCls { int j, rows, ct, ch, cw, Tc, Hc, Wc; FastDiv dT, dH, dW; };
    auto par_cls = [&](int mt) -> ParCls {
        ParCls k;
        int c;
        parity_tile(p.par, mt, c, k.j);
        k.cw = c & (g.sw - 1); k.ch = (c >> g.lsw) & (g.sh - 1); k.ct = c >> (g.lsw + g.lsh);   // strides are 1 or 2
        k.Tc = k.ct ? p.par.dimc[0][1] : p.par.dimc[0][0]; k.dT = k.ct ? p.par.div[0][1] : p.par.div[0][0];
        k.Hc = k.ch ? p.par.dimc[1][1] : p.par.dimc[1][0]; k.dH = k.ch ? p.par.div[1][1] : p.par.div[1][0];
        k.Wc = k.cw ? p.par.dimc[2][1] : p.par.dimc[2][0]; k.dW = k.cw ? p.par.div[2][1] : p.par.div[2][0];
        k.rows = p.par.N * k.Tc * k.Hc * k.Wc;
        return k;
    };
    auto par_decode = [&](const ParCls& k, int lr, int& n, int& tq, int& hq, int& wq) {
        const unsigned q1 = fdiv((unsigned)lr, k.dW);
        wq = lr - (int)q1 * k.Wc;
        const unsigned q2 = fdiv(q1, k.dH);
        hq = (int)q1 - (int)q2 * k.Hc;
        const unsigned nn = fdiv(q2, k.dT);
        tq = (int)q2 - (int)nn * k.Tc;
        n = (int)nn;
    };
    auto par_row = [&](const ParCls& k, int r) -> int {   // tile row -> output row of the class tile (g.M: no such row)
        const int lr = k.j * BM + r;
        if (lr >= k.rows) return g.M;
        int n, tq, hq, wq;
        par_decode(k, lr, n, tq, hq, wq);
        return (((n * g.RT + tq * g.st + k.ct) * g.RH + hq * g.sh + k.ch) * g.RW) + wq * g.sw + k.cw;
    };
    auto par_pk = [&](const ParCls& k, unsigned& pT, unsigned& pH, unsigned& pW) {
        pT = k.ct ? p.par.pk[0][1] : p.par.pk[0][0];
        pH = k.ch ? p.par.pk[1][1] : p.par.pk[1][0];
        pW = k.cw ? p.par.pk[2][1] : p.par.pk[2][0];
    };
    // tile row -> GEMM row (identity unless the tiles are grouped by output frame index / parity class)
    auto tile_row = [&](int mt, int r) -> int {
        if (!p.tgroup) return mt * BM + r;
        const int t = mt / p.tpt;
        const unsigned idx = (unsigned)((mt - t * p.tpt) * BM + r);   // pixel of frame t, counted across the clips
        const unsigned clip = fdiv(idx, p.d_hw);
        return (int)clip < p.nclip ? (int)((clip * (unsigned)g.RT + (unsigned)t) * (unsigned)p.hw + (idx - clip * (unsigned)p.hw)) : g.M;
    };
    // K chunks [lo, hi) of a tile: all of them, or those of the temporal taps that hit real frames
    auto tile_chunks = [&](int mt, int& lo, int& hi) {
        lo = 0; hi = nkc;
        if (PAR) {
            const ParCls k = par_cls(mt);
            unsigned pT, pH, pW;
            par_pk(k, pT, pH, pW);
            hi = par_pk_n(pT) * par_pk_n(pH) * par_pk_n(pW) * (g.Ci >> 6);
            return;
        }
        if (!p.tgroup) return;
        const int t = mt / p.tpt;
        int klo, khi;
        if (g.mode == 0) { klo = g.pt - t; khi = g.ST - 1 + g.pt - t; }
        else { klo = t + g.pt - (g.ST - 1); khi = t + g.pt; }
        if (klo < 0) klo = 0;
        if (khi > g.KT - 1) khi = g.KT - 1;
        lo = klo * p.cpkt; hi = (khi + 1) * p.cpkt;
    };
    int total = 0;  // chunks this workgroup walks, across its tiles
    for (int t = 0; t < my_tiles; ++t) {
        int lo, hi;
        tile_chunks(m_prog + t * p.gm, lo, hi);
        total += hi - lo;
    }
    const char* const zero = (const char*)dpc_zero16;

    if (wv >= 4) {
        // ------------------------------------------------------------------ loader waves
        const int lw = wv - 4;
        // piece j covers tile rows 8j..8j+7 (lane>>3 picks the row, lane&7 the 16-byte slot); this wave
        // owns pieces lw, lw+4, ...: 8 of A, 4 of B.  swz1(row) = (4j + (lane>>4)) & 7 and 4j mod 8 is
        // the same for all its pieces, so the logical unit u a lane fetches is a kernel constant.
        const int rl = lane >> 3;
        const int u = (lane & 7) ^ lds_swz1(8 * lw + rl);
        // sources as buffer resources: 32-bit offsets, hardware zero-fill for padding / rows beyond M / columns beyond Ncol
        const BufRsrc rs_a = make_buf_rsrc(p.src, p.src_bytes);
        const BufRsrc rs_b = make_buf_rsrc(p.wgt, p.wgt_bytes);
        unsigned wrow[4];
        DPC_UNROLL
        for (int i = 0; i < 4; ++i) {
            const int n = n_tile * BN + 8 * (lw + 4 * i) + rl;
            wrow[i] = n < p.Ncol ? (unsigned)(n * p.ldw + u * 8) * 2u : DPC_BUF_OOB;
        }
        unsigned rowoff[8];   // byte offset of (row, tap 0, this lane's channel group); wraps for rows that start in the padding
        unsigned vmask[8];
        // Row decode of a tile.  The 8 lanes that share a tile row (one per 16-byte slot) need the same 8 rows
        // (pieces lw, lw+4, ...): each of them decodes ONE of those rows and the group exchanges the results by
        // lane shuffles -- 1 decode + 16 shuffles per lane instead of 8 decodes (the decode sits on the loaders'
        // critical path once per tile: ~4000 cycles before, per tile of only 18 chunks on layer2).
        auto range_bits = [](int x0, int sgn, int K, int S) -> unsigned {  // bit k set <=> 0 <= x0 + sgn*k < S, k < K
            int lo, hi;                                                     // sgn = +1: k in [-x0, S-1-x0]; -1: k in [x0-S+1, x0]
            if (sgn > 0) { lo = -x0; hi = S - 1 - x0; } else { lo = x0 - S + 1; hi = x0; }
            if (lo < 0) lo = 0;
            if (hi > K - 1) hi = K - 1;
            return hi >= lo ? (((2u << hi) - 1u) & ~((1u << lo) - 1u)) : 0u;
        };
        unsigned l_pT = 0, l_pH = 0, l_pW = 0;          // PAR: packed tap tables of the class whose chunks are being issued,
        int l_jt = 0, l_jh = 0, l_jw = 0, l_cc = 0;   // and the (tap, 64-channel group) cursor of its next chunk
        auto decode_tile = [&](int mt) {
            const int mine = lane & 7;  // which of the group's 8 rows this lane decodes
            if (PAR) {
                const ParCls k = par_cls(mt);
                par_pk(k, l_pT, l_pH, l_pW);
                l_jt = l_jh = l_jw = l_cc = 0;
                const int lr = k.j * BM + 8 * (lw + 4 * mine) + rl;
                unsigned off = 0, m = 0;
                if (lr < k.rows) {
                    int n, tq, hq, wq;
                    par_decode(k, lr, n, tq, hq, wq);
                    off = (((((unsigned)(n * g.ST + tq) * (unsigned)g.SH + (unsigned)hq) * (unsigned)g.SW) + (unsigned)wq) * (unsigned)g.src_ld) * 2u;
                    DPC_UNROLL
                    for (int j = 0; j < 4; ++j) {
                        m |= (j < par_pk_n(l_pT) && (unsigned)(tq + par_pk_dl(l_pT, j)) < (unsigned)g.ST ? 1u : 0u) << j;
                        m |= (j < par_pk_n(l_pH) && (unsigned)(hq + par_pk_dl(l_pH, j)) < (unsigned)g.SH ? 1u : 0u) << (4 + j);
                        m |= (j < par_pk_n(l_pW) && (unsigned)(wq + par_pk_dl(l_pW, j)) < (unsigned)g.SW ? 1u : 0u) << (8 + j);
                    }
                }
                DPC_UNROLL
                for (int i = 0; i < 8; ++i) {
                    const int from = (lane & ~7) | i;
                    rowoff[i] = (unsigned)__shfl((int)off, from) + (unsigned)(u * 16);
                    vmask[i] = (unsigned)__shfl((int)m, from);
                }
                return;
            }
            const RowPos rp = decode_row(g, tile_row(mt, 8 * (lw + 4 * mine) + rl));
            const unsigned off = (((((unsigned)(rp.nbase + rp.t0) * (unsigned)g.SH + (unsigned)rp.h0) * (unsigned)g.SW) + (unsigned)rp.w0) *
                                  (unsigned)g.src_ld) * 2u;
            const int sgn = g.mode == 0 ? 1 : -1;
            const unsigned m = range_bits(rp.t0, sgn, g.KT, g.ST) | (range_bits(rp.h0, sgn, g.KH, g.SH) << g.KT) |
                               (range_bits(rp.w0, sgn, g.KW, g.SW) << (g.KT + g.KH));
            DPC_UNROLL
            for (int i = 0; i < 8; ++i) {
                const int from = (lane & ~7) | i;
                rowoff[i] = (unsigned)__shfl((int)off, from) + (unsigned)(u * 16);
                vmask[i] = (unsigned)__shfl((int)m, from);
            }
        };
        auto issue = [&](int kc, int stage) {
            unsigned char* st = lds + stage * STAGE;
            if (PAR) {   // chunks of a tile come in order: (tap of the class, 64-channel group) is a cursor, wave-uniform scalar work
                const int jt = l_jt, jh = l_jh, jw = l_jw, cofs = l_cc * BKE;
                if (++l_cc == (g.Ci >> 6)) {
                    l_cc = 0;
                    if (++l_jw == par_pk_n(l_pW)) {
                        l_jw = 0;
                        if (++l_jh == par_pk_n(l_pH)) { l_jh = 0; ++l_jt; }
                    }
                }
                const int kt = par_pk_kl(l_pT, jt), kh = par_pk_kl(l_pH, jh), kw = par_pk_kl(l_pW, jw);
                const int dt = par_pk_dl(l_pT, jt), dh = par_pk_dl(l_pH, jh), dw = par_pk_dl(l_pW, jw);
                const unsigned sel = (1u << jt) | (1u << (4 + jh)) | (1u << (8 + jw));
                const unsigned tapoff = (unsigned)(((dt * g.SH + dh) * g.SW + dw) * g.src_ld + cofs) * 2u;
                const unsigned kd = (unsigned)(((kt * g.KH + kh) * g.KW + kw) * g.Ci + cofs);
                DPC_UNROLL
                for (int i = 0; i < 8; ++i) {
                    const bool ok = (vmask[i] & sel) == sel;
                    if (!WS_DBG(4 | 128)) glds16_buf(rs_a, ok ? rowoff[i] + tapoff : DPC_BUF_OOB, 0u, st + (lw + 4 * i) * 1024, lane);
                }
                DPC_UNROLL
                for (int i = 0; i < 4; ++i)
                    if (!WS_DBG(4 | 256)) glds16_buf(rs_b, wrow[i], kd * 2u, st + BM * 128 + (lw + 4 * i) * 1024, lane);
                return;
            }
            // one tap per chunk (Ci is a multiple of 64): the tap decode is wave-uniform scalar work
            const int kd = kc * BKE;
            const int tap = (g.taps == 1) ? 0 : (kd >> g.log2C);
            const unsigned q = fdiv((unsigned)tap, g.dKW);
            const int kw = tap - (int)q * g.KW;
            const unsigned kt = fdiv(q, g.dKH);
            const int kh = (int)q - (int)kt * g.KH;
            const unsigned sel = (1u << kt) | (1u << (g.KT + kh)) | (1u << (g.KT + g.KH + kw));
            const int sgn = g.mode == 0 ? 1 : -1;
            const int cbase = (g.taps == 1) ? 0 : (tap << g.log2C);
            const unsigned tapoff = (unsigned)(sgn * ((((int)kt * g.SH + kh) * g.SW + kw) * g.src_ld) + (kd - cbase)) * 2u;
            DPC_UNROLL
            for (int i = 0; i < 8; ++i) {
                const bool ok = (vmask[i] & sel) == sel;
                if (!WS_DBG(4 | 128)) glds16_buf(rs_a, ok ? rowoff[i] + tapoff : DPC_BUF_OOB, 0u, st + (lw + 4 * i) * 1024, lane);
            }
            DPC_UNROLL
            for (int i = 0; i < 4; ++i)
                if (!WS_DBG(4 | 256)) glds16_buf(rs_b, wrow[i], (unsigned)kd * 2u, st + BM * 128 + (lw + 4 * i) * 1024, lane);
        };
        // chunk counter -> (tile, kc) of the NEXT chunk to issue
        int it_tile = 0, it_kc = 0, it_hi = 0;
        bool it_new = true;
        auto issue_next = [&](int gc) {
            if (it_new) {
                const int mt = m_prog + it_tile * p.gm;
                decode_tile(mt);
                tile_chunks(mt, it_kc, it_hi);
                it_new = false;
            }
            issue(it_kc, gc % NST);
            if (++it_kc == it_hi) { it_new = true; ++it_tile; }
        };
        if (total > 0) issue_next(0);
        if (total > 1) issue_next(1);
        int kc_done = 0, cur_tile = 0, cur_n;  // position of chunk gc inside its tile, chunks of that tile
        {
            int lo, hi;
            tile_chunks(m_prog, lo, hi);
            cur_n = hi - lo;
        }
        for (int gc = 0; gc < total; ++gc) {
            if (gc + 1 < total) wait_vmcnt<12>(); else wait_vmcnt<0>();
            ws_barrier();  // chunk gc is published; every reader is done with chunk gc-1
            if (gc + 2 < total) issue_next(gc + 2);
            if (++kc_done == cur_n) {
                kc_done = 0;
                ws_barrier();  // matches the compute waves' "tile fully read" barrier
                if (++cur_tile < my_tiles) {
                    int lo, hi;
                    tile_chunks(m_prog + cur_tile * p.gm, lo, hi);
                    cur_n = hi - lo;
                }
            }
        }
    }

    // ---------------------------------------------------------------------- compute waves
    const int l31 = lane & 31, lhi = lane >> 5;
    // fragment addresses of K step 0 inside a stage (rows / columns +32: +4096 B); step kk: ^ (kk << 5), see step_il
    const ldsa_t lds0 = ldsa(lds);
    const int fa0 = lds_unit_off(wv * 64 + l31, lhi);
    const int fb0 = BM * 128 + lds_unit_off(l31, lhi);
    // epilogue lane constants: a pass stages 32 rows x 128 columns (8 KB) per wave
    const int cu = lane & 15, er = lane >> 4;  // output unit column, first row of the lane inside a pass (rows er + 4*it)
    const int col0 = n_tile * BN + cu * EPO;
    float s1[EPO], s2[EPO];
    DPC_UNROLL
    for (int e = 0; e < EPO; ++e) { s1[e] = 0.f; s2[e] = 0.f; }

    int gc = 0;
#ifndef DPC_SIMT_EMU
    __builtin_amdgcn_s_setprio(3);   // the compute waves win the SIMD's issue arbitration against their loader partner (-1..2 %)
#endif
    for (int t = 0; wv < 4 && t < my_tiles; ++t) {
        const int mt = m_prog + t * p.gm;
        ParCls ck = {};
        if (PAR) ck = par_cls(mt);
        int kc_lo, kc_hi;
        tile_chunks(mt, kc_lo, kc_hi);
        const int nkc_t = kc_hi - kc_lo;
        f32x16 acc[2][4];
        DPC_UNROLL
        for (int i = 0; i < 2; ++i)
            DPC_UNROLL
            for (int j = 0; j < 4; ++j)
                DPC_UNROLL
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
        // The chunk loop is software-pipelined ACROSS the chunk barrier: the barrier that publishes chunk
        // gc+1 sits between MFMA steps 2 and 3 of chunk gc (all LDS reads of chunk gc are complete by then, so
        // it is also the "stage gc may be refilled" point), and the first fragments of chunk gc+1 are read
        // between the 8 MFMAs of step 3 (step_il: one read per MFMA gap, per-operand counted waits).  Left to itself
        // hipcc reuses one register set (every step then waits out a full LDS round trip: 64 % of the MFMA
        // rate measured) or, given two sets, waits lgkmcnt(0) -- i.e. also for the reads it has just issued.
        // Barrier sequence per tile, identical to the loaders': B(c0) B(c1) ... B(c_last) B("tile fully read").
        // residual addend (input-gradient of a block's first conv): requested ahead of the epilogue -- issued inside
        // it, 8 x 16 B per lane in flight made the extra tensor read cost +150 us on layer2
        u32x4 av0[8], av1[8];
        auto fetch_addend = [&](int i, u32x4 (&dst)[8]) {
            DPC_UNROLL
            for (int it = 0; it < 8; ++it) {
                const int row = PAR ? par_row(ck, wv * 64 + i * 32 + er + 4 * it) : tile_row(mt, wv * 64 + i * 32 + er + 4 * it);
                const bool ok = row < g.M && col0 < p.Ncol;
                const char* a = (const char*)p.addend + ((long long)row * p.ldo + col0) * 2;
                // rows outside the tensor read its unit 0 (never used: their outputs are not stored).  Not the zero page: a select between
                // a kernel-argument pointer and a device constant is a GENERIC pointer, its flat loads count on lgkmcnt as well and hipcc
                // waits lgkmcnt(0) for them -- behind the hand-issued fragment reads of the chunk (scripts/asm_drain_lint.py)
                dst[it] = *(const u32x4*)(ok ? a : (const char*)p.addend);
            }
        };
        int stage_last = 0;
        {
            FragSet f0, f1;
            ws_barrier();  // first chunk of the tile published
            ldsa_t pa = lds0 + (gc % NST) * STAGE + fa0, pb = lds0 + (gc % NST) * STAGE + fb0;
            frag_read_p2<4096>(f0, pa, pb);
            for (int kc = 0; kc < nkc_t; ++kc) {
                if (HAS_ADD && kc + 1 == nkc_t) fetch_addend(0, av0);  // lands under the last chunk's 32 MFMAs
                step_il<true, 4096>(acc, f0, f1, pa ^ 32, pb ^ 32);
                step_il<true, 4096>(acc, f1, f0, pa ^ 64, pb ^ 64);
                step_il<true, 4096>(acc, f0, f1, pa ^ 96, pb ^ 96);
                frag_wait<0>(f1);   // this wave's last reads of the chunk have landed
                ws_barrier();  // next chunk published -- or, after the tile's last chunk, "tile fully read"
                stage_last = gc % NST;
                ++gc;
                pa = lds0 + (gc % NST) * STAGE + fa0;
                pb = lds0 + (gc % NST) * STAGE + fb0;
                // after the tile's last chunk these reads fetch nothing useful (the next tile re-reads after ITS first barrier):
                // unconditional, so that the MFMAs stay out of a branch (accumulators joined after a branch are copied and spill)
                step_il<true, 4096>(acc, f1, f0, pa, pb);
            }
            WS_RETIRE_TAIL_READS(f0);   // the reads past the tile's last chunk land before their registers are handed on
        }

        if (WS_DBG(2)) {  // probe: no epilogue at all (the accumulators stay live through one store)
            if (acc[0][0][0] == 12345.f) *(float*)p.out = acc[1][3][5];
            continue;
        }
        // ---- epilogue: two passes of 32 rows through this wave's 8 KB of the free stage
        unsigned char* mine = lds + stage_last * STAGE + wv * WS_STG_WAVE;   // 34 KB of the 48 KB stage every wave has finished reading
        DPC_UNROLL
        for (int i = 0; i < 2; ++i) {
            stage_block(mine, acc[i], l31, lhi);
            wave_lds_fence();  // wave-private region: LDS operations of one wave complete in order, no workgroup barrier
            if (HAS_ADD && i == 0) fetch_addend(1, av1);  // the first pass's accumulators are dead: room for the second pass's addend
            u32x4 ov[8];
            DPC_UNROLL
            for (int it = 0; it < 8; ++it) {
                const int row_l = er + 4 * it;
                ov[it] = *(const u32x4*)(mine + row_l * WS_STG_ROW + cu * EPO * 2);
            }
            wave_lds_fence();  // the second pass overwrites what this one has just read
            DPC_UNROLL
            for (int it = 0; it < 8; ++it) {
                const int row = PAR ? par_row(ck, wv * 64 + i * 32 + er + 4 * it) : tile_row(mt, wv * 64 + i * 32 + er + 4 * it);
                if (row < g.M && col0 < p.Ncol) {
                    u32x4 o = ov[it];
                    if (HAS_ADD) {
                        const u32x4 a = i == 0 ? av0[it] : av1[it];
                        float sv[EPO];
                        DPC_UNROLL
                        for (int e = 0; e < EPO; ++e) sv[e] = unit_get<TO>(o, e) + unit_get<TO>(a, e);
                        o = unit_pack<TO>(sv);
                    }
                    *(u32x4*)((char*)p.out + ((long long)row * p.ldo + col0) * 2) = o;
                    if (!HAS_ADD) {  // batch-norm partial sums belong to forward convs; those have no residual addend
                        DPC_UNROLL
                        for (int e = 0; e < EPO; ++e) {
                            const float v = unit_get<TO>(o, e);
                            s1[e] += v;
                            s2[e] += v * v;
                        }
                    }
                }
            }
        }
    }

    if (!HAS_ADD && p.stats) {
        // 16 lanes x 4 compute waves hold partial sums of the same 8 columns: fold the 4 row-lanes of a wave
        // by shuffles, then the 4 waves through LDS (the loader waves only keep the barriers company).
        DPC_UNROLL
        for (int e = 0; e < EPO; ++e) {
            s1[e] += __shfl_xor(s1[e], 16); s1[e] += __shfl_xor(s1[e], 32);
            s2[e] += __shfl_xor(s2[e], 16); s2[e] += __shfl_xor(s2[e], 32);
        }
        float* red = (float*)lds;  // [4 waves][2][128]
        __syncthreads();
        if (wv < 4 && lane < 16) {
            DPC_UNROLL
            for (int e = 0; e < EPO; ++e) {
                red[(wv * 2 + 0) * BN + cu * EPO + e] = s1[e];
                red[(wv * 2 + 1) * BN + cu * EPO + e] = s2[e];
            }
        }
        __syncthreads();
        if (tid < BN) {
            const int col = n_tile * BN + tid;
            if (col < p.Ncol) {
                float a = 0.f, b = 0.f;
                DPC_UNROLL
                for (int w = 0; w < 4; ++w) {
                    a += red[(w * 2 + 0) * BN + tid];
                    b += red[(w * 2 + 1) * BN + tid];
                }
                p.stats[((long long)m_prog * 2 + 0) * p.Ncol + col] = a;
                p.stats[((long long)m_prog * 2 + 1) * p.Ncol + col] = b;
            }
        }
    }
}


// =====================================================================================================================
// Plane variant: 1x3x3 unit-stride convs (forward and input-gradient) over 16 x 16 planes -- layer2 of the 128 x 128
// configurations (backbone/resnet_2d3d.py:14-32: conv3x3x3 with a 1x3x3 kernel in the 2d blocks).
//
// igemm_ws_kernel re-reads every source row once per tap: 18 chunks x (32 KB of rows + 16 KB of weights) = 864 one-KB
// LDS-DMA pieces per 256-row tile, and the loaders cannot issue them faster than ~80 pieces/us per CU -- its chunk period on
// layer2 is 1.0 us against 0.43 us of MFMA work (40 % of peak, profiles/r02_r18_128_step_timeline.txt).  Here a tile IS one
// 16 x 16 plane and its source is staged ONCE as a zero-padded 18 x 18 patch, 64 channels at a time (41 pieces): the K order
// becomes (channel group, tap), all nine taps of a group read the same patch at tap-dependent offsets, and only the weights
// (16 pieces per chunk) stream through the ring -- 371 pieces per tile instead of 864, so the loop is bound by the matrix
// cores.  LDS: two patches (ping-pong by group, 41 KB each) + a ring of four 16 KB weight stages = 146 KB.
//   * loaders: after the barrier that publishes chunk c they issue <= 2 pieces of the NEXT group's patch (its buffer was read
//     last by the previous group, i.e. free since this group's first barrier) and then the weights of chunk c+3.  LDS-DMA of a
//     wave completes in issue order, so "weights of chunk c landed" (counted vmcnt: the ops issued after them are known)
//     implies the patch pieces issued before them landed -- a group's patch is complete six chunks before its first use.
//   * compute waves: as igemm_ws_kernel (fragment reads one step ahead, barrier between MFMA steps 2 and 3); the patch
//     position of a lane's row moves by a tap-dependent constant, and the XOR swizzle (slot = unit ^ ((patch column >> 1) & 7))
//     is re-derived per chunk (a dozen VALU operations against 32 MFMAs).
//   * epilogue: staged through the patch buffer of the tile's LAST group; the loaders refill that buffer only after the
//     first barrier of the next tile, which the compute waves reach after their epilogue.
// MEASURED DEAD END: the same scheme for the 3x3x3 convs of layer3 (four 8 x 8 planes of one output frame index per tile, one
// 50-piece patch set per (channel group, temporal tap), conflict-free slot swizzle ((column >> 1) + 4 (row & 1)) found by
// exhaustive search).  Leaving the A-operand DMA out of igemm_ws_kernel saves 123 of 424 us on that layer, but the variant
// with 9x fewer A pieces ran 431 us against 438 us (4 415 vs 4 415 clips/s, same box): what the probe removes with the DMA is
// also its LDS traffic and the data-dependent switching power, not just issue slots.  Not kept.
template <bool HAS_ADD>
__global__ __launch_bounds__(512, 2) void igemm_wsp_kernel(WsParams p) {
    typedef bf16_t TO;
    constexpr int BM = 256, BN = 128;
    constexpr int PWD = 18, NPP = PWD * PWD, NPIECE = (NPP + 7) / 8, PATCH = NPIECE * 1024;  // 324 positions, 41 pieces
    constexpr int BST = BN * 128, NSB = 4;
    constexpr int EPO = 8;
    constexpr int MAXP = (NPIECE + 3) / 4;  // patch pieces per loader wave
    static_assert(4 * WS_STG_WAVE <= PATCH, "epilogue staging fits a patch buffer");
    static_assert(2 * PATCH + NSB * BST <= 160 * 1024, "patches + ring fit the CU's LDS");
    __shared__ __attribute__((aligned(1024))) unsigned char lds[2 * PATCH + NSB * BST];
    unsigned char* const bring = lds + 2 * PATCH;

    const GatherGeom& g = p.g;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
#ifdef DPC_SIMT_EMU
    const int wv = tid >> 6;
#else
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
#endif
    const int n_tile = blockIdx.x / p.gm;
    const int m_prog = blockIdx.x % p.gm;
    const int G = g.Ci >> 6;       // channel groups of 64
    const int nkc = 9 * G;         // chunks per tile: (group, tap)
    const int my_tiles = (p.ntm - m_prog + p.gm - 1) / p.gm;
    const int total = my_tiles * nkc;
    const char* const zero = (const char*)dpc_zero16;

    if (wv >= 4) {
        // ------------------------------------------------------------------ loader waves
        const int lw = wv - 4;
        const int rl = lane >> 3;
        const BufRsrc rs_a = make_buf_rsrc(p.src, p.src_bytes);
        const BufRsrc rs_b = make_buf_rsrc(p.wgt, p.wgt_bytes);
        const int ub = (lane & 7) ^ lds_swz1(8 * lw + rl);   // weights: same stage image as igemm_ws_kernel
        unsigned wrow[4];
        DPC_UNROLL
        for (int i = 0; i < 4; ++i) {
            const int n = n_tile * BN + 8 * (lw + 4 * i) + rl;
            wrow[i] = n < p.Ncol ? (unsigned)(n * p.ldw + ub * 8) * 2u : DPC_BUF_OOB;
        }
        // patch piece j = lw + 4i holds patch positions 8j .. 8j+7; a lane fetches unit (slot ^ swizzle(patch column)) of its
        // position.  The offsets are relative to the plane: the same for every tile.
        unsigned poff[MAXP];
        DPC_UNROLL
        for (int i = 0; i < MAXP; ++i) {
            const int pp = 8 * (lw + 4 * i) + rl;
            const int pr = pp / PWD, pc = pp - pr * PWD;
            const bool ok = pp < NPP && pr >= 1 && pr <= 16 && pc >= 1 && pc <= 16;
            const int up = (lane & 7) ^ ((pc >> 1) & 7);
            poff[i] = ok ? (unsigned)((((pr - 1) * 16 + (pc - 1)) * g.src_ld + up * 8) * 2) : DPC_BUF_OOB;
        }
        const int n_mine = (NPIECE - lw + 3) / 4;
        int issued = 0;               // operations issued in the current slot
        auto issue_patch = [&](int gg, int i0, int i1) {   // pieces [i0, i1) of this wave, of global group gg
            const int tl = gg / G, grp = gg - tl * G;
            const int mt = WS_DBG(512) ? m_prog & 7 : m_prog + tl * p.gm;
            const unsigned soff = (unsigned)((mt * BM * g.src_ld + grp * 64) * 2);
            unsigned char* dst = lds + (gg & 1) * PATCH;
            DPC_UNROLL
            for (int i = 0; i < MAXP; ++i)
                if (i >= i0 && i < i1 && i < n_mine && !WS_DBG(4 | 128)) {
                    glds16_buf(rs_a, poff[i], soff, dst + (lw + 4 * i) * 1024, lane);
                    ++issued;
                }
        };
        auto issue_b = [&](int gc) {
            const int gg = gc / 9, tap = gc - gg * 9;
            const int grp = gg % G;
            const int kd = tap * g.Ci + grp * 64;
            unsigned char* st = bring + (gc % NSB) * BST;
            if (WS_DBG(4 | 256)) return;
            DPC_UNROLL
            for (int i = 0; i < 4; ++i) glds16_buf(rs_b, wrow[i], (unsigned)kd * 2u, st + (lw + 4 * i) * 1024, lane);
            issued += 4;
        };
        // A slot = what is issued between two chunk barriers: patch pieces first, the weights of chunk c+3 last.  The weights of
        // chunk c are the last operations of slot c-3, so "at most (slot c-2) + (slot c-1) operations outstanding" means landed.
        // The prologue counts as slots -3 (patch of group 0 + chunk 0), -2 (chunk 1), -1 (chunk 2).
        if (total > 0) {
            issue_patch(0, 0, MAXP);
            issue_b(0);
        }
        issued = 0;
        if (total > 1) issue_b(1);
        int o2 = issued;
        issued = 0;
        if (total > 2) issue_b(2);
        int o1 = issued;
        for (int gc = 0; gc < total; ++gc) {
            wait_vmcnt_upto(o1 + o2);
            if (!WS_DBG(32)) ws_barrier();  // chunk gc (and, at a group's first chunk, its patch) is published; every reader is done with chunk gc-1
            const int gg = gc / 9, q = gc - gg * 9;
            issued = 0;
            if (q < (MAXP + 1) / 2 && (gg + 1) * 9 < total) issue_patch(gg + 1, 2 * q, 2 * q + 2);
            if (gc + 3 < total) issue_b(gc + 3);
            o2 = o1;
            o1 = issued;
            if (q == 8 && (gg + 1) % G == 0) if (!WS_DBG(32)) ws_barrier();  // matches the compute waves' "tile fully read" barrier
        }
    }

    // ---------------------------------------------------------------------- compute waves
    const int l31 = lane & 31, lhi = lane >> 5;
    // A-fragment addresses.  Row r of the tile is pixel (r >> 4, r & 15); for tap (kh, kw) it reads patch position
    // (r>>4 + dh) * 18 + (r&15) + dw with (dh, dw) = (kh, kw) forward, (2-kh, 2-kw) input-gradient; the second fragment (+32 rows =
    // +2 image rows) is 36 positions = 4608 bytes further in the same patch column.  The 16-byte slots of a position are
    // XOR-swizzled by (patch column >> 1) & 7: a ds_read_b128 lane group is 8 + 8 lanes of two image rows
    // (MI355X_MICROARCH.md, LDS) whose columns c+{0..3,12..15} and c+{4..11} then cover 16 distinct (column parity, slot) pairs =
    // all 64 banks; swizzling by the position (row pitch 18) does not.  Everything that depends on the lane is tabulated per kw
    // (12 registers); the kh / patch-buffer part is a scalar added per chunk.
    // slot bits = ((2kk + lhi) ^ p7) << 4 = ((lhi ^ p7) << 4) ^ (kk << 5), and every base below is a multiple of 128: the address
    // of step kk is (chunk base + va[kw]) ^ (kk << 5) -- one register per kw instead of a 12-entry table.
    int va[3];
    {
        const int pos0 = ((wv * 64 + l31) >> 4) * PWD + (l31 & 15);
        DPC_UNROLL
        for (int kw = 0; kw < 3; ++kw) {
            const int dw = g.mode == 0 ? kw : 2 - kw;
            const int p7 = (((l31 & 15) + dw) >> 1) & 7;
            va[kw] = (pos0 + dw) * 128 + ((lhi ^ p7) << 4);
        }
    }
    const ldsa_t lds0 = ldsa(lds);
    const int vb = lds_unit_off(l31, lhi);   // B fragment of step 0; step kk: ^ (kk << 5) (stage bases are multiples of 16 KB)
    const int cu = lane & 15, er = lane >> 4;
    const int col0 = n_tile * BN + cu * EPO;
    float s1[EPO], s2[EPO];
    DPC_UNROLL
    for (int e = 0; e < EPO; ++e) { s1[e] = 0.f; s2[e] = 0.f; }

    int gc = 0;
#ifndef DPC_SIMT_EMU
    __builtin_amdgcn_s_setprio(3);   // the compute waves win the SIMD's issue arbitration against their loader partner (-1..2 %)
#endif
    for (int t = 0; wv < 4 && t < my_tiles; ++t) {
        const int mt = m_prog + t * p.gm;
        f32x16 acc[2][4];
        DPC_UNROLL
        for (int i = 0; i < 2; ++i)
            DPC_UNROLL
            for (int j = 0; j < 4; ++j)
                DPC_UNROLL
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
        u32x4 av0[8], av1[8];
        auto fetch_addend = [&](int i, u32x4 (&dst)[8]) {
            DPC_UNROLL
            for (int it = 0; it < 8; ++it) {
                const int row = mt * BM + wv * 64 + i * 32 + er + 4 * it;
                const bool ok = row < g.M && col0 < p.Ncol;
                const char* a = (const char*)p.addend + ((long long)row * p.ldo + col0) * 2;
                // rows outside the tensor read its unit 0 (never used: their outputs are not stored).  Not the zero page: a select between
                // a kernel-argument pointer and a device constant is a GENERIC pointer, its flat loads count on lgkmcnt as well and hipcc
                // waits lgkmcnt(0) for them -- behind the hand-issued fragment reads of the chunk (scripts/asm_drain_lint.py)
                dst[it] = *(const u32x4*)(ok ? a : (const char*)p.addend);
            }
        };
        // Chunk loop: (channel group, kh) outer, kw unrolled (its address table is then a compile-time choice).  Four steps per
        // chunk, each = 8 MFMAs with the next step's reads interleaved; the barrier that publishes chunk c+1 sits in front of step
        // 3 of chunk c (whose interleaved reads are the first fragments of chunk c+1), after this wave's last reads of chunk c
        // have landed -- the same protocol as igemm_ws_kernel.
        {
            FragSet f0, f1;
            if (!WS_DBG(32)) ws_barrier();  // first chunk of the tile (and its patch) published
            ldsa_t a_row = lds0 + ((t * G) & 1) * PATCH + (g.mode == 0 ? 0 : 2) * (PWD * 128);
            ldsa_t b_st = lds0 + 2 * PATCH + (gc % NSB) * BST;
            ldsa_t pa = a_row + va[0];
            ldsa_t pb = b_st + vb;
            frag_read_p2<4608>(f0, pa, pb);
            for (int gk = 0; gk < 3 * G; ++gk) {   // (group, kh)
                const int grp = gk / 3, kh = gk - grp * 3;
                const bool last_gk = gk + 1 == 3 * G;
                const int kh_n = kh == 2 ? 0 : kh + 1, gg_n = t * G + (kh == 2 ? grp + 1 : grp);
                const ldsa_t a_row_n = lds0 + (gg_n & 1) * PATCH + (g.mode == 0 ? kh_n : 2 - kh_n) * (PWD * 128);
                static_for<3>([&](auto KWc) {
                    constexpr int KW = decltype(KWc)::value;
                    const bool last = last_gk && KW == 2;   // last chunk of the tile
                    if (HAS_ADD && last) fetch_addend(0, av0);
                    step_il<true, 4608>(acc, f0, f1, pa ^ 32, pb ^ 32);
                    step_il<true, 4608>(acc, f1, f0, pa ^ 64, pb ^ 64);
                    step_il<true, 4608>(acc, f0, f1, pa ^ 96, pb ^ 96);
                    frag_wait<0>(f1);   // this wave's last reads of the chunk have landed
                    if (!WS_DBG(32)) ws_barrier();  // next chunk published -- or, after the tile's last chunk, "tile fully read"
                    ++gc;
                    b_st = lds0 + 2 * PATCH + (gc % NSB) * BST;
                    if (KW == 2) a_row = a_row_n;
                    pa = a_row + va[(KW + 1) % 3];
                    pb = b_st + vb;
                    // after the tile's last chunk these reads fetch nothing useful (the next tile re-reads after ITS first barrier):
                    // keeping them unconditional keeps the MFMAs out of a branch -- accumulators that flow through both arms
                    // of a branch are copied at the join (hundreds of spilled registers)
                    step_il<true, 4608>(acc, f1, f0, pa, pb);
                });
            }
            WS_RETIRE_TAIL_READS(f0);   // see the macro: reads past the tile's last chunk land before their registers are handed on
        }

        // ---- epilogue: two passes of 32 rows through this wave's 8 KB of the last group's patch buffer
        unsigned char* mine = lds + ((t * G + G - 1) & 1) * PATCH + wv * WS_STG_WAVE;
        if (WS_DBG(2)) {  // probe: no epilogue at all (the accumulators stay live through one store)
            if (acc[0][0][0] == 12345.f) *(float*)p.out = acc[1][3][5];
            continue;
        }
        DPC_UNROLL
        for (int i = 0; i < 2; ++i) {
            stage_block(mine, acc[i], l31, lhi);
            wave_lds_fence();
            if (HAS_ADD && i == 0) fetch_addend(1, av1);
            u32x4 ov[8];
            DPC_UNROLL
            for (int it = 0; it < 8; ++it) {
                const int row_l = er + 4 * it;
                ov[it] = *(const u32x4*)(mine + row_l * WS_STG_ROW + cu * EPO * 2);
            }
            wave_lds_fence();
            DPC_UNROLL
            for (int it = 0; it < 8; ++it) {
                const int row = mt * BM + wv * 64 + i * 32 + er + 4 * it;
                if (row < g.M && col0 < p.Ncol && !WS_DBG(1)) {
                    u32x4 o = ov[it];
                    if (HAS_ADD) {
                        const u32x4 a = i == 0 ? av0[it] : av1[it];
                        float sv[EPO];
                        DPC_UNROLL
                        for (int e = 0; e < EPO; ++e) sv[e] = unit_get<TO>(o, e) + unit_get<TO>(a, e);
                        o = unit_pack<TO>(sv);
                    }
                    *(u32x4*)((char*)p.out + ((long long)row * p.ldo + col0) * 2) = o;
                    if (!HAS_ADD) {
                        DPC_UNROLL
                        for (int e = 0; e < EPO; ++e) {
                            const float v = unit_get<TO>(o, e);
                            s1[e] += v;
                            s2[e] += v * v;
                        }
                    }
                }
            }
        }
    }

    if (!HAS_ADD && p.stats) {
        DPC_UNROLL
        for (int e = 0; e < EPO; ++e) {
            s1[e] += __shfl_xor(s1[e], 16); s1[e] += __shfl_xor(s1[e], 32);
            s2[e] += __shfl_xor(s2[e], 16); s2[e] += __shfl_xor(s2[e], 32);
        }
        float* red = (float*)lds;  // [4 waves][2][128]
        __syncthreads();
        if (wv < 4 && lane < 16) {
            DPC_UNROLL
            for (int e = 0; e < EPO; ++e) {
                red[(wv * 2 + 0) * BN + cu * EPO + e] = s1[e];
                red[(wv * 2 + 1) * BN + cu * EPO + e] = s2[e];
            }
        }
        __syncthreads();
        if (tid < BN) {
            const int col = n_tile * BN + tid;
            if (col < p.Ncol) {
                float a = 0.f, b = 0.f;
                DPC_UNROLL
                for (int w = 0; w < 4; ++w) {
                    a += red[(w * 2 + 0) * BN + tid];
                    b += red[(w * 2 + 1) * BN + tid];
                }
                p.stats[((long long)m_prog * 2 + 0) * p.Ncol + col] = a;
                p.stats[((long long)m_prog * 2 + 1) * p.Ncol + col] = b;
            }
        }
    }
}


// =====================================================================================================================
// Strided input-gradient over 16 x 16 gradient planes ("wsd", round 6): the input-gradient of layer2.0.conv1 of the 128 x 128
// configurations -- Conv3d(64, 128, (1,3,3), stride (1,2,2), pad (0,1,1)), backbone/resnet_2d3d.py:24-32,241-244 -- which the
// generic kernel ran in 520 us (0.37 PFLOP/s: parity-class tiles of 2..8 K chunks, each with its own decode, first-chunk round
// trip and epilogue; HBM floor 160 us) and igemm_ws_kernel<., true> on its half-empty 128-column tile in 643.
//
// The transposed convolution as a DENSE-ish unit-stride one.  Output position (2a + py, 2b + px) of class (py, px) reads the
// gradient positions (a + sy, b + sx) with sy <= py, sx <= px through tap ky = py ? (sy ? 0 : 2) : 1 (kx alike): over the 16 x 16
// positions (a, b) of a gradient plane that is a 2 x 2 unit-stride convolution from 128 channels to 4 classes x 64 columns whose
// weight matrix has 9 non-zero 128 x 64 blocks of 16.  igemm_wsp_kernel's machinery runs it: a tile IS one gradient plane staged
// once as a zero-padded 18 x 18 patch per 64-channel group (the shifts are fragment-address constants), the weights stream
// through the ring, rows of wd[ci][tap][co] picked per class and shift by the loaders (a class that does not see a shift gets
// zero rows from the buffer resource), and the epilogue scatters a tile row's two 64-column halves to the two output positions
// of its classes.  Classes are paired so that both kinds of column tile see at least three shifts (a patch must be issued three
// chunk slots before its first use): A = {(1,1), (0,0)} over the four shifts, B = {(1,0), (0,1)} over three -- 14 chunks per plane
// for 9 chunks' worth of products (64 % useful; the {(1,1),(1,0)} / {(0,1),(0,0)} pairing has 12 but a two-chunk group).  A
// workgroup serves ONE kind (blockIdx < gm: A), its planes strided by the number of workgroups of its kind; the split between
// the kinds follows their chunk counts (plus the epilogue both pay per plane).
__global__ __launch_bounds__(512, 2) void igemm_wsd_kernel(WsParams p) {
    typedef bf16_t TO;
    constexpr int BM = 256, BN = 128;
    constexpr int PWD = 18, NPP = PWD * PWD, NPIECE = (NPP + 7) / 8, PATCH = NPIECE * 1024;
    constexpr int BST = BN * 128, NSB = 4;
    constexpr int EPO = 8;
    constexpr int MAXP = (NPIECE + 3) / 4;
    static_assert(4 * WS_STG_WAVE <= PATCH, "epilogue staging fits a patch buffer");
    static_assert(2 * PATCH + NSB * BST <= 160 * 1024, "patches + ring fit the CU's LDS");
    __shared__ __attribute__((aligned(1024))) unsigned char lds[2 * PATCH + NSB * BST];
    unsigned char* const bring = lds + 2 * PATCH;

    const GatherGeom& g = p.g;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
#ifdef DPC_SIMT_EMU
    const int wv = tid >> 6;
#else
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
#endif
    const bool kindB = (int)blockIdx.x >= p.gm;
    const int m_prog = kindB ? (int)blockIdx.x - p.gm : (int)blockIdx.x;
    const int gmk = kindB ? p.ntn : p.gm;          // workgroups of this kind (p.ntn carries kind B's count)
    const int CPG = kindB ? 3 : 4;                 // chunks (= shifts) per channel group
    const int G = g.Ci >> 6;
    const int nkc = CPG * G;
    const int my_tiles = m_prog < p.ntm ? (p.ntm - m_prog + gmk - 1) / gmk : 0;
    const int total = my_tiles * nkc;
    // classes of the two 64-column halves: kind A (1,1) | (0,0), kind B (1,0) | (0,1)
    const int py0 = 1, px0 = kindB ? 0 : 1, py1 = 0, px1 = kindB ? 1 : 0;
    // shift of chunk q of a group: A (0,0) (0,1) (1,0) (1,1); B (0,0) (1,0) (0,1)
    auto shift_of = [&](int q, int& sy, int& sx) {
        if (kindB) { sy = q == 1; sx = q == 2; } else { sy = q >> 1; sx = q & 1; }
    };

    if (wv >= 4) {
        // ------------------------------------------------------------------ loader waves
        const int lw = wv - 4;
        const int rl = lane >> 3;
        const BufRsrc rs_a = make_buf_rsrc(p.src, p.src_bytes);
        const BufRsrc rs_b = make_buf_rsrc(p.wgt, p.wgt_bytes);
        const int ub = (lane & 7) ^ lds_swz1(8 * lw + rl);
        // weight-tile rows 8 (lw + 4 i) + rl: i = 0, 1 are the first class half, i = 2, 3 the second; row inside the half = output column ci
        unsigned wrow[4];
        DPC_UNROLL
        for (int i = 0; i < 4; ++i) {
            const int n = 8 * (lw + 4 * (i & 1)) + rl;
            wrow[i] = (unsigned)(n * p.ldw + ub * 8) * 2u;
        }
        unsigned poff[MAXP];
        DPC_UNROLL
        for (int i = 0; i < MAXP; ++i) {
            const int pp = 8 * (lw + 4 * i) + rl;
            const int pr = pp / PWD, pc = pp - pr * PWD;
            const bool ok = pp < NPP && pr >= 1 && pr <= 16 && pc >= 1 && pc <= 16;
            const int up = (lane & 7) ^ ((pc >> 1) & 7);
            poff[i] = ok ? (unsigned)((((pr - 1) * 16 + (pc - 1)) * g.src_ld + up * 8) * 2) : DPC_BUF_OOB;
        }
        const int n_mine = (NPIECE - lw + 3) / 4;
        int issued = 0;
        auto issue_patch = [&](int gg, int i0, int i1) {
            const int tl = gg / G, grp = gg - tl * G;
            const int mt = m_prog + tl * gmk;
            const unsigned soff = (unsigned)((mt * BM * g.src_ld + grp * 64) * 2);
            unsigned char* dst = lds + (gg & 1) * PATCH;
            DPC_UNROLL
            for (int i = 0; i < MAXP; ++i)
                if (i >= i0 && i < i1 && i < n_mine) {
                    glds16_buf(rs_a, poff[i], soff, dst + (lw + 4 * i) * 1024, lane);
                    ++issued;
                }
        };
        auto issue_b = [&](int gc) {
            const int gg = gc / CPG, q = gc - gg * CPG;
            const int grp = gg % G;
            int sy, sx;
            shift_of(q, sy, sx);
            unsigned char* st = bring + (gc % NSB) * BST;
            DPC_UNROLL
            for (int i = 0; i < 4; ++i) {
                const int py = i < 2 ? py0 : py1, px = i < 2 ? px0 : px1;
                const bool valid = sy <= py && sx <= px;
                const int ky = py ? (sy ? 0 : 2) : 1, kx = px ? (sx ? 0 : 2) : 1;
                const unsigned kd = (unsigned)((ky * 3 + kx) * g.Ci + grp * 64);
                glds16_buf(rs_b, valid ? wrow[i] : DPC_BUF_OOB, kd * 2u, st + (lw + 4 * i) * 1024, lane);
            }
            issued += 4;
        };
        // slots as in igemm_wsp_kernel: what is issued between two chunk barriers, patch pieces first, the weights of chunk c+3 last;
        // "at most (slot c-2) + (slot c-1) operations outstanding" = the weights of chunk c, and every patch piece issued before them,
        // have landed.  A group has only CPG = 4 / 3 chunks, and its patch must be complete at its first chunk's barrier, i.e. issued
        // in slots <= first - 3: all of the NEXT group's pieces go out in the first CPG - 2 slots of this group.
        const int pslots = CPG - 2, ppc = (MAXP + pslots - 1) / pslots;
        if (total > 0) {
            issue_patch(0, 0, MAXP);
            issue_b(0);
        }
        issued = 0;
        if (total > 1) issue_b(1);
        int o2 = issued;
        issued = 0;
        if (total > 2) issue_b(2);
        int o1 = issued;
        for (int gc = 0; gc < total; ++gc) {
            wait_vmcnt_upto(o1 + o2);
            ws_barrier();  // chunk gc (and, at a group's first chunk, its patch) is published; every reader is done with chunk gc-1
            const int gg = gc / CPG, q = gc - gg * CPG;
            issued = 0;
            if (q < pslots && (gg + 1) * CPG < total) issue_patch(gg + 1, ppc * q, ppc * q + ppc);
            if (gc + 3 < total) issue_b(gc + 3);
            o2 = o1;
            o1 = issued;
            if (q == CPG - 1 && (gg + 1) % G == 0) ws_barrier();  // matches the compute waves' "tile fully read" barrier
        }
        return;
    }

    // ---------------------------------------------------------------------- compute waves
    const int l31 = lane & 31, lhi = lane >> 5;
    // A fragments as in igemm_wsp_kernel: tile row r = gradient position (r >> 4, r & 15); shift (sy, sx) reads patch position
    // (r >> 4 + 1 + sy) * 18 + (r & 15) + 1 + sx; slot swizzle by (patch column >> 1) & 7
    int va[2];
    {
        const int pos0 = ((wv * 64 + l31) >> 4) * PWD + (l31 & 15);
        DPC_UNROLL
        for (int sx = 0; sx < 2; ++sx) {
            const int dw = 1 + sx;
            const int p7 = (((l31 & 15) + dw) >> 1) & 7;
            va[sx] = (pos0 + dw) * 128 + ((lhi ^ p7) << 4);
        }
    }
    const ldsa_t lds0 = ldsa(lds);
    const int vb = lds_unit_off(l31, lhi);
    const int cu = lane & 15, er = lane >> 4;
    const int hpy = (cu >> 3) ? py1 : py0, hpx = (cu >> 3) ? px1 : px0;   // class of this lane's output unit
    const int ccol = (cu & 7) * EPO;

    int gc = 0;
#ifndef DPC_SIMT_EMU
    __builtin_amdgcn_s_setprio(3);
#endif
    for (int t = 0; t < my_tiles; ++t) {
        const int mt = m_prog + t * gmk;
        f32x16 acc[2][4];
        DPC_UNROLL
        for (int i = 0; i < 2; ++i)
            DPC_UNROLL
            for (int j = 0; j < 4; ++j)
                DPC_UNROLL
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
        {
            FragSet f0, f1;
            ws_barrier();  // first chunk of the tile (and its patch) published
            int grp = 0, q = 0;
            ldsa_t a_pat = lds0 + ((t * G) & 1) * PATCH;
            ldsa_t pa = a_pat + PWD * 128 + va[0];   // shift (0, 0): patch row + 1
            ldsa_t pb = lds0 + 2 * PATCH + (gc % NSB) * BST + vb;
            frag_read_p2<4608>(f0, pa, pb);
            for (int c = 0; c < nkc; ++c) {
                step_il<true, 4608>(acc, f0, f1, pa ^ 32, pb ^ 32);
                step_il<true, 4608>(acc, f1, f0, pa ^ 64, pb ^ 64);
                step_il<true, 4608>(acc, f0, f1, pa ^ 96, pb ^ 96);
                frag_wait<0>(f1);   // this wave's last reads of the chunk have landed
                ws_barrier();  // next chunk published -- or, after the tile's last chunk, "tile fully read"
                ++gc;
                if (++q == CPG) { q = 0; ++grp; }
                int sy, sx;
                shift_of(q, sy, sx);
                a_pat = lds0 + ((t * G + grp) & 1) * PATCH;
                pa = a_pat + (1 + sy) * (PWD * 128) + (sx ? va[1] : va[0]);
                pb = lds0 + 2 * PATCH + (gc % NSB) * BST + vb;
                // after the tile's last chunk these reads fetch nothing useful (the next tile re-reads after ITS first barrier):
                // unconditional, so that the MFMAs stay out of a branch
                step_il<true, 4608>(acc, f1, f0, pa, pb);
            }
            WS_RETIRE_TAIL_READS(f0);
        }

        // ---- epilogue: two passes of 32 rows through this wave's 8 KB of the last group's patch buffer; a row's two 64-column
        // halves go to the output positions of their classes
        unsigned char* mine = lds + ((t * G + G - 1) & 1) * PATCH + wv * WS_STG_WAVE;
        DPC_UNROLL
        for (int i = 0; i < 2; ++i) {
            stage_block(mine, acc[i], l31, lhi);
            wave_lds_fence();
            u32x4 ov[8];
            DPC_UNROLL
            for (int it = 0; it < 8; ++it) {
                const int row_l = er + 4 * it;
                ov[it] = *(const u32x4*)(mine + row_l * WS_STG_ROW + cu * EPO * 2);
            }
            wave_lds_fence();
            DPC_UNROLL
            for (int it = 0; it < 8; ++it) {
                const int r = wv * 64 + i * 32 + er + 4 * it;
                const int a = r >> 4, b = r & 15;
                const long long orow = ((long long)mt * g.RH + 2 * a + hpy) * g.RW + 2 * b + hpx;
                if (mt < p.ntm) *(u32x4*)((char*)p.out + (orow * p.ldo + ccol) * 2) = ov[it];
            }
        }
    }
}

static int env_int(const char* name, int dflt) {
    const char* e = getenv(name);
    return e ? atoi(e) : dflt;
}
// knobs for experiments and for the small shapes of the test tiers
static int ws_enabled() { static int v = env_int("DPC_IGEMM_WS", 1); return v; }
static int ws_min_rows() { static int v = env_int("DPC_IGEMM_WS_MINROWS", 256 * 64); return v; }
static int ws_max_programs() { static int v = env_int("DPC_IGEMM_WS_GM", 256); return v; }

static bool ws_plan(const dpc_conv_desc* d, WsParams* p) {
    if (!ws_enabled()) return false;
    if (d->dtype_in != DPC_BF16 || d->dtype_out != DPC_BF16) return false;
    if (d->Co < 64 || d->Co % 8 || d->ldo % 8 || d->ldw % 8) return false;
    GatherGeom& g = p->g;
    if (make_gather_geom(d, &g)) return false;
    if (g.Kp % 64 || (g.taps > 1 && g.Ci % 64)) return false;
    const bool unit_strides = g.st == 1 && g.sh == 1 && g.sw == 1;
    // byte offsets stay below DPC_BUF_OOB (2 GB), which therefore is out of range for the buffer resource
    const bool fits32 = (long long)(g.M / (g.RT * g.RH * g.RW)) * g.ST * g.SH * g.SW * g.src_ld < (1ll << 30);
    if (!fits32 || g.KT + g.KH + g.KW > 32) return false;
    p->parity = 0;
    int par_tiles = 0;
    if (!(g.mode == 0 || unit_strides)) {   // strided input-gradient: parity classes on 256-row tiles
        static const int par_on = env_int("DPC_IGEMM_WS_PAR", 1);
        if (!par_on || g.taps == 1 || !dpc_plan_parity(g, p->par, 64, 256, false, &par_tiles)) return false;
        if (!p->par.pk_ok) return false;
        for (int c = 0; c < p->par.ncls; ++c)   // a class without taps would be a tile without chunks: not a case of this kernel's barrier sequence
            if (p->par.cnt[0][c / (g.sw * g.sh)] * p->par.cnt[1][(c / g.sw) % g.sh] * p->par.cnt[2][c % g.sw] == 0) return false;
        p->parity = 1;
    }
    // 64 output columns (layer2.0.conv1's input-gradient, K = 128 per tap: tiles of 2..8 chunks) would run on the 128-column tile
    // with the upper weight rows zero-filled by the buffer resource (DPC_IGEMM_WS_PAR_MINCO=64).  Measured slower than the generic
    // kernel (643 vs 566 us): this kernel's tile boundary -- loader decode, "tile fully read" barrier, epilogue on the compute
    // waves -- costs ~5 us, which 13-chunk tiles (layer3.0: 395 -> 298 us) amortise and 4.5-chunk tiles do not.
    static const int par_min_co = env_int("DPC_IGEMM_WS_PAR_MINCO", 128);
    if (d->Co < (p->parity ? par_min_co : 128)) return false;
    if (g.M < ws_min_rows()) return false;  // too few 256-row tiles to feed 256 CUs: the 128-row kernel balances better
    p->Ncol = d->Co; p->ldw = d->ldw; p->ldo = d->ldo;
    const long long wbytes = ((long long)(d->Co - 1) * d->ldw + g.Kp) * 2;
    if (wbytes >= (1ll << 31)) return false;
    p->src_bytes = (unsigned)((long long)(g.M / (g.RT * g.RH * g.RW)) * g.ST * g.SH * g.SW * g.src_ld * 2);
    p->wgt_bytes = (unsigned)wbytes;
    p->ntn = (d->Co + 127) / 128;
    p->ntm = p->parity ? par_tiles : (g.M + 255) / 256;
    p->tgroup = 0; p->hw = 1; p->nclip = 0; p->tpt = 1; p->cpkt = 1; p->d_hw = make_fastdiv(1);
    {
        static const int plane_on = env_int("DPC_IGEMM_WS_PLANE", 1);
        p->plane = !p->parity && plane_on && g.KT == 1 && g.KH == 3 && g.KW == 3 && g.pt == 0 && g.ph == 1 && g.pw == 1 && unit_strides && g.RT == g.ST &&
                   g.RH == 16 && g.RW == 16 && g.SH == 16 && g.SW == 16 && g.Ci % 64 == 0;
    }
    {
        static const int tg_on = env_int("DPC_IGEMM_WS_TGROUP", 1);
        const int hw = g.RH * g.RW;
        if (tg_on && !p->parity && g.KT > 1 && unit_strides && g.RT == g.ST && g.RT > 1) {
            p->tgroup = 1; p->hw = hw; p->d_hw = make_fastdiv((uint32_t)hw);
            p->nclip = g.M / (g.RT * hw);
            p->tpt = (int)(((long long)p->nclip * hw + 255) / 256);
            p->cpkt = g.KH * g.KW * (g.Ci / 64);
            p->ntm = g.RT * p->tpt;
        }
    }
    // one resident workgroup per CU (144 KB of LDS); XCD x gets workgroups x, x+8, ...: keep gm a multiple
    // of 8 so that the ntn column tiles of one row tile (blockIdx differing by gm) share an L2
    int gm = dpc_persistent_grid(ws_max_programs()) / p->ntn;
    if (gm < 1) gm = 1;
    if (gm > p->ntm) gm = p->ntm;
    if (gm >= 8) gm &= ~7;
    p->gm = gm;
    if (p->parity) dpc_parity_rounds(p->par, gm);
    return true;
}

// igemm_wsd_kernel: bf16 input-gradient of a 1 x 3 x 3 convolution with stride (1, 2, 2), pad (0, 1, 1) from 128 to 64 channels over
// 16 x 16 gradient planes (32 x 32 output planes).  p->gm / p->ntn = workgroups of kind A / B, p->ntm = planes.
static bool wsd_plan(const dpc_conv_desc* d, WsParams* p) {
    static const int on = env_int("DPC_IGEMM_WSD", 1);
    if (!on || !ws_enabled()) return false;
    if (d->dtype_in != DPC_BF16 || d->dtype_out != DPC_BF16 || d->mode != 1) return false;
    if (d->KT != 1 || d->KH != 3 || d->KW != 3 || d->st != 1 || d->sh != 2 || d->sw != 2 || d->pt != 0 || d->ph != 1 || d->pw != 1) return false;
    if (d->RT != d->ST || d->SH != 16 || d->SW != 16 || d->RH != 32 || d->RW != 32) return false;
    if (d->Ci != 128 || d->src_ld != 128 || d->Co != 64 || d->ldo % 8 || d->ldo < 64 || d->ldw % 8 || d->ldw < 9 * 128) return false;
    GatherGeom& g = p->g;
    if (make_gather_geom(d, &g)) return false;
    const long long planes = (long long)d->N * d->RT;
    if (planes * 256 * 128 * 2 >= (1ll << 30) || planes * 1024 * (long long)d->ldo * 2 >= (1ll << 40)) return false;   // 32-bit source offsets
    static const int min_planes = env_int("DPC_WSD_MINPLANES", 512);
    if (planes < min_planes) return false;   // too few planes to feed both kinds of workgroup: the generic kernel balances better
    p->parity = 0; p->plane = 0; p->tgroup = 0; p->hw = 1; p->nclip = 0; p->tpt = 1; p->cpkt = 1; p->d_hw = make_fastdiv(1);
    p->Ncol = d->Co; p->ldw = d->ldw; p->ldo = d->ldo;
    p->src_bytes = (unsigned)(planes * 256 * 128 * 2);
    p->wgt_bytes = (unsigned)(((long long)(d->Co - 1) * d->ldw + 9 * 128) * 2);
    p->ntm = (int)planes;
    // kind A walks 4 shifts x 2 groups = 8 chunks per plane, kind B 6, and both pay one epilogue (~2 chunks' worth): 10 : 8
    static const int a_share = env_int("DPC_WSD_A_PERMILLE", 556);
    const int wgs = dpc_persistent_grid(ws_max_programs());
    int gma = (int)((long long)wgs * a_share / 1000);
    if (gma < 1) gma = 1;
    if (gma > wgs - 1) gma = wgs - 1;
    p->gm = gma;
    p->ntn = wgs - gma;
    return wgs >= 2;
}

int dpc_conv_ws_rows(const dpc_conv_desc* d) {
    WsParams p;
    if (!d || !ws_plan(d, &p)) return 0;
    return p.gm;
}

// returns 1 when the shape is not served by this kernel
int dpc_conv_ws_try(const dpc_conv_desc* d, const void* src, const void* wgt, void* out, const void* addend, float* stats,
                    const EpiExtra& epi, hipStream_t stream) {
    WsParams p;
    if (!addend && !stats && !epi_any(epi) && wsd_plan(d, &p)) {
        if (((uintptr_t)out % 16) || ((uintptr_t)src % 16) || ((uintptr_t)wgt % 16)) return 1;
        p.src = src; p.wgt = wgt; p.out = out; p.addend = nullptr; p.stats = nullptr; p.dbg = 0;
        DPC_LAUNCH(igemm_wsd_kernel, dim3((unsigned)(p.gm + p.ntn)), dim3(512), stream, p);
        return dpc_launch_status();
    }
    if (!ws_plan(d, &p)) return 1;
    if (epi_any(epi)) return 1;  // the fused backward pieces of dpc_conv_igemm_ex are not built into these kernels (yet)
    if (addend && stats) return 1;  // not a combination of this path: the generic kernel serves it
    if (p.parity && (addend || stats)) return 1;
    if (((uintptr_t)out % 16) || ((uintptr_t)addend % 16) || ((uintptr_t)src % 16) || ((uintptr_t)wgt % 16)) return 1;
    p.src = src; p.wgt = wgt; p.out = out; p.addend = addend; p.stats = stats;
#ifdef DPC_WS_PROBE
    p.dbg = getenv("DPC_WS_DBG") ? atoi(getenv("DPC_WS_DBG")) : 0;
#else
    p.dbg = 0;
#endif
    dim3 grid((unsigned)(p.gm * p.ntn)), block(512);
    if (p.parity) {
        DPC_LAUNCH((igemm_ws_kernel<false, true>), grid, block, stream, p);
    } else if (p.plane) {
        if (addend) {
            DPC_LAUNCH((igemm_wsp_kernel<true>), grid, block, stream, p);
        } else {
            DPC_LAUNCH((igemm_wsp_kernel<false>), grid, block, stream, p);
        }
    } else if (addend) {
        DPC_LAUNCH((igemm_ws_kernel<true>), grid, block, stream, p);
    } else {
        DPC_LAUNCH((igemm_ws_kernel<false>), grid, block, stream, p);
    }
    return dpc_launch_status();
}
