// conv_igemm.hip -- implicit-GEMM convolution / NT GEMM on the gfx950 matrix cores.
//
// One kernel serves every conv on the DPC-RNN path (stem via space-to-depth, 1x3x3,
// 3x3x3, strided 1x1x1, their input-gradients, the 1x1 ConvGRU/pred GEMMs and the
// score matmul).  Reference call sites: backbone/resnet_2d3d.py:14-32,211,241-244,
// backbone/convrnn.py:13-15, dpc/model_3d.py:36-40,83.
//
// Tiling (wave64, 4 waves = 2(M) x 2(N) per workgroup):
//   block tile 128 x BN (BN = 64 | 128), K advanced in 128-byte chunks (32 f32 / 64 bf16);
//   each wave owns 64 x BN/2 = 2 x (BN/64) MFMA 32x32 accumulators;
//   operands staged global -> VGPR -> LDS (double buffered, one barrier per chunk);
//   LDS rows are 128 B with the unit slot XOR-swizzled by (row>>1)&7 so that the
//   ds_read_b128 fragment reads of 16 rows x same unit hit all 64 banks once
//   (measured SQ_LDS_BANK_CONFLICT = 0);
//   f32 uses v_mfma_f32_32x32x2_f32 (exact f32, parity mode), bf16 v_mfma_f32_32x32x16_bf16.
//   K order inside a chunk is permuted identically for A and B (a lane's 16-byte unit
//   feeds its own k-group), which leaves the dot products unchanged.
// The first version of this kernel spent 15 VALU instructions per MFMA (rocprofv3
// SQ_INSTS_VALU / SQ_INSTS_MFMA) and ran issue-bound at ~27 % MFMA utilisation, so:
//   * GATHER 1/2 (affine gather: forward conv or unit-stride input-gradient): a row keeps one
//     element offset and one validity bit-mask per tile; a chunk costs and+cmp+add per row.
//     GATHER 1 additionally has one tap per chunk (Ci >= chunk) => tap decode is wave-uniform SALU.
//   * padded / out-of-range units are fetched from a 16-byte zero page instead of being masked;
//   * the K loop is unrolled by two so LDS buffer offsets are instruction immediates, fragment and
//     staging addresses are loop-invariant registers;
//   * the epilogue stages the tile through LDS and leaves as whole 16-byte units (coalesced rows),
//     the residual/addend comes in the same way, batch-norm partial sums are taken from the
//     stored (rounded) values.
// A workgroup walks m-tiles m_prog, m_prog+gm, ... so the per-channel batch-norm partial
// sums accumulate in registers and leave as ONE row of `stats` per program: deterministic.
#include "conv_common.h"

struct IGemmParams {
    GatherGeom g;
    ParityInfo par;
    const void* src;
    const void* wgt;
    void* out;
    const void* addend;
    float* stats;
    EpiExtra epi;
    int Ncol, ldw, ldo;
    int gm, ntn, ntm;
    int vec_out;  // Ncol and ldo are multiples of the 16-byte output unit
    // split-K (dpc_gemm_nt_splitk): workgroup group ks reduces K chunks [ks*kcps, (ks+1)*kcps) into slab ks of `out`
    int nks, kcps;
    long long slab;  // elements of TO between slabs
    int xcd_cols;    // column tiles of a row tile on one XCD (launch_igemm_bn)
    int x6;          // f32 operands: contractions as bf16x6 (dpc_rt.h) instead of f32 MFMA chains
};

#ifndef DPC_IGEMM_DMA
#define DPC_IGEMM_DMA 1
#endif
#ifndef DPC_IGEMM_PIPE2
#define DPC_IGEMM_PIPE2 1   // main loop with two chunks in flight (0: the round-1 loop, one chunk in flight)
#endif

// EPI: the fused backward pieces of dpc_conv_igemm_ex in the epilogue (their registers cost the plain instantiation occupancy /
// spills when they are merely run-time options: +50 % on the strided input-gradients, measured)
template <class T, class TO, int BN, int GATHER, bool EPI = false>
__global__ __launch_bounds__(256, (BN == 64 ? 3 : 2)) void igemm_kernel(IGemmParams p) {
    // DMA: operand units go global -> LDS directly (global_load_lds_dwordx4), no VGPR staging and no
    // ds_write.  The LDS destination of a wave is lane-linear (8 rows x 8 slots), so the XOR swizzle
    // moves to the SOURCE: the lane that owns slot s of row r fetches logical unit s ^ swz(r).
    constexpr bool DMA = DPC_IGEMM_DMA != 0;
    constexpr int EPU = Elt<T>::PER16;
    constexpr int BKE = 8 * EPU;
    constexpr int BM = 128;
    constexpr int NT = BN / 64;     // 32-wide n-tiles per wave
    constexpr int BROWS = BN / 32;  // B-tile rows per thread
    constexpr int BUF = (BM + BN) * 128;
    constexpr int EPO = 16 / (int)sizeof(TO);          // output elements per 16-byte unit
    constexpr int UPR = BN / EPO;                      // output units per tile row
    constexpr int OIT = BM * UPR / 256;                // output units per thread
    __shared__ __attribute__((aligned(16))) unsigned char lds[2 * BUF];
    static_assert(BM * BN * (int)sizeof(TO) <= 2 * BUF, "epilogue staging must fit the operand buffers");

    const GatherGeom& g = p.g;
    const int tid = threadIdx.x;
    const int r0 = tid >> 3;
    const int u = DMA ? ((tid & 7) ^ lds_swz1(r0)) : (tid & 7);  // logical 16-byte unit of the chunk this thread moves
    const int lane = tid & 63, wv = tid >> 6;
    const int wm = wv >> 1, wn = wv & 1;
    const int l31 = lane & 31, lhi = lane >> 5;
    // Workgroup b runs on XCD b & 7.  With more than one column tile the ntn workgroups that read the same source rows are
    // neighbours in the linear order -- different XCDs, so the rows came from HBM ntn times (d_pred = dS @ feature: 182 MB read for a
    // 75 MB dS, and the GEMM is bound by exactly that).  Remapped, workgroups b, b + 8, ... of one XCD are the column tiles of one
    // row tile: the second reader hits the L2 the first one filled.
    int bx = blockIdx.x;
    if (p.xcd_cols) {
        const int q = bx >> 3;
        bx = ((q / p.ntn) * 8 + (bx & 7)) * p.ntn + q % p.ntn;
    }
    const int n_tile = bx % p.ntn;
    const int ks = (bx / p.ntn) / p.gm;        // 0 unless the reduction is split over workgroups
    const int m_prog = (bx / p.ntn) - ks * p.gm;
    const int kc0 = ks * p.kcps;
    const int nkc_all = (g.Kp + BKE - 1) / BKE;
    const int nkc = (kc0 + p.kcps < nkc_all ? kc0 + p.kcps : nkc_all) - kc0;
    char* const outp = (char*)p.out + (long long)ks * p.slab * (int)sizeof(TO);
    const int esz = (int)sizeof(T);
    const char* const zero = (const char*)dpc_zero16;
    __shared__ int rowmap[GATHER == 3 ? BM : 1];      // tile row -> output row (parity classes permute rows)
    __shared__ int taptbl[GATHER == 3 ? 3 * 64 : 1];  // per class tap: {weight k base, source offset, validity select}

    // loop-invariant LDS byte offsets (buffer 0); +32 rows == +4096 B keeps the swizzle
    int frag_a[4], frag_b[4];
    DPC_UNROLL
    for (int kk = 0; kk < 4; ++kk) {
        frag_a[kk] = lds_unit_off(wm * 64 + l31, 2 * kk + lhi);
        frag_b[kk] = BM * 128 + lds_unit_off(wn * (BN / 2) + l31, 2 * kk + lhi);
    }
    const int stage_off = lds_unit_off(r0, u);  // rows r0+32*i: +4096*i

    // B (weights) rows of this thread: constant over the whole kernel
    long long wrow[BROWS];
    DPC_UNROLL
    for (int i = 0; i < BROWS; ++i) {
        const int n = n_tile * BN + r0 + 32 * i;
        wrow[i] = n < p.Ncol ? (long long)n * p.ldw : -1;
    }

    // batch-norm partial sums of this thread's output columns (fixed: 256 % UPR == 0)
    float s1[EPO], s2[EPO];
    DPC_UNROLL
    for (int e = 0; e < EPO; ++e) { s1[e] = 0.f; s2[e] = 0.f; }

    for (int mt = m_prog; mt < p.ntm; mt += p.gm) {
        const int m0 = mt * BM;
        RowPos rp[4];
        int rowbase[4];
        unsigned vmask[4];
        int nkc_t = nkc, lcpt = 0;
        if (GATHER == 3) {
            const ParityInfo& par = p.par;
            int c, jt_;
            parity_tile(par, mt, c, jt_);
            const int cw_ = c % g.sw, ch_ = (c / g.sw) % g.sh, ct_ = c / (g.sw * g.sh);
            const int Tc = par.dimc[0][ct_], Hc = par.dimc[1][ch_], Wc = par.dimc[2][cw_];
            const int rows_c = par.N * Tc * Hc * Wc;
            const int lr0 = jt_ * BM;
            const int ct_n = par.cnt[0][ct_], ch_n = par.cnt[1][ch_], cw_n = par.cnt[2][cw_];
            // the 8 lanes of a tile row need the same 4 rows (r0 + 32 i): each decodes one, the group exchanges
            // them by lane shuffles (the decode is per tile, and a parity-class tile has as few as 2 K chunks)
            {
                const int my_i = tid & 3;
                const int lr = lr0 + r0 + 32 * my_i;
                int orow = -1, rb_mine = 0;
                unsigned vm_mine = 0;
                if (lr < rows_c) {
                    const unsigned q1 = fdiv((unsigned)lr, par.div[2][cw_]);
                    const int wq = lr - (int)q1 * Wc;
                    const unsigned q2 = fdiv(q1, par.div[1][ch_]);
                    const int hq = (int)q1 - (int)q2 * Hc;
                    const unsigned n = fdiv(q2, par.div[0][ct_]);
                    const int tq = (int)q2 - (int)n * Tc;
                    rb_mine = (int)(((((unsigned)((int)n * g.ST + tq) * (unsigned)g.SH + (unsigned)hq) * (unsigned)g.SW) + (unsigned)wq) *
                                    (unsigned)g.src_ld);
                    unsigned m = 0;
                    for (int j = 0; j < ct_n; ++j) m |= ((unsigned)(tq + par.dl[0][ct_][j]) < (unsigned)g.ST ? 1u : 0u) << j;
                    for (int j = 0; j < ch_n; ++j) m |= ((unsigned)(hq + par.dl[1][ch_][j]) < (unsigned)g.SH ? 1u : 0u) << (4 + j);
                    for (int j = 0; j < cw_n; ++j) m |= ((unsigned)(wq + par.dl[2][cw_][j]) < (unsigned)g.SW ? 1u : 0u) << (8 + j);
                    vm_mine = m;
                    orow = ((((int)n * g.RT + tq * g.st + ct_) * g.RH + hq * g.sh + ch_) * g.RW) + wq * g.sw + cw_;
                }
                if ((tid & 7) < 4) rowmap[r0 + 32 * my_i] = orow;
                DPC_UNROLL
                for (int i = 0; i < 4; ++i) {
                    const int from = (lane & ~7) | i;
                    rowbase[i] = __shfl(rb_mine, from);
                    vmask[i] = (unsigned)__shfl((int)vm_mine, from);
                }
            }
            const int nt = ct_n * ch_n * cw_n;
            if (tid < nt) {
                const int jw = tid % cw_n, q = tid / cw_n, jh = q % ch_n, jt = q / ch_n;
                const int kt = par.kl[0][ct_][jt], kh = par.kl[1][ch_][jh], kw = par.kl[2][cw_][jw];
                const int dt = par.dl[0][ct_][jt], dh = par.dl[1][ch_][jh], dw = par.dl[2][cw_][jw];
                taptbl[3 * tid + 0] = ((kt * g.KH + kh) * g.KW + kw) * g.Ci;
                taptbl[3 * tid + 1] = ((dt * g.SH + dh) * g.SW + dw) * g.src_ld;
                taptbl[3 * tid + 2] = (int)((1u << jt) | (1u << (4 + jh)) | (1u << (8 + jw)));
            }
            lcpt = g.log2C - (EPU == 8 ? 6 : 5);  // log2(chunks per tap)
            nkc_t = nt << lcpt;
            __syncthreads();
        }
        if (GATHER == 1 || GATHER == 2) {  // affine gather: one row decoded per lane, shared inside the 8-lane row group
            const int my_i = tid & 3;
            const RowPos r1 = decode_row(g, m0 + r0 + 32 * my_i);
            const int rb_mine = (int)(((((unsigned)(r1.nbase + r1.t0) * (unsigned)g.SH + (unsigned)r1.h0) * (unsigned)g.SW) + (unsigned)r1.w0) *
                                      (unsigned)g.src_ld);
            const int sgn = g.mode == 0 ? 1 : -1;
            unsigned m = 0;
            for (int k = 0; k < g.KT; ++k) m |= ((unsigned)(r1.t0 + sgn * k) < (unsigned)g.ST ? 1u : 0u) << k;
            for (int k = 0; k < g.KH; ++k) m |= ((unsigned)(r1.h0 + sgn * k) < (unsigned)g.SH ? 1u : 0u) << (g.KT + k);
            for (int k = 0; k < g.KW; ++k) m |= ((unsigned)(r1.w0 + sgn * k) < (unsigned)g.SW ? 1u : 0u) << (g.KT + g.KH + k);
            DPC_UNROLL
            for (int i = 0; i < 4; ++i) {
                const int from = (lane & ~7) | i;
                rowbase[i] = __shfl(rb_mine, from);
                vmask[i] = (unsigned)__shfl((int)m, from);
            }
        }
        if (GATHER == 0) {
            DPC_UNROLL
            for (int i = 0; i < 4; ++i) rp[i] = decode_row(g, m0 + r0 + 32 * i);
        }

        f32x16 acc[2][NT];
        DPC_UNROLL
        for (int i = 0; i < 2; ++i)
            DPC_UNROLL
            for (int j = 0; j < NT; ++j)
                DPC_UNROLL
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

        u32x4 ra[4], rb[BROWS];
        auto load_chunk = [&](int kc_rel, int dma_buf) {
            const int kc = kc_rel + (GATHER == 3 ? 0 : kc0);
            int k = kc * BKE + u * EPU;
            bool kok = k < g.Kp;
            if (GATHER == 3) {
                const int tapv = kc >> lcpt;
                const int cofs = (kc & ((1 << lcpt) - 1)) * BKE + u * EPU;
                const int tapoff = taptbl[3 * tapv + 1] + cofs;
                const unsigned sel = (unsigned)taptbl[3 * tapv + 2];
                k = taptbl[3 * tapv + 0] + cofs;
                kok = true;
                DPC_UNROLL
                for (int i = 0; i < 4; ++i) {
                    const bool ok = (vmask[i] & sel) == sel;
                    const char* a = (const char*)p.src + (long long)(rowbase[i] + tapoff) * esz;
                    if (DMA) glds16(ok ? a : zero, lds + dma_buf + (32 * i + wv * 8) * 128, lane);
                    else ra[i] = *(const u32x4*)(ok ? a : zero);
                }
            } else if (GATHER != 0) {
                const int kd = (GATHER == 1) ? kc * BKE : k;  // GATHER 1: one tap per chunk -> wave-uniform decode
                const int tap = (g.taps == 1) ? 0 : (kd >> g.log2C);
                const unsigned q = fdiv((unsigned)tap, g.dKW);
                const int kw = tap - (int)q * g.KW;
                const unsigned kt = fdiv(q, g.dKH);
                const int kh = (int)q - (int)kt * g.KH;
                const unsigned sel = (1u << kt) | (1u << (g.KT + kh)) | (1u << (g.KT + g.KH + kw));
                const int sgn = g.mode == 0 ? 1 : -1;
                const int cbase = (g.taps == 1) ? 0 : (tap << g.log2C);
                const int tapoff = sgn * ((((int)kt * g.SH + kh) * g.SW + kw) * g.src_ld) + (k - cbase);
                DPC_UNROLL
                for (int i = 0; i < 4; ++i) {
                    const bool ok = kok && (vmask[i] & sel) == sel;
                    const char* a = (const char*)p.src + (long long)(rowbase[i] + tapoff) * esz;
                    if (DMA) glds16(ok ? a : zero, lds + dma_buf + (32 * i + wv * 8) * 128, lane);
                    else ra[i] = *(const u32x4*)(ok ? a : zero);
                }
            } else {
                const TapPos tp = decode_k(g, k);
                DPC_UNROLL
                for (int i = 0; i < 4; ++i) {
                    const long long off = gather_off(g, rp[i], tp);
                    const char* a = (const char*)p.src + off * esz;
                    if (DMA) glds16(off >= 0 ? a : zero, lds + dma_buf + (32 * i + wv * 8) * 128, lane);
                    else ra[i] = *(const u32x4*)(off >= 0 ? a : zero);
                }
            }
            DPC_UNROLL
            for (int i = 0; i < BROWS; ++i) {
                const char* b = (const char*)p.wgt + (wrow[i] + k) * esz;
                const bool ok = kok && wrow[i] >= 0;
                if (DMA) glds16(ok ? b : zero, lds + dma_buf + (BM + 32 * i + wv * 8) * 128, lane);
                else rb[i] = *(const u32x4*)(ok ? b : zero);
            }
        };
        auto store_chunk = [&](int bufoff) {
            if (DMA) return;
            unsigned char* base = lds + bufoff + stage_off;
            DPC_UNROLL
            for (int i = 0; i < 4; ++i) *(u32x4*)(base + 4096 * i) = ra[i];
            DPC_UNROLL
            for (int i = 0; i < BROWS; ++i) *(u32x4*)(base + BM * 128 + 4096 * i) = rb[i];
        };
        auto mma_chunk = [&](int bufoff) {
#ifndef DPC_SIMT_EMU
            if constexpr (sizeof(T) == 2 && sizeof(TO) == 2 && NT == 2) {   // measured: -8 % on the 128-wide tile, +2 % on the 64-wide one (three
                                                                          // workgroups per CU); the f32-output variants are register-bound: both left to hipcc
                // bf16: the chunk's MFMAs with hand-pipelined fragment reads (scheme of conv_wgrad_patch.hip / conv_wgrad.hip:
                // operand loads in program order per K step -- A0 B0 [B1] A1 --, LOOKAHEAD operands beyond the one an MFMA first
                // needs are in flight, counted lgkmcnt, a scheduling barrier after every instruction).  hipcc's own order is
                // "all reads of a K step, wait, its MFMAs": an LDS round trip per 2 NT x 32 cycles of matrix work.
                constexpr int LOOKAHEAD = 4, NOP = 2 + NT, NLOAD = 4 * NOP, NMMA = 8 * NT;
                const uint32_t l0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)lds + (uint32_t)bufoff;
                u32x4 fa[2][2], fb[2][NT];   // [K-step parity][fragment]
                auto load = [&](auto Ic) {
                    constexpr int I = decltype(Ic)::value;
                    constexpr int kk = I / NOP, r = I % NOP;
                    if constexpr (r == 0 || r == NOP - 1) {       // A0 first, A1 last
                        constexpr int i = r == 0 ? 0 : 1;
                        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fa[kk & 1][i]) : "v"(l0 + (uint32_t)frag_a[kk]), "n"(4096 * i) : "memory");
                    } else {
                        constexpr int j = r - 1;
                        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fb[kk & 1][j]) : "v"(l0 + (uint32_t)frag_b[kk]), "n"(4096 * j) : "memory");
                    }
                    __builtin_amdgcn_sched_barrier(0);
                };
                __builtin_amdgcn_sched_barrier(0);
                static_for<NMMA>([&](auto Mc) {
                    constexpr int m = decltype(Mc)::value;
                    constexpr int kk = m / (2 * NT), q = m % (2 * NT), i = q / NT, j = q % NT;
                    // operand this MFMA is the first to need: (0,0) -> B0; (0,j>0) -> Bj; (1,0) -> A1; (1,j>0) -> nothing new
                    constexpr int need = NOP * kk + (i == 0 ? 1 + j : NOP - 1);
                    constexpr int mp = m == 0 ? 0 : m - 1;
                    constexpr int kkp = mp / (2 * NT), qp = mp % (2 * NT), ip = qp / NT, jp = qp % NT;
                    constexpr int need_prev = m == 0 ? -1 : NOP * kkp + (ip == 0 ? 1 + jp : NOP - 1);
                    constexpr int f_prev = m == 0 ? 0 : (need_prev + 1 + LOOKAHEAD < NLOAD ? need_prev + 1 + LOOKAHEAD : NLOAD);
                    constexpr int f_now = need + 1 + LOOKAHEAD < NLOAD ? need + 1 + LOOKAHEAD : NLOAD;
                    static_for<(f_now > f_prev ? f_now - f_prev : 0)>([&](auto Dc) { load(std::integral_constant<int, f_prev + decltype(Dc)::value>{}); });
                    asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(f_now - need - 1) : "memory");
                    __builtin_amdgcn_sched_barrier(0);
                    acc[i][j] = mfma_unit<T>(fa[kk & 1][i], fb[kk & 1][j], acc[i][j]);
                    __builtin_amdgcn_sched_barrier(0);
                });
                return;
            }
#endif
            if constexpr (sizeof(T) == 4) {
                if (p.x6) {   // bf16x6: K steps in pairs (two units = 8 f32 per lane), every fragment split once, six MFMAs per block pair
                    DPC_UNROLL
                    for (int kp = 0; kp < 2; ++kp) {
                        Split3 sa[2], sb[NT];
                        DPC_UNROLL
                        for (int i = 0; i < 2; ++i)
                            sa[i] = split3_f32x8(*(const u32x4*)(lds + bufoff + frag_a[2 * kp] + 4096 * i),
                                                 *(const u32x4*)(lds + bufoff + frag_a[2 * kp + 1] + 4096 * i));
                        DPC_UNROLL
                        for (int j = 0; j < NT; ++j)
                            sb[j] = split3_f32x8(*(const u32x4*)(lds + bufoff + frag_b[2 * kp] + 4096 * j),
                                                 *(const u32x4*)(lds + bufoff + frag_b[2 * kp + 1] + 4096 * j));
                        DPC_UNROLL
                        for (int i = 0; i < 2; ++i)
                            DPC_UNROLL
                            for (int j = 0; j < NT; ++j) acc[i][j] = mfma_f32x6(sa[i], sb[j], acc[i][j]);
                    }
                    return;
                }
            }
            DPC_UNROLL
            for (int kk = 0; kk < 4; ++kk) {
                u32x4 fa[2], fb[NT];
                DPC_UNROLL
                for (int i = 0; i < 2; ++i) fa[i] = *(const u32x4*)(lds + bufoff + frag_a[kk] + 4096 * i);
                DPC_UNROLL
                for (int j = 0; j < NT; ++j) fb[j] = *(const u32x4*)(lds + bufoff + frag_b[kk] + 4096 * j);
                DPC_UNROLL
                for (int i = 0; i < 2; ++i)
                    DPC_UNROLL
                    for (int j = 0; j < NT; ++j) acc[i][j] = mfma_unit<T>(fa[i], fb[j], acc[i][j]);
            }
        };

        if constexpr (DMA && DPC_IGEMM_PIPE2) {
            // Two chunks in flight: chunk k+2 is requested as soon as every wave is done with the buffer of chunk k, and the wait in
            // front of chunk k+1's MFMAs counts (vmcnt(NLD): only the pieces of chunk k+2 may be outstanding) -- a chunk has two
            // iterations to land.  With "request k+1, MFMAs of k, wait for everything" an iteration lasts one memory round trip
            // whatever the MFMA time is: the short reductions (parity-class tiles of 2..32 chunks, split-K GEMMs) ran at 14 % MFMA
            // utilisation with one chunk in flight per workgroup.
            constexpr int NLD = 4 + BROWS;   // LDS-DMA instructions per thread and chunk
            if (nkc_t > 0) load_chunk(0, 0);   // a parity class may see no tap at all (strided 1x1x1): its rows are just 0 (+ addend)
            if (nkc_t > 1) load_chunk(1, BUF);
            for (int kc = 0; kc < nkc_t; kc += 2) {
                if (kc + 1 < nkc_t) wait_vmcnt<NLD>(); else wait_vmcnt<0>();
                barrier_lds_only();
                mma_chunk(0);
                if (kc + 2 < nkc_t) {
                    barrier_lds_only();
                    load_chunk(kc + 2, 0);
                }
                if (kc + 1 < nkc_t) {
                    if (kc + 2 < nkc_t) wait_vmcnt<NLD>(); else wait_vmcnt<0>();
                    barrier_lds_only();
                    mma_chunk(BUF);
                    if (kc + 3 < nkc_t) {
                        barrier_lds_only();
                        load_chunk(kc + 3, BUF);
                    }
                }
            }
            __syncthreads();   // every wave is done with the operand buffers: the epilogue stages the tile through them
        } else {
        if (nkc_t > 0) {
            load_chunk(0, 0);
            store_chunk(0);
        }
        __syncthreads();
        for (int kc = 0; kc < nkc_t; kc += 2) {
            if (kc + 1 < nkc_t) load_chunk(kc + 1, BUF);
            mma_chunk(0);
            if (kc + 1 < nkc_t) store_chunk(BUF);
            __syncthreads();
            if (kc + 1 < nkc_t) {
                if (kc + 2 < nkc_t) load_chunk(kc + 2, 0);
                mma_chunk(BUF);
                if (kc + 2 < nkc_t) store_chunk(0);
                __syncthreads();
            }
        }
        }

        // ---- epilogue: accumulators -> LDS tile [128][BN] of TO -> 16-byte units -> global
        // C/D map: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
        TO* tile = (TO*)lds;
        DPC_UNROLL
        for (int j = 0; j < NT; ++j)
            DPC_UNROLL
            for (int i = 0; i < 2; ++i)
                DPC_UNROLL
                for (int r = 0; r < 16; ++r) {
                    const int row_l = wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
                    const int col_l = wn * (BN / 2) + j * 32 + l31;
                    tile[row_l * BN + col_l] = Elt<TO>::from_f32(acc[i][j][r]);
                }
        __syncthreads();
        const int cu = tid % UPR;
        const int col0 = n_tile * BN + cu * EPO;
        auto out_row = [&](int row_l) -> int {  // -1: no such row
            if (GATHER == 3) return rowmap[row_l];
            return (m0 + row_l < g.M) ? m0 + row_l : -1;
        };
        if (EPI && p.vec_out) {
            if constexpr (EPI) {
            const bool bnred = p.epi.bn_raw != nullptr;
            float mu[EPO], is[EPO];
            DPC_UNROLL
            for (int e = 0; e < EPO; ++e) {
                const bool okc = bnred && col0 + e < p.Ncol;
                mu[e] = okc ? p.epi.bn_mean[col0 + e] : 0.f;
                is[e] = okc ? p.epi.bn_invstd[col0 + e] : 0.f;
            }
            u32x4 ov[OIT], av[OIT], rv[OIT];
            unsigned ab[OIT], bb[OIT];
            DPC_UNROLL
            for (int it = 0; it < OIT; ++it) {
                const int row_l = (tid + 256 * it) / UPR;
                ov[it] = *(const u32x4*)(lds + (row_l * BN + cu * EPO) * (int)sizeof(TO));
                const int row = out_row(row_l);
                const bool ok = row >= 0 && col0 < p.Ncol;
                const long long eo = (long long)row * p.ldo + col0;  // element offset of the unit (dense tensors: ldo == Ncol with masks)
                ab[it] = ~0u; bb[it] = ~0u;
                if (p.addend) {
                    av[it] = *(const u32x4*)(ok ? (const char*)p.addend + eo * (int)sizeof(TO) : zero);
                    if (p.epi.addend_mask) ab[it] = ok ? (unsigned)p.epi.addend_mask[eo / EPO] : 0u;
                }
                if (bnred) {
                    rv[it] = *(const u32x4*)(ok ? (const char*)p.epi.bn_raw + eo * (int)sizeof(TO) : zero);
                    if (p.epi.bn_mask) bb[it] = ok ? (unsigned)p.epi.bn_mask[eo / EPO] : 0u;
                }
            }
            DPC_UNROLL
            for (int it = 0; it < OIT; ++it) {
                const int row = out_row((tid + 256 * it) / UPR);
                if (row >= 0 && col0 < p.Ncol) {
                    u32x4 o = ov[it];
                    epi_unit<TO, EPO>(o, p.addend != nullptr, av[it], ab[it], bnred, rv[it], bb[it], mu, is, s1, s2);
                    *(u32x4*)(outp + ((long long)row * p.ldo + col0) * (int)sizeof(TO)) = o;
                }
            }
            }
        } else if (p.vec_out) {
            u32x4 ov[OIT], av[OIT];
            DPC_UNROLL
            for (int it = 0; it < OIT; ++it) {
                const int row_l = (tid + 256 * it) / UPR;
                ov[it] = *(const u32x4*)(lds + (row_l * BN + cu * EPO) * (int)sizeof(TO));
                const int row = out_row(row_l);
                const bool ok = row >= 0 && col0 < p.Ncol;
                if (p.addend) {
                    const char* a = (const char*)p.addend + ((long long)row * p.ldo + col0) * (int)sizeof(TO);
                    av[it] = *(const u32x4*)(ok ? a : zero);
                }
            }
            DPC_UNROLL
            for (int it = 0; it < OIT; ++it) {
                const int row = out_row((tid + 256 * it) / UPR);
                if (row >= 0 && col0 < p.Ncol) {
                    u32x4 o = ov[it];
                    if (p.addend) {
                        float sv[EPO];
                        DPC_UNROLL
                        for (int e = 0; e < EPO; ++e) sv[e] = unit_get<TO>(o, e) + unit_get<TO>(av[it], e);
                        o = unit_pack<TO>(sv);
                    }
                    *(u32x4*)(outp + ((long long)row * p.ldo + col0) * (int)sizeof(TO)) = o;
                    DPC_UNROLL
                    for (int e = 0; e < EPO; ++e) {
                        const float v = unit_get<TO>(o, e);
                        s1[e] += v;
                        s2[e] += v * v;
                    }
                }
            }
        } else {  // ragged output width: element-wise tail path
            DPC_UNROLL
            for (int it = 0; it < OIT; ++it) {
                const int row_l = (tid + 256 * it) / UPR;
                const int row = out_row(row_l);
                DPC_UNROLL
                for (int e = 0; e < EPO; ++e) {
                    const int col = col0 + e;
                    if (row >= 0 && col < p.Ncol) {
                        const long long o = (long long)row * p.ldo + col;
                        float v = Elt<TO>::to_f32(tile[row_l * BN + cu * EPO + e]);
                        if (p.addend) v += Elt<TO>::to_f32(((const TO*)p.addend)[o]);
                        const TO q = Elt<TO>::from_f32(v);
                        ((TO*)outp)[o] = q;
                        const float vq = Elt<TO>::to_f32(q);
                        s1[e] += vq;
                        s2[e] += vq * vq;
                    }
                }
            }
        }
        __syncthreads();  // the tile buffer is the next m-tile's operand buffer
    }

    if (p.stats) {
        // 256/UPR threads hold partial sums for the same EPO columns: reduce through LDS
        float* red = (float*)lds;  // [2][256][EPO]
        DPC_UNROLL
        for (int e = 0; e < EPO; ++e) {
            red[tid * EPO + e] = s1[e];
            red[(256 + tid) * EPO + e] = s2[e];
        }
        __syncthreads();
        if (tid < BN) {
            const int cu2 = tid / EPO, e2 = tid % EPO;
            const int col = n_tile * BN + tid;
            if (col < p.Ncol) {
                float a = 0.f, b = 0.f;
                for (int t = cu2; t < 256; t += UPR) {
                    a += red[t * EPO + e2];
                    b += red[(256 + t) * EPO + e2];
                }
                p.stats[((long long)m_prog * 2 + 0) * p.Ncol + col] = a;
                p.stats[((long long)m_prog * 2 + 1) * p.Ncol + col] = b;
            }
        }
    }
}

static void igemm_grid(const dpc_conv_desc* d, int M, int* ntm, int* ntn, int* gm, int* bn) {
    *bn = d->Co > 64 ? 128 : 64;
    *ntn = (d->Co + *bn - 1) / *bn;
    *ntm = (M + 127) / 128;
    int cap = 2048 / *ntn;
    if (cap < 1) cap = 1;
    *gm = *ntm < cap ? *ntm : cap;
}

extern "C" int dpc_conv_stats_rows(const dpc_conv_desc* d) {
    GatherGeom g;
    int rc = make_gather_geom(d, &g);
    if (rc) return rc;
    const int hr = dpc_conv_halo_rows(d);
    if (hr > 0) return hr;
    const int wr = dpc_conv_ws_rows(d);
    if (wr > 0) return wr;
    int ntm, ntn, gm, bn;
    igemm_grid(d, g.M, &ntm, &ntn, &gm, &bn);
    return gm;
}

template <class T, class TO, int BN>
static int launch_igemm_bn(IGemmParams& p, int gather, hipStream_t stream) {
    dim3 grid((unsigned)(p.gm * p.ntn * p.nks)), block(256);
    p.xcd_cols = (p.ntn > 1 && (p.gm * p.nks) % 8 == 0) ? 1 : 0;
    dpc_plan_detail("T=%s TO=%s BN=%d", sizeof(T) == 2 ? "bf16" : "f32", sizeof(TO) == 2 ? "bf16" : "f32", BN);
    if (epi_any(p.epi)) {  // dpc_conv_igemm_ex (T == TO, vectorised output: checked by the entry)
        if constexpr (sizeof(T) == sizeof(TO)) {
            if (gather == 1) {
                DPC_LAUNCH((igemm_kernel<T, TO, BN, 1, true>), grid, block, stream, p);
            } else if (gather == 2) {
                DPC_LAUNCH((igemm_kernel<T, TO, BN, 2, true>), grid, block, stream, p);
            } else {
                DPC_LAUNCH((igemm_kernel<T, TO, BN, 0, true>), grid, block, stream, p);
            }
            return dpc_launch_status();
        } else {
            return DPC_ERR_UNSUPPORTED;
        }
    }
    if (gather == 3) {
        DPC_LAUNCH((igemm_kernel<T, TO, BN, 3>), grid, block, stream, p);
    } else if (gather == 1) {
        DPC_LAUNCH((igemm_kernel<T, TO, BN, 1>), grid, block, stream, p);
    } else if (gather == 2) {
        DPC_LAUNCH((igemm_kernel<T, TO, BN, 2>), grid, block, stream, p);
    } else {
        DPC_LAUNCH((igemm_kernel<T, TO, BN, 0>), grid, block, stream, p);
    }
    return dpc_launch_status();
}

// fills p.par and re-plans the m-tiles class by class; false when the shape is outside the scheme
static bool plan_parity(IGemmParams& p, int bke) {
    // in-place accumulation (addend == out): positions no tap reaches keep their value, so their classes get no tiles --
    // a 1x1 stride-2 downsample's input-gradient touches 1/4 (2D) or 1/8 (3D) of dx instead of rewriting all of it
    if (!dpc_plan_parity(p.g, p.par, bke, 128, p.addend && p.addend == p.out, &p.ntm)) return false;
    static const int gm_cap = getenv("DPC_IGEMM_GM_CAP") ? atoi(getenv("DPC_IGEMM_GM_CAP")) : 2048;   // test tiers: several rounds on small shapes
    int cap = gm_cap / p.ntn;
    if (cap < 1) cap = 1;
    if (cap >= 8) cap &= ~7;  // interleaved class tiles rely on tile mt running on XCD mt % 8
    p.gm = p.ntm < cap ? p.ntm : cap;
    dpc_parity_rounds(p.par, p.gm);
    return true;
}

template <class T, class TO>
static int launch_igemm(IGemmParams& p, int bn, hipStream_t stream) {
    const GatherGeom& g = p.g;
    const int bke = 8 * Elt<T>::PER16;
    const bool unit_strides = g.st == 1 && g.sh == 1 && g.sw == 1;
    const bool fits32 = (long long)(g.M / (g.RT * g.RH * g.RW)) * g.ST * g.SH * g.SW * g.src_ld < (1ll << 31);
    const bool affine = fits32 && (g.mode == 0 || unit_strides) && (g.KT + g.KH + g.KW <= 32);
    int gather = !affine ? 0 : ((g.taps == 1 || g.Ci >= bke) ? 1 : 2);
    if (gather == 0 && fits32 && g.mode == 1 && !p.stats && !epi_any(p.epi) && plan_parity(p, bke)) gather = 3;
    if (bn == 64) return launch_igemm_bn<T, TO, 64>(p, gather, stream);
    return launch_igemm_bn<T, TO, 128>(p, gather, stream);
}

static int conv_igemm_impl(const dpc_conv_desc* d, const void* src, const void* wgt, void* out, const void* addend, float* stats,
                           const EpiExtra& epi, hipStream_t stream);

extern "C" int dpc_conv_igemm(const dpc_conv_desc* d, const void* src, const void* wgt, void* out,
                              const void* addend, float* stats, dpc_stream_t stream_) {
    return conv_igemm_impl(d, src, wgt, out, addend, stats, epi_none(), (hipStream_t)stream_);
}

extern "C" int dpc_conv_igemm_ex(const dpc_conv_desc* d, const void* src, const void* wgt, void* out, const dpc_conv_epilogue* e,
                                 dpc_stream_t stream_) {
    if (!d || !e) return DPC_ERR_ARG;
    EpiExtra x = {e->addend_mask, e->bn_raw, e->bn_mask, e->bn_mean, e->bn_invstd};
    if (x.addend_mask && !e->addend) return DPC_ERR_ARG;
    if (x.bn_raw && (!x.bn_mean || !x.bn_invstd || !e->stats)) return DPC_ERR_ARG;
    if (x.bn_mask && !x.bn_raw) return DPC_ERR_ARG;
    if (epi_any(x)) {
        // masks index dense [rows][Co] tensors by 16-byte unit; the fused pieces live in the vectorised epilogue only
        const int epo = d->dtype_out == DPC_BF16 ? 8 : 4;
        if (d->ldo != d->Co || d->Co % epo || d->dtype_in != d->dtype_out) return DPC_ERR_UNSUPPORTED;
        if (((uintptr_t)out % 16) || ((uintptr_t)e->addend % 16) || ((uintptr_t)x.bn_raw % 16)) return DPC_ERR_UNSUPPORTED;
    }
    return conv_igemm_impl(d, src, wgt, out, e->addend, e->stats, x, (hipStream_t)stream_);
}

static int conv_igemm_impl(const dpc_conv_desc* d, const void* src, const void* wgt, void* out, const void* addend, float* stats,
                           const EpiExtra& epi, hipStream_t stream) {
    IGemmParams p;
    p.x6 = dpc_f32_matmul_mode();
    int rc = make_gather_geom(d, &p.g);
    if (rc) return rc;
    if (!src || !wgt || !out) return DPC_ERR_ARG;
    const int per16 = d->dtype_in == DPC_BF16 ? 8 : 4;
    if (d->ldw % per16) return DPC_ERR_UNSUPPORTED;
    if (d->ldo < d->Co || d->ldw < p.g.Kp) return DPC_ERR_ARG;
    if (!addend && !stats && !epi_any(epi)) {
        rc = dpc_score_gemm_try(d, src, wgt, out, stream);                  // large bf16 -> f32 NT GEMM with K = 256: the materialised score
        if (rc != 1) return rc;
    }
    rc = dpc_conv_halo_try(d, src, wgt, out, addend, stats, epi, stream);  // LDS-staged patch kernel when the shape allows
    if (rc != 1) return rc;
    rc = dpc_conv_ws_try(d, src, wgt, out, addend, stats, epi, stream);    // loader/compute specialised kernel for the wide layers
    if (rc != 1) return rc;
    p.src = src; p.wgt = wgt; p.out = out; p.addend = addend; p.stats = stats; p.epi = epi;
    p.nks = 1; p.kcps = 1 << 28; p.slab = 0;
    p.Ncol = d->Co; p.ldw = d->ldw; p.ldo = d->ldo;
    const int epo = d->dtype_out == DPC_BF16 ? 8 : 4;
    p.vec_out = (d->Co % epo == 0 && d->ldo % epo == 0 && ((uintptr_t)out % 16 == 0) && ((uintptr_t)addend % 16 == 0)) ? 1 : 0;
    int bn;
    igemm_grid(d, p.g.M, &p.ntm, &p.ntn, &p.gm, &bn);
    {   // dpc_conv_stats_rows() promised a row count for this shape (the patch kernel's, the loader/compute kernel's or this
        // kernel's own).  If a specialised kernel declined at launch time (alignment, addend + stats) the promise is kept:
        // the generic kernel is persistent over any gm, and rows it has no tile for are zero-filled, so that
        // dpc_bn_finalize never reads partial sums another layer left in a shared statistics buffer.
        const int promised = dpc_conv_stats_rows(d);
        if (promised > 0 && promised != p.gm) {
            p.gm = promised < p.ntm ? promised : p.ntm;
            if (stats && p.gm < promised && !dpc_tls_plan_only) {
                if (hipMemsetAsync(stats + (size_t)p.gm * 2 * d->Co, 0, (size_t)(promised - p.gm) * 2 * d->Co * sizeof(float), stream) != hipSuccess)
                    return DPC_ERR_LAUNCH;
            }
        }
    }
    if (d->dtype_in == DPC_F32 && d->dtype_out == DPC_F32) return launch_igemm<float, float>(p, bn, stream);
    if (d->dtype_in == DPC_BF16 && d->dtype_out == DPC_BF16) return launch_igemm<bf16_t, bf16_t>(p, bn, stream);
    if (d->dtype_in == DPC_BF16 && d->dtype_out == DPC_F32) return launch_igemm<bf16_t, float>(p, bn, stream);
    return DPC_ERR_UNSUPPORTED;
}

// NT GEMM with the reduction split over workgroups (f32 partial slabs): products whose [M][N] output has too few 128 x BN tiles to
// fill 256 CUs while K is long -- d_pred = dS @ feature_inf of the contrastive loss: M = R = 6 144, N = 256, K = R (96 tiles;
// through dpc_conv_igemm it ran 89 us = 217 TFLOP/s on a third of the chip).
extern "C" int dpc_gemm_nt_splitk(int32_t dtype, int32_t M, int32_t N, int32_t K, const void* A, int32_t lda, const void* B, int32_t ldb,
                                  float* part, int32_t* nsplit, dpc_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (M <= 0 || N <= 0 || K <= 0 || (dtype != DPC_F32 && dtype != DPC_BF16)) return DPC_ERR_ARG;
    const int per16 = dtype == DPC_BF16 ? 8 : 4;
    if (K % per16 || lda % per16 || ldb % per16 || lda < K || ldb < K) return DPC_ERR_UNSUPPORTED;
    if (dtype == DPC_BF16) {   // long reductions with a narrow output: 256 x 128 tiles on the loader / compute kernel (gemm_ws.hip)
        const int rcw = dpc_gemm_ws_try(0, M, N, K, A, lda, B, ldb, part, nsplit, stream);
        if (rcw != 1) return rcw;
    }
    dpc_conv_desc d = {dtype, DPC_F32, 0, M, 1, 1, 1, 1, 1, 1, K, lda, N, ldb, N, 1, 1, 1, 1, 1, 1, 0, 0, 0};
    IGemmParams p;
    p.x6 = dpc_f32_matmul_mode();
    int rc = make_gather_geom(&d, &p.g);
    if (rc) return rc;
    int bn;
    igemm_grid(&d, p.g.M, &p.ntm, &p.ntn, &p.gm, &bn);
    p.gm = p.ntm;  // one m-tile per workgroup: the split supplies the parallelism
    const int bke = 8 * per16;
    const int nchunks = (K + bke - 1) / bke;
    int want = 512 / (p.ntm * p.ntn);           // two resident workgroups per CU
    if (want > nchunks / 8) want = nchunks / 8;  // every split amortises its f32 slab over >= 8 chunks
    if (want < 1) want = 1;
    p.kcps = (nchunks + want - 1) / want;
    p.nks = (nchunks + p.kcps - 1) / p.kcps;
    if (nsplit) *nsplit = p.nks;
    if (!part) return DPC_OK;  // size query: nsplit * M * N floats
    if (!A || !B) return DPC_ERR_ARG;
    p.src = A; p.wgt = B; p.out = part; p.addend = nullptr; p.stats = nullptr; p.epi = epi_none();
    p.Ncol = N; p.ldw = ldb; p.ldo = N; p.slab = (long long)M * N;
    p.vec_out = (N % 4 == 0 && ((uintptr_t)part % 16 == 0)) ? 1 : 0;
    dpc_plan_detail("splitk=%d", p.nks);
    if (dtype == DPC_F32) return launch_igemm<float, float>(p, bn, stream);
    return launch_igemm<bf16_t, float>(p, bn, stream);
}

extern "C" int dpc_abi_version(void) { return 1; }
