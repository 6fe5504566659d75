// conv_wgrad_patch.hip -- weight gradient of the 1x3x3 and 3x3x3 stride-1 "same" convolutions (BasicBlock2d /
// BasicBlock3d convs of layer1..3, backbone/resnet_2d3d.py:14-32,83-116; autograd of F.conv3d w.r.t. the
// weight), bf16.  A 3x3x3 kernel is three 1x3x3 problems, one per temporal tap, whose source plane is t+kt-1.
//
//   part[ks][co][tap*Ci + ci] = sum_{m in split ks} dy[m][co] * x[m shifted by tap][ci]
//
// conv_wgrad.hip's generic kernel gathers the source once PER TAP: for a 3x3 kernel 9 of the 10 operand
// sub-tiles of a chunk are shifted copies of the same pixels, the LDS-DMA stream is 32 KB per 48 MFMAs and
// the CU's texture addresser (64 B/clk), not the matrix cores, sets the pace (~500 TFLOP/s measured on
// layer1).  Here a workgroup owns a 64(co) x 64(ci) block of ALL NINE taps:
//   * per chunk of 64 output positions (whole image rows) it stages dy[64][64 co] (8 KB) and ONE patch of
//     the source, (rows+2) x (W+4) positions x 64 ci (17 KB for W = 32), zero borders included, by
//     LDS-DMA through buffer resources (out-of-image pixels are out-of-range lanes -> hardware zero fill);
//   * both stay in memory orientation [position][channel]; the MFMA fragments (reduction index = position
//     contiguous per lane) come from ds_read_b64_tr_b16, and the fragment of tap (kh,kw) is the SAME patch
//     read (kh*(W+4) + kw) positions further -- an instruction immediate, no gather arithmetic;
//   * 4 waves = 2 (co halves) x 2 (ci halves), each 32 x 32 x 9 taps = 9 accumulators (144 VGPRs);
//     144 MFMAs per 25 KB staged instead of 48 per 32 KB.
// LDS rows are 128 B (64 channels); the 16-byte slot of position p is XORed with 2*(p&3): the four
// positions a transpose read touches are consecutive, so any tap shift keeps the reads conflict free.
// The patch is W+4 wide so that row starts stay multiples of 4 positions (the swizzle phase of a read then
// depends on kw only).  Split-K partial slabs and their reduction are shared with conv_wgrad.hip.
#include "conv_common.h"
#include <stdlib.h>

struct WgradPatchParams {
    const void* src;
    const void* dy;
    float* part;
    int Co, Ci, dy_ld, src_ld, Kp;
    int RH, W, NF;       // image rows, real image width (<= the padded width the kernel is instantiated for), frames (N*T)
    int M;
    int nks, kcps;
    int ntm, ntc;
    unsigned src_bytes, dy_bytes;
    FastDiv d_cpf;       // chunks per frame
    int cpf;
    int KT, pt, T;       // temporal taps: a workgroup owns ONE kt; T = frames per clip
    FastDiv d_T;
    int dbg;             // DPC_WS_PROBE builds only: phases to leave out, for timing (scripts/probes/wgrad_probe.py)
};
#ifdef DPC_WS_PROBE
#define WP_DBG(bit) (p.dbg & (bit))
#else
#define WP_DBG(bit) 0
#endif

// RW: image width rounded up to 8/16/32/64.  A chunk is 64/RW image rows of RW columns; columns >= the real
// width and rows >= the image height are out-of-range lanes of the dy DMA (zeros: they add nothing), so any
// image up to 64 wide is served (the 224-pixel configurations: 56, 28, 14, 7 -> 64, 32, 16, 8 at 12.5 % padding).
template <int RW>
__global__ __launch_bounds__(256, 2) void wgrad_patch_kernel(WgradPatchParams p) {
    constexpr int ROWS = 64 / RW;          // image rows per chunk (RW = 8: a chunk is one 8x8 plane)
    constexpr int PR = ROWS + 2, PW = RW + 4;
    constexpr int NPOS = PR * PW;          // patch positions
    constexpr int NPB = (NPOS + 7) / 8;    // B pieces (8 positions = 1 KB each)
    constexpr int NIB = (NPB + 3) / 4;     // B pieces per wave
    constexpr int STAGE = 8192 + NPB * 1024;
    // ring of three stages (two chunks in flight) for the 32-wide images, two stages elsewhere
    constexpr int NST = (RW == 32 && 2 * 3 * STAGE <= 160 * 1024) ? 3 : 2;   // measured: 32 -> -3 %, 16 -> +4 %, 8 -> 0
    __shared__ __attribute__((aligned(1024))) unsigned char lds[NST * STAGE];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
#ifdef DPC_SIMT_EMU
    const int wv = tid >> 6;
#else
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
#endif
    const int wi = wv >> 1, wj = wv & 1;
    const int l31 = lane & 31, lhi = lane >> 5;
    // an XCD owns a contiguous range of logical ids: the tiles of one K-split share their dy / source rows in one L2
    int id = blockIdx.x;
    {
        const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, x = id & 7;
        id = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (id >> 3);
    }
    const int tile_c = id % p.ntc; id /= p.ntc;
    const int tile_m = id % p.ntm; id /= p.ntm;
    const int kt = id % p.KT;
    const int ks = id / p.KT;
    const int nchunks = p.NF * p.cpf;
    const int c_begin = ks * p.kcps;
    const int c_end = (c_begin + p.kcps < nchunks) ? c_begin + p.kcps : nchunks;

    // ---- DMA lane constants
    const BufRsrc rs_a = make_buf_rsrc(p.dy, p.dy_bytes);
    const BufRsrc rs_b = make_buf_rsrc(p.src, p.src_bytes);
    const int pl = lane >> 3;                              // position inside a piece
    const int lslot = (lane & 7) ^ (2 * (pl & 3));         // channel group this lane fetches (swizzle on the source side)
    unsigned a_off[2];
    int a_row[2];  // image row relative to the chunk's first row, or a value no image row can have (padding column)
    DPC_UNROLL
    for (int i = 0; i < 2; ++i) {
        const int pp = 8 * (wv + 4 * i) + pl;
        const int r = pp / RW, c = pp % RW;
        a_row[i] = c < p.W ? r : (1 << 28);
        a_off[i] = (unsigned)((r * p.W + c) * p.dy_ld + tile_m * 64 + lslot * 8) * 2u;
    }
    unsigned b_off[NIB];
    int b_row[NIB];  // patch row - 1 (image row relative to the chunk's first row), or a value no image row can have
    DPC_UNROLL
    for (int i = 0; i < NIB; ++i) {
        const int pp = 8 * (wv + 4 * i) + pl;
        const int prow = pp / PW, pcol = pp % PW;
        const bool ok = pp < NPOS && pcol >= 1 && pcol <= p.W;
        b_row[i] = ok ? prow - 1 : (1 << 28);
        b_off[i] = (unsigned)(((prow - 1) * p.W + (pcol - 1)) * p.src_ld + tile_c * 64 + lslot * 8) * 2u;
    }

    f32x16 acc[9];
    DPC_UNROLL
    for (int t = 0; t < 9; ++t)
        DPC_UNROLL
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    // DMA of one chunk = 2 pieces of dy + up to NIB pieces of the patch per wave.  prep() resolves the chunk's per-piece offsets
    // (out-of-image rows become out-of-range lanes), piece(i) issues one of them: the pieces of the NEXT chunk are issued one at
    // a time between the MFMAs of the current one instead of as a burst in front of them.  MEASURED (scripts/probes/wgrad_probe.py,
    // layer1 / layer2 / layer3 of cfg2): MFMA + operand reads alone 257 / 239 / 385 us, DMA + barriers alone 250 / 176 / 229 us,
    // together 381 / 325 / 446 us -- and that sum does not move with the placement of the DMA instructions (burst: 362 / 320 / 455)
    // nor with a third stage: the two streams share the LDS (each MFMA needs 1 KB of transposed reads; ds_read_b64_tr_b16 reaches
    // its peak only from ~4 waves per SIMD, this kernel runs 2), not the issue slots.
    constexpr int NPC = 2 + NIB;
    unsigned dma_v[NPC];
    unsigned char* dma_stage = lds;
    auto dma_prep = [&](int chunk, int buf, bool real) {   // !real (no chunk left): every lane out of range -- the pieces write zeros
        dma_stage = lds + buf * STAGE;                      // into the free stage, and the MFMA stream needs no branch around them
        const unsigned frame = fdiv((unsigned)chunk, p.d_cpf);
        const int h0 = (chunk - (int)frame * p.cpf) * ROWS;
        const unsigned a_base = (unsigned)(((int)frame * p.RH + h0) * p.W) * (unsigned)p.dy_ld * 2u;
        const unsigned b_base = (unsigned)((((int)frame + kt - p.pt) * p.RH + h0) * p.W) * (unsigned)p.src_ld * 2u;
        DPC_UNROLL
        for (int i = 0; i < 2; ++i) {
            const bool ok = (unsigned)(h0 + a_row[i]) < (unsigned)p.RH;
            dma_v[i] = ok && real ? a_base + a_off[i] : DPC_BUF_OOB;
        }
        DPC_UNROLL
        for (int i = 0; i < NIB; ++i) {
            const bool ok = (unsigned)(h0 + b_row[i]) < (unsigned)p.RH;
            dma_v[2 + i] = ok && real ? b_base + b_off[i] : DPC_BUF_OOB;
        }
    };
    auto dma_piece = [&](auto Ic) {
        constexpr int I = decltype(Ic)::value;
        if (WP_DBG(1)) return;
        if constexpr (I < 2) {
            glds16_buf(rs_a, dma_v[I], 0u, dma_stage + (wv + 4 * I) * 1024, lane);
        } else {
            if constexpr (4 * (I - 2) + 3 < NPB) {   // every wave has this piece
                glds16_buf(rs_b, dma_v[I], 0u, dma_stage + 8192 + (wv + 4 * (I - 2)) * 1024, lane);
            } else {
                if (wv + 4 * (I - 2) < NPB) glds16_buf(rs_b, dma_v[I], 0u, dma_stage + 8192 + (wv + 4 * (I - 2)) * 1024, lane);
            }
        }
    };
    auto issue = [&](int chunk, int buf) {
        dma_prep(chunk, buf, true);
        static_for<NPC>([&](auto Ic) { dma_piece(Ic); });
    };

    // ---- fragment lane offsets
    const int gq = lane >> 4, s16 = lane & 15, ph = s16 >> 2;
    const int lin = ((gq >> 1) * 8 + ph) * 128;
    auto slot_off = [&](int c0, int phase) {
        const int colb = (c0 + (gq & 1) * 16 + 4 * (s16 & 3)) * 2;
        return ((((colb >> 4) ^ (2 * (phase & 3))) & 7) << 4) + (colb & 15);
    };
    const int fa = lin + slot_off(wi * 32, ph);
    // the lane's 8-position group starts (gq>>1)*8 positions into the 16 of a step: for W = 8 that is the next image row
    const int linb = ((((gq >> 1) * 8) / RW) * PW + ((gq >> 1) * 8) % RW + ph) * 128;
    int fb[3];
    DPC_UNROLL
    for (int kw = 0; kw < 3; ++kw) fb[kw] = 8192 + linb + slot_off(wj * 32, ph + kw);

#ifdef DPC_SIMT_EMU
    auto compute = [&](int buf) {
        static_for<NPC>([&](auto Ic) { dma_piece(Ic); });
        const unsigned char* st = lds + buf * STAGE;
        DPC_UNROLL
        for (int kk = 0; kk < 4; ++kk) {
            const u32x2 a0 = lds_read_tr16(st + fa + (kk * 16) * 128);
            const u32x2 a1 = lds_read_tr16(st + fa + (kk * 16 + 4) * 128);
            const u32x4 av = {a0[0], a0[1], a1[0], a1[1]};
            const int r = (kk * 16) / RW, w0 = (kk * 16) % RW;
            DPC_UNROLL
            for (int kh = 0; kh < 3; ++kh)
                DPC_UNROLL
                for (int kw = 0; kw < 3; ++kw) {
                    const int pos = (r + kh) * PW + w0 + kw;
                    const u32x2 b0 = lds_read_tr16(st + fb[kw] + pos * 128);
                    const u32x2 b1 = lds_read_tr16(st + fb[kw] + (pos + 4) * 128);
                    const u32x4 bv = {b0[0], b0[1], b1[0], b1[1]};
                    acc[kh * 3 + kw] = mfma_32x32x16_bf16(av, bv, acc[kh * 3 + kw]);
                }
        }
    };
#else
    // The 36 MFMAs of a chunk with their operand reads software-pipelined by hand.  Left to hipcc the loop is "4 reads, wait for
    // them, 1 MFMA" (an LDS round trip per 32-cycle MFMA; 49-61 % matrix-pipe busy measured on the three instantiations).
    // Operand loads in program order: per K step kk the A fragment, then the nine tap fragments B(kk, tap) -- 40 loads of two
    // ds_read_b64_tr_b16 each.  Before MFMA m the loads up to LOOKAHEAD operands beyond its own have been issued, and the counted
    // wait lets exactly those stay in flight (a wave's LDS operations complete in order).  A scheduling barrier after every
    // instruction pins the order written here; the reads are untracked asm, so nothing may touch a destination before its wait.
    constexpr int LOOKAHEAD = 4, RB = 8;
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)lds;
    auto compute = [&](int buf) {
        const uint32_t sa = lds0 + buf * STAGE + fa;
        uint32_t sb[3];
        DPC_UNROLL
        for (int kw = 0; kw < 3; ++kw) sb[kw] = lds0 + buf * STAGE + fb[kw];
        u32x2 alo[2], ahi[2], blo[RB], bhi[RB];
        auto load = [&](auto Ic) {
            constexpr int I = decltype(Ic)::value;
            constexpr int kk = I / 10, r10 = I % 10;
            if constexpr (r10 == 0) {
                asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(alo[kk & 1]) : "v"(sa), "n"((kk * 16) * 128) : "memory");
                __builtin_amdgcn_sched_barrier(0);
                asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(ahi[kk & 1]) : "v"(sa), "n"((kk * 16 + 4) * 128) : "memory");
                __builtin_amdgcn_sched_barrier(0);
            } else {
                constexpr int tap = r10 - 1, kh = tap / 3, kw = tap % 3, j = 9 * kk + tap;
                constexpr int pos = ((kk * 16) / RW + kh) * PW + (kk * 16) % RW + kw;
                asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(blo[j % RB]) : "v"(sb[kw]), "n"(pos * 128) : "memory");
                __builtin_amdgcn_sched_barrier(0);
                asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(bhi[j % RB]) : "v"(sb[kw]), "n"((pos + 4) * 128) : "memory");
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        __builtin_amdgcn_sched_barrier(0);
        static_for<36>([&](auto Mc) {
            constexpr int m = decltype(Mc)::value;
            constexpr int kk = m / 9, tap = m % 9, idx = 10 * kk + 1 + tap;
            constexpr int f_prev = m == 0 ? 0 : ((idx - 1 + (tap == 0 ? -1 : 0)) + 1 + LOOKAHEAD < 40 ? (idx - 1 + (tap == 0 ? -1 : 0)) + 1 + LOOKAHEAD : 40);
            constexpr int f_now = idx + 1 + LOOKAHEAD < 40 ? idx + 1 + LOOKAHEAD : 40;
            static_for<f_now - f_prev>([&](auto Dc) { load(std::integral_constant<int, f_prev + decltype(Dc)::value>{}); });
            asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(2 * (f_now - idx - 1)) : "memory");
            __builtin_amdgcn_sched_barrier(0);
            const u32x4 av = {alo[kk & 1][0], alo[kk & 1][1], ahi[kk & 1][0], ahi[kk & 1][1]};
            const u32x4 bv = {blo[m % RB][0], blo[m % RB][1], bhi[m % RB][0], bhi[m % RB][1]};
            acc[tap] = mfma_32x32x16_bf16(av, bv, acc[tap]);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (m % 4 == 1 && m / 4 < NPC) {   // next chunk's DMA piece m/4, behind an MFMA that has just been issued
                dma_piece(std::integral_constant<int, m / 4>{});
                __builtin_amdgcn_sched_barrier(0);
            }
        });
    };
#endif

    // chunks whose source plane t + kt - pt falls outside the clip contribute nothing: skip them
    auto next_valid = [&](int c) {
        if (p.KT == 1) return c;
        while (c < c_end) {
            const unsigned frame = fdiv((unsigned)c, p.d_cpf);
            const int t = (int)frame - (int)fdiv(frame, p.d_T) * p.T;
            if ((unsigned)(t + kt - p.pt) < (unsigned)p.T) break;
            ++c;
        }
        return c;
    };
    // Chunk c lives in stage c mod NST.  Three stages: chunk c+2 is issued while chunk c is computed, i.e. a DMA piece has two
    // chunk periods (~1-2 us) to land instead of one (worth 3 % on the 32-wide images only).
    // The wait before the barrier is counted -- the pieces of the newest chunk stay in flight (LDS-DMA of a wave lands in issue
    // order) -- and the barrier itself is the raw instruction: __syncthreads() would drain every outstanding load.
    int n_mine = 2;   // DMA pieces this wave issues per chunk
    DPC_UNROLL
    for (int i = 0; i < NIB; ++i) n_mine += (wv + 4 * i < NPB) ? 1 : 0;
    int cur = next_valid(c_begin), buf = 0;
    if (NST == 3) {
        int nxt = c_end;
        if (cur < c_end) {
            issue(cur, 0);
            nxt = next_valid(cur + 1);
            if (nxt < c_end) issue(nxt, 1);
        }
        while (cur < c_end) {
            if (nxt < c_end) wait_vmcnt_upto(n_mine); else wait_vmcnt<0>();
            if (!WP_DBG(4)) barrier_lds_only();   // chunk cur landed for every wave; everyone is done with the stage chunk nxt2 goes to
            const int nxt2 = nxt < c_end ? next_valid(nxt + 1) : c_end;
            dma_prep(nxt2 < c_end ? nxt2 : cur, buf >= 1 ? buf - 1 : 2, nxt2 < c_end);
            if (!WP_DBG(2)) compute(buf);
            else static_for<NPC>([&](auto Ic) { dma_piece(Ic); });
            cur = nxt;
            nxt = nxt2;
            buf = buf == 2 ? 0 : buf + 1;
        }
    } else {
        if (cur < c_end) issue(cur, 0);
        __syncthreads();
        while (cur < c_end) {
            const int nxt = next_valid(cur + 1);
            dma_prep(nxt < c_end ? nxt : cur, buf ^ 1, nxt < c_end);
            if (!WP_DBG(2)) compute(buf);
            else static_for<NPC>([&](auto Ic) { dma_piece(Ic); });
            __syncthreads();
            cur = nxt;
            buf ^= 1;
        }
    }

    // ---- partial slab: rows = co, columns = tap*Ci + ci
    const int ci = tile_c * 64 + wj * 32 + l31;
    DPC_UNROLL
    for (int t = 0; t < 9; ++t)
        DPC_UNROLL
        for (int r = 0; r < 16; ++r) {
            const int co = tile_m * 64 + wi * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
            p.part[((long long)ks * p.Co + co) * p.Kp + (kt * 9 + t) * p.Ci + ci] = acc[t][r];
        }
}

static int wgrad_patch_enabled() {
    static int v = -1;
    if (v < 0) {
        const char* e = getenv("DPC_WGRAD_PATCH");
        v = (e && e[0] == '0') ? 0 : 1;
    }
    return v;
}

// returns 1 when the shape is not served (caller runs the generic kernels); with part == NULL only *nsplit is set
int dpc_wgrad_patch_try(const dpc_conv_desc* d, const void* src, const void* dy, int dy_ld, float* part, int32_t* nsplit,
                        hipStream_t stream) {
    if (!wgrad_patch_enabled()) return 1;
    if (d->dtype_in != DPC_BF16 || d->mode != 0) return 1;
    if (d->KH != 3 || d->KW != 3 || d->st != 1 || d->sh != 1 || d->sw != 1 || d->ph != 1 || d->pw != 1) return 1;
    if (!((d->KT == 1 && d->pt == 0) || (d->KT == 3 && d->pt == 1))) return 1;
    if (d->RT != d->ST || d->RH != d->SH || d->RW != d->SW) return 1;
    if (d->RW > 64) return 1;
    if (d->Ci % 64 || d->Co % 64 || d->src_ld % 8 || dy_ld % 8 || dy_ld < d->Co) return 1;
    const int rwp = d->RW <= 8 ? 8 : d->RW <= 16 ? 16 : d->RW <= 32 ? 32 : 64;  // padded width the kernel is instantiated for
    const int rows = 64 / rwp;
    // padded columns / rows are wasted MFMA work: below 60 % useful positions the generic kernels are faster (4x4 images: 25 %)
    if (d->RH * d->RW * 10 < ((d->RH + rows - 1) / rows) * 64 * 6) return 1;
    const long long M = (long long)d->N * d->RT * d->RH * d->RW;
    const long long sb = M * d->src_ld * 2, db = M * dy_ld * 2;
    if (sb >= (1ll << 31) || db >= (1ll << 31)) return 1;  // 32-bit buffer offsets, DPC_BUF_OOB stays out of range
    WgradPatchParams p;
    p.Co = d->Co; p.Ci = d->Ci; p.dy_ld = dy_ld; p.src_ld = d->src_ld; p.Kp = 9 * d->KT * d->Ci;
    p.KT = d->KT; p.pt = d->pt; p.T = d->RT; p.d_T = make_fastdiv((uint32_t)d->RT);
    p.RH = d->RH; p.W = d->RW; p.NF = d->N * d->RT; p.M = (int)M;
    p.ntm = d->Co / 64; p.ntc = d->Ci / 64;
    p.src_bytes = (unsigned)sb; p.dy_bytes = (unsigned)db;
    p.cpf = (d->RH + rows - 1) / rows;
    p.d_cpf = make_fastdiv((uint32_t)p.cpf);
    const int nchunks = p.NF * p.cpf;
    static const int target_blocks = getenv("DPC_WGRAD_PATCH_BLOCKS") ? atoi(getenv("DPC_WGRAD_PATCH_BLOCKS")) : 512;
    int want = target_blocks / (p.ntm * p.ntc * p.KT);
    if (want < 1) want = 1;
    if (want > nchunks) want = nchunks;
    p.kcps = (nchunks + want - 1) / want;
    p.nks = (nchunks + p.kcps - 1) / p.kcps;
    if (nsplit) *nsplit = p.nks;
    if (!part) return DPC_OK;
    if (!src || !dy) return DPC_ERR_ARG;
    if (((uintptr_t)src % 16) || ((uintptr_t)dy % 16)) return DPC_ERR_UNSUPPORTED;
    p.src = src; p.dy = dy; p.part = part;
#ifdef DPC_WS_PROBE
    p.dbg = getenv("DPC_WS_DBG") ? atoi(getenv("DPC_WS_DBG")) : 0;
#else
    p.dbg = 0;
#endif
    dim3 grid((unsigned)(p.ntm * p.ntc * p.KT * p.nks)), block(256);
    if (rwp == 64) {
        DPC_LAUNCH((wgrad_patch_kernel<64>), grid, block, stream, p);
    } else if (rwp == 32) {
        DPC_LAUNCH((wgrad_patch_kernel<32>), grid, block, stream, p);
    } else if (rwp == 8) {
        DPC_LAUNCH((wgrad_patch_kernel<8>), grid, block, stream, p);
    } else {
        DPC_LAUNCH((wgrad_patch_kernel<16>), grid, block, stream, p);
    }
    return dpc_launch_status();
}
