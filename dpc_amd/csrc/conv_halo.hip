// conv_halo.hip -- 1xKHxKW stride-1 "same" convolutions (BasicBlock2d convs of layer1/layer2,
// backbone/resnet_2d3d.py:24-32,83-116, and their input-gradients) with the input patch staged ONCE
// in LDS.
//
// Why a second conv kernel: in conv_igemm.hip every tap re-gathers its 128 x 128-byte A chunk
// through the cache hierarchy (9x amplification for 3x3) and pays the gather arithmetic per chunk;
// for the small-channel layers (C=64/128, K = 576/1152) that, not the matrix cores, bounds the kernel
// (~450 TFLOP/s).  Here a workgroup owns a TR x TW patch of one frame (128 output positions):
//   * the (TR+KH-1) x (TW+KW-1) input halo is loaded once per tile as 16-byte units into LDS,
//     channels-last, each position padded by 16 B so that the ds_read_b128 of 16 neighbouring
//     positions touch 16 different bank slots;
//   * the A fragment of tap (kh,kw) is the same LDS image read at a constant offset
//     (kh*halo_w + kw) positions further: no per-tap global loads, no gather arithmetic;
//   * the weights (B) live in REGISTERS for the whole kernel: with 128 bytes per position and <= 64
//     output channels a wave's B operand (32 channels x KH*KW*128 B) is KH*KW*16 VGPRs per lane
//     (144 for 3x3), loaded once per workgroup; the tap loop then issues only ds_read_b128 + MFMA --
//     no global loads, no LDS writes, no barrier.  (Measured dead ends, both slower than the generic
//     kernel: streaming B through LDS per tap = one L2 round trip + barrier per 256 MFMA cycles and
//     LDS-write-bound; per-lane B fragments fetched from L2 every tap = TA-bound 16-byte gathers.)
//   * the halo itself arrives by LDS-DMA (global_load_lds_dwordx4, no VGPR staging -- the register
//     file belongs to B) into a second buffer while the current tile is in its MFMA phase; fetching
//     it through registers in small batches cost ~4 exposed HBM round trips per tile;
//   * MFMA fragments, C/D map, LDS-staged coalesced epilogue, fused BN partial sums and residual add
//     are the same as in conv_igemm.hip.
// Serves layer1 of the 2d3d-ResNet (C=64 bf16) -- the layer family where the LDS write bandwidth
// (~80 B/clk/CU for ds_write_b128) of re-staging A per tap bounded the generic kernel -- and, with
// UPP = 2 (32-byte positions: the 16-channel space-to-depth image), the stem's 1x4x4 convolution,
// where one 128-byte K chunk is the four kw taps of a kernel row = four neighbouring positions.
// An input-gradient with unit strides is the same convolution with pad' = K-1-pad and the tap order
// reversed (the [Ci][tap][Co] weight pack is indexed with the flipped tap).
#include "conv_common.h"
#include <stdlib.h>

struct HaloParams {
    const void* src;
    const void* wgt;
    void* out;
    const void* addend;
    float* stats;
    EpiExtra epi;
    int NF, H, W, C, Co, ldw, ldo;
    int ph, pw, flip;
    int TR, TW, lTW, HR, HWd;
    int tiles_w, tiles_per_frame, ntm, gm;
    FastDiv d_tpf, d_tw, d_hwd;
    int vec_out;
    unsigned src_bytes;
    int ws;  // shape served by the role-specialised kernel
    int dbg; // DPC_WS_PROBE builds only: phases to leave out, for timing (scripts/probes/halo_probe.py)
    void* probe_buf;   // DPC_WS_PROBE builds only: where the BNIN load probe (bit 256) stores its interior units
};
#ifdef DPC_WS_PROBE
#define HP_DBG(bit) (p.dbg & (bit))
#else
#define HP_DBG(bit) 0
#endif

// UPP = 16-byte units per position: 8 (one tap = one 128-byte chunk) or 2 (one chunk = 4 kw taps, KW == 4)
template <class T, class TO, int KH, int KW, int UPP>
__global__ __launch_bounds__(256, 2) void conv_halo_kernel(HaloParams p) {
    constexpr int EPU = Elt<T>::PER16;
    constexpr int BM = 128, BN = 64;
    constexpr int SPP = UPP + 1;        // LDS slots per position: data units + one 16-byte pad slot
    constexpr int CBP = SPP * 16;       // padded LDS pitch of a position (144 B / 48 B: conflict-free b128 reads)
    constexpr int MAXPOS = UPP == 8 ? 228 : 252;
    constexpr int NCH = UPP == 8 ? KW : 1;          // 128-byte K chunks per kernel row
    constexpr int NTAPS = KH * NCH;                 // B chunks held in registers
    constexpr int KKSTEP = UPP == 8 ? 32 : CBP;     // A-fragment address step per kk (two units / one position)
    static_assert(UPP == 8 || (UPP == 2 && KW == 4), "a chunk is one tap, or the 4 kw taps of 32-byte positions");
    constexpr int EPO = 16 / (int)sizeof(TO);
    constexpr int UPR = BN / EPO;
    constexpr int OIT = BM * UPR / 256;
    constexpr int DMA_IT = (MAXPOS * SPP + 255) / 256;      // DMA_IT x 256 lanes x 16 B cover MAXPOS positions
    constexpr int STAGE_BYTES = BM * BN * (int)sizeof(TO);
    constexpr int HALO_BYTES = DMA_IT * 256 * 16 > STAGE_BYTES ? DMA_IT * 256 * 16 : STAGE_BYTES;  // doubles as the epilogue staging tile
    __shared__ __attribute__((aligned(16))) unsigned char lds2[2 * HALO_BYTES];
    __shared__ int rowmap[BM];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wv = tid >> 6;
    const int wm = wv >> 1, wn = wv & 1;
    const int l31 = lane & 31, lhi = lane >> 5;
    const int m_prog = blockIdx.x;
    const int esz = (int)sizeof(T);
    const char* const zero = (const char*)dpc_zero16;

    // ---- B operand: this lane's output channel, its k-group of every (tap, kk): resident in VGPRs
    u32x4 fbr[NTAPS][4];
    {
        const int n = wn * 32 + l31;
        const char* wp = n < p.Co ? (const char*)p.wgt + ((long long)n * p.ldw + lhi * EPU) * esz : nullptr;
        DPC_UNROLL
        for (int tap = 0; tap < NTAPS; ++tap) {
            const int tapw = p.flip ? (NTAPS - 1 - tap) : tap;
            DPC_UNROLL
            for (int kk = 0; kk < 4; ++kk)
                fbr[tap][kk] = *(const u32x4*)(wp ? wp + (long long)tapw * 128 + kk * 32 : zero);  // chunk = 128 B of K
        }
    }
    int frag_a[2];
    DPC_UNROLL
    for (int i = 0; i < 2; ++i) {
        const int row = wm * 64 + i * 32 + l31;
        const int r = row >> p.lTW, c = row & (p.TW - 1);
        frag_a[i] = (r * p.HWd + c) * CBP + lhi * 16;
    }
    const int npos = p.HR * p.HWd;
    const int rowpitch = p.HWd * CBP;

    float s1[EPO], s2[EPO];
    DPC_UNROLL
    for (int e = 0; e < EPO; ++e) { s1[e] = 0.f; s2[e] = 0.f; }

    // LDS-DMA of the halo of tile `t` into buffer `bufoff`: slot = position*9 + unit, unit 8 = pad
    auto issue_halo = [&](int t, int bufoff) {
        const unsigned frame = fdiv((unsigned)t, p.d_tpf);
        const int tin = t - (int)frame * p.tiles_per_frame;
        const unsigned th = fdiv((unsigned)tin, p.d_tw);
        const int tw = tin - (int)th * p.tiles_w;
        const int hb = (int)th * p.TR - p.ph, wb = tw * p.TW - p.pw;
        DPC_NOUNROLL
        for (int it = 0; it < DMA_IT; ++it) {  // rolled on purpose: the register file belongs to the B operand
            const int slot = it * 256 + tid;
            const int hpos = slot / SPP, cu = slot - hpos * SPP;
            const unsigned hr = fdiv((unsigned)hpos, p.d_hwd);
            const int hc = hpos - (int)hr * p.HWd;
            const int h = hb + (int)hr, w = wb + hc;
            const bool ok = cu < UPP && hpos < npos && (unsigned)h < (unsigned)p.H && (unsigned)w < (unsigned)p.W;
            const char* a = (const char*)p.src + ((long long)(((int)frame * p.H + h) * p.W + w) * p.C) * esz + cu * 16;
            glds16(ok ? a : zero, lds2 + bufoff + (it * 256 + wv * 64) * 16, lane);
        }
    };

    int bufoff = 0;
    if (m_prog < p.ntm) issue_halo(m_prog, 0);
    for (int mt = m_prog; mt < p.ntm; mt += p.gm) {
        unsigned char* const lds = lds2 + bufoff;
        const unsigned frame = fdiv((unsigned)mt, p.d_tpf);
        const int tin = mt - (int)frame * p.tiles_per_frame;
        const unsigned th = fdiv((unsigned)tin, p.d_tw);
        const int tw = tin - (int)th * p.tiles_w;
        const int h0 = (int)th * p.TR, w0 = tw * p.TW;

        if (tid < BM) {
            const int r = tid >> p.lTW, c = tid & (p.TW - 1);
            const int h = h0 + r, w = w0 + c;
            rowmap[tid] = (h < p.H && w < p.W) ? ((int)frame * p.H + h) * p.W + w : -1;
        }

        f32x16 acc[2];
        DPC_UNROLL
        for (int i = 0; i < 2; ++i)
            DPC_UNROLL
            for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
        __syncthreads();  // this tile's halo has landed (the barrier drains the DMA), rowmap visible
        if (mt + p.gm < p.ntm) issue_halo(mt + p.gm, bufoff ^ HALO_BYTES);  // next tile's halo flies during the MFMA phase
        if constexpr (sizeof(T) == 2) {
            // 2 ds_read_b128 + 2 MFMA per step: fragments are fetched two steps ahead into a ring of three
            // register sets (hand-counted waits, see dpc_rt.h) -- with the compiler's own placement every step
            // waited out an LDS round trip behind only 64 cycles of MFMA work.
            constexpr int S = KH * NCH * 4, D = 2, R = 3;
            u32x4 ring[R][2];
            const unsigned char* rowp[KH][2];
            DPC_UNROLL
            for (int kh = 0; kh < KH; ++kh)
                DPC_UNROLL
                for (int i = 0; i < 2; ++i) rowp[kh][i] = lds + frag_a[i] + kh * rowpitch;
            auto fetch = [&](int s) {
                const int kh = s / (NCH * 4), ch = (s / 4) % NCH, kk = s % 4;
                DPC_UNROLL
                for (int i = 0; i < 2; ++i) lds_read_b128_async(ring[s % R][i], rowp[kh][i] + ch * CBP + kk * KKSTEP);
            };
            DPC_UNROLL
            for (int s = 0; s < D; ++s) fetch(s);
            DPC_UNROLL
            for (int s = 0; s < S; ++s) {
                if (s + D < S) fetch(s + D);
                const int ahead = (S - 1 - s) < D ? (S - 1 - s) : D;
                lds_wait_tie_n(2 * ahead, ring[s % R][0], ring[s % R][1]);
                DPC_UNROLL
                for (int i = 0; i < 2; ++i) acc[i] = mfma_unit<T>(ring[s % R][i], fbr[s / 4][s % 4], acc[i]);
                sched_fence();
            }
        } else {
            DPC_UNROLL
            for (int kh = 0; kh < KH; ++kh) {
                const int roff = kh * rowpitch;  // wave-uniform
                DPC_UNROLL
                for (int ch = 0; ch < NCH; ++ch) {
                    DPC_UNROLL
                    for (int kk = 0; kk < 4; ++kk) {
                        u32x4 fa[2];
                        DPC_UNROLL
                        for (int i = 0; i < 2; ++i) fa[i] = *(const u32x4*)(lds + frag_a[i] + roff + ch * CBP + kk * KKSTEP);
                        DPC_UNROLL
                        for (int i = 0; i < 2; ++i) acc[i] = mfma_unit<T>(fa[i], fbr[kh * NCH + ch][kk], acc[i]);
                    }
                }
            }
        }
        barrier_lds_only();  // every wave is done with the halo before it becomes the staging tile (the DMA keeps flying)

        // ---- epilogue (same scheme as conv_igemm.hip): accumulators -> LDS tile -> 16-byte units
        TO* tile = (TO*)lds;
        DPC_UNROLL
        for (int i = 0; i < 2; ++i)
            DPC_UNROLL
            for (int r = 0; r < 16; ++r) {
                const int row_l = wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
                tile[row_l * BN + wn * 32 + l31] = Elt<TO>::from_f32(acc[i][r]);
            }
        barrier_lds_only();
        const int cu = tid % UPR;
        const int col0 = cu * EPO;
        if (p.vec_out) {
          const bool bnred = p.epi.bn_raw != nullptr;
          float mu[EPO], is[EPO];
          DPC_UNROLL
          for (int e = 0; e < EPO; ++e) {
              const bool okc = bnred && col0 + e < p.Co;
              mu[e] = okc ? p.epi.bn_mean[col0 + e] : 0.f;
              is[e] = okc ? p.epi.bn_invstd[col0 + e] : 0.f;
          }
          DPC_UNROLL
          for (int hb = 0; hb < OIT; hb += 2) {  // two units at a time: the B operand owns most of the register file
            u32x4 ov[2], av[2], rv[2];
            unsigned ab[2], bb[2];
            DPC_UNROLL
            for (int it = 0; it < 2; ++it) {
                const int row_l = (tid + 256 * (hb + it)) / UPR;
                ov[it] = *(const u32x4*)(lds + (row_l * BN + cu * EPO) * (int)sizeof(TO));
                const int row = rowmap[row_l];
                const bool ok = row >= 0 && col0 < p.Co;
                const long long eo = (long long)row * p.ldo + col0;
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
            for (int it = 0; it < 2; ++it) {
                const int row = rowmap[(tid + 256 * (hb + it)) / UPR];
                if (row >= 0 && col0 < p.Co) {
                    u32x4 o = ov[it];
                    epi_unit<TO, EPO>(o, p.addend != nullptr, av[it], ab[it], bnred, rv[it], bb[it], mu, is, s1, s2);
                    *(u32x4*)((char*)p.out + ((long long)row * p.ldo + col0) * (int)sizeof(TO)) = o;
                }
            }
          }
        } else {
            DPC_UNROLL
            for (int it = 0; it < OIT; ++it) {
                const int row_l = (tid + 256 * it) / UPR;
                const int row = rowmap[row_l];
                DPC_UNROLL
                for (int e = 0; e < EPO; ++e) {
                    const int col = col0 + e;
                    if (row >= 0 && col < p.Co) {
                        const long long o = (long long)row * p.ldo + col;
                        float v = Elt<TO>::to_f32(tile[row_l * BN + cu * EPO + e]);
                        if (p.addend) v += Elt<TO>::to_f32(((const TO*)p.addend)[o]);
                        const TO q = Elt<TO>::from_f32(v);
                        ((TO*)p.out)[o] = q;
                        const float vq = Elt<TO>::to_f32(q);
                        s1[e] += vq;
                        s2[e] += vq * vq;
                    }
                }
            }
        }
        barrier_lds_only();  // staging tile and rowmap are free again
        bufoff ^= HALO_BYTES;
    }

    if (p.stats) {
        float* red = (float*)lds2;  // [2][256][EPO]
        DPC_UNROLL
        for (int e = 0; e < EPO; ++e) {
            red[tid * EPO + e] = s1[e];
            red[(256 + tid) * EPO + e] = s2[e];
        }
        __syncthreads();
        if (tid < BN && tid < p.Co) {
            const int cu2 = tid / EPO, e2 = tid % EPO;
            float a = 0.f, b = 0.f;
            for (int t = cu2; t < 256; t += UPR) {
                a += red[t * EPO + e2];
                b += red[(256 + t) * EPO + e2];
            }
            p.stats[((long long)m_prog * 2 + 0) * p.Co + tid] = a;
            p.stats[((long long)m_prog * 2 + 1) * p.Co + tid] = b;
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// Role-specialised variant for the 3x3 / 128-byte-position case (layer1 of the 2d3d-ResNet, bf16).
//
// Timing the phases of conv_halo_kernel separately on layer1 (MI355X, 5.2 M positions): fixed per-tile work
// (halo address arithmetic, barriers) 114 us, halo DMA issue +80, MFMA loop 216, epilogue 135 -- and the
// whole kernel 600: the phases of a workgroup run one after the other and two workgroups per CU overlap
// them only by accident, while the HBM floor of the layer (read x once, write y once) is ~270 us.
// Here a workgroup is 8 waves and every phase has its own waves, pipelined over consecutive tiles:
//   waves 0-3  compute   : B operand resident in VGPRs, 36 steps of {2 ds_read_b128, 2 MFMA} on patch t,
//                          accumulators -> bf16 staging tile t&1;
//   waves 4-5  load      : LDS-DMA of patch t+2 (ring of three patch buffers) through a buffer resource
//                          (image borders = out-of-range lanes = hardware zero fill), offsets precomputed
//                          per lane, counted s_waitcnt vmcnt(15) -- patch t+1 may still be in flight;
//   waves 6-7  store     : epilogue of tile t-1 from the other staging tile: residual add, 16-byte row
//                          stores, batch-norm partial sums.
// Two workgroup barriers per tile: B1(t) "patch t landed, staging t&1 is free", B2(t) "staging t&1 written,
// patch t released".  One workgroup per CU (125 KB of LDS), persistent over its tiles.
#ifndef HALO_WS_D
#define HALO_WS_D 3
#endif
#ifndef HALO_WS_NPB
#define HALO_WS_NPB 3
#endif
// UPP = 8: 3x3 taps over 128-byte positions (one tap = one K chunk).  UPP = 2: the space-to-depth stem, 4x4 taps over
// 32-byte positions (one K chunk = the four kw taps of a kernel row = four neighbouring positions).
// BM = output positions per tile: 128, or 256 for the stem (its MFMA phase is only 16 steps, so the fixed cost of a
// tile -- two workgroup barriers, staging, address arithmetic: ~0.5 us -- was 40 % of its 1.35 us tile period).
// EPI: the fused backward pieces of dpc_conv_igemm_ex (gated residual addend, BatchNorm-backward partial sums); like the addend
// they belong to the helper waves: the mask bytes and the raw unit of tile j are requested one interval before its epilogue.
template <bool HAS_ADD, int UPP, int BM, bool EPI = false>
__global__ __launch_bounds__(512, 2) void conv_halo_ws_kernel(HaloParams p) {
    typedef bf16_t T;
    typedef bf16_t TO;
    constexpr int KH = UPP == 8 ? 3 : 4, NCH = UPP == 8 ? 3 : 1, NTAPS = KH * NCH;
    constexpr int SPP = UPP + 1, CBP = SPP * 16, KKSTEP = UPP == 8 ? 32 : CBP;
    constexpr int BN = 64, EPO = 8, UPR = BN / EPO;
    constexpr int MI = BM / 64;                   // 32-row accumulator blocks per compute wave
    constexpr int NQ = BM / 32;                   // output units per helper lane and tile
    constexpr int LIT = UPP == 8 ? 8 : (BM == 128 ? 3 : 5);       // DMA pieces per helper wave and patch: LIT x 256 lanes x 16 B >= positions x (UPP+1) slots
    constexpr int PATCH = LIT * 256 * 16;
    constexpr int STG = BM * BN * 2;
    constexpr int NPB = HALO_WS_NPB;              // patch ring: two tiles of lookahead (a fourth buffer = all 160 KB of LDS was measured: no change)
    static_assert(NPB * PATCH + 2 * STG <= 160 * 1024 && 2 * (NPB * PATCH + 2 * STG) > 160 * 1024, "one workgroup per CU");
    // (round 3 claimed all 160 KB here as a precaution; the failure it guarded against was a register hazard of igemm_ws_kernel,
    // conv_igemm_ws.hip WS_RETIRE_TAIL_READS -- this kernel's asm reads are all consumed, scripts/asm_hazard_lint.py)
#ifdef DPC_WS_PROBE
    __shared__ __attribute__((aligned(1024))) unsigned char lds[NPB * PATCH + 2 * STG + 1024];   // + the probe's coefficient table
#else
    __shared__ __attribute__((aligned(1024))) unsigned char lds[NPB * PATCH + 2 * STG];
#endif

    const int tid = threadIdx.x;
    const int lane = tid & 63;
#ifdef DPC_SIMT_EMU
    const int wv = tid >> 6;
#else
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
#endif
#ifdef DPC_WS_PROBE
    if (tid < 128) ((float*)(lds + NPB * PATCH + 2 * STG))[tid] = tid < 64 ? 1.0f : 0.01f;
#endif
    // Workgroup b runs on XCD b & 7 (its own L2).  The tiles of one frame are consecutive tile indices and neighbouring row bands
    // share two of their six patch rows: give every XCD a contiguous range of tile slots, so that those re-reads hit its L2
    // instead of going to HBM a second time (round-robin slots put neighbouring bands on different XCDs: 1.5x input traffic).
    int m_prog = blockIdx.x;
    if ((p.gm & 7) == 0 && !HP_DBG(16)) m_prog = (m_prog & 7) * (p.gm >> 3) + (m_prog >> 3);
    const int ntiles = (p.ntm - m_prog + p.gm - 1) / p.gm;  // >= 1
    const char* const zero = (const char*)dpc_zero16;

    auto tile_origin = [&](int mt, int& frame, int& h0, int& w0) {
        const unsigned f = fdiv((unsigned)mt, p.d_tpf);
        const int tin = mt - (int)f * p.tiles_per_frame;
        const unsigned th = fdiv((unsigned)tin, p.d_tw);
        frame = (int)f;
        h0 = (int)th * p.TR;
        w0 = (tin - (int)th * p.tiles_w) * p.TW;
    };

    if (wv < 4) {
        // ------------------------------------------------------------------ compute waves
#ifndef DPC_SIMT_EMU
        if (!HP_DBG(32)) __builtin_amdgcn_s_setprio(3);   // win the SIMD's issue arbitration against the helper wave
#endif
        const int wm = wv >> 1, wn = wv & 1;
        const int l31 = lane & 31, lhi = lane >> 5;
        u32x4 fbr[NTAPS][4];
        {
            const int n = wn * 32 + l31;
            const char* wp = n < p.Co ? (const char*)p.wgt + ((long long)n * p.ldw + lhi * 8) * 2 : nullptr;
            DPC_UNROLL
            for (int tap = 0; tap < NTAPS; ++tap) {
                const int tapw = p.flip ? (NTAPS - 1 - tap) : tap;
                DPC_UNROLL
                for (int kk = 0; kk < 4; ++kk) fbr[tap][kk] = *(const u32x4*)(wp ? wp + (long long)tapw * 128 + kk * 32 : zero);
            }
        }
        int frag_a[MI];
        DPC_UNROLL
        for (int i = 0; i < MI; ++i) {
            const int row = wm * (BM / 2) + i * 32 + l31;
            const int r = row >> p.lTW, c = row & (p.TW - 1);
            frag_a[i] = (r * p.HWd + c) * CBP + lhi * 16;
        }
        const int rowpitch = p.HWd * CBP;
#ifndef DPC_SIMT_EMU
        // The weights are waited for HERE: their loads go through generic pointers (the zero page for channels beyond Co), flat loads
        // count on vmcnt AND lgkmcnt, and with the first use inside the tile loop hipcc put "s_waitcnt vmcnt(0) lgkmcnt(0)" in front of
        // every tile's first MFMA -- behind the eight hand-issued fragment reads of which that MFMA needs two.
        DPC_UNROLL
        for (int tap = 0; tap < NTAPS; ++tap)
            DPC_UNROLL
            for (int kk = 0; kk < 4; ++kk) asm volatile("" : "+v"(fbr[tap][kk]));
#endif
        for (int j = 0; j < ntiles; ++j) {
            f32x16 acc[MI];
            DPC_UNROLL
            for (int i = 0; i < MI; ++i)
                DPC_UNROLL
                for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
            barrier_lds_only();  // B1(j)
            const unsigned char* patch = lds + (j % NPB) * PATCH;
            // 36 steps of {2 ds_read_b128, 2 MFMA}; fragments three steps ahead in a ring of four register sets (two: +3 %; four: no further gain),
            // the constant part of every address in the instruction's offset field (hand-counted waits, dpc_rt.h)
            constexpr int S = KH * NCH * 4, D = HALO_WS_D, R = HALO_WS_D + 1;
            u32x4 ring[R][MI];
            const unsigned char* rowp[KH][MI];
            DPC_UNROLL
            for (int kh = 0; kh < KH; ++kh)
                DPC_UNROLL
                for (int i = 0; i < MI; ++i) rowp[kh][i] = patch + frag_a[i] + kh * rowpitch;
            auto fetch = [&](auto sc) {
                constexpr int s = decltype(sc)::value;
                constexpr int kh = s / (NCH * 4), ch = (s / 4) % NCH, kk = s % 4;
                // probe bit 64: only the first of the MI fragments is read (timing of the loop with half its LDS reads; wrong results)
                DPC_UNROLL
                for (int i = 0; i < MI; ++i)
                    if (i == 0 || !HP_DBG(64)) lds_read_b128_async_off<ch * CBP + kk * KKSTEP>(ring[s % R][i], rowp[kh][i]);
            };
            if (!HP_DBG(2)) static_for<D>(fetch);
            if (!HP_DBG(2)) static_for<S>([&](auto sc) {
                constexpr int s = decltype(sc)::value;
                if constexpr (s + D < S) fetch(std::integral_constant<int, s + D>{});
                constexpr int ahead = (S - 1 - s) < D ? (S - 1 - s) : D;
                DPC_UNROLL
                for (int i = 0; i < MI; i += 2) lds_wait_tie<MI * ahead>(ring[s % R][i], ring[s % R][i + 1]);
                DPC_UNROLL
                for (int i = 0; i < MI; ++i) acc[i] = mfma_32x32x16_bf16(ring[s % R][i], fbr[s / 4][s % 4], acc[i]);
                sched_fence();
            });
            TO* tile = (TO*)(lds + NPB * PATCH + (j & 1) * STG);
            if (!HP_DBG(4))
            DPC_UNROLL
            for (int i = 0; i < MI; ++i)
                DPC_UNROLL
                for (int r = 0; r < 16; ++r) {
                    const int row_l = wm * (BM / 2) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
                    tile[row_l * BN + wn * 32 + l31] = f32_to_bf16(acc[i][r]);
                }
            barrier_lds_only_tracked();  // B2(j): hipcc knows the staging stores have retired (no wait of its own behind the next tile's first fragment reads)
        }
        barrier_lds_only();
        barrier_lds_only();
    } else {
        // ------------------------------------------------------------------ helper waves (one per SIMD)
        // each moves a quarter of every patch and stores a quarter of every tile, so the VALU work that shares
        // the issue port with the MFMAs is the same on all four SIMDs
        const int hw = wv - 4, htid = hw * 64 + lane;
        const BufRsrc rs = make_buf_rsrc(p.src, p.src_bytes);
        const int npos = p.HR * p.HWd;
        unsigned rel[LIT];
        int hrc[LIT];  // (patch row << 16) | patch column; a row no image has when the slot carries no data
        DPC_UNROLL
        for (int it = 0; it < LIT; ++it) {
            const int slot = it * 256 + htid;
            const int hpos = slot / SPP, cu = slot - hpos * SPP;
            const unsigned hr = fdiv((unsigned)hpos, p.d_hwd);
            const int hc = hpos - (int)hr * p.HWd;
            const bool ok = cu < UPP && hpos < npos;
            hrc[it] = ok ? (((int)hr << 16) | hc) : (0x4000 << 16);
            rel[it] = (unsigned)((((int)hr * p.W + hc) * p.C + cu * 8) * 2);
        }
        auto issue = [&](int j) {
            int frame, h0, w0;
            tile_origin(m_prog + j * p.gm, frame, h0, w0);
            const int hb = h0 - p.ph, wb = w0 - p.pw;
            const unsigned base = (unsigned)(((frame * p.H + hb) * p.W + wb) * p.C * 2);
            unsigned char* patch = lds + (j % NPB) * PATCH;
            DPC_UNROLL
            for (int it = 0; it < LIT; ++it) {
                const int h = hb + (hrc[it] >> 16), w = wb + (hrc[it] & 0xffff);
                const bool ok = ((unsigned)h < (unsigned)p.H) & ((unsigned)w < (unsigned)p.W);
                if (!HP_DBG(1)) glds16_buf(rs, ok ? base + rel[it] : DPC_BUF_OOB, 0u, patch + (it * 256 + hw * 64) * 16, lane);
            }
        };
        const int cu = htid & 7, rbase = htid >> 3;  // output unit column; tile rows rbase + 32*q
        const int col0 = cu * EPO;
        float s1[EPO], s2[EPO];  // batch-norm partial sums live only in the helper waves' registers
        DPC_UNROLL
        for (int e = 0; e < EPO; ++e) { s1[e] = 0.f; s2[e] = 0.f; }
        // The residual addend of tile j is requested one interval before it is consumed: with the loads issued
        // inside the epilogue only 4 x 16 B per lane were in flight and the extra 0.67 GB of an input-gradient
        // with residual cost +270 us (the kernel then moves 2 GB and is HBM-bound).
        struct Pre {             // what the epilogue of one tile needs from HBM besides the staged tile
            u32x4 add[NQ], raw[NQ];
            unsigned ab[NQ], bb[NQ];
        };
        // Two buffers, tile t's operands in pre[t & 1], and the tile loop unrolled by two so that each is a fixed set of registers.
        // (Rounds 1-5 kept "cur" and "nxt" and copied nxt into cur at the end of every iteration: a copy of registers whose loads are
        // in flight, so hipcc put s_waitcnt vmcnt(0) in front of it -- right behind the LDS-DMA of patch j+2, which is there to stay
        // in flight for two intervals.  Every tile of the residual / fused-reduction variants drained its whole memory pipeline:
        // their memory side ran 3.3 us per tile against 1.45 for the plain kernel, profiles/r06_halo_epi_probe.txt.)
        Pre pre0, pre1;
        DPC_UNROLL
        for (int q = 0; q < NQ; ++q) {
            pre0.add[q] = u32x4{0u, 0u, 0u, 0u}; pre0.raw[q] = u32x4{0u, 0u, 0u, 0u}; pre0.ab[q] = ~0u; pre0.bb[q] = ~0u;
            pre1.add[q] = u32x4{0u, 0u, 0u, 0u}; pre1.raw[q] = u32x4{0u, 0u, 0u, 0u}; pre1.ab[q] = ~0u; pre1.bb[q] = ~0u;
        }
        const bool bnred = EPI && p.epi.bn_raw != nullptr;
        float mu[EPO], is[EPO];
        DPC_UNROLL
        for (int e = 0; e < EPO; ++e) {
            mu[e] = bnred ? p.epi.bn_mean[col0 + e] : 0.f;
            is[e] = bnred ? p.epi.bn_invstd[col0 + e] : 0.f;
        }
        auto tile_rows = [&](int j, int (&rows)[NQ]) {
            int frame, h0, w0;
            tile_origin(m_prog + j * p.gm, frame, h0, w0);
            DPC_UNROLL
            for (int q = 0; q < NQ; ++q) {
                const int row_l = rbase + 32 * q;
                const int h = h0 + (row_l >> p.lTW), w = w0 + (row_l & (p.TW - 1));
                const bool ok = h < p.H && w < p.W && col0 < p.Co;
                rows[q] = ok ? (frame * p.H + h) * p.W + w : -1;
            }
        };
        auto prefetch = [&](int j, Pre& d) {
            int rows[NQ];
            tile_rows(j, rows);
            // branch-free: rows outside the image read unit 0 of the tensors (their values are never used) -- a load behind a
            // divergent branch makes hipcc lose count of the operations in flight and wait vmcnt(0) at the first use
            DPC_UNROLL
            for (int q = 0; q < NQ; ++q) {
                const long long eo = rows[q] >= 0 ? (long long)rows[q] * p.ldo + col0 : 0ll;
                if (HAS_ADD) {
                    d.add[q] = *(const u32x4*)((const char*)p.addend + eo * 2);
                    if (EPI && p.epi.addend_mask) d.ab[q] = (unsigned)p.epi.addend_mask[eo >> 3];
                }
                if (bnred) {
                    d.raw[q] = *(const u32x4*)((const char*)p.epi.bn_raw + eo * 2);
                    if (p.epi.bn_mask) d.bb[q] = (unsigned)p.epi.bn_mask[eo >> 3];
                }
            }
        };
        auto epilogue = [&](int j, const Pre& d) {
            int rows[NQ];
            tile_rows(j, rows);
            const unsigned char* stg = lds + NPB * PATCH + (j & 1) * STG;
            u32x4 ov[NQ];
            DPC_UNROLL
            for (int q = 0; q < NQ; ++q) ov[q] = *(const u32x4*)(stg + ((rbase + 32 * q) * BN + col0) * 2);
            DPC_UNROLL
            for (int q = 0; q < NQ; ++q) {
                if (rows[q] >= 0) {
                    u32x4 o = ov[q];
                    epi_unit<TO, EPO>(o, HAS_ADD, d.add[q], d.ab[q], bnred, d.raw[q], d.bb[q], mu, is, s1, s2);
                    *(u32x4*)((char*)p.out + ((long long)rows[q] * p.ldo + col0) * 2) = o;
                }
            }
        };
        constexpr bool PRE = HAS_ADD || EPI;   // anything to request ahead of the epilogue
        issue(0);
        DPC_UNROLL
        for (int a = 1; a < NPB - 1; ++a)
            if (ntiles > a) issue(a);
        if (PRE) prefetch(0, pre0);
        // one interval: patch j is awaited and published, the operands of tile j are requested into `fill`, tile j-1 leaves with the
        // operands in `use`, patch j+NPB-1 is issued.  STEADY (1 <= j <= ntiles - NPB): every one of those happens, unconditionally --
        // hipcc's wait-count pass then KNOWS what is in flight (a load or its use behind a condition leaves it guessing, and its
        // guess is vmcnt(0): the patch DMA drained once per tile again).  The first interval and the last NPB run the general form.
        auto touch = [&](const Pre& d) {   // the loads into `d` are waited for HERE, on every path (then their registers are free to be re-used)
#ifndef DPC_SIMT_EMU
            DPC_UNROLL
            for (int q = 0; q < NQ; ++q) asm volatile("" ::"v"(d.add[q]), "v"(d.raw[q]), "v"(d.ab[q]), "v"(d.bb[q]));
#else
            (void)d;
#endif
        };
        auto interval = [&](auto steady_c, int j, Pre& fill, const Pre& use) {
            constexpr bool STEADY = decltype(steady_c)::value;
            // patch j must have landed.  Newer than its pieces are: this wave's stores of older tiles and the LIT pieces of each
            // of the patches j+1 .. j+NPB-2.  Loads (LDS-DMA included) complete in order among themselves, so "at most that many
            // outstanding" implies every piece of patch j is done whatever the stores do.  (The addend / raw / mask loads of the
            // residual and fused-reduction variants are requested between two patches: counting them as absent only makes the
            // wait stricter.)
            if constexpr (STEADY) {
                if (NPB - 2 >= 2) wait_vmcnt<2 * LIT>(); else wait_vmcnt<LIT>();
            } else if (j < ntiles) {
                const int newer = ntiles - 1 - j < NPB - 2 ? ntiles - 1 - j : NPB - 2;
                if (newer >= 2) wait_vmcnt<2 * LIT>(); else if (newer == 1) wait_vmcnt<LIT>(); else wait_vmcnt<0>();
            }
            barrier_lds_only();  // B1(j)
#ifdef DPC_WS_PROBE
            // LOAD PROBE (bit 256; results are wrong by design): what would a BatchNorm-apply + ReLU of the SOURCE cost if it ran here,
            // on the staged patch, instead of as its own kernel?  Every helper lane rewrites the units its own DMA
            // pieces brought (patch j+1: read, 8 x {fma, max}, round, write back; coefficients from a 512-byte LDS table) and stores the
            // interior ones -- the activation the backward pass wants -- to a scratch tensor (bit 512: without that store).
            if (HP_DBG(256) && UPP == 8 && j + 1 < ntiles) {
                wait_vmcnt<0>();
                unsigned char* pn = lds + ((j + 1) % NPB) * PATCH;
                const float* tab = (const float*)(lds + NPB * PATCH + 2 * STG);
                int frame, h0, w0;
                tile_origin(m_prog + (j + 1) * p.gm, frame, h0, w0);
                const unsigned base = (unsigned)(((frame * p.H + h0 - p.ph) * p.W + w0 - p.pw) * p.C * 2);
                DPC_UNROLL
                for (int it = 0; it < LIT; ++it) {
                    const int slot = it * 256 + htid;
                    const int cu_ = slot % SPP;
                    if (cu_ >= UPP) continue;
                    u32x4 v = *(const u32x4*)(pn + slot * 16);
                    const f32x4 sc0 = *(const f32x4*)(tab + cu_ * 8), sc1 = *(const f32x4*)(tab + cu_ * 8 + 4);
                    const f32x4 sh0 = *(const f32x4*)(tab + 64 + cu_ * 8), sh1 = *(const f32x4*)(tab + 64 + cu_ * 8 + 4);
                    float x[8];
                    DPC_UNROLL
                    for (int e = 0; e < 4; ++e) {
                        x[2 * e] = __builtin_bit_cast(float, v[e] << 16);
                        x[2 * e + 1] = __builtin_bit_cast(float, v[e] & 0xffff0000u);
                    }
                    DPC_UNROLL
                    for (int e = 0; e < 4; ++e) {
                        x[e] = fmaxf(fmaf(x[e], sc0[e], sh0[e]), 0.f);
                        x[4 + e] = fmaxf(fmaf(x[4 + e], sc1[e], sh1[e]), 0.f);
                    }
                    DPC_UNROLL
                    for (int e = 0; e < 4; ++e) v[e] = bf16x2_pack(x[2 * e], x[2 * e + 1]);
                    *(u32x4*)(pn + slot * 16) = v;
                    const int hr = hrc[it] >> 16, hc = hrc[it] & 0xffff;
                    const bool inner = hr >= p.ph && hr < p.ph + p.TR && hc >= p.pw && hc < p.pw + p.TW;
                    if (inner && !HP_DBG(512) && p.probe_buf) *(u32x4*)((char*)p.probe_buf + base + rel[it]) = v;
                }
            }
#endif
            if (PRE && (STEADY || (j < ntiles && j >= 1))) prefetch(j, fill);
            if (PRE) touch(use);
            if ((STEADY || j >= 1) && !HP_DBG(8)) epilogue(j - 1, use);
            if (STEADY || j + NPB - 1 < ntiles) issue(j + NPB - 1);  // into the buffer of patch j-1, released at B2(j-1)
            barrier_lds_only();  // B2(j)
        };
        const std::true_type steady = {};
        const std::false_type general = {};
        interval(general, 0, pre0, pre1);
        // tile 0's operands are waited for HERE, outside the loop (hipcc: vmcnt(0), once): on the loop's entry edge nothing of pre0 is
        // pending then, and on its back edge pre0's loads are followed by a patch DMA and the next tile's requests -- without this
        // the merge of the two edges made the first wait INSIDE the loop a vmcnt(0) again (scripts/asm_drain_lint.py found it in the
        // reduction-only variant: the patch DMA drained every second tile)
        if (PRE) touch(pre0);
        int j = 1;
        for (; j + 1 + NPB - 1 < ntiles; j += 2) {   // j odd: tile j's operands in pre1, tile j-1's in pre0
            interval(steady, j, pre1, pre0);
            interval(steady, j + 1, pre0, pre1);
        }
        for (; j <= ntiles; ++j) {
            if (j & 1) interval(general, j, pre1, pre0); else interval(general, j, pre0, pre1);
        }
        // every patch has been consumed (last B2 passed): the first patch buffer becomes the reduction scratch
        float* red = (float*)lds;  // [2][256][EPO]
        DPC_UNROLL
        for (int e = 0; e < EPO; ++e) {
            red[htid * EPO + e] = s1[e];
            red[(256 + htid) * EPO + e] = s2[e];
        }
    }

    if (p.stats) {
        // 32 helper lanes hold partial sums of the same 8 columns
        const float* red = (const float*)lds;
        __syncthreads();
        if (tid < BN && tid < p.Co) {
            const int cu2 = tid / EPO, e2 = tid % EPO;
            float a = 0.f, b = 0.f;
            for (int t = cu2; t < 256; t += UPR) {
                a += red[t * EPO + e2];
                b += red[(256 + t) * EPO + e2];
            }
            p.stats[((long long)m_prog * 2 + 0) * p.Co + tid] = a;
            p.stats[((long long)m_prog * 2 + 1) * p.Co + tid] = b;
        }
    }
}

// ---------------------------------------------------------------- host side
static bool halo_plan(const dpc_conv_desc* d, HaloParams* p, bool allow_ws = true) {
    if (d->KT != 1 || d->st != 1 || d->sh != 1 || d->sw != 1 || d->pt != 0) return false;
    if (d->RT != d->ST || d->RH != d->SH || d->RW != d->SW) return false;
    if (d->src_ld != d->Ci || d->Co > 64) return false;
    if (d->dtype_in != d->dtype_out) return false;
    const int esz = d->dtype_in == DPC_BF16 ? 2 : 4;
    const bool k33 = d->KH == 3 && d->KW == 3 && d->Ci * esz == 128;                   // BasicBlock2d conv, 128 B per position
    const bool k44 = d->KH == 4 && d->KW == 4 && d->Ci * esz == 32 && d->mode == 0;    // space-to-depth stem, 32 B per position
    if (!k33 && !k44) return false;
    int tw = 32;
    while (tw > d->RW && tw > 4) tw >>= 1;
    p->TW = tw;
    p->lTW = ilog2_exact(tw);
    static const int ws_on = getenv("DPC_HALO_WS") ? atoi(getenv("DPC_HALO_WS")) : 1;
    const bool ws_ok = allow_ws && ws_on && d->dtype_in == DPC_BF16 && d->ldo % 8 == 0 && d->Co % 8 == 0;
    const int bm = (ws_ok && k44) ? 256 : 128;  // the role-specialised stem variant works on 256-position tiles
    p->TR = bm / tw;
    p->HR = p->TR + d->KH - 1;
    p->HWd = p->TW + d->KW - 1;
    if (bm == 128 && p->HR * p->HWd > (k33 ? 228 : 252)) return false;
    p->NF = d->N * d->RT;
    p->H = d->RH; p->W = d->RW; p->C = d->Ci; p->Co = d->Co; p->ldw = d->ldw; p->ldo = d->ldo;
    p->flip = d->mode == 1;
    p->ph = d->mode == 1 ? d->KH - 1 - d->ph : d->ph;
    p->pw = d->mode == 1 ? d->KW - 1 - d->pw : d->pw;
    p->tiles_w = (d->RW + p->TW - 1) / p->TW;
    const int tiles_h = (d->RH + p->TR - 1) / p->TR;
    p->tiles_per_frame = tiles_h * p->tiles_w;
    const long long ntm = (long long)p->NF * p->tiles_per_frame;
    if (ntm >= (1ll << 30) || (long long)p->NF * d->RH * d->RW * d->Ci >= (1ll << 31)) return false;  // 32-bit element offsets
    p->ntm = (int)ntm;
    p->gm = p->ntm < 1024 ? p->ntm : 1024;  // persistent workgroups: B is loaded once per workgroup
    const long long sbytes = (long long)p->NF * d->RH * d->RW * d->Ci * esz;
    p->src_bytes = sbytes < (1ll << 31) ? (unsigned)sbytes : 0u;
    p->ws = ws_ok && p->src_bytes > 0 && ((k33 && p->HR * p->HWd * 9 <= 8 * 256) || (k44 && p->HR * p->HWd * 3 <= 5 * 256));
    if (bm == 256 && !p->ws) return halo_plan(d, p, false);
    static const int ws_gm = getenv("DPC_HALO_WS_GM") ? atoi(getenv("DPC_HALO_WS_GM")) : 256;  // test tiers shrink it
    if (p->ws) {  // one resident workgroup per CU (minus the CUs reserved for a concurrent collective)
        const int cap = dpc_persistent_grid(ws_gm);
        p->gm = p->ntm < cap ? p->ntm : cap;
    }
    p->d_tpf = make_fastdiv(p->tiles_per_frame);
    p->d_tw = make_fastdiv(p->tiles_w);
    p->d_hwd = make_fastdiv(p->HWd);
    return true;
}

int dpc_conv_halo_rows(const dpc_conv_desc* d) {
    HaloParams p;
    if (!d || !halo_plan(d, &p)) return 0;
    return p.gm;
}

#ifdef DPC_WS_PROBE
static void* g_halo_probe_buf = nullptr;
extern "C" int dpc_probe_set_halo_buf(void* buf) { g_halo_probe_buf = buf; return 0; }   // scripts/probes/halo_probe.py
#endif

// returns 1 when the shape is not served by this kernel (caller falls back to dpc_conv_igemm's generic path)
int dpc_conv_halo_try(const dpc_conv_desc* d, const void* src, const void* wgt, void* out, const void* addend, float* stats,
                      const EpiExtra& epi, hipStream_t stream) {
    HaloParams p;
    if (!halo_plan(d, &p)) return 1;
    p.src = src; p.wgt = wgt; p.out = out; p.addend = addend; p.stats = stats; p.epi = epi;
#ifdef DPC_WS_PROBE
    p.dbg = getenv("DPC_WS_DBG") ? atoi(getenv("DPC_WS_DBG")) : 0;
    p.probe_buf = g_halo_probe_buf;
#else
    p.dbg = 0;
    p.probe_buf = nullptr;
#endif
    const int epo = d->dtype_out == DPC_BF16 ? 8 : 4;
    p.vec_out = (d->Co % epo == 0 && d->ldo % epo == 0 && ((uintptr_t)out % 16 == 0) && ((uintptr_t)addend % 16 == 0)) ? 1 : 0;
    const bool ws_go = p.ws && p.vec_out && ((uintptr_t)src % 16 == 0) && !(d->KH == 4 && addend);
    if (p.ws && !ws_go) {  // the specialised kernel declined at launch: generic patch kernel, same number of stats rows as promised
        const int promised = p.gm, vo = p.vec_out;
        if (!halo_plan(d, &p, false)) return 1;
        p.src = src; p.wgt = wgt; p.out = out; p.addend = addend; p.stats = stats; p.epi = epi;
        p.vec_out = vo;
        if (p.gm > promised) p.gm = promised;
    }
    dim3 grid((unsigned)p.gm), block(256);
    if (ws_go) {
        if (d->KH == 4) {
            if (epi_any(epi)) return DPC_ERR_UNSUPPORTED;  // the stem has no input-gradient
            DPC_LAUNCH((conv_halo_ws_kernel<false, 2, 256>), grid, dim3(512), stream, p);
        } else if (epi_any(epi)) {
            if (addend) {
                DPC_LAUNCH((conv_halo_ws_kernel<true, 8, 128, true>), grid, dim3(512), stream, p);
            } else {
                DPC_LAUNCH((conv_halo_ws_kernel<false, 8, 128, true>), grid, dim3(512), stream, p);
            }
        } else if (addend) {
            DPC_LAUNCH((conv_halo_ws_kernel<true, 8, 128>), grid, dim3(512), stream, p);
        } else {
            DPC_LAUNCH((conv_halo_ws_kernel<false, 8, 128>), grid, dim3(512), stream, p);
        }
        return dpc_launch_status();
    }
    dpc_plan_detail("ws_declined=%d", (int)(p.ws && !ws_go));
    if (d->KH == 3) {
        if (d->dtype_in == DPC_F32) {
            DPC_LAUNCH((conv_halo_kernel<float, float, 3, 3, 8>), grid, block, stream, p);
        } else {
            DPC_LAUNCH((conv_halo_kernel<bf16_t, bf16_t, 3, 3, 8>), grid, block, stream, p);
        }
    } else {
        if (d->dtype_in == DPC_F32) {
            DPC_LAUNCH((conv_halo_kernel<float, float, 4, 4, 2>), grid, block, stream, p);
        } else {
            DPC_LAUNCH((conv_halo_kernel<bf16_t, bf16_t, 4, 4, 2>), grid, block, stream, p);
        }
    }
    return dpc_launch_status();
}
