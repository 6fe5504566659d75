// conv_wgrad_stem.hip -- weight gradient of the stem convolution in its space-to-depth form (1x4x4 taps over
// the 16-channel 2x2 space-to-depth image, stride 1; backbone/resnet_2d3d.py:211 = Conv3d(3,64,(1,7,7),
// s(1,2,2),p(0,3,3)), see dpc_pack_input_s2d), bf16.
//
//   part[ks][co][tap*16 + ch] = sum_{positions of split ks} dy[pos][co] * xs[pos shifted by tap][ch]
//
// The generic kernel (conv_wgrad.hip) re-gathers the source once per tap: per 64 positions it stages 8 KB of
// dy and 32 KB of shifted copies of 2 KB of source, 10 LDS-DMA pieces per wave for 16 MFMAs -- DMA-issue bound
// (1.06 ms for 3.2 GB of unique bytes).  Here a chunk is 64 consecutive positions of ONE image row; the source
// arrives once as a 4 x 68 position patch (8.7 KB, image borders = out-of-range buffer lanes = zeros), every
// tap is the same patch read a few positions further (an instruction immediate), and the fragments come from
// ds_read_b64_tr_b16 as in conv_wgrad_patch.hip: 17 pieces per chunk instead of 40.
// A workgroup owns 64 co x all 256 reduction columns: wave (wi, wj) = co half wi x kernel rows {2wj, 2wj+1}
// (four 32 x 32 accumulators: a column block is the pair of taps kw = 2c, 2c+1 of one kernel row x 16 channels).
// The patch needs no swizzle: with 32-byte positions a transpose read's 16 lanes cover 128 contiguous bytes.
#include "conv_common.h"
#include <stdlib.h>

struct WgradStemParams {
    const void* src;
    const void* dy;
    float* part;
    int Co, dy_ld;
    int H, W, NF, nseg;   // image, frames, 64-position segments per row
    int ph, pw;
    int nks, kcps, ntm;
    long long src_bytes, dy_bytes;
    FastDiv d_cpf, d_nseg;
    int cpf;              // chunks per frame = H * nseg
    // FUSED: dy is never read -- it is produced on the fly from the stem's max-pool / BatchNorm backward
    const void* raw;      // [NF][H][W][Co] raw conv output (same geometry as dy)
    const void* dpool;    // [NF][Ho][Wo][Co] gradient at the pooled output
    const uint8_t* argmax;
    const float *mean, *invstd, *gamma, *coef;
    int Ho, Wo;
    int dbg;              // DPC_WS_PROBE builds only (scripts/stem_bench.py --probe): phases to leave out, for timing
};
#ifdef DPC_WS_PROBE
#define SF_DBG(bit) (p.dbg & (bit))
#else
#define SF_DBG(bit) 0
#endif

__global__ __launch_bounds__(256, 3) void wgrad_stem_kernel(WgradStemParams p) {
    constexpr int PW = 68, NPOS = 4 * PW;       // patch: 4 rows x (64 + 3, padded to 68) positions of 32 bytes
    constexpr int NPB = (NPOS + 31) / 32;        // 9 pieces of 32 positions
    constexpr int NIB = (NPB + 3) / 4;
    constexpr int STAGE = 8192 + NPB * 1024;
    __shared__ __attribute__((aligned(1024))) unsigned char lds[2 * STAGE];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
#ifdef DPC_SIMT_EMU
    const int wv = tid >> 6;
#else
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
#endif
    const int wi = wv >> 1, wj = wv & 1;
    const int l31 = lane & 31, lhi = lane >> 5;
    int id = blockIdx.x;
    {
        const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, x = id & 7;
        id = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (id >> 3);
    }
    const int tile_m = id % p.ntm;
    const int ks = id / p.ntm;
    const int nchunks = p.NF * p.cpf;
    const int c_begin = ks * p.kcps;
    const int c_end = (c_begin + p.kcps < nchunks) ? c_begin + p.kcps : nchunks;

    // The stem's dy is 2.7 GB at batch 128 and buffer offsets are 32-bit: every workgroup addresses its own window,
    // which starts at its first chunk (minus the patch's top/left margin for the source) and is far below 2 GB long.
    long long a_lo, b_lo;
    {
        const unsigned frame = fdiv((unsigned)c_begin, p.d_cpf);
        const int rem = c_begin - (int)frame * p.cpf;
        const unsigned h = fdiv((unsigned)rem, p.d_nseg);
        const int w0 = (rem - (int)h * p.nseg) * 64;
        const long long pos = ((long long)frame * p.H + (int)h) * p.W + w0;
        a_lo = pos * p.dy_ld * 2;
        b_lo = (pos - (long long)p.ph * p.W - p.pw) * 32;
        if (b_lo < 0) b_lo = 0;
    }
    const long long a_len = p.dy_bytes - a_lo, b_len = p.src_bytes - b_lo;
    const BufRsrc rs_a = make_buf_rsrc((const char*)p.dy + a_lo, (unsigned)(a_len < 0x7fffffffll ? a_len : 0x7fffffffll));
    const BufRsrc rs_b = make_buf_rsrc((const char*)p.src + b_lo, (unsigned)(b_len < 0x7fffffffll ? b_len : 0x7fffffffll));
    // dy sub-tile [64 positions][64 co], 128-byte rows, slot swizzled by 2*(position & 3) (source side)
    const int pl = lane >> 3;
    const int lslot = (lane & 7) ^ (2 * (pl & 3));
    unsigned a_off[2];
    int a_pos[2];
    DPC_UNROLL
    for (int i = 0; i < 2; ++i) {
        a_pos[i] = 8 * (wv + 4 * i) + pl;
        a_off[i] = (unsigned)(a_pos[i] * p.dy_ld + tile_m * 64 + lslot * 8) * 2u;
    }
    // source patch [4 rows][68 positions][16 ch], linear
    unsigned b_off[NIB];
    int b_row[NIB], b_col[NIB];  // patch row / column of this lane's position; row = huge when the slot is beyond the patch
    DPC_UNROLL
    for (int i = 0; i < NIB; ++i) {
        const int pp = 32 * (wv + 4 * i) + (lane >> 1);
        const int prow = pp / PW, pcol = pp % PW;
        b_row[i] = pp < NPOS ? prow : (1 << 28);
        b_col[i] = pcol;
        b_off[i] = (unsigned)((prow * p.W + pcol) * 32 + (lane & 1) * 16);
    }

    f32x16 acc[4];
    DPC_UNROLL
    for (int t = 0; t < 4; ++t)
        DPC_UNROLL
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    auto issue = [&](int chunk, int buf) {
        unsigned char* stage = lds + buf * STAGE;
        const unsigned frame = fdiv((unsigned)chunk, p.d_cpf);
        const int rem = chunk - (int)frame * p.cpf;
        const unsigned h = fdiv((unsigned)rem, p.d_nseg);
        const int w0 = (rem - (int)h * p.nseg) * 64;
        const long long pos = ((long long)frame * p.H + (int)h) * p.W + w0;
        const unsigned a_base = (unsigned)(pos * p.dy_ld * 2 - a_lo);
        // patch origin = image (h - ph, w0 - pw); offsets may wrap below zero for border positions (those lanes are masked)
        const unsigned b_base = (unsigned)((pos - (long long)p.ph * p.W - p.pw) * 32 - b_lo);
        DPC_UNROLL
        for (int i = 0; i < 2; ++i) {
            const bool ok = w0 + a_pos[i] < p.W;
            glds16_buf(rs_a, ok ? a_base + a_off[i] : DPC_BUF_OOB, 0u, stage + (wv + 4 * i) * 1024, lane);
        }
        DPC_UNROLL
        for (int i = 0; i < NIB; ++i) {
            if (wv + 4 * i < NPB) {
                const bool ok = ((unsigned)((int)h - p.ph + b_row[i]) < (unsigned)p.H) & ((unsigned)(w0 - p.pw + b_col[i]) < (unsigned)p.W);
                glds16_buf(rs_b, ok ? b_base + b_off[i] : DPC_BUF_OOB, 0u, stage + 8192 + (wv + 4 * i) * 1024, lane);
            }
        }
    };

    // fragment lane offsets
    const int gq = lane >> 4, s16 = lane & 15, ph4 = s16 >> 2;
    int fa;
    {
        const int colb = (wi * 32 + (gq & 1) * 16 + 4 * (s16 & 3)) * 2;
        fa = ((gq >> 1) * 8 + ph4) * 128 + ((((colb >> 4) ^ (2 * ph4)) & 7) << 4) + (colb & 15);
    }
    // patch: position (gq>>1)*8 + ph4, + (gq&1) for the odd tap of the column block's pair, channels 4*(s16&3)..
    const int fb = 8192 + ((gq >> 1) * 8 + ph4 + (gq & 1)) * 32 + 8 * (s16 & 3);

#ifdef DPC_SIMT_EMU
    auto compute = [&](int buf) {
        const unsigned char* st = lds + buf * STAGE;
        DPC_UNROLL
        for (int kk = 0; kk < 4; ++kk) {
            const u32x2 a0 = lds_read_tr16(st + fa + (kk * 16) * 128);
            const u32x2 a1 = lds_read_tr16(st + fa + (kk * 16 + 4) * 128);
            const u32x4 av = {a0[0], a0[1], a1[0], a1[1]};
            DPC_UNROLL
            for (int c = 0; c < 4; ++c) {  // column block: kernel row 2*wj + (c>>1), taps kw = 2*(c&1), 2*(c&1)+1
                const unsigned char* bp = st + fb + (((c >> 1) * PW + kk * 16 + 2 * (c & 1)) * 32) + wj * (2 * PW * 32);
                const u32x2 b0 = lds_read_tr16(bp);
                const u32x2 b1 = lds_read_tr16(bp + 4 * 32);
                const u32x4 bv = {b0[0], b0[1], b1[0], b1[1]};
                acc[c] = mfma_32x32x16_bf16(av, bv, acc[c]);
            }
        }
    };
#else
    // The 16 MFMAs of a chunk with their operand reads software-pipelined by hand (same scheme as conv_wgrad_patch.hip: loads in
    // program order A(kk), B(kk, 0..3); before MFMA m the loads up to LOOKAHEAD operands beyond its own are issued and the counted
    // wait lets exactly those stay in flight; a scheduling barrier after every instruction pins the order).
    constexpr int LOOKAHEAD = 4, RB = 8;
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)lds;
    auto compute = [&](int buf) {
        const uint32_t sa = lds0 + buf * STAGE + fa;
        const uint32_t sb = lds0 + buf * STAGE + fb + wj * (2 * PW * 32);
        u32x2 alo[2], ahi[2], blo[RB], bhi[RB];
        auto load = [&](auto Ic) {
            constexpr int I = decltype(Ic)::value;
            constexpr int kk = I / 5, r5 = I % 5;
            if constexpr (r5 == 0) {
                asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(alo[kk & 1]) : "v"(sa), "n"((kk * 16) * 128) : "memory");
                __builtin_amdgcn_sched_barrier(0);
                asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(ahi[kk & 1]) : "v"(sa), "n"((kk * 16 + 4) * 128) : "memory");
                __builtin_amdgcn_sched_barrier(0);
            } else {
                constexpr int c = r5 - 1, j = 4 * kk + c;
                constexpr int off = ((c >> 1) * PW + kk * 16 + 2 * (c & 1)) * 32;
                asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(blo[j % RB]) : "v"(sb), "n"(off) : "memory");
                __builtin_amdgcn_sched_barrier(0);
                asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(bhi[j % RB]) : "v"(sb), "n"(off + 4 * 32) : "memory");
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        __builtin_amdgcn_sched_barrier(0);
        static_for<16>([&](auto Mc) {
            constexpr int m = decltype(Mc)::value;
            constexpr int kk = m / 4, c = m % 4, idx = 5 * kk + 1 + c;
            constexpr int idx_prev = idx - 1 + (c == 0 ? -1 : 0);
            constexpr int f_prev = m == 0 ? 0 : (idx_prev + 1 + LOOKAHEAD < 20 ? idx_prev + 1 + LOOKAHEAD : 20);
            constexpr int f_now = idx + 1 + LOOKAHEAD < 20 ? idx + 1 + LOOKAHEAD : 20;
            static_for<f_now - f_prev>([&](auto Dc) { load(std::integral_constant<int, f_prev + decltype(Dc)::value>{}); });
            asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(2 * (f_now - idx - 1)) : "memory");
            __builtin_amdgcn_sched_barrier(0);
            const u32x4 av = {alo[kk & 1][0], alo[kk & 1][1], ahi[kk & 1][0], ahi[kk & 1][1]};
            const u32x4 bv = {blo[m % RB][0], blo[m % RB][1], bhi[m % RB][0], bhi[m % RB][1]};
            acc[c] = mfma_32x32x16_bf16(av, bv, acc[c]);
            __builtin_amdgcn_sched_barrier(0);
        });
    };
#endif

    if (c_begin < c_end) issue(c_begin, 0);
    __syncthreads();
    for (int ch = c_begin; ch < c_end; ++ch) {
        const int buf = (ch - c_begin) & 1;
        if (ch + 1 < c_end) issue(ch + 1, buf ^ 1);
        compute(buf);
        __syncthreads();
    }

    // partial slab rows = co, columns = tap*16 + ch = (kernel row)*64 + (c&1)*32 + lane column
    DPC_UNROLL
    for (int c = 0; c < 4; ++c) {
        const int col = (2 * wj + (c >> 1)) * 64 + (c & 1) * 32 + l31;
        DPC_UNROLL
        for (int r = 0; r < 16; ++r) {
            const int co = tile_m * 64 + wi * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
            p.part[((long long)ks * p.Co + co) * 256 + col] = acc[c][r];
        }
    }
}


// ---------------------------------------------------------------------------------------------------------------------
// Fused form (dpc_stem_wgrad_fused): the [64 positions][64 co] gradient sub-tile is not read from a tensor but BUILT in LDS
// from the stem's raw output, the gradient at the pooled output (routed through the saved argmax bytes) and the BatchNorm-
// backward coefficients -- what dpc_pool_bn_bwd_apply wrote to a 2.7 GB dz tensor (1.2 ms at batch 128) that the plain
// kernel then read back (0.66 ms).  Round 2 built this with per-lane register gathers one chunk ahead: 2.14 ms, slower than the
// two kernels (the gathers saw HBM latency that one chunk of MFMA work -- 0.5 us -- does not cover).  Here EVERY input of a
// chunk arrives by LDS-DMA three chunks ahead (ring of four 33 KB stages, one workgroup of eight waves per CU):
//     raw tile 8 KB (8 pieces) | source patch 9 pieces | pooled gradient, 2 rows x 33 positions (10 pieces) | argmax bytes (6)
// and the TWELVE waves have roles: waves 4-11 (two per SIMD) issue the DMA and turn the raw tile of chunk c+1 into the gradient
// tile IN PLACE (one 16-byte unit per lane; the pooling windows of a position are read from the staged pooled rows; position
// parity = wave parity and the image row is chunk-uniform, so the tap a window must point at is a wave-uniform constant) while
// waves 0-3 (one per SIMD) run the 64 MFMAs of chunk c exactly as the plain kernel does (co half x kernel-row pair,
// hand-pipelined transpose reads).  One barrier per chunk.  Same chunking, same K order per accumulator and the same arithmetic on the same bf16
// values as the two-kernel form: bit-identical results (case_stem_wgrad_fused).
// History (MI355X, batch 128, 2.7 GB raw tensor; the two kernels: 1.23 + 0.65 ms):
//   4.2 ms  first build: a table-driven DMA loop hipcc did not unroll (table in scratch, LDS destination in a waterfall loop) and
//           __syncthreads -- both wait for EVERY outstanding LDS-DMA piece, i.e. expose the memory latency of chunk c+3 per chunk;
//   2.19 ms straight-line DMA issue, LDS-only barriers, raw LDS accessors.  Phase ablation (scripts/stem_bench.py --probe): loop +
//           barriers 0.34, DMA issue 0.34, memory traffic 0.26, LDS reads 0.40, builder VALU 0.67, MFMAs 0.35 ms -- they ADD UP
//           (2.36): with all eight waves doing the same phase at the same time nothing overlaps;
//   2.39 ms roles with ONE builder wave per SIMD (two units per lane): slower -- a lone wave cannot hide its own VALU dependency
//           and LDS latencies (builder math alone 1.03 ms);
//   roles with two builder waves + one MFMA wave per SIMD (this version).
constexpr int SF_PW = 68, SF_NPOS = 4 * SF_PW;
constexpr int SF_A = 0, SF_X = 8192, SF_P = SF_X + 9 * 1024, SF_M = SF_P + 10 * 1024, SF_STAGE = SF_M + 6 * 1024, SF_NS = 4;

__global__ __launch_bounds__(768) void wgrad_stem_fused_kernel(WgradStemParams p) {
    static_assert(SF_NS * SF_STAGE <= 160 * 1024, "ring fits the CU's LDS");
    __shared__ __attribute__((aligned(1024))) unsigned char lds[SF_NS * SF_STAGE];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
#ifdef DPC_SIMT_EMU
    const int wv = tid >> 6;
#else
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
#endif
    int id = blockIdx.x;
    {
        const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, x = id & 7;
        id = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (id >> 3);
    }
    const int tile_m = id % p.ntm;
    const int ks = id / p.ntm;
    const int nchunks = p.NF * p.cpf;
    const int c_begin = ks * p.kcps;
    const int c_end = (c_begin + p.kcps < nchunks) ? c_begin + p.kcps : nchunks;
    const int nck = c_end > c_begin ? c_end - c_begin : 0;

    // 32-bit buffer offsets: the raw tensor (2.7 GB) and the source are addressed through per-workgroup windows (as in the plain
    // kernel); the pooled tensors are below 2 GB (checked on the host)
    long long a_lo, b_lo;
    {
        const unsigned frame = fdiv((unsigned)c_begin, p.d_cpf);
        const int rem = c_begin - (int)frame * p.cpf;
        const unsigned h = fdiv((unsigned)rem, p.d_nseg);
        const int w0 = (rem - (int)h * p.nseg) * 64;
        const long long pos = ((long long)frame * p.H + (int)h) * p.W + w0;
        a_lo = pos * p.dy_ld * 2;
        b_lo = (pos - (long long)p.ph * p.W - p.pw) * 32;
        if (b_lo < 0) b_lo = 0;
    }
    const long long a_len = p.dy_bytes - a_lo, b_len = p.src_bytes - b_lo;
    const long long pooled_elems = (long long)p.NF * p.Ho * p.Wo * p.Co;
    const BufRsrc rs_a = make_buf_rsrc((const char*)p.raw + a_lo, (unsigned)(a_len < 0x7fffffffll ? a_len : 0x7fffffffll));
    const BufRsrc rs_b = make_buf_rsrc((const char*)p.src + b_lo, (unsigned)(b_len < 0x7fffffffll ? b_len : 0x7fffffffll));
    const BufRsrc rs_p = make_buf_rsrc(p.dpool, (unsigned)(pooled_elems * 2));
    const BufRsrc rs_m = make_buf_rsrc(p.argmax, (unsigned)pooled_elems);

    // ---- this wave's DMA pieces (33 per chunk over 8 waves; straight-line code with named per-lane constants: a loop over a
    //      table of slots was not unrolled by hipcc, the table went to scratch, the LDS destination into a waterfall loop --
    //      and a scratch load waits for every LDS-DMA piece issued before it)
    //   A  raw piece wv: positions 8 wv .. 8 wv + 7, slot = logical unit ^ 2 (position & 3)     (the dy tile image of the plain kernel)
    //   B  source patch piece wv (32 positions x 32 bytes); wave 0 also piece 8 (E)
    //   C  pooled gradient piece wv: row wv / 5, positions 8 (wv % 5) + (lane >> 3), unit lane & 7 of this co tile
    //   D  waves 0-1: pooled gradient pieces 8, 9 (row 1); waves 2-7: argmax piece wv - 2: row (wv-2) / 3, positions
    //      16 ((wv-2) % 3) + (lane >> 2), 16 bytes (lane & 3) of this co tile
    // waves 0-3 run the MFMAs, waves 4-11 (bw = 0..7, two per SIMD) issue the DMA and build the gradient tiles
    const int bw = wv >= 4 ? wv - 4 : 0;
    const int posA = 8 * bw + (lane >> 3);
    const unsigned relA = (unsigned)(posA * p.dy_ld + tile_m * 64 + ((lane & 7) ^ (2 * ((lane >> 3) & 3))) * 8) * 2u;
    auto patch_consts = [&](int piece, unsigned& rel, int& prow, int& pcol) {
        const int pp = 32 * piece + (lane >> 1);
        prow = pp < SF_NPOS ? pp / SF_PW : (1 << 20);
        pcol = pp % SF_PW;
        rel = (unsigned)(((pp / SF_PW) * p.W + pcol) * 32 + (lane & 1) * 16);
    };
    unsigned relB, relE;
    int prowB, pcolB, prowE, pcolE;
    patch_consts(bw, relB, prowB, pcolB);
    patch_consts(8, relE, prowE, pcolE);
    const int rC = bw / 5, ocC = 8 * (bw % 5) + (lane >> 3);
    const unsigned relC = (unsigned)(ocC * p.Co + tile_m * 64 + (lane & 7) * 8) * 2u;
    const bool dIsP = bw < 2;
    const int rD = dIsP ? 1 : (bw - 2) / 3;
    const int ocD = dIsP ? 8 * (3 + bw) + (lane >> 3) : 16 * ((bw - 2) % 3) + (lane >> 2);
    const unsigned relD = dIsP ? (unsigned)(ocD * p.Co + tile_m * 64 + (lane & 7) * 8) * 2u : (unsigned)(ocD * p.Co + tile_m * 64 + (lane & 3) * 16);
    const int n_mine = bw == 0 ? 5 : 4;   // DMA instructions this wave issues per chunk

    // ---- chunk cursors.  A chunk's frame / row / segment and its position relative to the workgroup's first chunk are carried
    //      along and advanced per iteration: the scalar ALU is shared by the whole CU, and the first role-specialised version
    //      spent 1 266 scalar instructions per chunk (divisions, 64-bit offsets, stage multiplies, exec-mask juggling; rocprofv3
    //      SQ_INSTS_SALU) -- a third of its run time.
    struct Cur { int frame, h, seg, posrel; };
    auto cur_at_begin = [&]() {
        const unsigned f = fdiv((unsigned)c_begin, p.d_cpf);
        const int rem = c_begin - (int)f * p.cpf;
        const unsigned hh = fdiv((unsigned)rem, p.d_nseg);
        Cur c = {(int)f, (int)hh, rem - (int)hh * p.nseg, 0};
        return c;
    };
    auto advance = [&](Cur& c) {
        if (c.seg + 1 < p.nseg) { ++c.seg; c.posrel += 64; }
        else {
            c.posrel += p.W - 64 * c.seg;
            c.seg = 0;
            if (++c.h == p.H) { c.h = 0; ++c.frame; }
        }
    };
    int b_off0;   // (first position - patch margin) * 32 - b_lo: what the source offsets of the first chunk start from (may be negative: masked lanes)
    {
        const Cur c0 = cur_at_begin();
        const long long pos0 = ((long long)c0.frame * p.H + c0.h) * p.W + c0.seg * 64;
        b_off0 = (int)((pos0 - (long long)p.ph * p.W - p.pw) * 32 - b_lo);
    }
    const int a_step = p.dy_ld * 2;
    auto issue = [&](const Cur& c, int so) {
        unsigned char* stage = lds + so;
        const int h = c.h, w0 = c.seg * 64;
        const unsigned a_base = (unsigned)(c.posrel * a_step);
        const unsigned b_base = (unsigned)(b_off0 + c.posrel * 32);
        const int oh0 = h >> 1, ow0 = w0 >> 1;
        if (SF_DBG(64)) return;  // probe: no DMA instructions at all
        // A
        glds16_buf(rs_a, (w0 + posA < p.W && !SF_DBG(1)) ? a_base + relA : DPC_BUF_OOB, 0u, stage + SF_A + bw * 1024, lane);
        // B
        {
            const bool ok = ((unsigned)(h - p.ph + prowB) < (unsigned)p.H) & ((unsigned)(w0 - p.pw + pcolB) < (unsigned)p.W) & !SF_DBG(1);
            glds16_buf(rs_b, ok ? b_base + relB : DPC_BUF_OOB, 0u, stage + SF_X + bw * 1024, lane);
        }
        // pooled rows: row 1 of the pair is only looked at from odd image rows; positions beyond the 33 a chunk can touch, beyond
        // the pooled row or the pooled image are out-of-range lanes (zero fill, no memory traffic)
        // probe bits 1 / 2: every piece / the pooled pieces out of range (same instruction stream, no memory traffic)
        const unsigned pb0 = (unsigned)(((c.frame * p.Ho + oh0) * p.Wo + ow0) * p.Co), prow = (unsigned)(p.Wo * p.Co);
        const bool r1 = (h & 1) && oh0 + 1 < p.Ho && !SF_DBG(1 | 2), r0 = oh0 < p.Ho && !SF_DBG(1 | 2);
        auto pooled_ok = [&](int r, int oc) { return (r ? r1 : r0) && oc < 33 && ow0 + oc < p.Wo; };
        // C
        glds16_buf(rs_p, pooled_ok(rC, ocC) ? (pb0 + (rC ? prow : 0u)) * 2u + relC : DPC_BUF_OOB, 0u, stage + SF_P + bw * 1024, lane);
        // D
        if (dIsP) glds16_buf(rs_p, pooled_ok(1, ocD) ? (pb0 + prow) * 2u + relD : DPC_BUF_OOB, 0u, stage + SF_P + (8 + bw) * 1024, lane);
        else glds16_buf(rs_m, pooled_ok(rD, ocD) ? pb0 + (rD ? prow : 0u) + relD : DPC_BUF_OOB, 0u, stage + SF_M + (bw - 2) * 1024, lane);
        // E
        if (bw == 0) {
            const bool ok = ((unsigned)(h - p.ph + prowE) < (unsigned)p.H) & ((unsigned)(w0 - p.pw + pcolE) < (unsigned)p.W);
            glds16_buf(rs_b, ok ? b_base + relE : DPC_BUF_OOB, 0u, stage + SF_X + 8 * 1024, lane);
        }
    };
    // own pieces of a chunk have landed when at most those of the `newer` (0, 1, 2) chunks issued after it are outstanding
    const auto wait_landed = [&](int newer) {
        if (newer <= 0) wait_vmcnt<0>();
        else if (newer == 1) { if (bw == 0) wait_vmcnt<5>(); else wait_vmcnt<4>(); }
        else { if (bw == 0) wait_vmcnt<10>(); else wait_vmcnt<8>(); }
    };
    constexpr int SF_RING = SF_NS * SF_STAGE;
    auto next_stage = [](int so) { return so + SF_STAGE == SF_RING ? 0 : so + SF_STAGE; };

    if (wv >= 4) {
        // ------------------------------------------------------------------ builder waves
        // lane makes logical unit fu of position pos = 2 idx + parity of every chunk
        const int parity = bw & 1, fu = lane & 7;
        const int idx = (lane >> 3) + 8 * (bw >> 1);
        const int a_unit = SF_A + (2 * idx + parity) * 128 + (((fu ^ (2 * ((2 * idx + parity) & 3))) & 7) << 4);
        const int pg_off = SF_P + idx * 128 + fu * 16, pm_off = SF_M + idx * 64 + fu * 8;
        float f_mu[8], f_is[8], f_ga[8], f_c1[8], f_c2[8];
        DPC_UNROLL
        for (int e = 0; e < 8; ++e) {
            const int c = tile_m * 64 + fu * 8 + e;
            f_mu[e] = p.mean[c]; f_is[e] = p.invstd[c]; f_ga[e] = p.gamma[c] * f_is[e]; f_c1[e] = p.coef[c]; f_c2[e] = p.coef[p.Co + c];
        }
        // All LDS traffic goes through the raw accessors of dpc_rt.h (hipcc may order an LDS access it can see after every
        // outstanding LDS-DMA piece); the four windows of a unit are read unconditionally into NAMED registers -- an array of asm
        // outputs behind run-time conditions went to scratch, and a scratch load waits like vmcnt(0).
        auto build = [&](const Cur& c, int so) {
            unsigned char* stage = lds + so;
            const int h = c.h;
            const int w = c.seg * 64 + 2 * idx + parity;
            const bool ok = w < p.W;
            u32x4 rv, g00, g01, g10, g11;
            u32x2 m00, m01, m10, m11;
            if (!SF_DBG(32)) {
                lds_read_b128_raw(rv, stage + a_unit);
                lds_read_b128_raw(g00, stage + pg_off);
                lds_read_b128_raw(g01, stage + pg_off + 128);
                lds_read_b128_raw(g10, stage + pg_off + 5120);
                lds_read_b128_raw(g11, stage + pg_off + 5120 + 128);
                lds_read_b64_raw(m00, stage + pm_off);
                lds_read_b64_raw(m01, stage + pm_off + 64);
                lds_read_b64_raw(m10, stage + pm_off + 3072);
                lds_read_b64_raw(m11, stage + pm_off + 3072 + 64);
                lds_wait0_5(rv, g00, g01, g10, g11);
                lds_wait0_4x2(m00, m01, m10, m11);
            }
            if (SF_DBG(4)) return;
            float g[8];
            DPC_UNROLL
            for (int e = 0; e < 8; ++e) g[e] = 0.f;
            // window (a, b) = pooled row (h >> 1) + a, pooled column idx + b; it holds this position at tap (kh, kw) -- constants of the
            // chunk (h) and of the wave (column parity).  Summation order (0,0) (0,1) (1,0) (1,1): pool_routed_grad's.  A position
            // outside the image or a window outside the pooled image matches nothing: its argmax bytes are replaced by 9 ("no
            // gradient") ONCE, so the per-element work is compare + select + add with no exec-mask / scalar-mask arithmetic.
            const u32x2 nine = {0x09090909u, 0x09090909u};
            auto route = [&](const u32x4& gv, const u32x2& am_in, int a, int b) {
                const int kh = a == 0 ? (h & 1) + 1 : 0, kw = b == 0 ? parity + 1 : 0;
                const unsigned want = (unsigned)(kh * 3 + kw);
                const bool use = ok && (w >> 1) + b < p.Wo;
                const u32x2 am = {use ? am_in[0] : nine[0], use ? am_in[1] : nine[1]};
                DPC_UNROLL
                for (int e = 0; e < 8; ++e) {
                    const float v = unit_get<bf16_t>(gv, e);
                    g[e] += (((am[e >> 2] >> (8 * (e & 3))) & 0xffu) == want) ? v : 0.f;
                }
            };
            if ((h >> 1) < p.Ho) {                               // chunk-uniform
                route(g00, m00, 0, 0);
                if (parity) route(g01, m01, 0, 1);               // wave-uniform
            }
            if ((h & 1) && (h >> 1) + 1 < p.Ho) {
                route(g10, m10, 1, 0);
                if (parity) route(g11, m11, 1, 1);
            }
            float ov[8];
            DPC_UNROLL
            for (int e = 0; e < 8; ++e) {
                const float xh = (unit_get<bf16_t>(rv, e) - f_mu[e]) * f_is[e];
                ov[e] = f_ga[e] * (g[e] - f_c1[e] - xh * f_c2[e]);
            }
            u32x4 o = unit_pack<bf16_t>(ov);
            if (!ok) o = u32x4{0u, 0u, 0u, 0u};
            lds_write_b128_raw(stage + a_unit, o);
        };
        Cur ci = cur_at_begin(), cb = ci;   // next chunk to issue / to build
        int so_i = 0, so_b = 0;
        for (int j = 0; j < 3 && j < nck; ++j) { issue(ci, so_i); advance(ci); so_i = next_stage(so_i); }
        if (nck > 0) {
            wait_landed(nck - 1 < 2 ? nck - 1 : 2);
            barrier_lds_only();   // P: chunk 0 landed.  (LDS-only barriers: __syncthreads would wait for the pieces issued two and three chunks ahead)
            build(cb, so_b); advance(cb); so_b = next_stage(so_b);
        }
        for (int j = 0; j < nck; ++j) {
            if (j + 1 < nck) wait_landed(nck - 2 - j < 1 ? nck - 2 - j : 1);   // own pieces of chunk j+1 landed (only j+2 may be newer)
            barrier_lds_only();   // B(j): tile j built and published; stage (j-1) % 4 free; chunk j+1 complete
            if (j + 3 < nck) { issue(ci, so_i); advance(ci); so_i = next_stage(so_i); }
            if (j + 1 < nck) { build(cb, so_b); advance(cb); so_b = next_stage(so_b); }
        }
        return;
    }

    // ---------------------------------------------------------------------- MFMA waves (the plain kernel's wave tile and read pipeline)
    const int wi = wv >> 1, wj = wv & 1;
    const int l31 = lane & 31, lhi = lane >> 5;
    const int gq = lane >> 4, s16 = lane & 15, ph4 = s16 >> 2;
    int fa;
    {
        const int colb = (wi * 32 + (gq & 1) * 16 + 4 * (s16 & 3)) * 2;
        fa = SF_A + ((gq >> 1) * 8 + ph4) * 128 + ((((colb >> 4) ^ (2 * ph4)) & 7) << 4) + (colb & 15);
    }
    const int fb = SF_X + ((gq >> 1) * 8 + ph4 + (gq & 1)) * 32 + 8 * (s16 & 3) + wj * (2 * SF_PW * 32);
    f32x16 acc[4];
    DPC_UNROLL
    for (int t = 0; t < 4; ++t)
        DPC_UNROLL
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
#ifdef DPC_SIMT_EMU
    auto compute = [&](int so) {
        const unsigned char* sp = lds + so;
        DPC_UNROLL
        for (int kk = 0; kk < 4; ++kk) {
            const u32x2 a0 = lds_read_tr16(sp + fa + (kk * 16) * 128);
            const u32x2 a1 = lds_read_tr16(sp + fa + (kk * 16 + 4) * 128);
            const u32x4 av = {a0[0], a0[1], a1[0], a1[1]};
            DPC_UNROLL
            for (int c = 0; c < 4; ++c) {
                const unsigned char* bp = sp + fb + (((c >> 1) * SF_PW + kk * 16 + 2 * (c & 1)) * 32);
                const u32x2 b0 = lds_read_tr16(bp);
                const u32x2 b1 = lds_read_tr16(bp + 4 * 32);
                const u32x4 bv = {b0[0], b0[1], b1[0], b1[1]};
                acc[c] = mfma_32x32x16_bf16(av, bv, acc[c]);
            }
        }
    };
#else
    constexpr int LOOKAHEAD = 4, RB = 8;
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)lds;
    auto compute = [&](int so) {
        if (SF_DBG(8)) return;
        const uint32_t sa = lds0 + so + fa;
        const uint32_t sb = lds0 + so + fb;
        u32x2 alo[2], ahi[2], blo[RB], bhi[RB];
        auto load = [&](auto Ic) {
            constexpr int I = decltype(Ic)::value;
            constexpr int kk = I / 5, r5 = I % 5;
            if constexpr (r5 == 0) {
                asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(alo[kk & 1]) : "v"(sa), "n"((kk * 16) * 128) : "memory");
                __builtin_amdgcn_sched_barrier(0);
                asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(ahi[kk & 1]) : "v"(sa), "n"((kk * 16 + 4) * 128) : "memory");
                __builtin_amdgcn_sched_barrier(0);
            } else {
                constexpr int c = r5 - 1, jx = 4 * kk + c;
                constexpr int off = ((c >> 1) * SF_PW + kk * 16 + 2 * (c & 1)) * 32;
                asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(blo[jx % RB]) : "v"(sb), "n"(off) : "memory");
                __builtin_amdgcn_sched_barrier(0);
                asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(bhi[jx % RB]) : "v"(sb), "n"(off + 4 * 32) : "memory");
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        __builtin_amdgcn_sched_barrier(0);
        static_for<16>([&](auto Mc) {
            constexpr int m = decltype(Mc)::value;
            constexpr int kk = m / 4, c = m % 4, ix = 5 * kk + 1 + c;
            constexpr int ix_prev = ix - 1 + (c == 0 ? -1 : 0);
            constexpr int f_prev = m == 0 ? 0 : (ix_prev + 1 + LOOKAHEAD < 20 ? ix_prev + 1 + LOOKAHEAD : 20);
            constexpr int f_now = ix + 1 + LOOKAHEAD < 20 ? ix + 1 + LOOKAHEAD : 20;
            static_for<f_now - f_prev>([&](auto Dc) { load(std::integral_constant<int, f_prev + decltype(Dc)::value>{}); });
            asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(2 * (f_now - ix - 1)) : "memory");
            __builtin_amdgcn_sched_barrier(0);
            const u32x4 av = {alo[kk & 1][0], alo[kk & 1][1], ahi[kk & 1][0], ahi[kk & 1][1]};
            const u32x4 bv = {blo[m % RB][0], blo[m % RB][1], bhi[m % RB][0], bhi[m % RB][1]};
            acc[c] = mfma_32x32x16_bf16(av, bv, acc[c]);
            __builtin_amdgcn_sched_barrier(0);
        });
    };
#endif
    if (nck > 0) barrier_lds_only();   // P
    int so_c = 0;
    for (int j = 0; j < nck; ++j) {
        barrier_lds_only();   // B(j): tile j is built (the builders waited for every piece of it and transformed it)
        compute(so_c);
        so_c = so_c + SF_STAGE == SF_NS * SF_STAGE ? 0 : so_c + SF_STAGE;
    }

    // partial slab rows = co, columns = tap*16 + ch = (kernel row)*64 + (c&1)*32 + lane column
    DPC_UNROLL
    for (int c = 0; c < 4; ++c) {
        const int col = (2 * wj + (c >> 1)) * 64 + (c & 1) * 32 + l31;
        DPC_UNROLL
        for (int r = 0; r < 16; ++r) {
            const int co = tile_m * 64 + wi * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
            p.part[((long long)ks * p.Co + co) * 256 + col] = acc[c][r];
        }
    }
}

static int stem_plan(const dpc_conv_desc* d, int dy_ld, WgradStemParams* p) {
    static const int on = getenv("DPC_WGRAD_STEM") ? atoi(getenv("DPC_WGRAD_STEM")) : 1;
    if (!on || d->dtype_in != DPC_BF16 || d->mode != 0) return 1;
    if (d->KT != 1 || d->KH != 4 || d->KW != 4 || d->st != 1 || d->sh != 1 || d->sw != 1 || d->pt != 0) return 1;
    if (d->Ci != 16 || d->src_ld != 16 || d->Co % 64 || dy_ld % 8 || dy_ld < d->Co) return 1;
    if (d->RT != d->ST || d->RH != d->SH || d->RW != d->SW || d->RW < 48) return 1;  // narrow images: mostly padding, generic kernel
    const long long M = (long long)d->N * d->RT * d->RH * d->RW;
    const long long sb = M * 32, db = M * dy_ld * 2;
    p->Co = d->Co; p->dy_ld = dy_ld; p->H = d->RH; p->W = d->RW; p->NF = d->N * d->RT;
    p->nseg = (d->RW + 63) / 64; p->ph = d->ph; p->pw = d->pw;
    p->cpf = d->RH * p->nseg;
    p->d_cpf = make_fastdiv((uint32_t)p->cpf); p->d_nseg = make_fastdiv((uint32_t)p->nseg);
    p->ntm = d->Co / 64;
    p->src_bytes = sb; p->dy_bytes = db;
    const int nchunks = p->NF * p->cpf;
    int want = 1024 / p->ntm;
    if (want > nchunks / 16) want = nchunks / 16;
    if (want < 1) want = 1;
    p->kcps = (nchunks + want - 1) / want;
    p->nks = (nchunks + p->kcps - 1) / p->kcps;
    if ((long long)(p->kcps + 2 * p->nseg * 4) * 64 * dy_ld * 2 >= (1ll << 31)) return 1;  // a workgroup's window must stay 32-bit addressable
    return 0;
}

// returns 1 when the shape is not served; with part == NULL only *nsplit is set
int dpc_wgrad_stem_try(const dpc_conv_desc* d, const void* src, const void* dy, int dy_ld, float* part, int32_t* nsplit,
                       hipStream_t stream) {
    WgradStemParams p = {};
    if (stem_plan(d, dy_ld, &p)) return 1;
    if (nsplit) *nsplit = p.nks;
    if (!part) return DPC_OK;
    if (!src || !dy) return DPC_ERR_ARG;
    if (((uintptr_t)src % 16) || ((uintptr_t)dy % 16)) return DPC_ERR_UNSUPPORTED;
    p.src = src; p.dy = dy; p.part = part;
    DPC_LAUNCH(wgrad_stem_kernel, dim3((unsigned)(p.ntm * p.nks)), dim3(256), stream, p);
    return dpc_launch_status();
}

// Weight gradient of the stem convolution straight from the gradient at the POOLED output: BatchNorm backward and the
// max-pool routing (dpc_pool_bn_bwd_apply's arithmetic, bit for bit: dz is rounded to bf16 before the MFMAs in both forms)
// happen inside the kernel, the full-resolution dz tensor is never written.  desc = the stem's weight-gradient descriptor
// (space-to-depth form); raw / mean / invstd / gamma / coef as for dpc_pool_bn_bwd_apply; dpool / argmax at the pooled
// resolution ((H-1)/2+1, (W-1)/2+1).  With part == NULL only *nsplit is set.  DPC_ERR_UNSUPPORTED: use the two-kernel form.
extern "C" int dpc_stem_wgrad_fused(const dpc_conv_desc* d, const void* src_s2d, const void* raw, const void* dpool, const uint8_t* argmax,
                                    const float* mean, const float* invstd, const float* gamma, const float* coef, float* part,
                                    int32_t* nsplit, dpc_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!d) return DPC_ERR_ARG;
    WgradStemParams p = {};
    if (d->dtype_out != DPC_F32 || stem_plan(d, d->Co, &p)) return DPC_ERR_UNSUPPORTED;
    const long long pooled = (long long)p.NF * ((p.H - 1) / 2 + 1) * ((p.W - 1) / 2 + 1) * d->Co;
    if (pooled * 2 >= (1ll << 31)) return DPC_ERR_UNSUPPORTED;  // 32-bit element offsets into the pooled tensors
    if (nsplit) *nsplit = p.nks;
    if (!part) return DPC_OK;
    if (!src_s2d || !raw || !dpool || !argmax || !mean || !invstd || !gamma || !coef) return DPC_ERR_ARG;
    if (((uintptr_t)src_s2d % 16) || ((uintptr_t)raw % 16) || ((uintptr_t)dpool % 16) || ((uintptr_t)argmax % 8)) return DPC_ERR_UNSUPPORTED;
    p.src = src_s2d; p.dy = raw; p.part = part;  // p.dy only sizes the (unused) dy window
    p.raw = raw; p.dpool = dpool; p.argmax = argmax; p.mean = mean; p.invstd = invstd; p.gamma = gamma; p.coef = coef;
    p.Ho = (p.H - 1) / 2 + 1; p.Wo = (p.W - 1) / 2 + 1;
#ifdef DPC_WS_PROBE
    p.dbg = getenv("DPC_SF_DBG") ? atoi(getenv("DPC_SF_DBG")) : 0;
#endif
    DPC_LAUNCH(wgrad_stem_fused_kernel, dim3((unsigned)(p.ntm * p.nks)), dim3(768), stream, p);
    return dpc_launch_status();
}
