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
//   ds_read_b128 fragment reads of 16 rows x same unit hit all 64 banks once;
//   f32 uses v_mfma_f32_32x32x2_f32 (exact f32, parity mode), bf16 v_mfma_f32_32x32x16_bf16.
//   K order inside a chunk is permuted identically for A and B (a lane's 16-byte unit
//   feeds its own k-group), which leaves the dot products unchanged.
// A workgroup walks m-tiles m_prog, m_prog+gm, ... so the per-channel batch-norm partial
// sums (sum, sum^2 of the *stored* values) accumulate in registers and leave as ONE row
// of `stats` per program: deterministic, no atomics.
#include "conv_common.h"

struct IGemmParams {
    GatherGeom g;
    const void* src;
    const void* wgt;
    void* out;
    const void* addend;
    float* stats;
    int Ncol, ldw, ldo;
    int gm, ntn, ntm;
};

// FAST: every 128-byte K chunk lies inside one tap (Ci >= chunk, or a plain GEMM) and the gather is
// affine in the tap (forward conv, or an input-gradient with unit strides).  Tap decode and the tap's
// address delta are then wave-uniform (SALU) and a row costs 3 adds + 3 compares per chunk instead of
// a full coordinate->offset recomputation: the kernel is issue-bound, not MFMA-bound, without this.
template <class T, class TO, int BN, bool FAST>
__global__ __launch_bounds__(256, (BN == 64 ? 3 : 2)) void igemm_kernel(IGemmParams p) {
    constexpr int EPU = Elt<T>::PER16;
    constexpr int BKE = 8 * EPU;
    constexpr int BM = 128;
    constexpr int NT = BN / 64;   // 32-wide n-tiles per wave
    constexpr int BROWS = BN / 32;  // B-tile rows per thread
    __shared__ __attribute__((aligned(16))) unsigned char lds[2 * (BM + BN) * 128];

    const GatherGeom& g = p.g;
    const int tid = threadIdx.x;
    const int u = tid & 7, r0 = tid >> 3;
    const int lane = tid & 63, wv = tid >> 6;
    const int wm = wv >> 1, wn = wv & 1;
    const int l31 = lane & 31, lhi = lane >> 5;
    const int n_tile = blockIdx.x % p.ntn;
    const int m_prog = blockIdx.x / p.ntn;
    const int nkc = (g.Kp + BKE - 1) / BKE;
    const int esz = (int)sizeof(T);

    float s1[NT], s2[NT];
    DPC_UNROLL
    for (int j = 0; j < NT; ++j) { s1[j] = 0.f; s2[j] = 0.f; }

    for (int mt = m_prog; mt < p.ntm; mt += p.gm) {
        const int m0 = mt * BM;
        RowPos rp[4];
        int rowbase[4];
        DPC_UNROLL
        for (int i = 0; i < 4; ++i) {
            rp[i] = decode_row(g, m0 + r0 + 32 * i);
            rowbase[i] = (int)(((((unsigned)(rp[i].nbase + rp[i].t0) * (unsigned)g.SH + (unsigned)rp[i].h0) * (unsigned)g.SW) +
                                (unsigned)rp[i].w0) * (unsigned)g.src_ld);
        }

        f32x16 acc[2][NT];
        DPC_UNROLL
        for (int i = 0; i < 2; ++i)
            DPC_UNROLL
            for (int j = 0; j < NT; ++j)
                DPC_UNROLL
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

        u32x4 ra[4], rb[BROWS];
        unsigned okbits = 0;  // validity of the units in flight, applied when they are stored to LDS
        auto load_chunk = [&](int kc) {
            const int k = kc * BKE + u * EPU;
            okbits = 0;
            TapPos tp;
            if (FAST) {
                const int k0 = kc * BKE;  // wave-uniform
                const int tap = (g.taps == 1) ? 0 : (k0 >> g.log2C);
                const unsigned q = fdiv((unsigned)tap, g.dKW);
                const int kw = tap - (int)q * g.KW;
                const unsigned kt = fdiv(q, g.dKH);
                const int kh = (int)q - (int)kt * g.KH;
                const int sgn = g.mode == 0 ? 1 : -1;
                const int dt = sgn * (int)kt, dh = sgn * kh, dw = sgn * kw;
                const int cbase = (g.taps == 1) ? 0 : (tap << g.log2C);
                const int tapoff = ((dt * g.SH + dh) * g.SW + dw) * g.src_ld + (k - cbase);
                tp.ok = k < g.Kp;
                DPC_UNROLL
                for (int i = 0; i < 4; ++i) {
                    const bool ok = tp.ok && (unsigned)(rp[i].t0 + dt) < (unsigned)g.ST && (unsigned)(rp[i].h0 + dh) < (unsigned)g.SH &&
                                    (unsigned)(rp[i].w0 + dw) < (unsigned)g.SW;
                    okbits |= (ok ? 1u : 0u) << i;
                    ra[i] = load_unit_raw(p.src, ok ? (long long)(rowbase[i] + tapoff) : -1, esz);
                }
            } else {
                tp = decode_k(g, k);
                DPC_UNROLL
                for (int i = 0; i < 4; ++i) {
                    const long long off = gather_off(g, rp[i], tp);
                    okbits |= (off >= 0 ? 1u : 0u) << i;
                    ra[i] = load_unit_raw(p.src, off, esz);
                }
            }
            DPC_UNROLL
            for (int i = 0; i < BROWS; ++i) {
                const int n = n_tile * BN + r0 + 32 * i;
                const bool ok = tp.ok && n < p.Ncol;
                okbits |= (ok ? 1u : 0u) << (4 + i);
                rb[i] = load_unit_raw(p.wgt, ok ? ((long long)n * p.ldw + k) : -1, esz);
            }
        };
        auto store_chunk = [&](int buf) {
            unsigned char* As = lds + buf * (BM + BN) * 128;
            unsigned char* Bs = As + BM * 128;
            DPC_UNROLL
            for (int i = 0; i < 4; ++i) *(u32x4*)(As + lds_unit_off(r0 + 32 * i, u)) = mask_unit(ra[i], (okbits >> i) & 1u);
            DPC_UNROLL
            for (int i = 0; i < BROWS; ++i) *(u32x4*)(Bs + lds_unit_off(r0 + 32 * i, u)) = mask_unit(rb[i], (okbits >> (4 + i)) & 1u);
        };

        load_chunk(0);
        store_chunk(0);
        __syncthreads();
        for (int kc = 0; kc < nkc; ++kc) {
            const int buf = kc & 1;
            if (kc + 1 < nkc) load_chunk(kc + 1);
            const unsigned char* As = lds + buf * (BM + BN) * 128;
            const unsigned char* Bs = As + BM * 128;
            DPC_UNROLL
            for (int kk = 0; kk < 4; ++kk) {
                const int unit = 2 * kk + lhi;
                u32x4 fa[2], fb[NT];
                DPC_UNROLL
                for (int i = 0; i < 2; ++i) fa[i] = *(const u32x4*)(As + lds_unit_off(wm * 64 + i * 32 + l31, unit));
                DPC_UNROLL
                for (int j = 0; j < NT; ++j) fb[j] = *(const u32x4*)(Bs + lds_unit_off(wn * (BN / 2) + j * 32 + l31, unit));
                DPC_UNROLL
                for (int i = 0; i < 2; ++i)
                    DPC_UNROLL
                    for (int j = 0; j < NT; ++j) acc[i][j] = mfma_unit<T>(fa[i], fb[j], acc[i][j]);
            }
            if (kc + 1 < nkc) store_chunk(buf ^ 1);
            __syncthreads();
        }

        // epilogue: C/D map col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
        if (p.addend) {  // all addend loads first (clamped, unconditional), one wait, then the adds
            DPC_UNROLL
            for (int j = 0; j < NT; ++j) {
                const int col = n_tile * BN + wn * (BN / 2) + j * 32 + l31;
                DPC_UNROLL
                for (int i = 0; i < 2; ++i) {
                    TO ad[16];
                    DPC_UNROLL
                    for (int r = 0; r < 16; ++r) {
                        const int row = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
                        const bool ok = row < g.M && col < p.Ncol;
                        ad[r] = ((const TO*)p.addend)[ok ? ((long long)row * p.ldo + col) : 0];
                    }
                    DPC_UNROLL
                    for (int r = 0; r < 16; ++r) acc[i][j][r] += Elt<TO>::to_f32(ad[r]);
                }
            }
        }
        DPC_UNROLL
        for (int j = 0; j < NT; ++j) {
            const int col = n_tile * BN + wn * (BN / 2) + j * 32 + l31;
            if (col < p.Ncol) {
                DPC_UNROLL
                for (int i = 0; i < 2; ++i) {
                    DPC_UNROLL
                    for (int r = 0; r < 16; ++r) {
                        const int row = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
                        if (row < g.M) {
                            const long long o = (long long)row * p.ldo + col;
                            float v = acc[i][j][r];
                            const TO q = Elt<TO>::from_f32(v);
                            ((TO*)p.out)[o] = q;
                            const float vq = Elt<TO>::to_f32(q);
                            s1[j] += vq;
                            s2[j] += vq * vq;
                        }
                    }
                }
            }
        }
    }

    if (p.stats) {
        float* red = (float*)lds;  // [wave][2][NT][32]
        DPC_UNROLL
        for (int j = 0; j < NT; ++j) {
            s1[j] += __shfl_xor(s1[j], 32);
            s2[j] += __shfl_xor(s2[j], 32);
        }
        __syncthreads();
        if (lhi == 0) {
            DPC_UNROLL
            for (int j = 0; j < NT; ++j) {
                red[((wv * 2 + 0) * NT + j) * 32 + l31] = s1[j];
                red[((wv * 2 + 1) * NT + j) * 32 + l31] = s2[j];
            }
        }
        __syncthreads();
        if (tid < BN) {
            const int wn2 = tid / (BN / 2), j2 = (tid % (BN / 2)) / 32, l2 = tid % 32;
            const int col = n_tile * BN + tid;
            if (col < p.Ncol) {
                float a = 0.f, b = 0.f;
                DPC_UNROLL
                for (int w2 = 0; w2 < 2; ++w2) {
                    const int wave = w2 * 2 + wn2;
                    a += red[((wave * 2 + 0) * NT + j2) * 32 + l2];
                    b += red[((wave * 2 + 1) * NT + j2) * 32 + l2];
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
    int ntm, ntn, gm, bn;
    igemm_grid(d, g.M, &ntm, &ntn, &gm, &bn);
    return gm;
}

template <class T, class TO>
static int launch_igemm(const IGemmParams& p, int bn, hipStream_t stream) {
    dim3 grid((unsigned)(p.gm * p.ntn)), block(256);
    const GatherGeom& g = p.g;
    const int bke = 8 * Elt<T>::PER16;
    const bool unit_strides = g.st == 1 && g.sh == 1 && g.sw == 1;
    const bool fits32 = (long long)(g.M / (g.RT * g.RH * g.RW)) * g.ST * g.SH * g.SW * g.src_ld < (1ll << 31);
    const bool fast = fits32 && (g.taps == 1 || g.Ci >= bke) && (g.mode == 0 || unit_strides);
    if (bn == 64) {
        if (fast) {
            DPC_LAUNCH((igemm_kernel<T, TO, 64, true>), grid, block, stream, p);
        } else {
            DPC_LAUNCH((igemm_kernel<T, TO, 64, false>), grid, block, stream, p);
        }
    } else {
        if (fast) {
            DPC_LAUNCH((igemm_kernel<T, TO, 128, true>), grid, block, stream, p);
        } else {
            DPC_LAUNCH((igemm_kernel<T, TO, 128, false>), grid, block, stream, p);
        }
    }
    return dpc_launch_status();
}

extern "C" int dpc_conv_igemm(const dpc_conv_desc* d, const void* src, const void* wgt, void* out,
                              const void* addend, float* stats, dpc_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    IGemmParams p;
    int rc = make_gather_geom(d, &p.g);
    if (rc) return rc;
    if (!src || !wgt || !out) return DPC_ERR_ARG;
    const int per16 = d->dtype_in == DPC_BF16 ? 8 : 4;
    if (d->ldw % per16) return DPC_ERR_UNSUPPORTED;
    if (d->ldo < d->Co || d->ldw < p.g.Kp) return DPC_ERR_ARG;
    p.src = src; p.wgt = wgt; p.out = out; p.addend = addend; p.stats = stats;
    p.Ncol = d->Co; p.ldw = d->ldw; p.ldo = d->ldo;
    int bn;
    igemm_grid(d, p.g.M, &p.ntm, &p.ntn, &p.gm, &bn);
    if (d->dtype_in == DPC_F32 && d->dtype_out == DPC_F32) return launch_igemm<float, float>(p, bn, stream);
    if (d->dtype_in == DPC_BF16 && d->dtype_out == DPC_BF16) return launch_igemm<bf16_t, bf16_t>(p, bn, stream);
    if (d->dtype_in == DPC_BF16 && d->dtype_out == DPC_F32) return launch_igemm<bf16_t, float>(p, bn, stream);
    return DPC_ERR_UNSUPPORTED;
}

extern "C" int dpc_abi_version(void) { return 1; }
