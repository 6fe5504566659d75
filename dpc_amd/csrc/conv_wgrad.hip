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

struct WgradParams {
    GatherGeom g;
    const void* src;
    const void* dy;
    float* part;
    int Co, dy_ld;
    int nks, kcps;  // K-splits, position-chunks per split
    int ntm, ntn;
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

template <class T, bool RF>
static int launch_wgrad_rf(const WgradParams& p, int tm, int tn, hipStream_t stream) {
    dim3 grid((unsigned)(p.ntm * p.ntn * p.nks)), block(256);
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
    if (!d || d->mode != 0) return DPC_ERR_ARG;
    int rc = make_gather_geom(d, &p.g);
    if (rc) return rc;
    const int per16 = d->dtype_in == DPC_BF16 ? 8 : 4;
    if (dy_ld % per16 || dy_ld < d->Co) return DPC_ERR_UNSUPPORTED;
    const int tm = d->Co > 64 ? 128 : 64;
    const int tn = (p.g.Kp >= 128 || tm == 128) ? 128 : 64;
    p.ntm = (d->Co + tm - 1) / tm;
    p.ntn = (p.g.Kp + tn - 1) / tn;
    const int bkp = 8 * per16;
    const int nchunks = (p.g.M + bkp - 1) / bkp;
    int want = 1536 / (p.ntm * p.ntn);
    if (want < 1) want = 1;
    if (want > nchunks) want = nchunks;
    p.kcps = (nchunks + want - 1) / want;
    p.nks = (nchunks + p.kcps - 1) / p.kcps;
    if (nsplit) *nsplit = p.nks;
    if (!part) return DPC_OK;  // size query
    if (!src || !dy) return DPC_ERR_ARG;
    p.src = src; p.dy = dy; p.part = part; p.Co = d->Co; p.dy_ld = dy_ld;
    if (d->dtype_in == DPC_F32) return launch_wgrad<float>(p, tm, tn, stream);
    return launch_wgrad<bf16_t>(p, tm, tn, stream);
}
