// pack.hip -- layout/precision repacking around the matrix-core kernels.  All HBM-bound,
// all tiny next to the conv stack except dpc_pack_input_s2d (one pass over the video).
#include "dpc_rt.h"
#include "../../include/dpc_hip.h"

template <class TO>
__global__ void pack3d_kernel(const float* in, TO* out, int d0, int d1, int d2, long long s0, long long s1, long long s2) {
    const long long n = (long long)d0 * d1 * d2;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const int i2 = (int)(i % d2);
        const long long q = i / d2;
        const int i1 = (int)(q % d1);
        const int i0 = (int)(q / d1);
        out[i] = Elt<TO>::from_f32(in[i0 * s0 + i1 * s1 + i2 * s2]);
    }
}

static inline unsigned grid_for(long long n, int block = 256, int cap = 4096) {
    long long g = (n + block - 1) / block;
    if (g < 1) g = 1;
    if (g > cap) g = cap;
    return (unsigned)g;
}

extern "C" int dpc_pack3d(const float* in, void* out, int32_t dtype_out, int32_t d0, int32_t d1, int32_t d2,
                          int64_t s0, int64_t s1, int64_t s2, dpc_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!in || !out || d0 <= 0 || d1 <= 0 || d2 <= 0) return DPC_ERR_ARG;
    const long long n = (long long)d0 * d1 * d2;
    if (dtype_out == DPC_F32) {
        DPC_LAUNCH((pack3d_kernel<float>), dim3(grid_for(n)), dim3(256), stream, in, (float*)out, d0, d1, d2, (long long)s0, (long long)s1, (long long)s2);
    } else if (dtype_out == DPC_BF16) {
        DPC_LAUNCH((pack3d_kernel<bf16_t>), dim3(grid_for(n)), dim3(256), stream, in, (bf16_t*)out, d0, d1, d2, (long long)s0, (long long)s1, (long long)s2);
    } else {
        return DPC_ERR_ARG;
    }
    return dpc_launch_status();
}

// ---- all per-step weight repacks of the conv stack in ONE launch.  The engine repacked 38 tensors per step with 38 pack3d
// launches of ~7 us each (0.26 ms of a 30 ms step); the table of (source, destination, extents, strides) is static, so it lives
// in device memory and a block finds its entry by its block range.
template <class TO>
__global__ void pack3d_multi_kernel(const dpc_pack_entry* tab, int n_entries, TO* /*type tag*/) {
    int e = 0;
    while (e + 1 < n_entries && (int)blockIdx.x >= tab[e + 1].block0) ++e;  // <= 80 entries: a scan per thread is cheaper than a barrier
    const dpc_pack_entry t = tab[e];
    const long long n = (long long)t.d0 * t.d1 * t.d2;
    const int nblk = (e + 1 < n_entries ? tab[e + 1].block0 : (int)gridDim.x) - t.block0;
    const float* in = (const float*)t.in;
    TO* out = (TO*)t.out;
    // Conv weights: [Co][Ci][taps] -> [Co][tap][Ci] (forward) and [Ci][tap][Co] (input-gradient).  In both the source is
    // contiguous along the MIDDLE output index (s1 == 1), so an output-linear walk reads with a stride of `taps` floats: every lane
    // its own 128-byte line, each line fetched ~27 times -- 847 MB of HBM reads per step for 44 MB of weights (rocprofv3 PMC,
    // 140 us).  Tiled: (all of d1) x (64 of d2) for one i0 goes through LDS -- runs of d1 contiguous floats in, 64 contiguous
    // elements out.
    constexpr int TJ = 64, TI_MAX = 64;
    __shared__ float tile[TJ][TI_MAX + 1];
    if (t.s1 == 1 && t.d1 <= TI_MAX && t.d1 > 1) {   // workgroup-uniform
        const int nj = (t.d2 + TJ - 1) / TJ;
        const int ntiles = t.d0 * nj;
        for (int tl = (int)blockIdx.x - t.block0; tl < ntiles; tl += nblk) {
            const int i0 = tl / nj, j0 = (tl - i0 * nj) * TJ;
            const int jn = t.d2 - j0 < TJ ? t.d2 - j0 : TJ;
            const float* src = in + (long long)i0 * t.s0 + (long long)j0 * t.s2;
            for (int idx = threadIdx.x; idx < jn * t.d1; idx += blockDim.x) {
                const int j = idx / t.d1, i1 = idx - j * t.d1;
                tile[j][i1] = src[(long long)j * t.s2 + i1];
            }
            __syncthreads();
            TO* dst = out + ((long long)i0 * t.d1) * t.d2 + j0;
            for (int idx = threadIdx.x; idx < t.d1 * TJ; idx += blockDim.x) {
                const int i1 = idx / TJ, j = idx - i1 * TJ;
                if (j < jn) dst[(long long)i1 * t.d2 + j] = Elt<TO>::from_f32(tile[j][i1]);
            }
            __syncthreads();
        }
        return;
    }
    for (long long i = (long long)((int)blockIdx.x - t.block0) * blockDim.x + threadIdx.x; i < n; i += (long long)nblk * blockDim.x) {
        const int i2 = (int)(i % t.d2);
        const long long q = i / t.d2;
        const int i1 = (int)(q % t.d1);
        const int i0 = (int)(q / t.d1);
        out[i] = Elt<TO>::from_f32(in[i0 * t.s0 + i1 * t.s1 + i2 * t.s2]);
    }
}

extern "C" int dpc_pack3d_multi(const dpc_pack_entry* table_dev, int32_t n_entries, int32_t total_blocks, int32_t dtype_out,
                                dpc_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!table_dev || n_entries <= 0 || total_blocks < n_entries) return DPC_ERR_ARG;
    if (dtype_out == DPC_F32) {
        DPC_LAUNCH((pack3d_multi_kernel<float>), dim3((unsigned)total_blocks), dim3(256), stream, table_dev, n_entries, (float*)nullptr);
    } else if (dtype_out == DPC_BF16) {
        DPC_LAUNCH((pack3d_multi_kernel<bf16_t>), dim3((unsigned)total_blocks), dim3(256), stream, table_dev, n_entries, (bf16_t*)nullptr);
    } else {
        return DPC_ERR_ARG;
    }
    return dpc_launch_status();
}

// 256 threads = 32 consecutive outputs x 8 split lanes: the split-K slabs are summed 8-wide in
// parallel (fixed order => deterministic), then one LDS hop.  A serial loop over up to ~400 slabs
// per output was latency-bound (0.3 ms for the 9.4k-element stem gradient).
__global__ __launch_bounds__(256) void reduce_unpack_kernel(const float* part, int nsplit, float* out, int d0, int d1, int d2, long long s0,
                                     long long s1, long long s2, int accumulate) {
    __shared__ float red[8][32];
    const long long n = (long long)d0 * d1 * d2;
    const int ol = threadIdx.x & 31, sl = threadIdx.x >> 5;
    const long long i = (long long)blockIdx.x * 32 + ol;
    float s = 0.f;
    if (i < n)
        for (int k = sl; k < nsplit; k += 8) s += part[(long long)k * n + i];
    red[sl][ol] = s;
    __syncthreads();
    if (sl == 0 && i < n) {
        float t = 0.f;
        DPC_UNROLL
        for (int k = 0; k < 8; ++k) t += red[k][ol];
        const int i2 = (int)(i % d2);
        const long long q = i / d2;
        const int i1 = (int)(q % d1);
        const int i0 = (int)(q / d1);
        float* o = out + i0 * s0 + i1 * s1 + i2 * s2;
        *o = accumulate ? (*o + t) : t;
    }
}

// Same reduction with 16-byte loads: a thread owns FOUR consecutive outputs (d2 % 4 == 0, so they share i0, i1), a workgroup
// 128 outputs x 8 split lanes; each split lane walks its slabs eight at a time.  The 4-byte form reads 128 bytes per 32 lanes and
// slab: 34 us for the 75 MB of layer1's 512 slabs (2.2 TB/s).  Summation order is fixed by (lane, slab index): deterministic.
__global__ __launch_bounds__(256) void reduce_unpack4_kernel(const float* part, int nsplit, float* out, int d0, int d1, int d2, long long s0,
                                      long long s1, long long s2, int accumulate) {
    __shared__ float red[8][32][4];
    const long long n = (long long)d0 * d1 * d2;
    const int ol = threadIdx.x & 31, sl = threadIdx.x >> 5;
    const long long i = ((long long)blockIdx.x * 32 + ol) * 4;
    // EIGHT slabs per split lane and trip, loaded before any is added: with two (rounds 3-5) the walk was a chain of HBM round trips --
    // 62 us for the 512 slabs x 32 KB of layer2.0's downsample gradient (64 workgroups), on the side stream's critical path there
    float acc[8][4];
    DPC_UNROLL
    for (int j = 0; j < 8; ++j)
        DPC_UNROLL
        for (int e = 0; e < 4; ++e) acc[j][e] = 0.f;
    if (i < n) {
        int k = sl;
        for (; k + 56 < nsplit; k += 64) {
            u32x4 v[8];
            DPC_UNROLL
            for (int j = 0; j < 8; ++j) v[j] = *(const u32x4*)(part + (long long)(k + 8 * j) * n + i);
            DPC_UNROLL
            for (int j = 0; j < 8; ++j)
                DPC_UNROLL
                for (int e = 0; e < 4; ++e) acc[j][e] += unit_get<float>(v[j], e);
        }
        DPC_UNROLL
        for (int j = 0; j < 8; ++j)
            if (k + 8 * j < nsplit) {
                const u32x4 v0 = *(const u32x4*)(part + (long long)(k + 8 * j) * n + i);
                DPC_UNROLL
                for (int e = 0; e < 4; ++e) acc[j][e] += unit_get<float>(v0, e);
            }
    }
    float a[4], b[4];
    DPC_UNROLL
    for (int e = 0; e < 4; ++e) {
        a[e] = (acc[0][e] + acc[1][e]) + (acc[2][e] + acc[3][e]);
        b[e] = (acc[4][e] + acc[5][e]) + (acc[6][e] + acc[7][e]);
    }
    DPC_UNROLL
    for (int e = 0; e < 4; ++e) red[sl][ol][e] = a[e] + b[e];
    __syncthreads();
    if (sl == 0 && i < n) {
        const int i2 = (int)(i % d2);
        const long long q = i / d2;
        const int i1 = (int)(q % d1);
        const int i0 = (int)(q / d1);
        float* o = out + i0 * s0 + i1 * s1 + i2 * s2;
        DPC_UNROLL
        for (int e = 0; e < 4; ++e) {
            float t = 0.f;
            DPC_UNROLL
            for (int k = 0; k < 8; ++k) t += red[k][ol][e];
            o[e * s2] = accumulate ? (o[e * s2] + t) : t;
        }
    }
}

// Few slabs (<= 8: the score's backward products, 5 slabs of 6 MB at cfg2; the small weight gradients of layer3 / layer4): one thread
// owns four consecutive outputs and reads ALL its slabs before adding any -- no split lanes (with nsplit = 5 three of the eight idle),
// no LDS hop.  Sums left to right, which is what the split-lane form computes for nsplit <= 8 (lane k holds slab k alone, the tree
// adds lanes 0..7 in order): bit-identical results.
template <int NS>
__global__ __launch_bounds__(256) void reduce_unpack_few_kernel(const float* part, float* out, int d0, int d1, int d2, long long s0, long long s1,
                                                                long long s2, int accumulate) {
    const long long n = (long long)d0 * d1 * d2;
    const long long i = ((long long)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i >= n) return;
    u32x4 v[NS];
    DPC_UNROLL
    for (int k = 0; k < NS; ++k) v[k] = *(const u32x4*)(part + (long long)k * n + i);
    const int i2 = (int)(i % d2);
    const long long q = i / d2;
    const int i1 = (int)(q % d1);
    const int i0 = (int)(q / d1);
    float* o = out + i0 * s0 + i1 * s1 + i2 * s2;
    DPC_UNROLL
    for (int e = 0; e < 4; ++e) {
        float t = 0.f;
        DPC_UNROLL
        for (int k = 0; k < NS; ++k) t += unit_get<float>(v[k], e);
        o[e * s2] = accumulate ? (o[e * s2] + t) : t;
    }
}

// The conv weight-gradient case: slabs [co][tap][ci] -> parameter layout [co][ci][tap] (s0 = d1*d2, s1 = 1, s2 = d1).  The
// generic kernel above writes it with 4-byte stores `taps` floats apart and reads 128 bytes per wave and slab (34 us for a
// 256 x 256 x 27 gradient, 25 launches per step); here a workgroup owns (co, 64 input channels): the slab sums are read as
// 256-byte rows, transposed through LDS (row stride taps + (taps even): odd => conflict-free) and written as ONE contiguous
// run of 64 * taps floats.  Summation order over the slabs is fixed (deterministic).
constexpr int RUT_CW = 64, RUT_MAXT = 32;
__global__ __launch_bounds__(256) void reduce_unpack_t_kernel(const float* part, int nsplit, float* out, int Co, int taps, int Ci, int accumulate) {
    __shared__ float tile[RUT_CW * (RUT_MAXT + 1)];
    const int co = blockIdx.x, c0 = blockIdx.y * RUT_CW;
    const int cw = Ci - c0 < RUT_CW ? Ci - c0 : RUT_CW;
    const int ldt = taps | 1;  // odd row stride
    const long long n = (long long)Co * taps * Ci;
    const float* base = part + ((long long)co * taps) * Ci + c0;
    for (int e = threadIdx.x; e < taps * RUT_CW; e += 256) {
        const int tap = e / RUT_CW, c = e - tap * RUT_CW;
        if (c < cw) {
            const float* p = base + (long long)tap * Ci + c;
            float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
            int k = 0;
            for (; k + 4 <= nsplit; k += 4) {
                a0 += p[(long long)k * n];
                a1 += p[(long long)(k + 1) * n];
                a2 += p[(long long)(k + 2) * n];
                a3 += p[(long long)(k + 3) * n];
            }
            for (; k < nsplit; ++k) a0 += p[(long long)k * n];
            tile[c * ldt + tap] = (a0 + a1) + (a2 + a3);
        }
    }
    __syncthreads();
    float* o = out + ((long long)co * Ci + c0) * taps;
    for (int j = threadIdx.x; j < cw * taps; j += 256) {
        const int c = j / taps, tap = j - c * taps;
        const float v = tile[c * ldt + tap];
        o[j] = accumulate ? o[j] + v : v;
    }
}

extern "C" int dpc_reduce_unpack(const float* part, int32_t nsplit, float* out, int32_t d0, int32_t d1, int32_t d2,
                                 int64_t s0, int64_t s1, int64_t s2, int32_t accumulate, dpc_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!part || !out || nsplit <= 0 || d0 <= 0 || d1 <= 0 || d2 <= 0) return DPC_ERR_ARG;
    // few slabs only: with hundreds of slabs (layer1: 512 x 147 KB) the 8 split lanes of the generic kernel win (measured)
    if (nsplit <= 16 && d1 > 1 && d1 <= RUT_MAXT && s1 == 1 && s2 == d1 && s0 == (int64_t)d1 * d2 && d0 <= 65535) {
        DPC_LAUNCH(reduce_unpack_t_kernel, dim3((unsigned)d0, (unsigned)((d2 + RUT_CW - 1) / RUT_CW)), dim3(256), stream, part, nsplit, out, d0, d1, d2,
                   accumulate);
        return dpc_launch_status();
    }
    const long long n = (long long)d0 * d1 * d2;
    if (d2 % 4 == 0 && ((uintptr_t)part % 16) == 0 && nsplit <= 8) {   // n % 4 == 0 follows: every slab starts 16-byte aligned
        const dim3 grid((unsigned)((n / 4 + 255) / 256)), block(256);
#define DPC_RUF(NS) case NS: DPC_LAUNCH((reduce_unpack_few_kernel<NS>), grid, block, stream, part, out, d0, d1, d2, (long long)s0, (long long)s1, (long long)s2, accumulate); break
        switch (nsplit) { DPC_RUF(1); DPC_RUF(2); DPC_RUF(3); DPC_RUF(4); DPC_RUF(5); DPC_RUF(6); DPC_RUF(7); DPC_RUF(8); }
#undef DPC_RUF
        return dpc_launch_status();
    }
    if (d2 % 4 == 0 && ((uintptr_t)part % 16) == 0) {
        DPC_LAUNCH(reduce_unpack4_kernel, dim3((unsigned)((n / 4 + 31) / 32)), dim3(256), stream, part, nsplit, out, d0, d1, d2, (long long)s0, (long long)s1,
                   (long long)s2, accumulate);
        return dpc_launch_status();
    }
    DPC_LAUNCH(reduce_unpack_kernel, dim3((unsigned)((n + 31) / 32)), dim3(256), stream, part, nsplit, out, d0, d1, d2, (long long)s0, (long long)s1, (long long)s2, accumulate);
    return dpc_launch_status();
}

template <class TI, class TO>
__global__ void transpose2d_kernel(const TI* in, int ld_in, TO* out, int ld_out, int rows, int cols) {
    __shared__ float tile[32][33];
    const int bx = blockIdx.x * 32, by = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 256 threads: 8 row groups
    for (int r = ty; r < 32; r += 8) {
        const int row = by + r, col = bx + tx;
        tile[r][tx] = (row < rows && col < cols) ? Elt<TI>::to_f32(in[(long long)row * ld_in + col]) : 0.f;
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        const int orow = bx + r, ocol = by + tx;  // out[col][row]
        if (orow < cols && ocol < rows) out[(long long)orow * ld_out + ocol] = Elt<TO>::from_f32(tile[tx][r]);
    }
}

extern "C" int dpc_transpose2d(const void* in, int32_t dtype_in, int32_t ld_in, void* out, int32_t dtype_out,
                               int32_t ld_out, int32_t rows, int32_t cols, dpc_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!in || !out || rows <= 0 || cols <= 0) return DPC_ERR_ARG;
    dim3 grid((cols + 31) / 32, (rows + 31) / 32), block(256);
    if (dtype_in == DPC_F32 && dtype_out == DPC_F32) {
        DPC_LAUNCH((transpose2d_kernel<float, float>), grid, block, stream, (const float*)in, ld_in, (float*)out, ld_out, rows, cols);
    } else if (dtype_in == DPC_F32 && dtype_out == DPC_BF16) {
        DPC_LAUNCH((transpose2d_kernel<float, bf16_t>), grid, block, stream, (const float*)in, ld_in, (bf16_t*)out, ld_out, rows, cols);
    } else if (dtype_in == DPC_BF16 && dtype_out == DPC_BF16) {
        DPC_LAUNCH((transpose2d_kernel<bf16_t, bf16_t>), grid, block, stream, (const bf16_t*)in, ld_in, (bf16_t*)out, ld_out, rows, cols);
    } else if (dtype_in == DPC_BF16 && dtype_out == DPC_F32) {
        DPC_LAUNCH((transpose2d_kernel<bf16_t, float>), grid, block, stream, (const bf16_t*)in, ld_in, (float*)out, ld_out, rows, cols);
    } else {
        return DPC_ERR_ARG;
    }
    return dpc_launch_status();
}

// two bf16 matrices of one shape transposed by ONE launch (blockIdx.z picks the pair): pred and feature_inf -> the K-contiguous
// operands of the score's backward products (dpc/model_3d.py:83's autograd); 16-byte loads, a 64 x 64 tile through LDS, 16-byte stores
__global__ __launch_bounds__(256) void transpose2d_bf16x2_kernel(const bf16_t* in0, const bf16_t* in1, int ld_in, bf16_t* out0, bf16_t* out1, int ld_out,
                                                                int rows, int cols) {
    __shared__ unsigned short tile[64][66];
    const bf16_t* in = blockIdx.z ? in1 : in0;
    bf16_t* out = blockIdx.z ? out1 : out0;
    const int bx = blockIdx.x * 64, by = blockIdx.y * 64;   // column / row origin of the input tile
    const int t = threadIdx.x;
    {   // 64 rows x 8 units of 8 columns: two rows of units per thread pass
        DPC_UNROLL
        for (int it = 0; it < 2; ++it) {
            const int r = (t >> 3) + 32 * it, u = t & 7;
            const int row = by + r, col = bx + u * 8;
            u32x4 v = {0u, 0u, 0u, 0u};
            if (row < rows && col + 7 < cols) v = *(const u32x4*)(in + (long long)row * ld_in + col);
            else if (row < rows) {
                unsigned short e[8];
                DPC_UNROLL
                for (int k = 0; k < 8; ++k) e[k] = col + k < cols ? *(const unsigned short*)(in + (long long)row * ld_in + col + k) : (unsigned short)0;
                v[0] = e[0] | ((unsigned)e[1] << 16); v[1] = e[2] | ((unsigned)e[3] << 16);
                v[2] = e[4] | ((unsigned)e[5] << 16); v[3] = e[6] | ((unsigned)e[7] << 16);
            }
            DPC_UNROLL
            for (int k = 0; k < 4; ++k) {
                tile[r][u * 8 + 2 * k] = (unsigned short)(v[k] & 0xffffu);
                tile[r][u * 8 + 2 * k + 1] = (unsigned short)(v[k] >> 16);
            }
        }
    }
    __syncthreads();
    DPC_UNROLL
    for (int it = 0; it < 2; ++it) {
        const int c = (t >> 3) + 32 * it, u = t & 7;     // output row = input column bx + c, 8 input rows by + 8u ..
        const int orow = bx + c, ocol = by + u * 8;
        if (orow >= cols) continue;
        unsigned short e[8];
        DPC_UNROLL
        for (int k = 0; k < 8; ++k) e[k] = tile[u * 8 + k][c];
        if (ocol + 7 < rows) {
            const u32x4 v = {e[0] | ((unsigned)e[1] << 16), e[2] | ((unsigned)e[3] << 16), e[4] | ((unsigned)e[5] << 16), e[6] | ((unsigned)e[7] << 16)};
            *(u32x4*)(out + (long long)orow * ld_out + ocol) = v;
        } else {
            DPC_UNROLL
            for (int k = 0; k < 8; ++k)
                if (ocol + k < rows) *(unsigned short*)(out + (long long)orow * ld_out + ocol + k) = e[k];
        }
    }
}

extern "C" int dpc_transpose2d_bf16x2(const void* in0, const void* in1, int32_t ld_in, void* out0, void* out1, int32_t ld_out, int32_t rows,
                                      int32_t cols, dpc_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!in0 || !out0 || rows <= 0 || cols <= 0 || ld_in < cols || ld_out < rows) return DPC_ERR_ARG;
    if (ld_in % 8 || ld_out % 8 || (((uintptr_t)in0 | (uintptr_t)in1 | (uintptr_t)out0 | (uintptr_t)out1) % 16)) return DPC_ERR_UNSUPPORTED;
    dim3 grid((cols + 63) / 64, (rows + 63) / 64, in1 ? 2 : 1), block(256);
    DPC_LAUNCH(transpose2d_bf16x2_kernel, grid, block, stream, (const bf16_t*)in0, (const bf16_t*)in1, ld_in, (bf16_t*)out0, (bf16_t*)out1, ld_out, rows, cols);
    return dpc_launch_status();
}

// ---- stem: NCDHW f32 video -> 2x2 space-to-depth channels-last, 16 channels (12 live)
// thread = one (n,t,hb,wb) cell; reads are float2 along W (coalesced NCDHW rows),
// the write is one contiguous 16-channel cell.
template <class TO>
__global__ void pack_input_s2d_kernel(const float* in, TO* out, int BN, int T, int H, int W, int rev) {
    const int Hb = H / 2, Wb = W / 2;
    const long long cells = (long long)BN * T * Hb * Wb;
    // rev (round 6): sweeps from the end of the tensor to its start, so that the HEAD of the operand is what the Infinity Cache holds
    // when the stem convolution starts reading it front to back
    const long long stride = (long long)gridDim.x * blockDim.x, first = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long nsweep = first < cells ? (cells - 1 - first) / stride + 1 : 0;
    for (long long k = 0; k < nsweep; ++k) {
        const long long i = first + (rev ? nsweep - 1 - k : k) * stride;
        const unsigned ci = (unsigned)i;  // cells < 2^31: 32-bit index math
        const unsigned q1 = ci / (unsigned)Wb;
        const int wb = (int)(ci - q1 * (unsigned)Wb);
        const unsigned q2 = q1 / (unsigned)Hb;
        const int hb = (int)(q1 - q2 * (unsigned)Hb);
        const unsigned n_ = q2 / (unsigned)T;
        const int t = (int)(q2 - n_ * (unsigned)T);
        const int n = (int)n_;
        float v[16];
        DPC_UNROLL
        for (int k = 12; k < 16; ++k) v[k] = 0.f;
        DPC_UNROLL
        for (int c = 0; c < 3; ++c)
            DPC_UNROLL
            for (int sy = 0; sy < 2; ++sy) {
                const float* p = in + ((((long long)n * 3 + c) * T + t) * H + (2 * hb + sy)) * W + 2 * wb;
                const float2 x = *(const float2*)p;
                v[(sy * 2 + 0) * 3 + c] = x.x;
                v[(sy * 2 + 1) * 3 + c] = x.y;
            }
        u32x4* o = (u32x4*)(out + i * 16);  // 16 channels = 2 (bf16) or 4 (f32) 16-byte units
        constexpr int EPO = Elt<TO>::PER16;
        DPC_UNROLL
        for (int k = 0; k < 16 / EPO; ++k) o[k] = unit_pack<TO>(v + k * EPO);
    }
}

extern "C" int dpc_pack_input_s2d(const float* block, void* out, int32_t dtype_out, int32_t BN, int32_t T, int32_t H,
                                  int32_t W, dpc_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!block || !out || BN <= 0 || T <= 0 || H <= 0 || W <= 0) return DPC_ERR_ARG;
    if ((H & 1) || (W & 1)) return DPC_ERR_UNSUPPORTED;
    const long long cells = (long long)BN * T * (H / 2) * (W / 2);
    static const int rev = getenv("DPC_PACK_REV") ? atoi(getenv("DPC_PACK_REV")) : 0;
    if (dtype_out == DPC_F32) {
        DPC_LAUNCH((pack_input_s2d_kernel<float>), dim3(grid_for(cells, 256, 16384)), dim3(256), stream, block, (float*)out, BN, T, H, W, rev);
    } else if (dtype_out == DPC_BF16) {
        DPC_LAUNCH((pack_input_s2d_kernel<bf16_t>), dim3(grid_for(cells, 256, 16384)), dim3(256), stream, block, (bf16_t*)out, BN, T, H, W, rev);
    } else {
        return DPC_ERR_ARG;
    }
    return dpc_launch_status();
}

// stem weight [Co][3][1][7][7] -> [Co][th*4+tw][(sy*2+sx)*3+c], ky = 2*th+sy-1, kx = 2*tw+sx-1
template <class TO>
__global__ void pack_stem_weight_kernel(const float* w, TO* out, int Co) {
    const int n = Co * 256;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const int c16 = i & 15, tap = (i >> 4) & 15, co = i >> 8;
        const int th = tap >> 2, tw = tap & 3;
        float v = 0.f;
        if (c16 < 12) {
            const int c = c16 % 3, s = c16 / 3, sy = s >> 1, sx = s & 1;
            const int ky = 2 * th + sy - 1, kx = 2 * tw + sx - 1;
            if (ky >= 0 && ky < 7 && kx >= 0 && kx < 7) v = w[((co * 3 + c) * 7 + ky) * 7 + kx];
        }
        out[i] = Elt<TO>::from_f32(v);
    }
}

extern "C" int dpc_pack_stem_weight(const float* w, void* out, int32_t dtype_out, int32_t Co, dpc_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!w || !out || Co <= 0) return DPC_ERR_ARG;
    if (dtype_out == DPC_F32) {
        DPC_LAUNCH((pack_stem_weight_kernel<float>), dim3(grid_for(Co * 256)), dim3(256), stream, w, (float*)out, Co);
    } else if (dtype_out == DPC_BF16) {
        DPC_LAUNCH((pack_stem_weight_kernel<bf16_t>), dim3(grid_for(Co * 256)), dim3(256), stream, w, (bf16_t*)out, Co);
    } else {
        return DPC_ERR_ARG;
    }
    return dpc_launch_status();
}

// The slabs are [Co][16 taps][16 slots] f32 (the space-to-depth layout); the gradient is [Co][3][7][7].  A thread owns four
// consecutive SOURCE elements (16-byte loads down the slabs, the layout of reduce_unpack4_kernel) and scatters its four sums --
// walking the slabs by destination index read every 128-byte line from several workgroups (250 MB of HBM reads for 67 MB of
// slabs, 56 us on 294 workgroups).  Summation order is fixed by (lane, slab index): deterministic.
__global__ __launch_bounds__(256) void unpack_stem_wgrad_kernel(const float* part, int nsplit, float* dw, int Co) {
    __shared__ float red[8][32][4];
    const int n = Co * 256;
    const int ol = threadIdx.x & 31, sl = threadIdx.x >> 5;
    const int i = (blockIdx.x * 32 + ol) * 4;
    // EIGHT slabs per split lane and trip, loaded before any is added: with two (rounds 3-5) the walk was a chain of HBM round trips --
    // 62 us for the 512 slabs x 32 KB of layer2.0's downsample gradient (64 workgroups), on the side stream's critical path there
    float acc[8][4];
    DPC_UNROLL
    for (int j = 0; j < 8; ++j)
        DPC_UNROLL
        for (int e = 0; e < 4; ++e) acc[j][e] = 0.f;
    if (i < n) {
        int k = sl;
        for (; k + 56 < nsplit; k += 64) {
            u32x4 v[8];
            DPC_UNROLL
            for (int j = 0; j < 8; ++j) v[j] = *(const u32x4*)(part + (long long)(k + 8 * j) * n + i);
            DPC_UNROLL
            for (int j = 0; j < 8; ++j)
                DPC_UNROLL
                for (int e = 0; e < 4; ++e) acc[j][e] += unit_get<float>(v[j], e);
        }
        DPC_UNROLL
        for (int j = 0; j < 8; ++j)
            if (k + 8 * j < nsplit) {
                const u32x4 v0 = *(const u32x4*)(part + (long long)(k + 8 * j) * n + i);
                DPC_UNROLL
                for (int e = 0; e < 4; ++e) acc[j][e] += unit_get<float>(v0, e);
            }
    }
    float a[4], b[4];
    DPC_UNROLL
    for (int e = 0; e < 4; ++e) {
        a[e] = (acc[0][e] + acc[1][e]) + (acc[2][e] + acc[3][e]);
        b[e] = (acc[4][e] + acc[5][e]) + (acc[6][e] + acc[7][e]);
    }
    DPC_UNROLL
    for (int e = 0; e < 4; ++e) red[sl][ol][e] = a[e] + b[e];
    __syncthreads();
    if (sl == 0 && i < n) {
        DPC_UNROLL
        for (int e = 0; e < 4; ++e) {
            float t = 0.f;
            DPC_UNROLL
            for (int k = 0; k < 8; ++k) t += red[k][ol][e];
            const int src = i + e;                       // (co * 16 + th * 4 + tw) * 16 + (sy * 2 + sx) * 3 + c
            const int slot = src & 15, tap = (src >> 4) & 15, co = src >> 8;
            const int c = slot % 3, sxy = slot / 3;      // slots 12..15 are padding
            const int ky = 2 * (tap >> 2) + (sxy >> 1) - 1, kx = 2 * (tap & 3) + (sxy & 1) - 1;
            if (slot < 12 && ky >= 0 && ky < 7 && kx >= 0 && kx < 7) dw[((co * 3 + c) * 7 + ky) * 7 + kx] = t;
        }
    }
}

extern "C" int dpc_unpack_stem_wgrad(const float* part, int32_t nsplit, float* dw, int32_t Co, dpc_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!part || !dw || nsplit <= 0 || Co <= 0) return DPC_ERR_ARG;
    DPC_LAUNCH(unpack_stem_wgrad_kernel, dim3((Co * 256 + 127) / 128), dim3(256), stream, part, nsplit, dw, Co);
    return dpc_launch_status();
}

// dst[r][c] = src[r][c] for an f32 [rows][cols] window (leading dimensions in elements): scatters the batched ConvGRU
// weight-gradient GEMM results into the reference's [D][2D] gate parameters (x half | h half)
__global__ void copy2d_f32_kernel(const float* src, long long src_ld, float* dst, long long dst_ld, int rows, int cols) {
    const long long n = (long long)rows * cols;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const long long r = i / cols, c = i % cols;
        dst[r * dst_ld + c] = src[r * src_ld + c];
    }
}

__global__ void copy2d_multi_kernel(const dpc_copy2d_entry* tab, int n_entries) {
    int e = 0;
    while (e + 1 < n_entries && (int)blockIdx.x >= tab[e + 1].block0) ++e;
    const dpc_copy2d_entry t = tab[e];
    const int nblk = (e + 1 < n_entries ? tab[e + 1].block0 : (int)gridDim.x) - t.block0;
    const long long n = (long long)t.rows * t.cols;
    for (long long i = (long long)((int)blockIdx.x - t.block0) * blockDim.x + threadIdx.x; i < n; i += (long long)nblk * blockDim.x) {
        const long long r = i / t.cols, c = i % t.cols;
        t.dst[r * t.dst_ld + c] = t.src[r * t.src_ld + c];
    }
}

extern "C" int dpc_copy2d_multi(const dpc_copy2d_entry* table_dev, int32_t n_entries, int32_t total_blocks, dpc_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!table_dev || n_entries <= 0 || total_blocks < n_entries) return DPC_ERR_ARG;
    DPC_LAUNCH(copy2d_multi_kernel, dim3((unsigned)total_blocks), dim3(256), stream, table_dev, n_entries);
    return dpc_launch_status();
}

extern "C" int dpc_copy2d_f32(const float* src, int64_t src_ld, float* dst, int64_t dst_ld, int32_t rows, int32_t cols,
                              dpc_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!src || !dst || rows <= 0 || cols <= 0 || src_ld < cols || dst_ld < cols) return DPC_ERR_ARG;
    const long long n = (long long)rows * cols;
    long long g = (n + 255) / 256;
    if (g > 2048) g = 2048;
    DPC_LAUNCH(copy2d_f32_kernel, dim3((unsigned)g), dim3(256), stream, src, (long long)src_ld, dst, (long long)dst_ld, rows, cols);
    return dpc_launch_status();
}
