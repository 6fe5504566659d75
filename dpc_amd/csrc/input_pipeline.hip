// input_pipeline.hip -- GPU side of the reference's data path: decoded RGB frames (uint8, HWC as
// PIL hands them over, dpc/dataset_3d.py:104) -> the model's input.  Everything per pixel happens here, bit for bit what
// PIL / torchvision compute on the host in the reference; the per-clip / per-frame random draws stay on the host
// (dpc_amd/data.py mirrors the reference's use of `random` / `np.random` call by call).
//   frame sampling   idx_sampler: frame(n, sl) = start + (n * seq_len + sl) * ds                (dataset_3d.py:85-92)
//   RandomCrop / the crop box of RandomSizedCrop, consistent over the clip                        (utils/augmentation.py:99-196)
//   resize           Scale (NEAREST) and RandomSizedCrop (BILINEAR, antialiased when shrinking) = PIL's separable
//                    ImagingResample, 8 bits per channel: per output column / row a window of source indices and 22-bit
//                    fixed-point coefficients (built on the host, dpc_amd/data.py: resample_tables), horizontal pass rounded and
//                    clipped to 8 bits, then the vertical pass -- the same integers PIL produces      (augmentation.py:20-43,144-196)
//   RandomHorizontalFlip(consistent=True), before the crop (ucf101 recipe) or after it (k400)     (augmentation.py:198-222)
//   RandomGray = channel splitting, one channel choice per frame                                  (augmentation.py:224-251)
//   ColorJitter      per frame: brightness / contrast / saturation / hue in a shuffled order (augmentation.py:253-351 through
//                    torchvision.transforms.functional = PIL ImageEnhance + an HSV round trip):
//                      blend(a, b, f) = trunc(a + f (b - a)) with f as a C float, clipped to [0, 255] when f > 1     (PIL Blend.c)
//                      brightness: blend(0, px, f); saturation: blend(L(px), px, f); contrast: blend(m, px, f) with
//                      m = int(mean of L over the frame + 0.5), L = (19595 R + 38470 G + 7471 B + 32768) >> 16      (PIL Convert.c)
//                      hue: RGB -> HSV (float / double steps exactly as Convert.c orders them), H += shift (uint8 wrap), HSV -> RGB
//                    -- verified against PIL for all 2^24 colours on the host (tests/golden/make_aug_golden.py)
//   ToTensor (/255) + Normalize(mean, std)                                                        (augmentation.py:368-379)
//   stack / view / transpose to [N, C, SL, H, W]                                                  (dataset_3d.py:107-111)
// written either as the boundary tensor block [B][N][3][SL][H][W] f32 (what DPC_RNN.forward takes) or DIRECTLY as the
// stem's space-to-depth operand (dpc_pack_input_s2d's output), so that a host-fed step moves 1/4 of the bytes over PCIe
// (uint8 instead of f32) and the f32 video never exists in HBM.
// ColorJitter's contrast needs a whole-frame mean in the MIDDLE of the per-pixel chain, so with jitter the work is three
// launches over a uint8 intermediate (geometry + gray -> u8 image; per-frame sum of L just before the contrast step; jitter
// + normalise + layout); without jitter it is one.  HBM-bound byte work.
#include "dpc_rt.h"
#include "../../include/dpc_hip.h"

namespace {

constexpr int RS_PREC = 32 - 8 - 2;  // PIL ImagingResample PRECISION_BITS

struct Geo {
    const uint8_t* frames;
    int B, F, H0, W0;
    const dpc_clip_aug* aug;
    const int8_t* gray;
    int N, SL, ds, H, W;
    // NEAREST tables (legacy entry) or general resampling tables (per clip); all NULL = no resize
    const int32_t *xtab, *ytab;
    const int32_t *xb, *xk, *yb, *yk;  // [B][W][2], [B][W][ksx], [B][H][2], [B][H][ksy]
    int ksx, ksy;
};

__device__ __forceinline__ int clip8(int v) {
    v >>= RS_PREC;
    return v < 0 ? 0 : (v > 255 ? 255 : v);
}

// the three channels of output pixel (y, x) of frame (b, f): crop / flip / resize, then the RandomGray channel choice
__device__ __forceinline__ void geo_pixel(const Geo& g, int b, int f, int gch, const dpc_clip_aug& a, int y, int x, int (&rgb)[3]) {
    const uint8_t* fr = g.frames + ((long long)b * g.F + f) * g.H0 * g.W0 * 3;
    // flip = 1: flip AFTER crop (+ resize) (the k400 recipe); flip = 2: flip of the full frame BEFORE the crop (ucf101)
    const int xo = a.flip == 1 ? g.W - 1 - x : x;
    if (g.xb) {
        const int* xb = g.xb + ((long long)b * g.W + xo) * 2;
        const int* yb = g.yb + ((long long)b * g.H + y) * 2;
        const int* xk = g.xk + ((long long)b * g.W + xo) * g.ksx;
        const int* yk = g.yk + ((long long)b * g.H + y) * g.ksy;
        int acc[3] = {1 << (RS_PREC - 1), 1 << (RS_PREC - 1), 1 << (RS_PREC - 1)};
        for (int vy = 0; vy < yb[1]; ++vy) {
            const int ysrc = a.y1 + yb[0] + vy;
            int hs[3] = {1 << (RS_PREC - 1), 1 << (RS_PREC - 1), 1 << (RS_PREC - 1)};
            for (int vx = 0; vx < xb[1]; ++vx) {
                const int xc = a.x1 + xb[0] + vx;
                const int xsrc = a.flip == 2 ? g.W0 - 1 - xc : xc;
                const uint8_t* px = fr + ((long long)ysrc * g.W0 + xsrc) * 3;
                const int k = xk[vx];
                DPC_UNROLL
                for (int c = 0; c < 3; ++c) hs[c] += (int)px[c] * k;
            }
            const int k = yk[vy];
            DPC_UNROLL
            for (int c = 0; c < 3; ++c) acc[c] += clip8(hs[c]) * k;   // the horizontal pass is stored as 8 bits before the vertical one
        }
        DPC_UNROLL
        for (int c = 0; c < 3; ++c) rgb[c] = clip8(acc[c]);
    } else {
        const int xc = g.xtab ? g.xtab[xo] : xo;            // column inside the crop box
        const int ysrc = a.y1 + (g.ytab ? g.ytab[y] : y);
        const int xsrc = a.flip == 2 ? g.W0 - 1 - (a.x1 + xc) : a.x1 + xc;
        const uint8_t* px = fr + ((long long)ysrc * g.W0 + xsrc) * 3;
        DPC_UNROLL
        for (int c = 0; c < 3; ++c) rgb[c] = px[c];
    }
    if (gch >= 0) { const int v = rgb[gch]; rgb[0] = v; rgb[1] = v; rgb[2] = v; }
}

// ---- PIL arithmetic of ColorJitter
__device__ __forceinline__ int pil_L(const int (&p)[3]) { return (p[0] * 19595 + p[1] * 38470 + p[2] * 7471 + 0x8000) >> 16; }
__device__ __forceinline__ int pil_blend(int in1, int in2, float alpha) {   // Blend.c: interpolation truncates, extrapolation clips first
    const float t = (float)in1 + alpha * (float)(in2 - in1);
    if (alpha >= 0.f && alpha <= 1.f) return (int)t;
    return t <= 0.f ? 0 : (t >= 255.f ? 255 : (int)t);
}
__device__ __forceinline__ void pil_rgb2hsv(const int (&p)[3], int (&o)[3]) {    // Convert.c rgb2hsv_row
    const int r = p[0], g = p[1], b = p[2];
    const int maxc = r > g ? (r > b ? r : b) : (g > b ? g : b), minc = r < g ? (r < b ? r : b) : (g < b ? g : b);
    o[2] = maxc;
    if (minc == maxc) { o[0] = 0; o[1] = 0; return; }
    const float cr = (float)(maxc - minc);
    const float s = cr / (float)maxc;
    const float rc = (float)(maxc - r) / cr, gc = (float)(maxc - g) / cr, bc = (float)(maxc - b) / cr;
    float h;
    if (r == maxc) h = bc - gc;                                   // float - float
    else if (g == maxc) h = (float)(2.0 + (double)rc - (double)bc);   // "2.0 + rc - bc": double arithmetic, stored as float
    else h = (float)(4.0 + (double)gc - (double)rc);
    h = (float)fmod((double)h / 6.0 + 1.0, 1.0);
    int uh = (int)((double)h * 255.0), us = (int)((double)s * 255.0);
    o[0] = uh < 0 ? 0 : (uh > 255 ? 255 : uh);
    o[1] = us < 0 ? 0 : (us > 255 ? 255 : us);
}
__device__ __forceinline__ int c_round(double x) { return (int)(x >= 0.0 ? floor(x + 0.5) : ceil(x - 0.5)); }
__device__ __forceinline__ void pil_hsv2rgb(const int (&q)[3], int (&o)[3]) {    // Convert.c hsv2rgb
    const int h = q[0], s = q[1], v = q[2];
    if (s == 0) { o[0] = v; o[1] = v; o[2] = v; return; }
    const double hf = (double)(float)h * 6.0 / 255.0;
    const int i = (int)floor(hf);
    const float f = (float)(hf - (double)(float)i);
    const float fs = (float)((double)(float)s / 255.0);
    int pp = c_round((double)v * (1.0 - (double)fs));
    int qq = c_round((double)v * (1.0 - (double)fs * (double)f));
    int tt = c_round((double)v * (1.0 - (double)fs * (1.0 - (double)f)));
    pp = pp < 0 ? 0 : (pp > 255 ? 255 : pp); qq = qq < 0 ? 0 : (qq > 255 ? 255 : qq); tt = tt < 0 ? 0 : (tt > 255 ? 255 : tt);
    switch (i % 6) {
        case 0: o[0] = v; o[1] = tt; o[2] = pp; break;
        case 1: o[0] = qq; o[1] = v; o[2] = pp; break;
        case 2: o[0] = pp; o[1] = v; o[2] = tt; break;
        case 3: o[0] = pp; o[1] = qq; o[2] = v; break;
        case 4: o[0] = tt; o[1] = pp; o[2] = v; break;
        default: o[0] = v; o[1] = pp; o[2] = qq; break;
    }
}
// steps [k0, k1) of a frame's shuffled ColorJitter chain on one pixel; lmean = int(mean L + 0.5) taken just before the contrast step
__device__ __forceinline__ void jitter_steps(const dpc_frame_jitter& j, int k0, int k1, int lmean, int (&p)[3]) {
    for (int k = k0; k < k1; ++k) {
        const int op = j.order[k];
        if (op == 0) {
            DPC_UNROLL
            for (int c = 0; c < 3; ++c) p[c] = pil_blend(0, p[c], j.factor[0]);
        } else if (op == 1) {
            DPC_UNROLL
            for (int c = 0; c < 3; ++c) p[c] = pil_blend(lmean, p[c], j.factor[1]);
        } else if (op == 2) {
            const int l = pil_L(p);
            DPC_UNROLL
            for (int c = 0; c < 3; ++c) p[c] = pil_blend(l, p[c], j.factor[2]);
        } else if (op == 3) {
            int hsv[3];
            pil_rgb2hsv(p, hsv);
            hsv[0] = (hsv[0] + j.hue_shift) & 255;
            pil_hsv2rgb(hsv, p);
        }
    }
}
__device__ __forceinline__ int contrast_pos(const dpc_frame_jitter& j) {
    DPC_UNROLL
    for (int k = 0; k < 4; ++k)
        if (j.order[k] == 1) return k;
    return 4;
}

__device__ __forceinline__ void decode_cell(unsigned ci, int Hb, int Wb, int SL, int N, int& b, int& n, int& sl, int& hb, int& wb) {
    const unsigned q1 = ci / (unsigned)Wb;
    wb = (int)(ci - q1 * (unsigned)Wb);
    const unsigned q2 = q1 / (unsigned)Hb;
    hb = (int)(q1 - q2 * (unsigned)Hb);
    const unsigned q3 = q2 / (unsigned)SL;
    sl = (int)(q2 - q3 * (unsigned)SL);
    b = (int)(q3 / (unsigned)N);
    n = (int)(q3 - (q3 / (unsigned)N) * (unsigned)N);
}

// normalise one 2x2 cell (12 u8 values -> f32) and write block / space-to-depth operand
template <class TO>
__device__ __forceinline__ void emit_cell(const int (&px)[4][3], long long i, int b, int n, int sl, int hb, int wb, int N, int SL, int H, int W,
                                          const float (&mean)[3], const float (&stdv)[3], float* block, TO* s2d) {
    float v[16];
    DPC_UNROLL
    for (int k = 12; k < 16; ++k) v[k] = 0.f;
    DPC_UNROLL
    for (int q = 0; q < 4; ++q) {
        const int y = 2 * hb + (q >> 1), x = 2 * wb + (q & 1);
        DPC_UNROLL
        for (int c = 0; c < 3; ++c) {
            const float t = ((float)px[q][c] / 255.f - mean[c]) / stdv[c];  // ToTensor then Normalize, f32 like torchvision
            v[q * 3 + c] = t;
            if (block) block[((((long long)(b * N + n) * 3 + c) * SL + sl) * H + y) * W + x] = t;
        }
    }
    if (s2d) {
        u32x4* o = (u32x4*)(s2d + i * 16);
        constexpr int EPO = Elt<TO>::PER16;
        DPC_UNROLL
        for (int k = 0; k < 16 / EPO; ++k) o[k] = unit_pack<TO>(v + k * EPO);
    }
}

// one thread per 2x2 output cell and frame.  U8OUT: write the geometry stage as a uint8 image [B][N*SL][H][W][3] (jitter follows)
template <class TO, bool U8OUT>
__global__ void frames_geo_kernel(Geo g, float m0, float m1, float m2, float s0, float s1, float s2, float* block, TO* s2d, uint8_t* u8) {
    const int Hb = g.H / 2, Wb = g.W / 2;
    const long long cells = (long long)g.B * g.N * g.SL * Hb * Wb;
    const float mean[3] = {m0, m1, m2}, stdv[3] = {s0, s1, s2};
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < cells; i += (long long)gridDim.x * blockDim.x) {
        int b, n, sl, hb, wb;
        decode_cell((unsigned)i, Hb, Wb, g.SL, g.N, b, n, sl, hb, wb);
        const dpc_clip_aug a = g.aug[b];
        const int f = a.start + (n * g.SL + sl) * g.ds;
        const int gch = g.gray ? (int)g.gray[(long long)b * g.N * g.SL + n * g.SL + sl] : -1;
        int px[4][3];
        DPC_UNROLL
        for (int q = 0; q < 4; ++q) geo_pixel(g, b, f, gch, a, 2 * hb + (q >> 1), 2 * wb + (q & 1), px[q]);
        if (U8OUT) {
            uint8_t* o = u8 + (((long long)b * g.N * g.SL + n * g.SL + sl) * g.H * g.W) * 3;
            DPC_UNROLL
            for (int q = 0; q < 4; ++q) {
                uint8_t* d = o + ((long long)(2 * hb + (q >> 1)) * g.W + 2 * wb + (q & 1)) * 3;
                d[0] = (uint8_t)px[q][0]; d[1] = (uint8_t)px[q][1]; d[2] = (uint8_t)px[q][2];
            }
        } else {
            emit_cell<TO>(px, i, b, n, sl, hb, wb, g.N, g.SL, g.H, g.W, mean, stdv, block, s2d);
        }
    }
}

// per frame: sum of L over the image as it is just before the contrast step of that frame's chain (exact integers)
__global__ __launch_bounds__(256) void jitter_lsum_kernel(const uint8_t* u8, const dpc_frame_jitter* jit, int npix, unsigned long long* lsum) {
    __shared__ unsigned long long sh[4];
    const int fi = blockIdx.x;
    const dpc_frame_jitter j = jit[fi];
    const int kc = contrast_pos(j);
    unsigned long long acc = 0;
    if (kc < 4) {
        const uint8_t* im = u8 + (long long)fi * npix * 3;
        for (int i = threadIdx.x; i < npix; i += 256) {
            int p[3] = {im[3 * i], im[3 * i + 1], im[3 * i + 2]};
            jitter_steps(j, 0, kc, 0, p);
            acc += (unsigned long long)pil_L(p);
        }
    }
    DPC_UNROLL
    for (int m = 32; m >= 1; m >>= 1) acc += __shfl_xor(acc, m);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) lsum[fi] = sh[0] + sh[1] + sh[2] + sh[3];
}

template <class TO>
__global__ void frames_jitter_kernel(const uint8_t* u8, const dpc_frame_jitter* jit, const unsigned long long* lsum, int B, int N, int SL, int H, int W,
                                     float m0, float m1, float m2, float s0, float s1, float s2, float* block, TO* s2d) {
    const int Hb = H / 2, Wb = W / 2;
    const long long cells = (long long)B * N * SL * Hb * Wb;
    const float mean[3] = {m0, m1, m2}, stdv[3] = {s0, s1, s2};
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < cells; i += (long long)gridDim.x * blockDim.x) {
        int b, n, sl, hb, wb;
        decode_cell((unsigned)i, Hb, Wb, SL, N, b, n, sl, hb, wb);
        const long long fi = (long long)b * N * SL + n * SL + sl;
        const dpc_frame_jitter j = jit[fi];
        // ImageStat.Stat(L).mean[0] = sum / count in double; int(mean + 0.5)
        const int lmean = (int)((double)lsum[fi] / (double)(H * W) + 0.5);
        const uint8_t* im = u8 + fi * H * W * 3;
        int px[4][3];
        DPC_UNROLL
        for (int q = 0; q < 4; ++q) {
            const uint8_t* s = im + ((long long)(2 * hb + (q >> 1)) * W + 2 * wb + (q & 1)) * 3;
            px[q][0] = s[0]; px[q][1] = s[1]; px[q][2] = s[2];
            jitter_steps(j, 0, 4, lmean, px[q]);
        }
        emit_cell<TO>(px, i, b, n, sl, hb, wb, N, SL, H, W, mean, stdv, block, s2d);
    }
}

int check_common(const uint8_t* frames, int B, int F, int H0, int W0, const dpc_clip_aug* aug, int N, int SL, int ds, int H, int W,
                 const float* mean3, const float* std3, float* block, void* s2d) {
    if (!frames || !aug || !mean3 || !std3 || (!block && !s2d) || B <= 0 || F <= 0 || H0 <= 0 || W0 <= 0 || N <= 0 || SL <= 0 || ds <= 0 ||
        H <= 0 || W <= 0)
        return DPC_ERR_ARG;
    if ((H & 1) || (W & 1)) return DPC_ERR_UNSUPPORTED;
    if ((long long)B * N * SL * (H / 2) * (W / 2) >= (1ll << 31)) return DPC_ERR_UNSUPPORTED;
    return DPC_OK;
}

unsigned grid_cells(long long cells) {
    long long g = (cells + 255) / 256;
    if (g > 16384) g = 16384;
    return (unsigned)(g < 1 ? 1 : g);
}

}  // namespace

extern "C" int dpc_frames_to_input(const uint8_t* frames, int32_t B, int32_t F, int32_t H0, int32_t W0, const dpc_clip_aug* aug,
                                   const int8_t* gray, int32_t N, int32_t SL, int32_t ds, int32_t H, int32_t W, const int32_t* xtab,
                                   const int32_t* ytab, int32_t crop_w, int32_t crop_h, const float* mean3, const float* std3, float* block,
                                   void* s2d, int32_t dtype_s2d, dpc_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (int rc = check_common(frames, B, F, H0, W0, aug, N, SL, ds, H, W, mean3, std3, block, s2d)) return rc;
    if (!xtab != !ytab) return DPC_ERR_ARG;
    if (!xtab) { crop_w = W; crop_h = H; }
    if (crop_h > H0 || crop_w > W0 || crop_w <= 0 || crop_h <= 0) return DPC_ERR_UNSUPPORTED;
    Geo g = {frames, B, F, H0, W0, aug, gray, N, SL, ds, H, W, xtab, ytab, nullptr, nullptr, nullptr, nullptr, 0, 0};
    const unsigned grid = grid_cells((long long)B * N * SL * (H / 2) * (W / 2));
    if (!s2d || dtype_s2d == DPC_F32) {
        DPC_LAUNCH((frames_geo_kernel<float, false>), dim3(grid), dim3(256), stream, g, mean3[0], mean3[1], mean3[2], std3[0], std3[1], std3[2], block,
                   (float*)s2d, (uint8_t*)nullptr);
    } else if (dtype_s2d == DPC_BF16) {
        DPC_LAUNCH((frames_geo_kernel<bf16_t, false>), dim3(grid), dim3(256), stream, g, mean3[0], mean3[1], mean3[2], std3[0], std3[1], std3[2], block,
                   (bf16_t*)s2d, (uint8_t*)nullptr);
    } else {
        return DPC_ERR_ARG;
    }
    return dpc_launch_status();
}

extern "C" int dpc_frames_to_input_ex(const uint8_t* frames, int32_t B, int32_t F, int32_t H0, int32_t W0, const dpc_clip_aug* aug,
                                      const int8_t* gray, int32_t N, int32_t SL, int32_t ds, int32_t H, int32_t W, const dpc_resample* rs,
                                      const dpc_frame_jitter* jitter, uint8_t* u8_ws, uint64_t* lsum_ws, const float* mean3,
                                      const float* std3, float* block, void* s2d, int32_t dtype_s2d, dpc_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (int rc = check_common(frames, B, F, H0, W0, aug, N, SL, ds, H, W, mean3, std3, block, s2d)) return rc;
    if (jitter && (!u8_ws || !lsum_ws)) return DPC_ERR_ARG;
    if (rs && (!rs->xb || !rs->xk || !rs->yb || !rs->yk || rs->ksx <= 0 || rs->ksy <= 0)) return DPC_ERR_ARG;
    if (s2d && dtype_s2d != DPC_F32 && dtype_s2d != DPC_BF16) return DPC_ERR_ARG;
    Geo g = {frames, B, F, H0, W0, aug, gray, N, SL, ds, H, W, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0, 0};
    if (rs) { g.xb = rs->xb; g.xk = rs->xk; g.yb = rs->yb; g.yk = rs->yk; g.ksx = rs->ksx; g.ksy = rs->ksy; }
    const long long cells = (long long)B * N * SL * (H / 2) * (W / 2);
    const unsigned grid = grid_cells(cells);
    const bool bf = s2d && dtype_s2d == DPC_BF16;
    if (!jitter) {
        if (bf) {
            DPC_LAUNCH((frames_geo_kernel<bf16_t, false>), dim3(grid), dim3(256), stream, g, mean3[0], mean3[1], mean3[2], std3[0], std3[1], std3[2], block,
                       (bf16_t*)s2d, (uint8_t*)nullptr);
        } else {
            DPC_LAUNCH((frames_geo_kernel<float, false>), dim3(grid), dim3(256), stream, g, mean3[0], mean3[1], mean3[2], std3[0], std3[1], std3[2], block,
                       (float*)s2d, (uint8_t*)nullptr);
        }
        return dpc_launch_status();
    }
    DPC_LAUNCH((frames_geo_kernel<float, true>), dim3(grid), dim3(256), stream, g, 0.f, 0.f, 0.f, 1.f, 1.f, 1.f, (float*)nullptr, (float*)nullptr, u8_ws);
    DPC_LAUNCH(jitter_lsum_kernel, dim3((unsigned)(B * N * SL)), dim3(256), stream, (const uint8_t*)u8_ws, jitter, H * W, (unsigned long long*)lsum_ws);
    if (bf) {
        DPC_LAUNCH((frames_jitter_kernel<bf16_t>), dim3(grid), dim3(256), stream, (const uint8_t*)u8_ws, jitter, (const unsigned long long*)lsum_ws, B, N, SL, H, W,
                   mean3[0], mean3[1], mean3[2], std3[0], std3[1], std3[2], block, (bf16_t*)s2d);
    } else {
        DPC_LAUNCH((frames_jitter_kernel<float>), dim3(grid), dim3(256), stream, (const uint8_t*)u8_ws, jitter, (const unsigned long long*)lsum_ws, B, N, SL, H, W,
                   mean3[0], mean3[1], mean3[2], std3[0], std3[1], std3[2], block, (float*)s2d);
    }
    return dpc_launch_status();
}
