// input_pipeline.hip -- GPU side of the reference's data path: decoded RGB frames (uint8, HWC as
// PIL hands them over, dpc/dataset_3d.py:104) -> the model's input, in one pass:
//   frame sampling   idx_sampler: frame(n, sl) = start + (n * seq_len + sl) * ds                (dataset_3d.py:85-92)
//   RandomCrop / the crop of RandomSizedCrop, consistent over the clip (a box per clip)           (utils/augmentation.py:99-143)
//   RandomHorizontalFlip(consistent=True)                                                         (augmentation.py:198-222)
//   RandomGray = channel splitting, one channel choice per frame                                  (augmentation.py:224-251)
//   Scale(size) with its default NEAREST interpolation (the ucf101 recipe: crop 224 -> 128, dpc/main.py:118): the
//     output -> source index tables are produced on the host by PIL ITSELF (a resized coordinate ramp), so the kernel
//     reproduces PIL's own rounding for any size                                                   (augmentation.py:20-43)
//   ToTensor (/255) + Normalize(mean, std)                                                        (augmentation.py:368-379)
//   stack / view / transpose to [N, C, SL, H, W]                                                  (dataset_3d.py:107-111)
// written either as the boundary tensor block [B][N][3][SL][H][W] f32 (what DPC_RNN.forward takes) or DIRECTLY as the
// stem's space-to-depth operand (dpc_pack_input_s2d's output), so that a host-fed step moves 1/4 of the bytes over PCIe
// (uint8 instead of f32) and the f32 video never exists in HBM.  Not covered (PIL-specific resampling / colour space
// code): the BILINEAR resize of RandomSizedCrop and ColorJitter -- they stay on the host when a run uses them.
// HBM-bound byte work: one thread per 2x2 output cell and frame, uint8 reads, 16/32-byte writes.
#include "dpc_rt.h"
#include "../../include/dpc_hip.h"

template <class TO>
__global__ void frames_to_input_kernel(const uint8_t* frames, int B, int F, int H0, int W0, const dpc_clip_aug* aug, const int8_t* gray,
                                       int N, int SL, int ds, int H, int W, const int32_t* xtab, const int32_t* ytab, int CW, float m0, float m1,
                                       float m2, float s0, float s1, float s2, float* block, TO* s2d) {
    const int Hb = H / 2, Wb = W / 2;
    const long long cells = (long long)B * N * SL * Hb * Wb;
    const float mean[3] = {m0, m1, m2}, stdv[3] = {s0, s1, s2};
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < cells; i += (long long)gridDim.x * blockDim.x) {
        const unsigned ci = (unsigned)i;
        const unsigned q1 = ci / (unsigned)Wb;
        const int wb = (int)(ci - q1 * (unsigned)Wb);
        const unsigned q2 = q1 / (unsigned)Hb;
        const int hb = (int)(q1 - q2 * (unsigned)Hb);
        const unsigned q3 = q2 / (unsigned)SL;
        const int sl = (int)(q2 - q3 * (unsigned)SL);
        const int b = (int)(q3 / (unsigned)N), n = (int)(q3 - (q3 / (unsigned)N) * (unsigned)N);
        const dpc_clip_aug a = aug[b];
        const int f = a.start + (n * SL + sl) * ds;
        const int gch = gray ? (int)gray[(long long)b * N * SL + n * SL + sl] : -1;
        const uint8_t* fr = frames + ((long long)b * F + f) * H0 * W0 * 3;
        float v[16];
        DPC_UNROLL
        for (int k = 12; k < 16; ++k) v[k] = 0.f;
        DPC_UNROLL
        for (int sy = 0; sy < 2; ++sy)
            DPC_UNROLL
            for (int sx = 0; sx < 2; ++sx) {
                const int y = 2 * hb + sy, x = 2 * wb + sx;
                // flip = 1: flip AFTER crop (+ scale) (the k400 recipe); flip = 2: flip of the full frame BEFORE the crop (ucf101)
                const int xo = a.flip == 1 ? W - 1 - x : x;
                const int xc = xtab ? xtab[xo] : xo;            // column inside the crop box
                const int ysrc = a.y1 + (ytab ? ytab[y] : y);
                const int xsrc = a.flip == 2 ? W0 - 1 - (a.x1 + xc) : a.x1 + xc;
                (void)CW;
                const uint8_t* px = fr + ((long long)ysrc * W0 + xsrc) * 3;
                DPC_UNROLL
                for (int c = 0; c < 3; ++c) {
                    const float u = (float)px[gch >= 0 ? gch : c];
                    const float t = (u / 255.f - mean[c]) / stdv[c];  // ToTensor then Normalize, f32 like torchvision
                    v[(sy * 2 + sx) * 3 + c] = t;
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
}

extern "C" int dpc_frames_to_input(const uint8_t* frames, int32_t B, int32_t F, int32_t H0, int32_t W0, const dpc_clip_aug* aug,
                                   const int8_t* gray, int32_t N, int32_t SL, int32_t ds, int32_t H, int32_t W, const int32_t* xtab,
                                   const int32_t* ytab, int32_t crop_w, int32_t crop_h, const float* mean3, const float* std3, float* block,
                                   void* s2d, int32_t dtype_s2d, dpc_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!frames || !aug || !mean3 || !std3 || (!block && !s2d) || B <= 0 || F <= 0 || H0 <= 0 || W0 <= 0 || N <= 0 || SL <= 0 || ds <= 0 ||
        H <= 0 || W <= 0)
        return DPC_ERR_ARG;
    if (!xtab != !ytab) return DPC_ERR_ARG;
    if (!xtab) { crop_w = W; crop_h = H; }
    if ((H & 1) || (W & 1) || crop_h > H0 || crop_w > W0 || crop_w <= 0 || crop_h <= 0) return DPC_ERR_UNSUPPORTED;
    const long long cells = (long long)B * N * SL * (H / 2) * (W / 2);
    if (cells >= (1ll << 31)) return DPC_ERR_UNSUPPORTED;
    long long g = (cells + 255) / 256;
    if (g > 16384) g = 16384;
    if (!s2d || dtype_s2d == DPC_F32) {
        DPC_LAUNCH((frames_to_input_kernel<float>), dim3((unsigned)g), dim3(256), stream, frames, B, F, H0, W0, aug, gray, N, SL, ds, H, W, xtab, ytab,
                   crop_w, mean3[0], mean3[1], mean3[2], std3[0], std3[1], std3[2], block, (float*)s2d);
    } else if (dtype_s2d == DPC_BF16) {
        DPC_LAUNCH((frames_to_input_kernel<bf16_t>), dim3((unsigned)g), dim3(256), stream, frames, B, F, H0, W0, aug, gray, N, SL, ds, H, W, xtab, ytab,
                   crop_w, mean3[0], mean3[1], mean3[2], std3[0], std3[1], std3[2], block, (bf16_t*)s2d);
    } else {
        return DPC_ERR_ARG;
    }
    return dpc_launch_status();
}
