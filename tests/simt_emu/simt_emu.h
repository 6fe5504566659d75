// simt_emu.h -- host-side functional simulator of the gfx950 execution model.
//
// TEST INFRASTRUCTURE ONLY (never linked into dpc_amd/libdpc_hip.so, never
// loaded by the product path).  It lets the `-m "not gpu"` test tier execute
// the *same* kernel sources that hipcc compiles for gfx950, on the CPU, so that
// indexing, tiling, barrier placement and MFMA fragment bookkeeping are checked
// against the oracle without a GPU:
//   * one fiber per work-item, 64-lane wavefronts, workgroups run one at a time;
//   * __syncthreads() / wave collectives are cooperative yield points;
//   * __shfl*, MFMA 32x32x2 f32 and 32x32x16 bf16 follow the lane->element
//     maps of /opt/skills/guides/cdna_hip_programming.md §3 (A[i=l&31][k-group=l>>5],
//     C/D col=l&31,row=(r&3)+8*(r>>2)+4*(l>>5));
//   * atomics are sequential.  Nothing here models timing.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>

struct dim3 {
    unsigned x, y, z;
    dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct simt_uint3 { unsigned x, y, z; };

struct float2 { float x, y; };
struct float4 { float x, y, z, w; };
static inline float2 make_float2(float x, float y) { return float2{x, y}; }
static inline float4 make_float4(float x, float y, float z, float w) { return float4{x, y, z, w}; }

typedef void* hipStream_t;
typedef int hipError_t;
#define hipSuccess 0
static inline hipError_t hipGetLastError() { return 0; }
static inline const char* hipGetErrorString(hipError_t) { return "emu"; }
static inline hipError_t hipMemsetAsync(void* p, int v, size_t n, hipStream_t) { std::memset(p, v, n); return 0; }

namespace simt {
struct Fiber;
struct Wave {
    // exchange area for wave collectives (one 16-byte slot x2 per lane)
    alignas(16) unsigned char xa[64][16];
    alignas(16) unsigned char xb[64][16];
};
struct Cur {
    simt_uint3 tid, bid, bdim, gdim;
    int lane, wave;
    Wave* w;
};
extern Cur cur;
void sync_block();
void sync_wave();
void launch(dim3 grid, dim3 block, const std::function<void()>& body);
// ---- LDS-DMA completion model (round 4).  On gfx950 a global_load_lds / buffer_load ... lds piece is asynchronous: its 16 bytes per
// lane land in LDS some time between the instruction and the s_waitcnt vmcnt(N) that retires it (pieces of a wave complete in issue
// order; vmcnt counts the wave's stores too).  The kernels hand-count those waits, and with an eager copy the simulator could not tell a
// correct count from one that is too weak.  Model: dma_issue() reads the source now and queues the piece; dma_wait(leave) lands every
// queued piece except the newest `leave`; __syncthreads() lands everything (it is s_waitcnt vmcnt(0) on the device); whatever is
// still queued when the work-item ends lands then.  A wait that leaves the wrong pieces in flight makes its consumer read stale LDS
// -- deterministically.  dma_note_stores(n) queues n place-holders for stores the kernel's count relies on (score_fused.hip).
// DPC_EMU_DMA=eager restores the immediate copy (the other extreme of what the hardware may do: a piece that lands at once overwrites
// a stage somebody is still reading); the kernel tier runs the LDS-DMA kernels under both.  DPC_EMU_DMA_WEAK=k makes every wait
// leave k more pieces in flight than asked: the positive control of tests/test_ws_emu.py.
void dma_issue(unsigned char* lds_dst, const void* src16);   // src16 == nullptr: 16 zero bytes (out-of-range buffer lane)
void dma_wait(int leave);
void dma_note_stores(int n);
// dynamic LDS: one zero-initialised buffer of `lds_bytes` for the launch (workgroups run one at a time)
extern unsigned char* dyn_smem;
void launch_dyn(dim3 grid, dim3 block, size_t lds_bytes, const std::function<void()>& body);
}  // namespace simt

#define threadIdx (simt::cur.tid)
#define blockIdx (simt::cur.bid)
#define blockDim (simt::cur.bdim)
#define gridDim (simt::cur.gdim)

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __shared__ static
#define __launch_bounds__(...)
#define __restrict__
#define __syncthreads() (simt::dma_wait(0), simt::sync_block())

template <class T>
static inline T simt_shfl_from(T v, int src) {
    static_assert(sizeof(T) <= 16, "shfl payload");
    std::memcpy(simt::cur.w->xa[simt::cur.lane], &v, sizeof(T));
    simt::sync_wave();
    T r;
    std::memcpy(&r, simt::cur.w->xa[src & 63], sizeof(T));
    simt::sync_wave();
    return r;
}
template <class T> static inline T __shfl_xor(T v, int m) { return simt_shfl_from(v, simt::cur.lane ^ m); }
template <class T> static inline T __shfl_down(T v, int d) {
    int s = simt::cur.lane + d;
    return simt_shfl_from(v, s < 64 ? s : simt::cur.lane);
}
template <class T> static inline T __shfl(T v, int src) { return simt_shfl_from(v, src); }

static inline float atomicAdd(float* p, float v) { float o = *p; *p = o + v; return o; }
static inline double atomicAdd(double* p, double v) { double o = *p; *p = o + v; return o; }
static inline int atomicAdd(int* p, int v) { int o = *p; *p = o + v; return o; }
static inline unsigned atomicAdd(unsigned* p, unsigned v) { unsigned o = *p; *p = o + v; return o; }

static inline unsigned __umulhi(unsigned a, unsigned b) { return (unsigned)(((unsigned long long)a * b) >> 32); }

typedef float simt_f32x16 __attribute__((ext_vector_type(16)));

// v_mfma_f32_32x32x2_f32: D = A(32x2) * B(2x32) + C, exact f32 fma chain in k order.
static inline simt_f32x16 simt_mfma_f32_32x32x2f32(float a, float b, simt_f32x16 c) {
    const int l = simt::cur.lane;
    std::memcpy(simt::cur.w->xa[l], &a, 4);
    std::memcpy(simt::cur.w->xb[l], &b, 4);
    simt::sync_wave();
    const int col = l & 31, hi = l >> 5;
    for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * hi;
        float acc = c[r];
        for (int k = 0; k < 2; ++k) {
            float av, bv;
            std::memcpy(&av, simt::cur.w->xa[row + 32 * k], 4);
            std::memcpy(&bv, simt::cur.w->xb[col + 32 * k], 4);
            acc = std::fmaf(av, bv, acc);
        }
        c[r] = acc;
    }
    simt::sync_wave();
    return c;
}

static inline float simt_bf16_to_f32(unsigned short h) {
    unsigned u = (unsigned)h << 16;
    float f;
    std::memcpy(&f, &u, 4);
    return f;
}

// v_mfma_f32_32x32x16_bf16: lane l holds A[i=l&31][k=(l>>5)*8+j], B[k][j=l&31]; f32 accumulate.
struct simt_bf16x8 { unsigned short v[8]; };
static inline simt_f32x16 simt_mfma_f32_32x32x16_bf16(simt_bf16x8 a, simt_bf16x8 b, simt_f32x16 c) {
    const int l = simt::cur.lane;
    std::memcpy(simt::cur.w->xa[l], &a, 16);
    std::memcpy(simt::cur.w->xb[l], &b, 16);
    simt::sync_wave();
    const int col = l & 31, hi = l >> 5;
    for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * hi;
        float acc = c[r];
        for (int kg = 0; kg < 2; ++kg) {
            simt_bf16x8 av, bv;
            std::memcpy(&av, simt::cur.w->xa[row + 32 * kg], 16);
            std::memcpy(&bv, simt::cur.w->xb[col + 32 * kg], 16);
            for (int j = 0; j < 8; ++j) acc = std::fmaf(simt_bf16_to_f32(av.v[j]), simt_bf16_to_f32(bv.v[j]), acc);
        }
        c[r] = acc;
    }
    simt::sync_wave();
    return c;
}

// ds_read_b64_tr_b16 (semantics probed on MI355X, scripts/probes/tr16.*): every lane supplies the
// address of 8 bytes (4 x 16-bit); inside each 16-lane group the 16 x 4 values form a 4 x 16 matrix
// M[r][c] with M[r][4q..4q+3] = the data of source lane 4r+q; lane i receives column i: elem j = M[j][i].
static inline uint64_t simt_ds_read_tr16_b64(const unsigned char* lane_addr) {
    const int l = simt::cur.lane;
    std::memcpy(simt::cur.w->xa[l], lane_addr, 8);
    simt::sync_wave();
    const int g = l >> 4, i = l & 15;
    unsigned short out[4];
    for (int j = 0; j < 4; ++j) {
        unsigned short v[4];
        std::memcpy(v, simt::cur.w->xa[g * 16 + 4 * j + (i >> 2)], 8);
        out[j] = v[i & 3];
    }
    simt::sync_wave();
    uint64_t r;
    std::memcpy(&r, out, 8);
    return r;
}
