// dpc_rt.h -- the one place that knows which toolchain is compiling the kernels.
//
// Product build: hipcc --offload-arch=gfx950 (real CDNA4 code, MFMA builtins).
// Test build   : host clang++ with -DDPC_SIMT_EMU (tests/simt_emu, CPU functional
//                simulator used only by the `-m "not gpu"` test tier).
// Kernel sources are written once, for gfx950; nothing below is a CUDA shim.
#pragma once
#include <stdint.h>
#include <utility>

// ---- CU carve-out for a concurrent collective (dpc_set_reserved_cus, csrc/plan.hip): the kernels that launch exactly one
// persistent workgroup per CU (conv_halo_ws, igemm_ws / igemm_wsp) shrink their grid by this many workgroups, so that RCCL's
// channel kernels find free CUs while the gradient all-reduce overlaps the rest of the backward pass.  0 by default.
int dpc_reserved_cus();
static inline int dpc_persistent_grid(int base) {   // a multiple of 8 (XCD-grouped tile slots), at least 8
    int g = base - dpc_reserved_cus();
    if (g < 8) g = 8;
    return base >= 8 ? (g & ~7) : base;
}

// ---- kernel-selection trace (dpc_conv_plan / dpc_last_kernel, csrc/plan.hip) --------------------------------------------
// Every launch site names the kernel it selected (the stringified kernel expression of DPC_LAUNCH, plus whatever the
// dispatcher adds with dpc_plan_detail: element types, tile shape, padded grid ...).  In plan-only mode (set by dpc_conv_plan
// around a call of the very same dispatch code with dummy pointers) the launch itself is skipped, so the query can never
// disagree with what a real call does.  Thread-local: concurrent callers do not see each other's trace.
extern thread_local int dpc_tls_plan_only;
void dpc_plan_note(const char* kernel_expr);
void dpc_plan_detail(const char* fmt, ...);

#ifdef DPC_SIMT_EMU
#include "simt_emu.h"
#define DPC_LAUNCH(kernel, grid, block, stream, ...)                                   \
    do {                                                                               \
        dpc_plan_note(#kernel);                                                        \
        if (!dpc_tls_plan_only) simt::launch((grid), (block), [=]() { (kernel)(__VA_ARGS__); }); \
    } while (0)
#define DPC_LAUNCH_DYN(kernel, grid, block, lds, stream, ...)                          \
    do {                                                                               \
        dpc_plan_note(#kernel);                                                        \
        if (!dpc_tls_plan_only) simt::launch_dyn((grid), (block), (lds), [=]() { (kernel)(__VA_ARGS__); }); \
    } while (0)
#define DPC_DYN_SMEM(name) unsigned char* name = simt::dyn_smem
#define DPC_UNROLL
#define DPC_NOUNROLL
#else
#include <hip/hip_runtime.h>
#define DPC_LAUNCH(kernel, grid, block, stream, ...)                                   \
    do {                                                                               \
        dpc_plan_note(#kernel);                                                        \
        if (!dpc_tls_plan_only) hipLaunchKernelGGL(kernel, (grid), (block), 0, (stream), __VA_ARGS__); \
    } while (0)
#define DPC_LAUNCH_DYN(kernel, grid, block, lds, stream, ...)                          \
    do {                                                                               \
        dpc_plan_note(#kernel);                                                        \
        if (!dpc_tls_plan_only) hipLaunchKernelGGL(kernel, (grid), (block), (lds), (stream), __VA_ARGS__); \
    } while (0)
#define DPC_DYN_SMEM(name) extern __shared__ __attribute__((aligned(16))) unsigned char name[]
#define DPC_UNROLL _Pragma("unroll")
#define DPC_NOUNROLL _Pragma("unroll 1")
#endif

// fast transcendental forms for gate math (gfx950: v_exp_f32 / v_rcp_f32, ~1 ulp); exact libm in the host simulator
#ifdef DPC_SIMT_EMU
static inline float fast_exp(float x) { return std::exp(x); }
static inline float fast_rcp(float x) { return 1.f / x; }
static inline float fast_exp2(float x) { return std::exp2(x); }
#else
__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }  // one v_exp_f32: no denormal fix-up
__device__ __forceinline__ float fast_exp(float x) { return __expf(x); }
__device__ __forceinline__ float fast_rcp(float x) { return __frcp_rn(x); }
#endif

#define DPC_OK 0
#define DPC_ERR_ARG (-1)
#define DPC_ERR_LAUNCH (-2)
#define DPC_ERR_UNSUPPORTED (-3)

static inline int dpc_launch_status() {
    if (dpc_tls_plan_only) return DPC_OK;  // nothing was launched (and the query must work without a device)
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? DPC_OK : DPC_ERR_LAUNCH;
}

// ---------------------------------------------------------------- element types
// Activations/operands are either f32 (parity mode) or bf16 (throughput mode); bf16
// is carried as its raw 16-bit pattern.
typedef uint16_t bf16_t;
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

__device__ __host__ __forceinline__ float bf16_to_f32(bf16_t h) {
    union { uint32_t u; float f; } c;
    c.u = (uint32_t)h << 16;
    return c.f;
}
// round-to-nearest-even (NaN kept quiet); matches torch's float->bfloat16.  On gfx950 this is the
// hardware v_cvt_pk_bf16_f32 (one instruction for two values); the arithmetic form below is what the
// host-side simulator build executes.
__device__ __host__ __forceinline__ bf16_t f32_to_bf16_sw(float f) {
    union { uint32_t u; float f; } c;
    c.f = f;
    uint32_t u = c.u;
    if ((u & 0x7fffffffu) > 0x7f800000u) return (bf16_t)((u >> 16) | 0x40u);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (bf16_t)(u >> 16);
}
#if defined(__HIP_DEVICE_COMPILE__) && !defined(DPC_SIMT_EMU)
__device__ __forceinline__ bf16_t f32_to_bf16(float f) { return __builtin_bit_cast(bf16_t, (__bf16)f); }
__device__ __forceinline__ uint32_t bf16x2_pack(float lo, float hi) {
    typedef __bf16 bf2_t __attribute__((ext_vector_type(2)));
    typedef float f2_t __attribute__((ext_vector_type(2)));
    const f2_t v = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf2_t));
}
#else
__device__ __host__ __forceinline__ bf16_t f32_to_bf16(float f) { return f32_to_bf16_sw(f); }
__device__ __host__ __forceinline__ uint32_t bf16x2_pack(float lo, float hi) {
    return (uint32_t)f32_to_bf16_sw(lo) | ((uint32_t)f32_to_bf16_sw(hi) << 16);
}
#endif

template <class T> struct Elt;
template <> struct Elt<float> {
    static constexpr int PER16 = 4;  // elements per 16-byte unit
    __device__ __host__ static __forceinline__ float to_f32(float v) { return v; }
    __device__ __host__ static __forceinline__ float from_f32(float v) { return v; }
};
template <> struct Elt<bf16_t> {
    static constexpr int PER16 = 8;
    __device__ __host__ static __forceinline__ float to_f32(bf16_t v) { return bf16_to_f32(v); }
    __device__ __host__ static __forceinline__ bf16_t from_f32(float v) { return f32_to_bf16(v); }
};

// unpack element e of a 16-byte unit
template <class T> __device__ __forceinline__ float unit_get(const u32x4& u, int e);
template <> __device__ __forceinline__ float unit_get<float>(const u32x4& u, int e) {
    union { uint32_t u; float f; } c;
    c.u = u[e];
    return c.f;
}
template <> __device__ __forceinline__ float unit_get<bf16_t>(const u32x4& u, int e) {
    uint32_t w = u[e >> 1];
    return bf16_to_f32((bf16_t)((e & 1) ? (w >> 16) : (w & 0xffffu)));
}
template <class T> __device__ __forceinline__ void unit_set(u32x4& u, int e, float v);
template <> __device__ __forceinline__ void unit_set<float>(u32x4& u, int e, float v) {
    union { uint32_t u; float f; } c;
    c.f = v;
    u[e] = c.u;
}
template <> __device__ __forceinline__ void unit_set<bf16_t>(u32x4& u, int e, float v) {
    uint32_t h = f32_to_bf16(v);
    uint32_t w = u[e >> 1];
    u[e >> 1] = (e & 1) ? ((w & 0x0000ffffu) | (h << 16)) : ((w & 0xffff0000u) | h);
}

// pack E floats into one 16-byte unit (bf16: four v_cvt_pk_bf16_f32)
template <class T> __device__ __forceinline__ u32x4 unit_pack(const float* v);
template <> __device__ __forceinline__ u32x4 unit_pack<float>(const float* v) {
    u32x4 u;
    DPC_UNROLL
    for (int e = 0; e < 4; ++e) {
        union { uint32_t u; float f; } c;
        c.f = v[e];
        u[e] = c.u;
    }
    return u;
}
template <> __device__ __forceinline__ u32x4 unit_pack<bf16_t>(const float* v) {
    u32x4 u;
    DPC_UNROLL
    for (int q = 0; q < 4; ++q) u[q] = bf16x2_pack(v[2 * q], v[2 * q + 1]);
    return u;
}

// ---------------------------------------------------------------- exact division by a runtime constant
// q = x / d for 0 <= x < 2^31 (Granlund-Montgomery, p = 31 + ceil(log2 d)); built on the host.
struct FastDiv {
    uint32_t mul, shift, d, pad;
};
static inline FastDiv make_fastdiv(uint32_t d) {
    FastDiv f;
    uint32_t l = 0;
    while ((1ull << l) < d) ++l;
    f.shift = 31 + l;
    f.mul = (uint32_t)(((1ull << f.shift) + d - 1) / d);
    f.d = d;
    f.pad = 0;
    return f;
}
__device__ __host__ __forceinline__ uint32_t fdiv(uint32_t x, const FastDiv& f) {
    return (uint32_t)(((unsigned long long)x * f.mul) >> f.shift);
}

// ---------------------------------------------------------------- MFMA wrappers (gfx950)
// 32x32 tiles; lane l supplies A[i=l&31][k-group=l>>5] and B[k-group][j=l&31];
// C/D: col=l&31, row=(r&3)+8*(r>>2)+4*(l>>5).
__device__ __forceinline__ f32x16 mfma_32x32x2_f32(float a, float b, f32x16 c) {
#ifdef DPC_SIMT_EMU
    return simt_mfma_f32_32x32x2f32(a, b, c);
#else
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
#endif
}
// a,b: 16 bytes = 8 bf16 (k = (l>>5)*8 + j)
__device__ __forceinline__ f32x16 mfma_32x32x16_bf16(const u32x4& a, const u32x4& b, f32x16 c) {
#ifdef DPC_SIMT_EMU
    simt_bf16x8 av, bv;
    std::memcpy(&av, &a, 16);
    std::memcpy(&bv, &b, 16);
    return simt_mfma_f32_32x32x16_bf16(av, bv, c);
#else
    typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
#endif
}

// one 16-byte unit of operands -> the MFMA k-steps it feeds (f32: 4 steps of K=2; bf16: 1 step of K=16)
template <class T> __device__ __forceinline__ f32x16 mfma_unit(const u32x4& a, const u32x4& b, f32x16 c);
template <> __device__ __forceinline__ f32x16 mfma_unit<float>(const u32x4& a, const u32x4& b, f32x16 c) {
    DPC_UNROLL
    for (int s = 0; s < 4; ++s) c = mfma_32x32x2_f32(unit_get<float>(a, s), unit_get<float>(b, s), c);
    return c;
}
template <> __device__ __forceinline__ f32x16 mfma_unit<bf16_t>(const u32x4& a, const u32x4& b, f32x16 c) {
    return mfma_32x32x16_bf16(a, b, c);
}

// ---- "bf16x6": f32 operands on the bf16 matrix pipe at f32-grade accuracy (round 4; profiles/r04_bf16x3_feasibility.txt).
// x = x1 + x2 + x3 with x1 = bf16(x), x2 = bf16(x - x1), x3 = bf16(x - x1 - x2) (24 mantissa bits in three pieces); a product
// keeps the six terms a_i b_j with i + j <= 4 -- everything down to 2^-24 |a||b| -- accumulated in f32, smallest first.
// Six v_mfma_f32_32x32x16_bf16 (32 cycles each) replace eight v_mfma_f32_32x32x2_f32 (64 cycles each) per K = 16: 2.7x the f32
// matrix rate.  Measured against the reference's fp32 scores on the CPU model of this arithmetic: 1.1e-4 .. 1.7e-4 (the two-piece
// "bf16x3" misses north_star's 1e-3: 1.0e-3 .. 2.5e-3).  Two 16-byte units (8 f32, K index = the element order) per operand.
struct Split3 {
    u32x4 p[3];
};
__device__ __forceinline__ Split3 split3_f32x8(const u32x4& u0, const u32x4& u1) {
    Split3 s;
    DPC_UNROLL
    for (int q = 0; q < 4; ++q) {
        const float x0 = __builtin_bit_cast(float, q < 2 ? u0[2 * q] : u1[2 * q - 4]);
        const float x1 = __builtin_bit_cast(float, q < 2 ? u0[2 * q + 1] : u1[2 * q - 3]);
        const uint32_t h = bf16x2_pack(x0, x1);
        const float r0 = x0 - __builtin_bit_cast(float, h << 16), r1 = x1 - __builtin_bit_cast(float, h & 0xffff0000u);
        const uint32_t m = bf16x2_pack(r0, r1);
        const float t0 = r0 - __builtin_bit_cast(float, m << 16), t1 = r1 - __builtin_bit_cast(float, m & 0xffff0000u);
        s.p[0][q] = h;
        s.p[1][q] = m;
        s.p[2][q] = bf16x2_pack(t0, t1);
    }
    return s;
}
__device__ __forceinline__ Split3 split3_f32(const float (&x)[8]) {   // the same from eight loose values (K index = array index)
    const u32x4 u0 = {__builtin_bit_cast(uint32_t, x[0]), __builtin_bit_cast(uint32_t, x[1]), __builtin_bit_cast(uint32_t, x[2]), __builtin_bit_cast(uint32_t, x[3])};
    const u32x4 u1 = {__builtin_bit_cast(uint32_t, x[4]), __builtin_bit_cast(uint32_t, x[5]), __builtin_bit_cast(uint32_t, x[6]), __builtin_bit_cast(uint32_t, x[7])};
    return split3_f32x8(u0, u1);
}
__device__ __forceinline__ f32x16 mfma_f32x6(const Split3& a, const Split3& b, f32x16 c) {
    c = mfma_32x32x16_bf16(a.p[2], b.p[0], c);
    c = mfma_32x32x16_bf16(a.p[0], b.p[2], c);
    c = mfma_32x32x16_bf16(a.p[1], b.p[1], c);
    c = mfma_32x32x16_bf16(a.p[1], b.p[0], c);
    c = mfma_32x32x16_bf16(a.p[0], b.p[1], c);
    return mfma_32x32x16_bf16(a.p[0], b.p[0], c);
}
// process-wide switch (dpc_set_f32_matmul, csrc/plan.hip): 0 = exact f32 MFMA chains, 1 = bf16x6
int dpc_f32_matmul_mode();

// LDS tile rows are 128 bytes = 8 units of 16 B; unit u of row r lives at slot u ^ swz(r).
// swz1(r) = (r>>1)&7: a ds_read_b128 lane group (16 distinct rows, same logical unit) touches all 64
//   banks once, and swz1(r+32) == swz1(r) so +32-row fragments are immediate offsets (conv_igemm.hip).
// swz3(r) = ((r>>1)&7) ^ ((r>>4)&3): same read property AND the transposing ds_write_b128 of
//   conv_wgrad.hip (8 lanes -> rows 8 or 4 apart, same unit) is conflict-free (brute-force checked
//   against the lane groups of MI355X_MICROARCH.md; measured SQ_LDS_BANK_CONFLICT = 0 for swz1).
__device__ __forceinline__ int lds_swz1(int row) { return (row >> 1) & 7; }
__device__ __forceinline__ int lds_swz3(int row) { return ((row >> 1) & 7) ^ ((row >> 4) & 3); }
__device__ __forceinline__ int lds_unit_off(int row, int unit) { return row * 128 + (((unit ^ lds_swz1(row)) & 7) << 4); }
__device__ __forceinline__ int lds_unit_off3(int row, int unit) { return row * 128 + (((unit ^ lds_swz3(row)) & 7) << 4); }

// 16 bytes of zeros in device memory: the source of every padded / out-of-range operand unit, so the
// prefetch needs neither branches nor a post-load mask.
#ifdef DPC_SIMT_EMU
static const uint32_t dpc_zero16[4] __attribute__((aligned(16))) = {0u, 0u, 0u, 0u};
#else
static __device__ const uint32_t dpc_zero16[4] __attribute__((aligned(16))) = {0u, 0u, 0u, 0u};
#endif

// LDS-DMA (gfx950 global_load_lds_dwordx4): each lane's 16 global bytes land at
// wave_base + lane*16 in LDS without passing through VGPRs; asynchronous until the next
// s_waitcnt vmcnt / __syncthreads().  wave_base must be wave-uniform.
__device__ __forceinline__ void glds16(const void* gsrc, unsigned char* lds_wave_base, int lane) {
#ifdef DPC_SIMT_EMU
    simt::dma_issue(lds_wave_base + lane * 16, gsrc);   // lands at the wait that retires it (tests/simt_emu/simt_emu.h)
#else
    (void)lane;
    __builtin_amdgcn_global_load_lds(gsrc, (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
#endif
}

// Workgroup barrier that orders LDS traffic only: unlike __syncthreads() it does NOT wait for
// outstanding vector-memory operations, so an LDS-DMA prefetch issued earlier keeps flying across it
// (cdna_hip_programming.md §5 "Pipelining across barriers").  Use only where no global data produced
// by other waves is consumed after the barrier.
__device__ __forceinline__ void barrier_lds_only() {
#ifdef DPC_SIMT_EMU
    simt::sync_block();
#else
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#endif
}

// The same barrier with the wait as a compiler-visible instruction.  For a wave whose last LDS operations before the barrier are
// compiler-generated stores (a staging tile written from accumulator registers): behind the asm form hipcc still counts those stores
// as outstanding and puts its own s_waitcnt lgkmcnt(0) in front of the first instruction that overwrites their source registers --
// the first MFMA of the next tile, i.e. behind the hand-issued fragment reads that were meant to stay in flight under it.
__device__ __forceinline__ void barrier_lds_only_tracked() {
#ifdef DPC_SIMT_EMU
    simt::sync_block();
#else
    __builtin_amdgcn_s_waitcnt(0xC07F);   // lgkmcnt(0), vmcnt / expcnt untouched
    asm volatile("s_barrier" ::: "memory");
#endif
}

// LDS transpose read (gfx950 ds_read_b64_tr_b16): see tests/simt_emu/simt_emu.h for the lane map.
// Lane l of a 16-lane group passes the address of row (l>>2), columns 4*(l&3).. of a 4 x 16 block of
// 16-bit elements and receives column (l&15) of that block (4 elements, row order).
__device__ __forceinline__ u32x2 lds_read_tr16(const unsigned char* lane_addr) {
#ifdef DPC_SIMT_EMU
    const uint64_t v = simt_ds_read_tr16_b64(lane_addr);
    u32x2 r = {(uint32_t)v, (uint32_t)(v >> 32)};
    return r;
#else
    typedef short v4i16_t __attribute__((ext_vector_type(4)));
    const v4i16_t v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4i16_t*)lane_addr);
    return __builtin_bit_cast(u32x2, v);
#endif
}

// Ordering point for LDS traffic that stays inside ONE wave (write a wave-private region, read it back
// with another lane mapping).  The hardware executes a wave's LDS instructions in order and the compiler
// places the lgkmcnt wait; the host simulator runs lanes as fibers and needs the explicit rendezvous.
__device__ __forceinline__ void wave_lds_fence() {
#ifdef DPC_SIMT_EMU
    simt::sync_wave();
#else
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#endif
}

// LDS-DMA through a buffer resource (buffer_load_dwordx4 ... offen lds): like glds16, but the source is
// base + voff + soff with a 32-bit per-lane offset and a wave-uniform scalar offset, and a lane whose
// offset falls outside [0, nbytes) writes 16 zero bytes to LDS (probed on MI355X, scripts/probes/buflds.*):
// padding and out-of-range rows cost one v_cndmask instead of a 64-bit address select.
#ifdef DPC_SIMT_EMU
struct BufRsrc {
    const char* base;
    uint32_t nbytes;
};
static inline BufRsrc make_buf_rsrc(const void* p, uint32_t nbytes) {
    BufRsrc r = {(const char*)p, nbytes};
    return r;
}
static inline void glds16_buf(const BufRsrc& r, uint32_t voff, uint32_t soff, unsigned char* lds_wave_base, int lane) {
    const uint64_t o = (uint64_t)voff + soff;
    simt::dma_issue(lds_wave_base + lane * 16, (voff < r.nbytes && o + 16 <= r.nbytes) ? r.base + o : nullptr);
}
#else
typedef __amdgpu_buffer_rsrc_t BufRsrc;
__device__ __forceinline__ BufRsrc make_buf_rsrc(const void* p, uint32_t nbytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)nbytes, 0x00020000);
}
__device__ __forceinline__ void glds16_buf(const BufRsrc& r, uint32_t voff, uint32_t soff, unsigned char* lds_wave_base, int lane) {
    (void)lane;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)lds_wave_base, 16, (int)voff, (int)soff, 0, 0);
}
#endif
#define DPC_BUF_OOB 0x80000000u

// ---- hand-scheduled LDS fragment reads ------------------------------------------------------------
// hipcc serialises "ds_read -> s_waitcnt lgkmcnt(0) -> MFMA" per K step when the fragment registers are
// reused (and waits for ALL outstanding reads when they are not), which exposes one LDS round trip per
// step.  These reads are invisible to its wait-count pass; the caller counts: LDS operations of a wave
// complete in order, so after lds_wait_tie<N> every read except the newest N has landed.  The "+v" ties
// make later uses of the registers depend on the wait.  (Scalar loads share the counter but only make
// the wait more conservative.)
#ifdef DPC_SIMT_EMU
__device__ __forceinline__ void lds_read_b128_async(u32x4& d, const unsigned char* p) { d = *(const u32x4*)p; }
template <int OFF> __device__ __forceinline__ void lds_read_b128_async_off(u32x4& d, const unsigned char* p) { d = *(const u32x4*)(p + OFF); }
template <int N> __device__ __forceinline__ void lds_wait_tie(u32x4&, u32x4&) {}
__device__ __forceinline__ void sched_fence() {}
#else
// same with the constant part of the address as the instruction's 16-bit offset field (no VALU, no extra VGPR)
template <int OFF> __device__ __forceinline__ void lds_read_b128_async_off(u32x4& d, const unsigned char* p) {
    static_assert(OFF >= 0 && OFF < 65536, "ds offset field");
    const uint32_t a = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const unsigned char*)p;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=&v"(d) : "v"(a), "n"(OFF) : "memory");
}
__device__ __forceinline__ void lds_read_b128_async(u32x4& d, const unsigned char* p) {
    const uint32_t a = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const unsigned char*)p;
    asm volatile("ds_read_b128 %0, %1" : "=&v"(d) : "v"(a) : "memory");
}
template <int N> __device__ __forceinline__ void lds_wait_tie(u32x4& a, u32x4& b) {
    asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(a), "+v"(b) : "n"(N) : "memory");
}
// instruction-scheduler fence: nothing is moved across it (no instruction is emitted)
__device__ __forceinline__ void sched_fence() { __builtin_amdgcn_sched_barrier(0); }
#endif
// LDS accesses the compiler must not see as such: hipcc orders every LDS access it knows about after ALL outstanding LDS-DMA
// (s_waitcnt vmcnt(0) in front of it -- it cannot tell the DMA's destination from the address it reads), which drains a prefetch
// that is meant to stay in flight for several chunks.  The caller guarantees (counted vmcnt + barrier) that what it touches has
// landed.  lds_wait0_*: s_waitcnt lgkmcnt(0) tied to the registers the preceding raw reads wrote.
#ifdef DPC_SIMT_EMU
__device__ __forceinline__ void lds_read_b64_raw(u32x2& d, const unsigned char* p) { d = *(const u32x2*)p; }
__device__ __forceinline__ void lds_read_b128_raw(u32x4& d, const unsigned char* p) { d = *(const u32x4*)p; }
__device__ __forceinline__ void lds_write_b128_raw(unsigned char* p, const u32x4& v) { *(u32x4*)p = v; }
__device__ __forceinline__ void lds_read_tr16_raw(u32x2& d, const unsigned char* p) { d = lds_read_tr16(p); }
__device__ __forceinline__ void lds_wait0_5(u32x4&, u32x4&, u32x4&, u32x4&, u32x4&) {}
__device__ __forceinline__ void lds_wait0_4x2(u32x2&, u32x2&, u32x2&, u32x2&) {}
__device__ __forceinline__ void lds_wait0_6x2(u32x2&, u32x2&, u32x2&, u32x2&, u32x2&, u32x2&) {}
#else
__device__ __forceinline__ void lds_read_b64_raw(u32x2& d, const unsigned char* p) {
    const uint32_t a = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const unsigned char*)p;
    asm volatile("ds_read_b64 %0, %1" : "=&v"(d) : "v"(a) : "memory");
}
__device__ __forceinline__ void lds_read_b128_raw(u32x4& d, const unsigned char* p) { lds_read_b128_async(d, p); }
__device__ __forceinline__ void lds_write_b128_raw(unsigned char* p, const u32x4& v) {
    const uint32_t a = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)p;
    asm volatile("ds_write_b128 %0, %1" ::"v"(a), "v"(v) : "memory");
}
__device__ __forceinline__ void lds_read_tr16_raw(u32x2& d, const unsigned char* p) {
    const uint32_t a = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const unsigned char*)p;
    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=&v"(d) : "v"(a) : "memory");
}
__device__ __forceinline__ void lds_wait0_5(u32x4& a, u32x4& b, u32x4& c, u32x4& d, u32x4& e) {
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e)::"memory");
}
__device__ __forceinline__ void lds_wait0_4x2(u32x2& a, u32x2& b, u32x2& c, u32x2& d) {
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d)::"memory");
}
__device__ __forceinline__ void lds_wait0_6x2(u32x2& a, u32x2& b, u32x2& c, u32x2& d, u32x2& e, u32x2& f) {
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f)::"memory");
}
#endif
__device__ __forceinline__ void lds_wait_tie_n(int n, u32x4& a, u32x4& b) {  // n is a compile-time value after unrolling
    if (n >= 6) lds_wait_tie<6>(a, b);
    else if (n == 4) lds_wait_tie<4>(a, b);
    else if (n == 2) lds_wait_tie<2>(a, b);
    else lds_wait_tie<0>(a, b);
}



// wait until at most N of this wave's vector-memory operations (LDS-DMA pieces included) are outstanding
template <int N> __device__ __forceinline__ void wait_vmcnt() {
#ifndef DPC_SIMT_EMU
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
#else
    simt::dma_wait(N);
#endif
}
// n stores were just issued that a later counted wait relies on (vmcnt counts stores): nothing on the device, place-holders in the
// simulator's completion queue (tests/simt_emu/simt_emu.h)
__device__ __forceinline__ void vm_note_stores(int n) {
#ifdef DPC_SIMT_EMU
    simt::dma_note_stores(n);
#else
    (void)n;
#endif
}

// wait until at most n (wave-uniform, run-time) vector-memory operations of this wave are outstanding
__device__ __forceinline__ void wait_vmcnt_upto(int n) {
    switch (n) {
        case 0: wait_vmcnt<0>(); break;
        case 1: wait_vmcnt<1>(); break;
        case 2: wait_vmcnt<2>(); break;
        case 3: wait_vmcnt<3>(); break;
        case 4: wait_vmcnt<4>(); break;
        case 5: wait_vmcnt<5>(); break;
        case 6: wait_vmcnt<6>(); break;
        case 7: wait_vmcnt<7>(); break;
        case 8: wait_vmcnt<8>(); break;
        case 9: wait_vmcnt<9>(); break;
        case 10: wait_vmcnt<10>(); break;
        case 11: wait_vmcnt<11>(); break;
        case 12: wait_vmcnt<12>(); break;
        case 13: wait_vmcnt<13>(); break;
        case 14: wait_vmcnt<14>(); break;
        case 15: wait_vmcnt<15>(); break;
        case 16: case 17: wait_vmcnt<16>(); break;   // waiting for fewer outstanding operations is always safe
        case 18: case 19: wait_vmcnt<18>(); break;
        case 20: case 21: wait_vmcnt<20>(); break;
        case 22: case 23: wait_vmcnt<22>(); break;
        default: if (n >= 24) wait_vmcnt<24>(); else wait_vmcnt<0>(); break;   // (n < 0 cannot happen: drain)
    }
}

// compile-time unrolled loop: body(std::integral_constant<int, I>) for I = 0..N-1 (the index is usable as a
// template argument / instruction immediate inside the body)
template <class F, int... Is> __device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, Is...>) {
    (f(std::integral_constant<int, Is>{}), ...);
}
template <int N, class F> __device__ __forceinline__ void static_for(F&& f) {
    static_for_impl(static_cast<F&&>(f), std::make_integer_sequence<int, N>{});
}
