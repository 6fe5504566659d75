// ws_frag.h -- the compute-wave side of the loader / compute kernels (conv_igemm_ws.hip, gemm_ws.hip): fragment reads that hipcc
// does not track, interleaved one per MFMA gap with hand-counted waits, and the packed epilogue staging.  Moved out of
// conv_igemm_ws.hip in round 4 so that the split-K GEMMs of the score backward (dpc/model_3d.py:83's autograd) run on the same
// machinery.  Everything here is about ONE invariant: an asm LDS read's destination registers stay tied ("+v") until a wait has
// retired the read -- scripts/asm_hazard_lint.py checks the compiled code for it.
#pragma once
#include "conv_common.h"

// workgroup barrier that waits for nothing but this wave's LDS traffic
__device__ __forceinline__ void ws_barrier() { barrier_lds_only(); }

// ---- compute-wave fragment pipeline -------------------------------------------------------------
struct FragSet {
    u32x4 a[2], b[4];
};
#ifdef DPC_SIMT_EMU
__device__ __forceinline__ void frag_read(FragSet& f, const unsigned char* st, int off_a, int off_b) {
    for (int i = 0; i < 2; ++i) f.a[i] = *(const u32x4*)(st + off_a + 4096 * i);
    for (int j = 0; j < 4; ++j) f.b[j] = *(const u32x4*)(st + off_b + 4096 * j);
}
template <int N> __device__ __forceinline__ void frag_wait(FragSet&) {}
#else
// six ds_read_b128 the compiler does not track (it would otherwise add its own, coarser waits)
__device__ __forceinline__ void frag_read(FragSet& f, const unsigned char* st, int off_a, int off_b) {
    const uint32_t base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const unsigned char*)st;
    const uint32_t pa = base + (uint32_t)off_a, pb = base + (uint32_t)off_b;
    asm volatile("ds_read_b128 %0, %6\n\t"
                 "ds_read_b128 %1, %6 offset:4096\n\t"
                 "ds_read_b128 %2, %7\n\t"
                 "ds_read_b128 %3, %7 offset:4096\n\t"
                 "ds_read_b128 %4, %7 offset:8192\n\t"
                 "ds_read_b128 %5, %7 offset:12288"
                 : "=&v"(f.a[0]), "=&v"(f.a[1]), "=&v"(f.b[0]), "=&v"(f.b[1]), "=&v"(f.b[2]), "=&v"(f.b[3])
                 : "v"(pa), "v"(pb)
                 : "memory");
}
// wait until at most N of this wave's LDS reads are outstanding; the "+v" ties make every later use of
// the set depend on the wait
template <int N> __device__ __forceinline__ void frag_wait(FragSet& f) {
    asm volatile("s_waitcnt lgkmcnt(%6)"
                 : "+v"(f.a[0]), "+v"(f.a[1]), "+v"(f.b[0]), "+v"(f.b[1]), "+v"(f.b[2]), "+v"(f.b[3])
                 : "n"(N)
                 : "memory");
}
#endif
// THE HAZARD THAT LOOKED LIKE AN LDS RACE (round 3: "one wave's 64 x 128 block wrong in 38 of 15 000 launches beside a foreign
// LDS-using workgroup").  The last step_il of a tile issues the six fragment reads of a chunk that does not exist (unconditional,
// so that the MFMAs stay out of a branch).  Nothing consumes them, so for hipcc the six destination registers are DEAD at
// ;;#ASMEND -- and it gave two of them (v160/v161 in <true,false>) to the epilogue's LDS staging address.  The asm reads are
// invisible to its wait counts: if a0' (issued seven MFMAs, ~250 cycles, before the loop exit) has not landed when the
// `v_add_u32 v161` executes, the LDS data lands ON TOP of the address, stage_block's 32 ds_write_b64 go to whatever LDS offsets
// the bf16 pattern spells (dropped when out of range, otherwise into a stage the loaders are filling: "two rows of another tile"),
// and both epilogue passes store a stale staging tile: one wave's whole 64 x 128 block.  An LDS round trip is ~64-128 cycles on a
// quiet CU, so it never showed alone; a co-resident workgroup (LDS traffic, issue slots) stretches it past the seven MFMAs once in
// a few hundred launches.  Claiming the whole LDS removed one way of stretching it, not the hazard.  The rule (cdna_hip_programming.md
// "What hipcc does not do", 1): an asm load's destination must stay live, by a "+v" tie, until a wait has retired the load.
// scripts/asm_hazard_lint.py checks the compiled code for exactly this (tests/test_asm_hazards.py); DPC_WS_NOFIX re-creates the
// hazard in the probe library for the squatter repro (scripts/probes/squat_probe.py, profiles/r04_cotenant.txt).
#ifdef DPC_WS_NOFIX
#define WS_RETIRE_TAIL_READS(f) ((void)0)
#else
#define WS_RETIRE_TAIL_READS(f) frag_wait<0>(f)
#endif
__device__ __forceinline__ void mma_step(f32x16 (&acc)[2][4], const FragSet& f) {
    DPC_UNROLL
    for (int i = 0; i < 2; ++i)
        DPC_UNROLL
        for (int j = 0; j < 4; ++j) acc[i][j] = mfma_32x32x16_bf16(f.b[j], f.a[i], acc[i][j]);   // transposed block, see stage_block
}

// LDS addresses as integers (32-bit on the device: pointer arithmetic on generic pointers drags an address-space cast with a
// null check into every step; the simulator keeps host addresses)
#ifdef DPC_SIMT_EMU
typedef uintptr_t ldsa_t;
static inline ldsa_t ldsa(const void* p) { return (uintptr_t)p; }
#else
typedef uint32_t ldsa_t;
__device__ __forceinline__ ldsa_t ldsa(const void* p) { return (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const void*)p; }
#endif
// the six fragment reads of one K step on their own (first step of a tile)
template <int AOFF>
__device__ __forceinline__ void frag_read_p2(FragSet& f, ldsa_t xa, ldsa_t xb) {
#ifdef DPC_SIMT_EMU
    f.a[0] = *(const u32x4*)xa;
    f.a[1] = *(const u32x4*)(xa + AOFF);
    for (int j = 0; j < 4; ++j) f.b[j] = *(const u32x4*)(xb + 4096 * j);
#else
    asm volatile("ds_read_b128 %0, %6\n\t"             // same order as step_il: a0 b0 b1 a1 b2 b3
                 "ds_read_b128 %2, %7\n\t"
                 "ds_read_b128 %3, %7 offset:4096\n\t"
                 "ds_read_b128 %1, %6 offset:%8\n\t"
                 "ds_read_b128 %4, %7 offset:8192\n\t"
                 "ds_read_b128 %5, %7 offset:12288"
                 : "=&v"(f.a[0]), "=&v"(f.a[1]), "=&v"(f.b[0]), "=&v"(f.b[1]), "=&v"(f.b[2]), "=&v"(f.b[3])
                 : "v"(xa), "v"(xb), "n"(AOFF)
                 : "memory");
#endif
}
// One K step of a compute wave with the NEXT step's six fragment reads interleaved between its eight MFMAs (hipcc otherwise
// issues the reads as a burst in front of the MFMAs, and the matrix pipe idles while they issue -- probe: +61 us of 400 on
// layer2, scripts/probes/ws_probe.py).  `use` must have landed (caller waits); `ld` is written.
// xa: LDS address of the first A fragment (the second is AOFF bytes further); xb: of the first B fragment (+4096 per column block).
template <bool LOAD, int AOFF>
__device__ __forceinline__ void step_il(f32x16 (&acc)[2][4], const FragSet& use, FragSet& ld, ldsa_t xa, ldsa_t xb) {
#ifdef DPC_SIMT_EMU
    if (LOAD) {
        ld.a[0] = *(const u32x4*)xa;
        ld.a[1] = *(const u32x4*)(xa + AOFF);
        for (int j = 0; j < 4; ++j) ld.b[j] = *(const u32x4*)(xb + 4096 * j);
    }
    mma_step(acc, use);
#else
    // MFMAs stay compiler intrinsics (it allocates the accumulators and knows the matrix-pipe hazards); the reads are single
    // untracked ds_read_b128 statements, and a scheduling barrier after every instruction pins the order written here.
    // Counted waits: LDS operations of a wave complete in order.  On entry the only reads that may be outstanding are the six
    // of `use`, issued in the order a0 b0 b1 a1 b2 b3 (the order below, one per MFMA gap of the previous step); every MFMA
    // waits for exactly the operand it is the first to need, so each read has six MFMA gaps (~190 cycles) to land.
#define DPC_IL_READ(dst, addr, off) \
    if (LOAD) { asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off) : "memory"); } \
    __builtin_amdgcn_sched_barrier(0)
#define DPC_IL_WAIT(n) \
    asm volatile("s_waitcnt lgkmcnt(" #n ")" ::: "memory"); \
    __builtin_amdgcn_sched_barrier(0)
#define DPC_IL_MMA(i, j) \
    acc[i][j] = mfma_32x32x16_bf16(use.b[j], use.a[i], acc[i][j]); \
    __builtin_amdgcn_sched_barrier(0)
    __builtin_amdgcn_sched_barrier(0);
    DPC_IL_WAIT(4); DPC_IL_MMA(0, 0); DPC_IL_READ(ld.a[0], xa, 0);       // needs a0 b0; outstanding after: b1 a1 b2 b3 | a0'
    DPC_IL_WAIT(4); DPC_IL_MMA(0, 1); DPC_IL_READ(ld.b[0], xb, 0);       // needs b1
    DPC_IL_WAIT(4); DPC_IL_MMA(1, 0); DPC_IL_READ(ld.b[1], xb, 4096);    // needs a1
    DPC_IL_MMA(1, 1); DPC_IL_READ(ld.a[1], xa, AOFF);
    DPC_IL_WAIT(5); DPC_IL_MMA(0, 2); DPC_IL_READ(ld.b[2], xb, 8192);    // needs b2
    DPC_IL_MMA(1, 2); DPC_IL_READ(ld.b[3], xb, 12288);
    DPC_IL_WAIT(6); DPC_IL_MMA(0, 3);                                     // needs b3
    DPC_IL_MMA(1, 3);
#undef DPC_IL_READ
#undef DPC_IL_WAIT
#undef DPC_IL_MMA
#endif
}

// The MFMAs are issued with the operands swapped (weights as the first operand): acc[i][j] then holds the TRANSPOSED 32 x 32
// block -- a lane owns ONE tile row (l & 31) and four consecutive output columns per register quad -- so the epilogue stages a
// quad as one packed 8-byte LDS write (16 per 32-row pass instead of 64 two-byte writes; 37 of 347 us on layer2 were staging).
// Staging rows are 272 bytes apart: the 16 lanes of a ds_write_b64 group then hit 16 disjoint bank pairs, and the row reads
// (ds_read_b128, 4 rows per instruction) stay conflict-free.
constexpr int WS_STG_ROW = 272, WS_STG_WAVE = 32 * WS_STG_ROW;
__device__ __forceinline__ void stage_block(unsigned char* mine, const f32x16 (&acc)[4], int l31, int lhi) {
    DPC_UNROLL
    for (int j = 0; j < 4; ++j)
        DPC_UNROLL
        for (int k = 0; k < 4; ++k) {
            u32x2 v = {bf16x2_pack(acc[j][4 * k], acc[j][4 * k + 1]), bf16x2_pack(acc[j][4 * k + 2], acc[j][4 * k + 3])};
            *(u32x2*)(mine + l31 * WS_STG_ROW + (j * 32 + 8 * k + 4 * lhi) * 2) = v;
        }
}

