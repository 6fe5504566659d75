// simt_emu.cpp -- fiber scheduler for the host-side SIMT simulator (see simt_emu.h).
// TEST INFRASTRUCTURE ONLY.
#include "simt_emu.h"

#include <vector>

namespace simt {

Cur cur;

enum State { READY = 0, WAIT_WAVE = 1, WAIT_BLOCK = 2, DONE = 3 };

struct Fiber {
    void* sp;
    State st;
    simt_uint3 tid;
    int lane, wave;
};

static constexpr size_t STACK_BYTES = 256 * 1024;
static constexpr int MAX_THREADS = 1024;

static std::vector<Fiber> fibers;
static std::vector<Wave> waves;
static unsigned char* stacks = nullptr;
static void* sched_sp = nullptr;
static Fiber* running = nullptr;
static const std::function<void()>* body_fn = nullptr;

// x86-64 SysV context switch: save callee-saved regs on the current stack, swap sp.
extern "C" void simt_switch(void** save_sp, void* new_sp);
asm(R"(
.text
.globl simt_switch
.type simt_switch,@function
simt_switch:
    pushq %rbp
    pushq %rbx
    pushq %r12
    pushq %r13
    pushq %r14
    pushq %r15
    movq %rsp, (%rdi)
    movq %rsi, %rsp
    popq %r15
    popq %r14
    popq %r13
    popq %r12
    popq %rbx
    popq %rbp
    ret
.size simt_switch,.-simt_switch
)");

static void yield_to_sched(State st) {
    Fiber* f = running;
    f->st = st;
    simt_switch(&f->sp, sched_sp);
}

static void fiber_main() {
    (*body_fn)();
    dma_wait(0);   // pieces still in flight at the end of the work-item land (zero fills of a free stage, typically)
    yield_to_sched(DONE);
    std::fprintf(stderr, "simt: resumed a finished fiber\n");
    std::abort();
}

// ---- LDS-DMA completion model (simt_emu.h)
struct DmaRec {
    unsigned char* dst;   // nullptr: a store place-holder
    unsigned char data[16];
};
static std::vector<DmaRec> dmaq[MAX_THREADS];
static int dma_mode = -1, dma_weak = 0;   // mode 0 eager, 1 late

static int fiber_index() { return (int)(running - fibers.data()); }
static void dma_land(std::vector<DmaRec>& q, size_t n) {
    for (size_t i = 0; i < n; ++i)
        if (q[i].dst) std::memcpy(q[i].dst, q[i].data, 16);
    q.erase(q.begin(), q.begin() + (long)n);
}
void dma_issue(unsigned char* lds_dst, const void* src16) {
    if (dma_mode < 0) {
        const char* e = std::getenv("DPC_EMU_DMA");
        dma_mode = (e && e[0] == 'e') ? 0 : 1;
        const char* w = std::getenv("DPC_EMU_DMA_WEAK");
        dma_weak = w ? std::atoi(w) : 0;
    }
    if (dma_mode == 0) {
        if (src16) std::memcpy(lds_dst, src16, 16); else std::memset(lds_dst, 0, 16);
        return;
    }
    DmaRec r;
    r.dst = lds_dst;
    if (src16) std::memcpy(r.data, src16, 16); else std::memset(r.data, 0, 16);
    dmaq[fiber_index()].push_back(r);
}
void dma_note_stores(int n) {
    if (dma_mode == 0) return;
    DmaRec r;
    r.dst = nullptr;
    for (int i = 0; i < n; ++i) dmaq[fiber_index()].push_back(r);
}
void dma_wait(int leave) {
    if (!running) return;
    std::vector<DmaRec>& q = dmaq[fiber_index()];
    const size_t keep = (size_t)(leave > 0 ? leave + dma_weak : 0);   // DPC_EMU_DMA_WEAK: a counted wait that is too weak by that many pieces
    if (q.size() > keep) dma_land(q, q.size() - keep);
}

void sync_block() { yield_to_sched(WAIT_BLOCK); }
void sync_wave() { yield_to_sched(WAIT_WAVE); }

static void resume(Fiber& f) {
    running = &f;
    cur.tid = f.tid;
    cur.lane = f.lane;
    cur.wave = f.wave;
    cur.w = &waves[f.wave];
    simt_switch(&sched_sp, f.sp);
    running = nullptr;
}

static void prepare(Fiber& f, int idx) {
    unsigned char* top = stacks + (size_t)(idx + 1) * STACK_BYTES;
    uintptr_t t = ((uintptr_t)top) & ~(uintptr_t)15;
    void** sp = (void**)t;
    *--sp = nullptr;              // fake return address of fiber_main (keeps rsp = 8 mod 16 at entry)
    *--sp = (void*)&fiber_main;   // `ret` target of the first switch
    for (int i = 0; i < 6; ++i) *--sp = nullptr;  // rbp rbx r12..r15
    f.sp = (void*)sp;
    f.st = READY;
    dmaq[idx].clear();
}

unsigned char* dyn_smem = nullptr;

void launch_dyn(dim3 grid, dim3 block, size_t lds_bytes, const std::function<void()>& body) {
    if (lds_bytes > 160 * 1024) {
        std::fprintf(stderr, "simt: %zu bytes of dynamic LDS exceed the 160 KB of a gfx950 CU\n", lds_bytes);
        std::abort();
    }
    std::vector<unsigned char> buf(lds_bytes + 16, 0xA5);  // poison: kernels must not rely on zeroed LDS
    dyn_smem = (unsigned char*)(((uintptr_t)buf.data() + 15) & ~(uintptr_t)15);
    launch(grid, block, body);
    dyn_smem = nullptr;
}

void launch(dim3 grid, dim3 block, const std::function<void()>& body) {
    const int nthreads = (int)(block.x * block.y * block.z);
    if (nthreads <= 0 || nthreads > MAX_THREADS) {
        std::fprintf(stderr, "simt: bad block size %d\n", nthreads);
        std::abort();
    }
    if (!stacks) stacks = (unsigned char*)std::malloc(STACK_BYTES * MAX_THREADS);
    const int nwaves = (nthreads + 63) / 64;
    fibers.resize(nthreads);
    waves.resize(nwaves);
    body_fn = &body;
    cur.bdim = {block.x, block.y, block.z};
    cur.gdim = {grid.x, grid.y, grid.z};
    for (unsigned bz = 0; bz < grid.z; ++bz)
        for (unsigned by = 0; by < grid.y; ++by)
            for (unsigned bx = 0; bx < grid.x; ++bx) {
                cur.bid = {bx, by, bz};
                for (int t = 0; t < nthreads; ++t) {
                    Fiber& f = fibers[t];
                    f.tid = {(unsigned)t % block.x, ((unsigned)t / block.x) % block.y, (unsigned)t / (block.x * block.y)};
                    f.lane = t & 63;
                    f.wave = t >> 6;
                    prepare(f, t);
                }
                for (;;) {
                    int done = 0, at_block = 0;
                    for (int w = 0; w < nwaves; ++w) {
                        const int lo = w * 64, hi = (lo + 64 < nthreads) ? lo + 64 : nthreads;
                        for (;;) {
                            for (int t = lo; t < hi; ++t)
                                if (fibers[t].st == READY) resume(fibers[t]);
                            int nw = 0, nb = 0, nd = 0;
                            for (int t = lo; t < hi; ++t) {
                                nw += fibers[t].st == WAIT_WAVE;
                                nb += fibers[t].st == WAIT_BLOCK;
                                nd += fibers[t].st == DONE;
                            }
                            if (nw == 0) break;
                            if (nb != 0) {
                                std::fprintf(stderr, "simt: wave %d diverged: %d lanes at a wave collective, %d at __syncthreads\n", w, nw, nb);
                                std::abort();
                            }
                            for (int t = lo; t < hi; ++t)
                                if (fibers[t].st == WAIT_WAVE) fibers[t].st = READY;
                        }
                    }
                    for (int t = 0; t < nthreads; ++t) {
                        done += fibers[t].st == DONE;
                        at_block += fibers[t].st == WAIT_BLOCK;
                    }
                    if (done == nthreads) break;
                    if (done + at_block != nthreads) {
                        std::fprintf(stderr, "simt: scheduler stuck\n");
                        std::abort();
                    }
                    for (int t = 0; t < nthreads; ++t)
                        if (fibers[t].st == WAIT_BLOCK) fibers[t].st = READY;
                }
            }
    body_fn = nullptr;
}

}  // namespace simt
