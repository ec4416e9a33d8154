// hipemu.cpp — fiber scheduler for the host-side HIP-dialect simulator (see hipemu.h).
// TEST INFRASTRUCTURE ONLY.
#include "hipemu.h"

#include <vector>

// Minimal x86-64 SysV context switch: saves callee-saved GPRs on the current stack, stores the
// stack pointer through %rdi, loads %rsi as the new stack pointer, restores and returns.
extern "C" void hipemu_switch(void** save_sp, void* load_sp);
asm(R"(
    .text
    .globl hipemu_switch
    .type hipemu_switch,@function
hipemu_switch:
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
    .size hipemu_switch,.-hipemu_switch
)");

namespace hipemu {

Dim3 g_threadIdx, g_blockIdx, g_blockDim, g_gridDim;

namespace {

enum State { RUNNABLE = 0, WAIT_WAVE = 1, WAIT_BLOCK = 2, DONE = 3 };
constexpr size_t kStackBytes = 256 * 1024;

struct Fiber {
    void* sp = nullptr;
    char* stack = nullptr;
    State state = DONE;
    Dim3 tid;
    int lane = 0, wave = 0;
    unsigned seq = 0;  // number of wave collectives issued
};

struct Block {
    std::vector<Fiber> fibers;
    std::vector<std::vector<char>> wbuf;  // per wave: 2 * 64 * kSlotBytes
    void* sched_sp = nullptr;
    Fiber* cur = nullptr;
    const std::function<void()>* body = nullptr;
    int nthreads = 0, nwaves = 0;
};

Block g_blk;
std::vector<char*> g_stack_pool;

void yield_to_scheduler() {
    Fiber* f = g_blk.cur;
    hipemu_switch(&f->sp, g_blk.sched_sp);
}

void fiber_entry() {
    (*g_blk.body)();
    g_blk.cur->state = DONE;
    yield_to_scheduler();
    fprintf(stderr, "hipemu: resumed a finished fiber\n");
    abort();
}

void prepare_fiber(Fiber& f) {
    uintptr_t top = (uintptr_t)(f.stack + kStackBytes);
    top &= ~(uintptr_t)15;
    void** sp = (void**)top;
    *--sp = nullptr;               // fake return address of fiber_entry's "caller"
    *--sp = (void*)&fiber_entry;   // popped by `ret`
    for (int i = 0; i < 6; ++i) *--sp = nullptr;  // rbp rbx r12 r13 r14 r15
    f.sp = (void*)sp;
}

void switch_to(Fiber& f) {
    g_blk.cur = &f;
    g_threadIdx = f.tid;
    hipemu_switch(&g_blk.sched_sp, f.sp);
    g_blk.cur = nullptr;
}

void run_block() {
    Block& b = g_blk;
    for (;;) {
        bool progress = false;
        for (int w = 0; w < b.nwaves; ++w) {
            const int lo = w * 64, hi = (lo + 64 < b.nthreads) ? lo + 64 : b.nthreads;
            for (;;) {
                bool ran = false;
                for (int t = lo; t < hi; ++t) {
                    if (b.fibers[t].state == RUNNABLE) { switch_to(b.fibers[t]); ran = true; }
                }
                if (ran) progress = true;
                // wave collective complete?
                int waiting = 0, blocked = 0;
                for (int t = lo; t < hi; ++t) {
                    if (b.fibers[t].state == WAIT_WAVE) ++waiting;
                    else if (b.fibers[t].state == WAIT_BLOCK) ++blocked;
                }
                if (waiting > 0 && blocked == 0) {
                    unsigned seq = 0; bool first = true;
                    for (int t = lo; t < hi; ++t) if (b.fibers[t].state == WAIT_WAVE) {
                        if (first) { seq = b.fibers[t].seq; first = false; }
                        else if (b.fibers[t].seq != seq) { fprintf(stderr, "hipemu: divergent wave collectives (block %u,%u,%u)\n", g_blockIdx.x, g_blockIdx.y, g_blockIdx.z); abort(); }
                        b.fibers[t].state = RUNNABLE;
                    }
                    progress = true;
                    continue;  // keep running this wave
                }
                break;
            }
        }
        int blocked = 0, done = 0, waiting = 0;
        for (auto& f : b.fibers) {
            if (f.state == WAIT_BLOCK) ++blocked;
            else if (f.state == DONE) ++done;
            else if (f.state == WAIT_WAVE) ++waiting;
        }
        if (done == b.nthreads) return;
        if (blocked > 0 && blocked + done == b.nthreads) {
            for (auto& f : b.fibers) if (f.state == WAIT_BLOCK) f.state = RUNNABLE;
            continue;
        }
        if (!progress) {
            fprintf(stderr, "hipemu: deadlock in block (%u,%u,%u): %d at barrier, %d in wave collective, %d done of %d\n",
                    g_blockIdx.x, g_blockIdx.y, g_blockIdx.z, blocked, waiting, done, b.nthreads);
            abort();
        }
    }
}

}  // namespace

int lane_id() { return g_blk.cur->lane; }

void block_barrier() {
    g_blk.cur->state = WAIT_BLOCK;
    yield_to_scheduler();
}

const char* wave_exchange(const void* mine, int nbytes) {
    Fiber* f = g_blk.cur;
    if (nbytes > kSlotBytes) { fprintf(stderr, "hipemu: slot overflow\n"); abort(); }
    char* base = g_blk.wbuf[f->wave].data() + (size_t)(f->seq & 1) * 64 * kSlotBytes;
    memcpy(base + (size_t)f->lane * kSlotBytes, mine, nbytes);
    f->seq++;
    f->state = WAIT_WAVE;
    yield_to_scheduler();
    return base;
}

void launch(Dim3 grid, Dim3 block, const std::function<void()>& body) {
    Block& b = g_blk;
    b.nthreads = (int)(block.x * block.y * block.z);
    b.nwaves = (b.nthreads + 63) / 64;
    b.body = &body;
    b.fibers.assign(b.nthreads, Fiber());
    while ((int)g_stack_pool.size() < b.nthreads) g_stack_pool.push_back((char*)malloc(kStackBytes));
    b.wbuf.assign(b.nwaves, std::vector<char>(2 * 64 * kSlotBytes, 0));
    g_blockDim = block;
    g_gridDim = grid;
    for (unsigned bz = 0; bz < grid.z; ++bz)
        for (unsigned by = 0; by < grid.y; ++by)
            for (unsigned bx = 0; bx < grid.x; ++bx) {
                g_blockIdx = Dim3(bx, by, bz);
                int t = 0;
                for (unsigned tz = 0; tz < block.z; ++tz)
                    for (unsigned ty = 0; ty < block.y; ++ty)
                        for (unsigned tx = 0; tx < block.x; ++tx, ++t) {
                            Fiber& f = b.fibers[t];
                            f.stack = g_stack_pool[t];
                            f.tid = Dim3(tx, ty, tz);
                            f.lane = t & 63;
                            f.wave = t >> 6;
                            f.seq = 0;
                            f.state = RUNNABLE;
                            prepare_fiber(f);
                        }
                run_block();
            }
}

}  // namespace hipemu
