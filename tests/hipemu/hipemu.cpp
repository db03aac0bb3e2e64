// tests/hipemu/hipemu.cpp -- fiber scheduler of the HIP emulator (TEST INFRASTRUCTURE).
#include "hip/hip_runtime.h"

#include <stdio.h>

// clang announces its sanitizers through __has_feature, gcc through macros: one spelling below
#if defined(__has_feature)
#if __has_feature(thread_sanitizer) && !defined(__SANITIZE_THREAD__)
#define __SANITIZE_THREAD__ 1
#endif
#if __has_feature(address_sanitizer) && !defined(__SANITIZE_ADDRESS__)
#define __SANITIZE_ADDRESS__ 1
#endif
#endif
// ThreadSanitizer does not follow swapcontext by itself: tell it which fiber runs (tools/sanitize.sh builds this file with -fsanitize=thread)
#if defined(__SANITIZE_THREAD__)
extern "C" {
void *__tsan_get_current_fiber(void);
void *__tsan_create_fiber(unsigned flags);
void __tsan_destroy_fiber(void *fiber);
void __tsan_switch_to_fiber(void *fiber, unsigned flags);
}
#define PFV_TSAN_FIBERS 1
#else
#define PFV_TSAN_FIBERS 0
#endif

// Context switch.  glibc's swapcontext / getcontext make a signal-mask system call each: a third of the CPU suite's time was spent in the
// kernel.  On x86-64 without a sanitizer the fibers switch with a dozen instructions of their own (callee-saved registers + stack pointer);
// sanitizer builds and other targets keep ucontext (ASan follows swapcontext; TSan is told, above).
#if defined(__x86_64__) && !defined(__SANITIZE_ADDRESS__) && !defined(__SANITIZE_THREAD__) && !defined(PFV_EMU_UCONTEXT)
#define PFV_FAST_SWITCH 1
extern "C" void pfv_emu_switch(void **save_sp, void *to_sp);
asm(".text\n"
    ".hidden pfv_emu_switch\n"
    ".globl pfv_emu_switch\n"
    ".type pfv_emu_switch,@function\n"
    "pfv_emu_switch:\n"
    "    pushq %rbp\n"
    "    pushq %rbx\n"
    "    pushq %r12\n"
    "    pushq %r13\n"
    "    pushq %r14\n"
    "    pushq %r15\n"
    "    movq %rsp, (%rdi)\n"
    "    movq %rsi, %rsp\n"
    "    popq %r15\n"
    "    popq %r14\n"
    "    popq %r13\n"
    "    popq %r12\n"
    "    popq %rbx\n"
    "    popq %rbp\n"
    "    ret\n"
    ".size pfv_emu_switch,.-pfv_emu_switch\n");
#else
#define PFV_FAST_SWITCH 0
#endif

namespace hipemu {

Graph *g_capture = nullptr;
Idx g_threadIdx, g_blockIdx, g_blockDim, g_gridDim;

namespace {
constexpr size_t kStack = 128 * 1024;
struct Fiber {
#if PFV_FAST_SWITCH
    void *sp = nullptr;
#else
    ucontext_t ctx;
#endif
    void *stack = nullptr;
    bool done = false;
    Idx tid;
    void *tsan = nullptr;
};
void *sched_tsan = nullptr;
std::vector<Fiber> fibers;
#if PFV_FAST_SWITCH
void *sched_sp = nullptr;
#else
ucontext_t sched_ctx;
#endif
int cur = -1;
const std::function<void()> *cur_body = nullptr;

int n_threads = 0;
// block barrier
int bar_count = 0, bar_gen = 0;
// per-wave rendezvous
struct Wave { int count = 0, gen = 0; int slot[64]; int votes = 0; };
std::vector<Wave> waves;

void yield()
{
    int me = cur;
#if PFV_TSAN_FIBERS
    __tsan_switch_to_fiber(sched_tsan, 0);
#endif
#if PFV_FAST_SWITCH
    pfv_emu_switch(&fibers[me].sp, sched_sp);
#else
    swapcontext(&fibers[me].ctx, &sched_ctx);
#endif
    g_threadIdx = fibers[me].tid;
}
void trampoline()
{
    (*cur_body)();
    fibers[cur].done = true;
#if PFV_TSAN_FIBERS
    __tsan_switch_to_fiber(sched_tsan, 0);
#endif
#if PFV_FAST_SWITCH
    pfv_emu_switch(&fibers[cur].sp, sched_sp);
    __builtin_trap();          // a finished fiber is never resumed
#else
    swapcontext(&fibers[cur].ctx, &sched_ctx);
#endif
}
int lane_id() { return (int)(g_threadIdx.x & 63); }
Wave &my_wave() { return waves[g_threadIdx.x >> 6]; }
int wave_size(int w) { int lo = w * 64; int hi = lo + 64 < n_threads ? lo + 64 : n_threads; return hi - lo; }
}  // namespace

void block_barrier()
{
    int gen = bar_gen;
    if (++bar_count == n_threads) { bar_count = 0; bar_gen++; }
    else while (bar_gen == gen) yield();
}
void wave_barrier()
{
    Wave &w = my_wave();
    int gen = w.gen;
    if (++w.count == wave_size((int)(g_threadIdx.x >> 6))) { w.count = 0; w.gen++; }
    else while (w.gen == gen) yield();
}
int wave_exchange(int value, int src_lane)
{
    Wave &w = my_wave();
    w.slot[lane_id()] = value;
    wave_barrier();
    int r = w.slot[src_lane & 63];
    wave_barrier();
    return r;
}
bool wave_any(bool pred)
{
    int r = 0;
    for (int m = 1; m < 64; m <<= 1) { }   // (kept simple: gather through exchange of own predicate)
    Wave &w = my_wave();
    w.slot[lane_id()] = pred ? 1 : 0;
    wave_barrier();
    int n = wave_size((int)(g_threadIdx.x >> 6));
    for (int i = 0; i < n; i++) r |= w.slot[i];
    wave_barrier();
    return r != 0;
}

void launch(dim3 grid, dim3 block, const std::function<void()> &body)
{
    if (block.y != 1 || block.z != 1) abort();
    n_threads = (int)block.x;
    g_blockDim = Idx{block.x, block.y, block.z};
    g_gridDim = Idx{grid.x, grid.y, grid.z};
    if ((int)fibers.size() < n_threads) {
        fibers.resize(n_threads);
        for (auto &f : fibers)
            if (!f.stack) f.stack = malloc(kStack);
    }
    waves.assign((n_threads + 63) / 64, Wave());
    cur_body = &body;
#if PFV_TSAN_FIBERS
    sched_tsan = __tsan_get_current_fiber();     // the host thread that launches (one at a time: the emulator is single-threaded by design)
#endif
    for (unsigned bz = 0; bz < grid.z; bz++)
        for (unsigned by = 0; by < grid.y; by++)
            for (unsigned bx = 0; bx < grid.x; bx++) {
                g_blockIdx = Idx{bx, by, bz};
                bar_count = 0;
                for (auto &w : waves) { w.count = 0; }
                for (int t = 0; t < n_threads; t++) {
                    Fiber &f = fibers[t];
                    f.done = false;
                    f.tid = Idx{(unsigned)t, 0, 0};
#if PFV_FAST_SWITCH
                    // the frame pfv_emu_switch pops: r15 r14 r13 r12 rbx rbp, then `ret` into trampoline with the stack as after a call
                    uintptr_t top = ((uintptr_t)f.stack + kStack) & ~(uintptr_t)15;
                    void **fr = (void **)(top - 64);
                    for (int i = 0; i < 6; i++) fr[i] = nullptr;
                    fr[6] = (void *)&trampoline;
                    fr[7] = nullptr;                      // the return address trampoline never uses
                    f.sp = fr;
#else
                    getcontext(&f.ctx);
                    f.ctx.uc_stack.ss_sp = f.stack;
                    f.ctx.uc_stack.ss_size = kStack;
                    f.ctx.uc_link = nullptr;
                    makecontext(&f.ctx, trampoline, 0);
#endif
                }
                int remaining = n_threads;
                long spins = 0;
                while (remaining > 0) {
                    for (int t = 0; t < n_threads; t++) {
                        Fiber &f = fibers[t];
                        if (f.done) continue;
                        cur = t;
                        g_threadIdx = f.tid;
#if PFV_TSAN_FIBERS
                        if (!f.tsan) f.tsan = __tsan_create_fiber(0);
                        __tsan_switch_to_fiber(f.tsan, 0);
#endif
#if PFV_FAST_SWITCH
                        pfv_emu_switch(&sched_sp, f.sp);
#else
                        swapcontext(&sched_ctx, &f.ctx);
#endif
                        if (f.done) remaining--;
                    }
                    if (++spins > 10000000) { fprintf(stderr, "hipemu: deadlock (divergent barrier?)\n"); abort(); }
                }
            }
    cur_body = nullptr;
}

}  // namespace hipemu
