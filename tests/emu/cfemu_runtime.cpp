// cfemu_runtime.cpp — fiber scheduler behind tests/emu/hip/hip_runtime.h.
// TEST INFRASTRUCTURE ONLY (see the header's banner).
#include <hip/hip_runtime.h>

#include <sys/mman.h>
#include <ucontext.h>

#include <map>
#include <random>
#include <vector>

// the single dynamic-LDS window every kernel declares as `extern __shared__ ... cf_lds[]`
alignas(64) unsigned char cf_lds[160 * 1024];

namespace cfemu {

size_t g_bytes_live = 0;
static std::map<void*, size_t> g_allocs;

void* dev_alloc(size_t n) {
    if (n == 0) n = 1;
    void* p = nullptr;
    if (posix_memalign(&p, 256, n) != 0) return nullptr;
    // poison so that reads of uninitialised device memory are visible
    std::memset(p, 0xCD, n);
    g_allocs[p] = n;
    g_bytes_live += n;
    return p;
}
void dev_free(void* p) {
    if (!p) return;
    auto it = g_allocs.find(p);
    if (it != g_allocs.end()) { g_bytes_live -= it->second; g_allocs.erase(it); }
    std::free(p);
}

dim3 g_threadIdx, g_blockIdx, g_blockDim, g_gridDim;

enum State : uint8_t { READY, AT_BARRIER, AT_WAVE, DONE };

static const size_t kStack = 256 * 1024;
static const int kMaxThreads = 1024;

// Context switch between fibers.  x86-64: a dozen instructions (callee-saved registers + stack pointer); swapcontext()
// makes two rt_sigprocmask system calls per switch, and every ballot / shuffle / barrier of a kernel is a switch — the
// emulated distance kernel spent 90 % of its time there.  Elsewhere: ucontext.
#if defined(__x86_64__)
struct Ctx { void* sp; };
extern "C" void cfemu_switch(Ctx* from, Ctx* to);
asm(R"(
.text
.globl cfemu_switch
.type cfemu_switch,@function
cfemu_switch:
    pushq %rbp
    pushq %rbx
    pushq %r12
    pushq %r13
    pushq %r14
    pushq %r15
    movq %rsp, (%rdi)
    movq (%rsi), %rsp
    popq %r15
    popq %r14
    popq %r13
    popq %r12
    popq %rbx
    popq %rbp
    ret
.size cfemu_switch,.-cfemu_switch
)");
static void ctx_make(Ctx& c, unsigned char* stack, size_t size, void (*entry)()) {
    uintptr_t top = ((uintptr_t)stack + size) & ~(uintptr_t)15;
    void** sp = (void**)top;
    *--sp = nullptr;                 // keeps the entry's frame aligned as after a call
    *--sp = (void*)entry;            // popped by the ret of the first switch
    for (int i = 0; i < 6; ++i) *--sp = nullptr;      // rbp, rbx, r12 .. r15
    c.sp = (void*)sp;
}
static inline void ctx_switch(Ctx& from, Ctx& to) { cfemu_switch(&from, &to); }
#else
struct Ctx { ucontext_t uc; };
static void ctx_make(Ctx& c, unsigned char* stack, size_t size, void (*entry)()) {
    getcontext(&c.uc);
    c.uc.uc_stack.ss_sp = stack;
    c.uc.uc_stack.ss_size = size;
    c.uc.uc_link = nullptr;
    makecontext(&c.uc, entry, 0);
}
static inline void ctx_switch(Ctx& from, Ctx& to) { swapcontext(&from.uc, &to.uc); }
#endif

struct Fiber {
    Ctx ctx;
    State st;
};

static Ctx g_sched;
static Fiber g_fib[kMaxThreads];
static unsigned char* g_stacks = nullptr;
static int g_cur = -1;
static int g_nthreads = 0;
static const std::function<void()>* g_body = nullptr;

struct WaveSlot {
    uint64_t val[2][64];
    uint64_t mask[2];
    int gen;  // generation being collected
};
static WaveSlot g_wave[kMaxThreads / 64];
static const uint64_t* g_ret_vals[kMaxThreads];
static uint64_t g_ret_mask[kMaxThreads];

static void set_idx(int t) {
    g_threadIdx.x = (unsigned)t % g_blockDim.x;
    g_threadIdx.y = ((unsigned)t / g_blockDim.x) % g_blockDim.y;
    g_threadIdx.z = (unsigned)t / (g_blockDim.x * g_blockDim.y);
}

static void yield_to_sched() {
    int me = g_cur;
    ctx_switch(g_fib[me].ctx, g_sched);
    set_idx(me);
}

static void fiber_main() {
    (*g_body)();
    g_fib[g_cur].st = DONE;
    ctx_switch(g_fib[g_cur].ctx, g_sched);
    std::abort();       // a finished fiber is never resumed
}

void block_barrier() {
    g_fib[g_cur].st = AT_BARRIER;
    yield_to_sched();
}

unsigned lane_id() { return (unsigned)g_cur & 63u; }

const uint64_t* wave_exchange(uint64_t v, uint64_t* mask) {
    const int me = g_cur, w = me >> 6, lane = me & 63;
    WaveSlot& s = g_wave[w];
    const int b = s.gen & 1;
    s.val[b][lane] = v;
    s.mask[b] |= 1ull << lane;
    g_fib[me].st = AT_WAVE;
    yield_to_sched();
    *mask = g_ret_mask[me];
    return g_ret_vals[me];
}

static void ensure_stacks() {
    if (g_stacks) return;
    void* p = mmap(nullptr, kStack * kMaxThreads, PROT_READ | PROT_WRITE,
                   MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
    if (p == MAP_FAILED) { std::fprintf(stderr, "cfemu: cannot map fiber stacks\n"); std::abort(); }
    g_stacks = (unsigned char*)p;
}

void run_grid(dim3 grid, dim3 block, size_t lds_bytes, const std::function<void()>& body) {
    ensure_stacks();
    const int nt = (int)(block.x * block.y * block.z);
    if (nt <= 0 || nt > kMaxThreads) { std::fprintf(stderr, "cfemu: bad block size %d\n", nt); std::abort(); }
    if (lds_bytes > sizeof cf_lds) { std::fprintf(stderr, "cfemu: LDS request %zu > 160 KiB\n", lds_bytes); std::abort(); }
    static int order_mode = -1;
    static std::mt19937 rng(12345);
    if (order_mode < 0) {
        const char* e = std::getenv("CF_EMU_ORDER");
        order_mode = !e ? 0 : (!std::strcmp(e, "rev") ? 1 : (!std::strcmp(e, "rand") ? 2 : 0));
    }
    g_blockDim = block;
    g_gridDim = grid;
    g_body = &body;
    g_nthreads = nt;
    const int nw = (nt + 63) / 64;
    std::vector<int> order((size_t)nt);
    for (unsigned bz = 0; bz < grid.z; ++bz)
    for (unsigned by = 0; by < grid.y; ++by)
    for (unsigned bx = 0; bx < grid.x; ++bx) {
        g_blockIdx = dim3(bx, by, bz);
        // poison LDS: kernels must not rely on zero-initialised shared memory
        std::memset(cf_lds, 0xA5, lds_bytes ? lds_bytes : 0);
        for (int t = 0; t < nt; ++t) {
            ctx_make(g_fib[t].ctx, g_stacks + (size_t)t * kStack, kStack, fiber_main);
            g_fib[t].st = READY;
        }
        for (int w = 0; w < nw; ++w) { g_wave[w].mask[0] = g_wave[w].mask[1] = 0; g_wave[w].gen = 0; }
        int live = nt;
        while (live > 0) {
            for (int t = 0; t < nt; ++t) order[(size_t)t] = order_mode == 1 ? nt - 1 - t : t;
            if (order_mode == 2) std::shuffle(order.begin(), order.end(), rng);
            bool ran = false;
            for (int oi = 0; oi < nt; ++oi) {
                const int t = order[(size_t)oi];
                if (g_fib[t].st != READY) continue;
                ran = true;
                g_cur = t;
                set_idx(t);
                ctx_switch(g_sched, g_fib[t].ctx);
                if (g_fib[t].st == DONE) --live;
            }
            if (ran) continue;  // sweep again until quiescent
            // quiescent: release wave rendezvous first, then the block barrier
            bool released = false;
            for (int w = 0; w < nw; ++w) {
                WaveSlot& s = g_wave[w];
                const int b = s.gen & 1;
                if (!s.mask[b]) continue;
                for (int l = 0; l < 64; ++l) {
                    const int t = w * 64 + l;
                    if (t < nt && g_fib[t].st == AT_WAVE) {
                        g_ret_vals[t] = s.val[b];
                        g_ret_mask[t] = s.mask[b];
                        g_fib[t].st = READY;
                    }
                }
                s.gen++;
                s.mask[s.gen & 1] = 0;
                released = true;
            }
            if (released) continue;
            int at_bar = 0;
            for (int t = 0; t < nt; ++t) at_bar += g_fib[t].st == AT_BARRIER;
            if (at_bar == live && live > 0) {
                for (int t = 0; t < nt; ++t) if (g_fib[t].st == AT_BARRIER) g_fib[t].st = READY;
                continue;
            }
            std::fprintf(stderr, "cfemu: deadlock in block (%u,%u,%u): %d live, %d at barrier\n", bx, by, bz, live, at_bar);
            std::abort();
        }
    }
    g_body = nullptr;
    g_cur = -1;
}

}  // namespace cfemu
