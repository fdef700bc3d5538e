// hip/hip_runtime.h — HOST EMULATION of the small HIP subset used by
// centroflye_amd/csrc/hip/*.hip.  TEST INFRASTRUCTURE ONLY.
//
// Purpose: this container has no GPU.  To debug kernel LOGIC (indexing, hash probing, scans,
// carve-outs of LDS, barrier placement) before spending GPU-box minutes, the CPU test-suite
// compiles the unmodified kernel sources with g++ against this header
// (tests/emu/build_emu.sh -> tests/emu/libcfhip_emu.so) and runs every thread of a block as
// a ucontext fiber on one OS thread:
//   * __syncthreads()                -> all live threads of the block rendezvous
//   * __ballot/__shfl*/__any/__all   -> the live lanes of a 64-wide wave rendezvous
//   * atomics                        -> plain read-modify-write (one OS thread)
//   * blocks run one after another; lane order inside a block is configurable
//     (CF_EMU_ORDER=fwd|rev|rand) so that a missing barrier shows up as a wrong result.
// It proves nothing about memory-model behaviour or speed; the `-m gpu` tests do that on a
// real MI355X.  The product package never loads the emulated library.
#pragma once
#include <algorithm>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <tuple>
#include <type_traits>
#include <utility>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __shared__            /* only dynamic LDS is used: `extern __shared__ T name[]` */
#define __launch_bounds__(...)
#ifndef __restrict__
#define __restrict__
#endif

struct dim3 {
    unsigned x, y, z;
    constexpr dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};

// ------------------------------------------------------------------ runtime API (host side)
typedef int hipError_t;
enum { hipSuccess = 0, hipErrorOutOfMemory = 2, hipErrorInvalidValue = 1, hipErrorNotReady = 600 };
typedef struct cfemu_stream* hipStream_t;
struct cfemu_event { std::chrono::steady_clock::time_point t; };
typedef cfemu_event* hipEvent_t;
enum hipMemcpyKind { hipMemcpyHostToHost, hipMemcpyHostToDevice, hipMemcpyDeviceToHost, hipMemcpyDeviceToDevice, hipMemcpyDefault };
enum hipFuncAttribute { hipFuncAttributeMaxDynamicSharedMemorySize = 8 };
struct hipDeviceProp_t {
    char name[256];
    char gcnArchName[256];
    size_t totalGlobalMem;
    int multiProcessorCount;
    size_t sharedMemPerBlock;
    size_t maxSharedMemoryPerMultiProcessor;
    int warpSize;
    int clockRate;
};

namespace cfemu {
extern size_t g_bytes_live;
void* dev_alloc(size_t n);
void dev_free(void* p);
}  // namespace cfemu

inline const char* hipGetErrorString(hipError_t e) { return e == hipSuccess ? "hipSuccess" : "emulated HIP error"; }
inline hipError_t hipGetLastError() { return hipSuccess; }
inline hipError_t hipPeekAtLastError() { return hipSuccess; }
inline hipError_t hipSetDevice(int) { return hipSuccess; }
inline hipError_t hipGetDevice(int* d) { *d = 0; return hipSuccess; }
inline hipError_t hipGetDeviceCount(int* n) { *n = 1; return hipSuccess; }
inline hipError_t hipGetDeviceProperties(hipDeviceProp_t* p, int) {
    std::memset(p, 0, sizeof *p);
    std::snprintf(p->name, sizeof p->name, "cfemu host emulation");
    std::snprintf(p->gcnArchName, sizeof p->gcnArchName, "emu");
    p->totalGlobalMem = (size_t)64 << 30;
    p->multiProcessorCount = 4;
    p->sharedMemPerBlock = 64 << 10;
    p->maxSharedMemoryPerMultiProcessor = 160 << 10;
    p->warpSize = 64;
    p->clockRate = 1000000;
    return hipSuccess;
}
// (CFEMU_TOTAL_MB: the size of the emulated device, for tests of the "would not fit" paths; read at every call)
inline hipError_t hipMemGetInfo(size_t* fr, size_t* tot) {
    const char* mb = std::getenv("CFEMU_TOTAL_MB");
    *tot = mb && *mb ? (size_t)std::strtoull(mb, nullptr, 10) << 20 : (size_t)64 << 30;
    *fr = *tot > cfemu::g_bytes_live ? *tot - cfemu::g_bytes_live : 0;
    return hipSuccess;
}
inline hipError_t hipMalloc(void** p, size_t n) { *p = cfemu::dev_alloc(n); return (*p || !n) ? hipSuccess : hipErrorOutOfMemory; }
template <class T> inline hipError_t hipMalloc(T** p, size_t n) { return hipMalloc((void**)p, n); }
inline hipError_t hipFree(void* p) { cfemu::dev_free(p); return hipSuccess; }
inline hipError_t hipHostMalloc(void** p, size_t n, unsigned = 0) { *p = std::malloc(n ? n : 1); return *p ? hipSuccess : hipErrorOutOfMemory; }
inline hipError_t hipHostFree(void* p) { std::free(p); return hipSuccess; }
enum { hipHostRegisterDefault = 0 };
inline hipError_t hipHostRegister(void*, size_t, unsigned) { return hipSuccess; }
inline hipError_t hipHostUnregister(void*) { return hipSuccess; }
enum hipMemoryType { hipMemoryTypeHost = 1, hipMemoryTypeDevice = 2 };
struct hipPointerAttribute_t { hipMemoryType type; };
inline hipError_t hipPointerGetAttributes(hipPointerAttribute_t* a, const void*) { a->type = hipMemoryTypeHost; return hipSuccess; }   // "device" memory is host memory here
inline hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) { if (n) std::memmove(d, s, n); return hipSuccess; }
inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind k, hipStream_t) { return hipMemcpy(d, s, n, k); }
inline hipError_t hipMemset(void* d, int v, size_t n) { if (n) std::memset(d, v, n); return hipSuccess; }
inline hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t) { return hipMemset(d, v, n); }
inline hipError_t hipStreamCreate(hipStream_t* s) { *s = nullptr; return hipSuccess; }
inline hipError_t hipStreamDestroy(hipStream_t) { return hipSuccess; }
inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
inline hipError_t hipDeviceSynchronize() { return hipSuccess; }
inline hipError_t hipEventCreate(hipEvent_t* e) { *e = new cfemu_event(); return hipSuccess; }
inline hipError_t hipEventDestroy(hipEvent_t e) { delete e; return hipSuccess; }
inline hipError_t hipEventRecord(hipEvent_t e, hipStream_t = nullptr) { e->t = std::chrono::steady_clock::now(); return hipSuccess; }
inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
inline hipError_t hipEventElapsedTime(float* ms, hipEvent_t a, hipEvent_t b) {
    *ms = std::chrono::duration<float, std::milli>(b->t - a->t).count();
    return hipSuccess;
}
template <class F> inline hipError_t hipFuncSetAttribute(F, hipFuncAttribute, int) { return hipSuccess; }

// ------------------------------------------------------------------ device side
namespace cfemu {
extern dim3 g_threadIdx, g_blockIdx, g_blockDim, g_gridDim;
void run_grid(dim3 grid, dim3 block, size_t lds_bytes, const std::function<void()>& body);
void block_barrier();
// wave rendezvous: every live lane deposits v; returns pointer to the 64 deposited values and
// the mask of participating lanes (valid until the lane's next rendezvous)
const uint64_t* wave_exchange(uint64_t v, uint64_t* mask);
unsigned lane_id();
}  // namespace cfemu

#define threadIdx cfemu::g_threadIdx
#define blockIdx cfemu::g_blockIdx
#define blockDim cfemu::g_blockDim
#define gridDim cfemu::g_gridDim
static const int warpSize = 64;

template <class... KArgs, class... Args>
inline void cfemu_launch(void (*kern)(KArgs...), dim3 grid, dim3 block, size_t lds, hipStream_t, Args... args) {
    std::tuple<std::decay_t<KArgs>...> t{static_cast<std::decay_t<KArgs>>(args)...};
    cfemu::run_grid(grid, block, lds, [&]() { std::apply(kern, t); });
}
#define hipLaunchKernelGGL(kern, grid, block, lds, stream, ...) \
    cfemu_launch(kern, dim3(grid), dim3(block), (size_t)(lds), stream, ##__VA_ARGS__)

inline void __syncthreads() { cfemu::block_barrier(); }
inline void __threadfence() {}
inline void __threadfence_block() {}
inline unsigned __lane_id() { return cfemu::lane_id(); }

inline unsigned long long __ballot(int pred) {
    uint64_t mask;
    const uint64_t* v = cfemu::wave_exchange(pred ? 1 : 0, &mask);
    unsigned long long r = 0;
    for (int i = 0; i < 64; ++i) if (((mask >> i) & 1) && v[i]) r |= 1ull << i;
    return r;
}
inline int __any(int pred) { return __ballot(pred) != 0; }
// (cf_common.h: on the device the LDS window is addressed from the number 0)
struct uint2 { unsigned x, y; };
inline uint2 make_uint2(unsigned x, unsigned y) { uint2 v; v.x = x; v.y = y; return v; }
#define cf_lds_at(off) (cf_lds + (off))
#define cf_lds_base_ok() true
#define cf_ld_agent(p) (*(p))          /* cf_place2.hip: device-scope loads / the wave's drain of its memory operations */
#define CF_NO_BUFFER_LOAD 1      /* cf_dist.hip: the entry stream through pointer loads (no buffer descriptors on the host) */
#define cf_ballot(p) __ballot((p) ? 1 : 0)      /* cf_dist.hip: __builtin_amdgcn_ballot_w64 */
#define cf_bit_of(w, off) (((w) >> ((off) & 31u)) & 1u)      /* cf_dist.hip: v_bfe_u32 w, off, 1 (the offset's low five bits) */
#define cf_drain_vm() ((void)0)
#define cf_barrier_lds() __syncthreads()
#define __builtin_amdgcn_fence(order, scope) ((void)0)   /* lanes are fibers on one thread: program order is memory order */
inline void __builtin_amdgcn_wave_barrier() { (void)__ballot(1); }   // lanes of a wave run in lock step on the GPU: rendezvous here
inline int __all(int pred) {
    uint64_t mask;
    const uint64_t* v = cfemu::wave_exchange(pred ? 1 : 0, &mask);
    for (int i = 0; i < 64; ++i) if (((mask >> i) & 1) && !v[i]) return 0;
    return 1;
}
inline unsigned long long __activemask() { uint64_t m; cfemu::wave_exchange(0, &m); return m; }

template <class T> inline uint64_t cfemu_bits(T v) { uint64_t b = 0; static_assert(sizeof(T) <= 8, ""); std::memcpy(&b, &v, sizeof(T)); return b; }
template <class T> inline T cfemu_unbits(uint64_t b) { T v; std::memcpy(&v, &b, sizeof(T)); return v; }

template <class T> inline T __shfl(T v, int src, int width = 64) {
    uint64_t mask;
    const unsigned me = cfemu::lane_id();
    const uint64_t* a = cfemu::wave_exchange(cfemu_bits(v), &mask);
    int s = (int)(me & ~(unsigned)(width - 1)) + (src & (width - 1));
    return ((mask >> s) & 1) ? cfemu_unbits<T>(a[s]) : v;
}
template <class T> inline T __shfl_down(T v, unsigned d, int width = 64) {
    uint64_t mask;
    const unsigned me = cfemu::lane_id();
    const uint64_t* a = cfemu::wave_exchange(cfemu_bits(v), &mask);
    unsigned s = me + d;
    if ((s & ~(unsigned)(width - 1)) != (me & ~(unsigned)(width - 1)) || s >= 64) return v;
    return ((mask >> s) & 1) ? cfemu_unbits<T>(a[s]) : v;
}
template <class T> inline T __shfl_up(T v, unsigned d, int width = 64) {
    uint64_t mask;
    const unsigned me = cfemu::lane_id();
    const uint64_t* a = cfemu::wave_exchange(cfemu_bits(v), &mask);
    if ((me & (unsigned)(width - 1)) < d) return v;
    unsigned s = me - d;
    return ((mask >> s) & 1) ? cfemu_unbits<T>(a[s]) : v;
}
template <class T> inline T __shfl_xor(T v, int x, int width = 64) {
    uint64_t mask;
    const unsigned me = cfemu::lane_id();
    const uint64_t* a = cfemu::wave_exchange(cfemu_bits(v), &mask);
    unsigned s = me ^ (unsigned)x;
    (void)width;
    if (s >= 64) return v;
    return ((mask >> s) & 1) ? cfemu_unbits<T>(a[s]) : v;
}
inline int __builtin_amdgcn_readfirstlane(int v) {
    uint64_t mask;
    const uint64_t* a = cfemu::wave_exchange((uint64_t)(uint32_t)v, &mask);
    int first = __builtin_ctzll(mask);
    return (int)(uint32_t)a[first];
}

// only the control used by the kernels: 0x138 = wave_shr:1 (lane i reads lane i - 1; lane 0 keeps `old`)
inline int __builtin_amdgcn_update_dpp(int old, int src, int ctrl, int, int, bool) {
    uint64_t mask;
    const unsigned me = cfemu::lane_id();
    const uint64_t* a = cfemu::wave_exchange((uint64_t)(uint32_t)src, &mask);
    if (ctrl != 0x138) std::abort();
    return me == 0 ? old : (int)(uint32_t)a[me - 1];
}
inline int __builtin_amdgcn_readlane(int v, int lane) {
    uint64_t mask;
    const uint64_t* a = cfemu::wave_exchange((uint64_t)(uint32_t)v, &mask);
    return (int)(uint32_t)a[lane & 63];
}

// v_bfe_u32: offset and width are taken from the low 5 bits of their operands
inline unsigned __builtin_amdgcn_ubfe(unsigned v, unsigned offset, unsigned width) {
    offset &= 31u; width &= 31u;
    return width == 0 ? 0u : (v >> offset) & ((1u << width) - 1u);
}

inline int __popc(unsigned v) { return __builtin_popcount(v); }
inline int __popcll(unsigned long long v) { return __builtin_popcountll(v); }
// v_mbcnt_lo / v_mbcnt_hi: bits of the mask below this lane, added to `add`
inline unsigned __builtin_amdgcn_mbcnt_lo(unsigned mask, unsigned add) { const unsigned l = cfemu::lane_id(); return add + (unsigned)__builtin_popcount(l >= 32 ? mask : (mask & ((1u << l) - 1u))); }
inline unsigned __builtin_amdgcn_mbcnt_hi(unsigned mask, unsigned add) { const unsigned l = cfemu::lane_id(); return add + (l <= 32 ? 0u : (unsigned)__builtin_popcount(mask & ((1u << (l - 32)) - 1u))); }
// v_alignbyte_b32: ({hi, lo} >> 8 * (s & 3)), low word;  v_perm_b32: byte i of the result = byte sel.byte[i] of {s0 (bytes 7..4), s1 (bytes 3..0)}
// (selector values 0 .. 7 only; 12 = 0x00 and 13 .. 15 = 0xFF as the hardware defines them)
inline unsigned __builtin_amdgcn_alignbyte(unsigned hi, unsigned lo, unsigned s) { return (unsigned)(((((unsigned long long)hi) << 32) | lo) >> (8 * (s & 3u))); }
inline unsigned __builtin_amdgcn_perm(unsigned s0, unsigned s1, unsigned sel) {
    const unsigned long long pool = (((unsigned long long)s0) << 32) | s1;
    unsigned r = 0;
    for (int i = 0; i < 4; ++i) {
        const unsigned c = (sel >> (8 * i)) & 0xFFu;
        const unsigned b = c < 8u ? (unsigned)((pool >> (8 * c)) & 0xFFu) : (c == 12u ? 0u : (c >= 13u ? 0xFFu : 0u));
        r |= b << (8 * i);
    }
    return r;
}
inline int __ffs(int v) { return __builtin_ffs(v); }
inline int __ffsll(long long v) { return __builtin_ffsll(v); }
inline int __clz(int v) { return v ? __builtin_clz((unsigned)v) : 32; }
inline int __clzll(long long v) { return v ? __builtin_clzll((unsigned long long)v) : 64; }
inline void __builtin_amdgcn_s_sleep(int) {}


// atomics (single OS thread: plain RMW)
template <class T> inline T atomicAdd(T* p, T v) { T o = *p; *p = (T)(o + v); return o; }
template <class T> inline T atomicSub(T* p, T v) { T o = *p; *p = (T)(o - v); return o; }
template <class T> inline T atomicOr(T* p, T v) { T o = *p; *p = (T)(o | v); return o; }
template <class T> inline T atomicAnd(T* p, T v) { T o = *p; *p = (T)(o & v); return o; }
template <class T> inline T atomicMax(T* p, T v) { T o = *p; if (v > o) *p = v; return o; }
template <class T> inline T atomicMin(T* p, T v) { T o = *p; if (v < o) *p = v; return o; }
template <class T> inline T atomicExch(T* p, T v) { T o = *p; *p = v; return o; }
template <class T> inline T atomicCAS(T* p, T cmp, T v) { T o = *p; if (o == cmp) *p = v; return o; }

// scalar min/max overloads HIP puts in the global namespace
inline int min(int a, int b) { return a < b ? a : b; }
inline int max(int a, int b) { return a > b ? a : b; }
inline unsigned min(unsigned a, unsigned b) { return a < b ? a : b; }
inline unsigned max(unsigned a, unsigned b) { return a > b ? a : b; }
inline long long min(long long a, long long b) { return a < b ? a : b; }
inline long long max(long long a, long long b) { return a > b ? a : b; }
inline long min(long a, long b) { return a < b ? a : b; }
inline long max(long a, long b) { return a > b ? a : b; }
inline unsigned long long min(unsigned long long a, unsigned long long b) { return a < b ? a : b; }
inline unsigned long long max(unsigned long long a, unsigned long long b) { return a > b ? a : b; }
inline unsigned long min(unsigned long a, unsigned long b) { return a < b ? a : b; }
inline unsigned long max(unsigned long a, unsigned long b) { return a > b ? a : b; }
