// tools/emu/hip/hip_runtime.h — a small HOST stand-in for <hip/hip_runtime.h>: enough of the HIP device vocabulary to
// run a kernel of kimera_semantics_amd/csrc/*.h on the CPU, one workgroup at a time, every work-item a fiber.
// TEST TOOLING (tools/emu/README.md): lets a kernel's logic be checked where no GPU is at hand.  Not a product path.
//
// Model: the work-items of a block are ucontext fibers, resumed round-robin in thread order; a fiber runs until it
// reaches a wave collective (__ballot, __shfl*, readlane), __syncthreads() or KS_WAVE_LDS_ORDER(), where it waits for
// the other live work-items of its wavefront (block).  Collectives must be reached in uniform control flow (checked:
// every participant must come from the same source line).  A wavefront's implicit lock-step is NOT modelled: where a
// kernel relies on "the LDS runs one wavefront's operations in program order" between lanes without a collective in
// between, it says so with KS_WAVE_LDS_ORDER(), which is a wave barrier here and nothing on the GPU.
// Single OS thread: atomics are plain read-modify-writes.  Blocks run one after the other (a kernel that depends on
// inter-block timing cannot be modelled; the result is ONE valid interleaving).
#pragma once
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <ucontext.h>

#include <functional>
#include <vector>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __shared__ thread_local   // (all fibers of a block run on the launching OS thread: one instance, like LDS)
#define __launch_bounds__(...)
#ifndef __HIP_MEMORY_SCOPE_AGENT
#define __HIP_MEMORY_SCOPE_AGENT 4
#endif

struct dim3 {
  unsigned x, y, z;
  dim3(unsigned a = 1, unsigned b = 1, unsigned c = 1) : x(a), y(b), z(c) {}
};
struct uint4 { unsigned x, y, z, w; };
inline uint4 make_uint4(unsigned x, unsigned y, unsigned z, unsigned w) { return uint4{x, y, z, w}; }
struct ulonglong2 { unsigned long long x, y; };
inline ulonglong2 make_ulonglong2(unsigned long long x, unsigned long long y) { return ulonglong2{x, y}; }
struct float4 { float x, y, z, w; };
inline float4 make_float4(float x, float y, float z, float w) { return float4{x, y, z, w}; }
struct float2 { float x, y; };
inline float2 make_float2(float x, float y) { return float2{x, y}; }
struct float3 { float x, y, z; };
inline float3 make_float3(float x, float y, float z) { return float3{x, y, z}; }
struct uint3 { unsigned x, y, z; };
struct int4 { int x, y, z, w; };
inline int4 make_int4(int x, int y, int z, int w) { return int4{x, y, z, w}; }
struct uint2 { unsigned x, y; };
inline uint2 make_uint2(unsigned x, unsigned y) { return uint2{x, y}; }

// A fiber switch without the two sigprocmask system calls of swapcontext (a collective costs one switch per lane):
// callee-saved registers + stack pointer, x86-64 System V.
#if defined(__x86_64__)
asm(R"(
.text
.weak emu_ctx_switch
.type emu_ctx_switch,@function
emu_ctx_switch:
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
.size emu_ctx_switch,.-emu_ctx_switch
)");
extern "C" void emu_ctx_switch(void** save_sp, void* new_sp);
#define EMU_FAST_SWITCH 1
#else
#define EMU_FAST_SWITCH 0
#endif

#if defined(__has_feature)
#if __has_feature(address_sanitizer)
extern "C" void __asan_unpoison_memory_region(void const volatile* addr, size_t size);
#endif
#endif

#if defined(__has_feature)
#if __has_feature(thread_sanitizer)
#define EMU_TSAN 1
extern "C" void* __tsan_get_current_fiber(void);
extern "C" void* __tsan_create_fiber(unsigned flags);
extern "C" void __tsan_destroy_fiber(void* fiber);
extern "C" void __tsan_switch_to_fiber(void* fiber, unsigned flags);
#endif
#endif
#ifndef EMU_TSAN
#define EMU_TSAN 0
#endif

namespace emu {
inline thread_local dim3 g_threadIdx, g_blockIdx, g_blockDim, g_gridDim;   // per launching OS thread

// One collective site (source location) of a wavefront: what the arriving lanes contribute, and the last completed
// exchange.  Lanes of one wavefront may wait at DIFFERENT sites at the same time (divergent branches, e.g. the four
// 16-lane groups of a wavefront taking different paths): an exchange completes when every live lane of the wavefront
// is waiting somewhere, and the lanes of a site see only each other — the exec mask of that instruction.
struct Xchg {
  long site = 0;
  unsigned gen = 0;
  uint64_t in_mask = 0, res_mask = 0;
  uint64_t in[64], res[64];
};
constexpr int kSites = 96;
struct Wave {
  int alive = 0, blocked = 0, at_barrier = 0, n_sites = 0;
  Xchg x[kSites];
};
struct Block {
  std::vector<ucontext_t> ctx;
  std::vector<void*> sp;   // fast switch: saved stack pointers of the fibers
  std::vector<void*> tfib;   // ThreadSanitizer build: its view of the fibers
  std::vector<unsigned> tfib_uses;   // work-items each of them has run (a fiber is replaced every 256: its shadow call stack fills up)
  void* tsched = nullptr;
  void* sched_sp = nullptr;
  std::vector<char*> stacks;
  std::vector<char> done;
  std::vector<Wave> waves;
  int alive = 0, arrived = 0;
  unsigned gen = 0;
  ucontext_t sched;
  int cur = 0;
  std::function<void()> body;
  unsigned long long switches = 0;
};
inline thread_local Block* B = nullptr;
constexpr size_t kStack = 256 * 1024;

inline void yield() {
  Block* b = B;
  const int me = b->cur;
  ++b->switches;
#if EMU_TSAN
  __tsan_switch_to_fiber(b->tsched, 0);
#endif
#if EMU_FAST_SWITCH
  emu_ctx_switch(&b->sp[me], b->sched_sp);
#else
  swapcontext(&b->ctx[me], &b->sched);
#endif
}
inline void fail(const char* what, int line) {
  fprintf(stderr, "emu: %s (source line %d, block %u,%u thread %u)\n", what, line, g_blockIdx.x, g_blockIdx.y, g_threadIdx.x);
  abort();
}
inline void release_wave(Wave& w) {
  for (int i = 0; i < w.n_sites; ++i) {
    Xchg& x = w.x[i];
    if (!x.in_mask) continue;
    for (int l = 0; l < 64; ++l)
      if ((x.in_mask >> l) & 1ull) x.res[l] = x.in[l];
    x.res_mask = x.in_mask;
    x.in_mask = 0;
    ++x.gen;
  }
  w.blocked = 0;
}
// the calling lane contributes v at `site`; returns the completed exchange (valid until the lane's next collective)
inline const Xchg& wave_exchange(uint64_t v, long site) {
  Block* b = B;
  Wave& w = b->waves[b->cur >> 6];
  const int lane = b->cur & 63;
  Xchg* x = nullptr;
  for (int i = 0; i < w.n_sites; ++i)
    if (w.x[i].site == site) { x = &w.x[i]; break; }
  if (!x) {
    if (w.n_sites == kSites) fail("too many collective sites in one kernel", (int)site);
    x = &w.x[w.n_sites++];
    x->site = site;
    x->gen = 0;
    x->in_mask = x->res_mask = 0;
  }
  x->in[lane] = v;
  x->in_mask |= 1ull << lane;
  ++w.blocked;
  const unsigned g = x->gen;
  while (x->gen == g) {
    if (w.blocked + w.at_barrier >= w.alive) release_wave(w);   // every live lane of the wavefront waits somewhere
    else yield();
  }
  return *x;
}
inline unsigned long long ballot(bool p, long site) {
  const Xchg& x = wave_exchange(p ? 1u : 0u, site);
  unsigned long long m = 0;
  for (int l = 0; l < 64; ++l)
    if (((x.res_mask >> l) & 1ull) && (x.res[l] & 1ull)) m |= 1ull << l;
  return m;
}
template <typename T>
inline uint64_t to_bits(T x) {
  static_assert(sizeof(T) <= 8, "shuffle operand");
  uint64_t u = 0;
  memcpy(&u, &x, sizeof(T));
  return u;
}
template <typename T>
inline T from_bits(uint64_t u) {
  T x;
  memcpy(&x, &u, sizeof(T));
  return x;
}
template <typename T>
inline T shfl_from(T v, int src, long site) {   // src = absolute lane; out of range / not participating: own value
  const Xchg& x = wave_exchange(to_bits(v), site);
  if (src < 0 || src > 63 || !((x.res_mask >> src) & 1ull)) return v;
  return from_bits<T>(x.res[src]);
}
template <typename T>
inline T first_lane(T v, long site) {   // value of the lowest participating lane
  const Xchg& x = wave_exchange(to_bits(v), site);
  return from_bits<T>(x.res[__builtin_ctzll(x.res_mask)]);
}
inline int permute_fwd(int dst, int v, long site) {   // ds_permute_b32: lane i sends v to lane dst_i; untargeted lanes read 0
  const Xchg& x = wave_exchange(((uint64_t)(uint32_t)v << 8) | (uint64_t)(uint32_t)dst, site);
  const int me = B->cur & 63;
  int out = 0;
  for (int l = 0; l < 64; ++l)
    if (((x.res_mask >> l) & 1ull) && (int)(x.res[l] & 255u) == me) out = (int)(uint32_t)(x.res[l] >> 8);
  return out;
}
inline void wave_sync(long site) { (void)wave_exchange(0, site); }
inline unsigned lane() { return g_threadIdx.x & 63u; }
inline void block_sync(long site) {
  (void)site;   // (s_barrier counts arrivals: wavefronts may meet at different __syncthreads of the program, e.g. producer / consumer code)
  Block* b = B;
  Wave& w = b->waves[b->cur >> 6];
  const unsigned g = b->gen;
  ++b->arrived;
  ++w.at_barrier;
  while (b->gen == g) {
    if (b->arrived >= b->alive) {
      ++b->gen;
      b->arrived = 0;
      for (Wave& ww : b->waves) ww.at_barrier = 0;
      break;
    }
    if (w.blocked > 0 && w.blocked + w.at_barrier >= w.alive) release_wave(w);   // lanes of this wavefront wait in a collective
    yield();
  }
}

inline void fiber_main() {
  Block* b = B;
  b->body();
  const int me = b->cur;
  b->done[me] = 1;
  --b->alive;
  --b->waves[me >> 6].alive;
#if EMU_TSAN
  __tsan_switch_to_fiber(b->tsched, 0);
#endif
#if EMU_FAST_SWITCH
  void* dead = nullptr;
  emu_ctx_switch(&dead, b->sched_sp);   // never resumed
  abort();
#endif
  // (ucontext: returns to the scheduler through uc_link)
}

// runs body() for every work-item of a grid; blocks one after the other, x fastest
inline void launch(dim3 grid, dim3 block, const std::function<void()>& body) {
  static thread_local Block blk;
  Block* b = &blk;
  B = b;
  const unsigned nt = block.x;
  if (block.y != 1 || block.z != 1 || (nt & 63u)) fail("block shape not supported", 0);
  while (b->stacks.size() < nt) b->stacks.push_back((char*)malloc(kStack));
  b->ctx.resize(nt);
  b->sp.resize(nt);
#if EMU_TSAN
  b->tsched = __tsan_get_current_fiber();
  while (b->tfib.size() < nt) b->tfib.push_back(nullptr);
#endif
  b->done.assign(nt, 0);
  b->waves.assign(nt / 64, Wave());
  b->body = body;
  g_blockDim = block;
  g_gridDim = grid;
  for (unsigned bz = 0; bz < grid.z; ++bz)
    for (unsigned by = 0; by < grid.y; ++by)
      for (unsigned bx = 0; bx < grid.x; ++bx) {
        g_blockIdx = dim3(bx, by, bz);
        b->alive = (int)nt;
        b->arrived = 0;
        b->gen = 0;
        for (unsigned w = 0; w < nt / 64; ++w) {
          Wave& ww = b->waves[w];
          ww.alive = 64;
          ww.blocked = ww.at_barrier = ww.n_sites = 0;
        }
        for (unsigned t = 0; t < nt; ++t) {
          b->done[t] = 0;
#if EMU_TSAN
          // a work-item leaves its fiber without returning through the frames it entered: the sanitizer's shadow call stack of
          // a fiber that is used again keeps them (64 Ki entries: it overflows after some thousand work-items).  A fresh
          // fiber every 256 uses.
          if (b->tfib_uses.size() < b->tfib.size()) b->tfib_uses.resize(b->tfib.size(), 0u);
          if (!b->tfib[t] || ++b->tfib_uses[t] >= 256u) {
            if (b->tfib[t]) __tsan_destroy_fiber(b->tfib[t]);
            b->tfib[t] = __tsan_create_fiber(0);
            b->tfib_uses[t] = 0u;
          }
#endif
#if defined(__has_feature)
#if __has_feature(address_sanitizer)
          __asan_unpoison_memory_region(b->stacks[t], kStack);   // (frames abandoned by the previous fiber on this stack)
#endif
#endif
#if EMU_FAST_SWITCH
          // the first switch to the fiber pops six zeroed registers and "returns" into fiber_main with the stack
          // aligned as after a call
          uintptr_t top = ((uintptr_t)(b->stacks[t] + kStack) & ~(uintptr_t)15) - 16;
          void** frame = (void**)top;
          frame[1] = nullptr;                    // fiber_main's (unused) return address
          frame[0] = (void*)&fiber_main;
          for (int k = 1; k <= 6; ++k) frame[-k] = nullptr;
          b->sp[t] = (void*)(frame - 6);
#else
          getcontext(&b->ctx[t]);
          b->ctx[t].uc_stack.ss_sp = b->stacks[t];
          b->ctx[t].uc_stack.ss_size = kStack;
          b->ctx[t].uc_link = &b->sched;
          makecontext(&b->ctx[t], (void (*)())fiber_main, 0);
#endif
        }
        unsigned long long idle_passes = 0;
        while (b->alive > 0) {
          const unsigned long long before = b->switches;
          const int alive_before = b->alive;
          for (unsigned t = 0; t < nt; ++t) {
            if (b->done[t]) continue;
            b->cur = (int)t;
            g_threadIdx = dim3(t, 0, 0);
#if EMU_TSAN
            __tsan_switch_to_fiber(b->tfib[t], 0);
#endif
#if EMU_FAST_SWITCH
            emu_ctx_switch(&b->sched_sp, b->sp[t]);
#else
            swapcontext(&b->sched, &b->ctx[t]);
#endif
          }
          (void)before;
          if (b->alive == alive_before) {
            if (++idle_passes > 1000000ull) fail("no work-item finished in 1e6 scheduler passes (deadlock?)", 0);
          } else {
            idle_passes = 0;
          }
        }
      }
}
// EMU_PROFILE=1: wall time per kernel name, printed at exit
struct ProfEntry { const char* name; double s; unsigned long long launches, threads; };
inline std::vector<ProfEntry>& prof_table() { static std::vector<ProfEntry> t; return t; }
inline void prof_dump() {
  for (const ProfEntry& e : prof_table()) fprintf(stderr, "emu: %9.3f s  %6llu launches  %10llu work-items  %s\n", e.s, e.launches, e.threads, e.name);
}
inline void launch_named(const char* name, dim3 grid, dim3 block, const std::function<void()>& body) {
  static const bool on = getenv("EMU_PROFILE") != nullptr;
  static const bool trace = getenv("EMU_TRACE") != nullptr;   // EMU_TRACE=1: every launch, before it runs (which kernel was it?)
  if (trace) fprintf(stderr, "emu: launch %s grid %u x %u block %u\n", name, grid.x, grid.y, block.x);
  if (!on) { launch(grid, block, body); return; }
  static bool registered = false;
  if (!registered) { registered = true; atexit(prof_dump); }
  timespec a, b;
  clock_gettime(CLOCK_MONOTONIC, &a);
  launch(grid, block, body);
  clock_gettime(CLOCK_MONOTONIC, &b);
  const double dt = (double)(b.tv_sec - a.tv_sec) + 1e-9 * (double)(b.tv_nsec - a.tv_nsec);
  for (ProfEntry& e : prof_table())
    if (e.name == name) { e.s += dt; ++e.launches; e.threads += (unsigned long long)grid.x * grid.y * grid.z * block.x; return; }
  prof_table().push_back(ProfEntry{name, dt, 1, (unsigned long long)grid.x * grid.y * grid.z * block.x});
}
}  // namespace emu

#define threadIdx emu::g_threadIdx
#define blockIdx emu::g_blockIdx
#define blockDim emu::g_blockDim
#define gridDim emu::g_gridDim

// a collective site = source file + line
#define EMU_SITE ((long)(((uintptr_t)(const void*)__FILE__) * 1000003ul) ^ (long)__LINE__)
#define __syncthreads() emu::block_sync(EMU_SITE)
#define __ballot(emu_p) emu::ballot((emu_p), EMU_SITE)
#define __shfl(emu_v, emu_src) emu::shfl_from((emu_v), (int)(emu_src), EMU_SITE)
#define __shfl_xor(emu_v, emu_m) emu::shfl_from((emu_v), (int)(emu::lane() ^ (unsigned)(emu_m)), EMU_SITE)
#define __shfl_up(emu_v, emu_d) emu::shfl_from((emu_v), (int)emu::lane() - (int)(emu_d), EMU_SITE)
#define __shfl_down(emu_v, emu_d) emu::shfl_from((emu_v), (int)emu::lane() + (int)(emu_d), EMU_SITE)
#define __builtin_amdgcn_readlane(emu_v, emu_k) emu::shfl_from((emu_v), (int)(emu_k), EMU_SITE)
#define __builtin_amdgcn_readfirstlane(emu_v) emu::first_lane((emu_v), EMU_SITE)
#define __builtin_amdgcn_ds_bpermute(emu_addr, emu_v) emu::shfl_from((emu_v), (int)(((unsigned)(emu_addr)) >> 2) & 63, EMU_SITE)
#define __builtin_amdgcn_ds_permute(emu_addr, emu_v) emu::permute_fwd((int)(((unsigned)(emu_addr)) >> 2) & 63, (int)(emu_v), EMU_SITE)
#define __builtin_amdgcn_wave_barrier() emu::wave_sync(EMU_SITE)
#define KS_WAVE_LDS_ORDER() emu::wave_sync(EMU_SITE)
// DPP wave_shr:1 — lane k reads lane k-1's value, lane 0 (no source lane) keeps `first` (bound_ctrl off)
#define KS_LANE_BELOW(first_i, x_i) (emu::lane() == 0u ? ((void)emu::shfl_from((int)(x_i), 0, EMU_SITE), (int)(first_i)) : emu::shfl_from((int)(x_i), (int)emu::lane() - 1, EMU_SITE))
#define KS_WAIT_VMEM()
// sum over the wavefront, in every lane (ks_k_apply_xl.h: six DPP adds on the GPU)
inline int emu_wave_sum_i32(int v) {
  for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
#define KS_WAVE_SUM_I32(v) emu_wave_sum_i32(v)
#define KS_WAIT_LOADS()
#define KS_VALUE_BARRIER(x) (void)(x)

inline int __popcll(unsigned long long x) { return __builtin_popcountll(x); }
inline int __popc(unsigned x) { return __builtin_popcount(x); }
inline int __ffsll(long long x) { return __builtin_ffsll(x); }
inline int __ffs(int x) { return __builtin_ffs(x); }
inline int __clzll(long long x) { return x == 0 ? 64 : __builtin_clzll((unsigned long long)x); }
inline int __clz(int x) { return x == 0 ? 32 : __builtin_clz((unsigned)x); }
inline float __int_as_float(int x) { return emu::from_bits<float>((uint64_t)(uint32_t)x); }
inline int __float_as_int(float x) { return (int)(uint32_t)emu::to_bits(x); }
inline unsigned __float_as_uint(float x) { return (uint32_t)emu::to_bits(x); }
inline float __uint_as_float(unsigned x) { return emu::from_bits<float>((uint64_t)x); }

template <typename T> struct emu_same { typedef T type; };
template <typename T> inline T atomicMax(T* p, typename emu_same<T>::type v) { const T o = *p; if (v > o) *p = v; return o; }
template <typename T> inline T atomicMin(T* p, typename emu_same<T>::type v) { const T o = *p; if (v < o) *p = v; return o; }
template <typename T> inline T atomicAdd(T* p, typename emu_same<T>::type v) { const T o = *p; *p = o + v; return o; }
template <typename T> inline T atomicOr(T* p, typename emu_same<T>::type v) { const T o = *p; *p = o | v; return o; }
template <typename T> inline T atomicAnd(T* p, typename emu_same<T>::type v) { const T o = *p; *p = o & v; return o; }
template <typename T> inline T atomicExch(T* p, typename emu_same<T>::type v) { const T o = *p; *p = v; return o; }
template <typename T> inline T atomicCAS(T* p, typename emu_same<T>::type cmp, typename emu_same<T>::type v) { const T o = *p; if (o == cmp) *p = v; return o; }
#include <type_traits>
template <typename A, typename B2, typename = typename std::enable_if<std::is_arithmetic<A>::value && std::is_arithmetic<B2>::value>::type>
inline typename std::common_type<A, B2>::type max(A a, B2 b) { typedef typename std::common_type<A, B2>::type T; return (T)a < (T)b ? (T)b : (T)a; }
template <typename A, typename B2, typename = typename std::enable_if<std::is_arithmetic<A>::value && std::is_arithmetic<B2>::value>::type>
inline typename std::common_type<A, B2>::type min(A a, B2 b) { typedef typename std::common_type<A, B2>::type T; return (T)b < (T)a ? (T)b : (T)a; }
inline float __builtin_amdgcn_fmed3f_emu(float a, float b, float c) {   // v_med3_f32: a NaN operand -> min3 of the others
  if (a != a || b != b || c != c) {
    float m = __builtin_inff();
    if (a == a && a < m) m = a;
    if (b == b && b < m) m = b;
    if (c == c && c < m) m = c;
    return m;
  }
  const float lo = a < b ? a : b, hi = a < b ? b : a;
  return c < lo ? lo : (c > hi ? hi : c);
}
#define __builtin_amdgcn_fmed3f(a, b, c) __builtin_amdgcn_fmed3f_emu((a), (b), (c))
inline void __threadfence() {}   // (blocks run one after the other, lanes between two rendezvous points too: nothing to order)
#define __hip_atomic_load(p, order, scope) (*(p))
#define __hip_atomic_store(p, v, order, scope) (*(p) = (v))
#define __hip_atomic_fetch_max(p, v, order, scope) emu_fetch_max((p), (v))
namespace emu { inline void (*global_atomic_hook)(const void*) = nullptr; }   // statistics of a test harness (contention on an address)
template <typename P, typename T> inline T emu_fetch_max(P p, T v) {
  if (emu::global_atomic_hook) emu::global_atomic_hook((const void*)p);
  const T o = *p; if (v > o) *p = v; return o;
}


// ---------------------------------------------------------------------------------------------------------------
// The slice of the HIP runtime API that kimera_semantics_amd/csrc/ks_hip.hip uses, for the host functional model:
// device memory is host memory, a launch runs to completion before it returns (streams and events are no-ops),
// stream capture is refused (the library then launches directly).
// ---------------------------------------------------------------------------------------------------------------
typedef enum { hipSuccess = 0, hipErrorInvalidValue = 1, hipErrorOutOfMemory = 2, hipErrorNotSupported = 801, hipErrorUnknown = 999 } hipError_t;
#include <atomic>
// Streams and events carry one atomic each: every operation on a stream (or event) is an acquire-release read-modify-write
// of it, so that a ThreadSanitizer build sees the happens-before edges the program asks the GPU for — operations of one
// stream are ordered, hipEventRecord / hipStreamWaitEvent / *Synchronize carry order across streams — and nothing else: a
// buffer handed from one host thread's launches to another's without such an edge is reported as the race it would be on
// the GPU.  (Launches of ONE host thread run one after the other here: races between streams fed by the same thread
// cannot be seen.)
struct ihipStream_t { std::atomic<long> sync{0}; };
struct ihipEvent_t { std::atomic<long> sync{0}; };
namespace emu {
inline ihipStream_t g_null_stream;
inline void touch(ihipStream_t* s) { (s ? s : &g_null_stream)->sync.fetch_add(1, std::memory_order_acq_rel); }
inline void touch(ihipEvent_t* e) { if (e) e->sync.fetch_add(1, std::memory_order_acq_rel); }
}  // namespace emu
typedef ihipStream_t* hipStream_t;
typedef ihipEvent_t* hipEvent_t;
typedef struct ihipGraph* hipGraph_t;
typedef struct ihipGraphExec* hipGraphExec_t;
typedef enum { hipMemcpyHostToHost = 0, hipMemcpyHostToDevice = 1, hipMemcpyDeviceToHost = 2, hipMemcpyDeviceToDevice = 3, hipMemcpyDefault = 4 } hipMemcpyKind;
typedef enum { hipStreamCaptureModeGlobal = 0, hipStreamCaptureModeThreadLocal = 1, hipStreamCaptureModeRelaxed = 2 } hipStreamCaptureMode;
typedef enum { hipMemoryTypeUnregistered = 0, hipMemoryTypeHost = 1, hipMemoryTypeDevice = 2, hipMemoryTypeManaged = 3 } hipMemoryType;
struct hipPointerAttribute_t { hipMemoryType type; int device; void* devicePointer; void* hostPointer; int isManaged; unsigned allocationFlags; };
#define hipStreamNonBlocking 1u
#define hipEventDisableTiming 2u
#define hipHostMallocDefault 0u
#define HIP_SYMBOL(x) (x)
inline const char* hipGetErrorString(hipError_t e) { return e == hipSuccess ? "hipSuccess" : "hip error (host functional model)"; }
inline hipError_t hipGetLastError() { return hipSuccess; }
inline hipError_t hipGetDeviceCount(int* n) { *n = 1; return hipSuccess; }
inline hipError_t hipSetDevice(int) { return hipSuccess; }
inline hipError_t hipDeviceGetStreamPriorityRange(int* lo, int* hi) { *lo = 0; *hi = 0; return hipSuccess; }
inline hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) { *s = new ihipStream_t; return hipSuccess; }
inline hipError_t hipStreamCreateWithPriority(hipStream_t* s, unsigned, int) { *s = new ihipStream_t; return hipSuccess; }
inline hipError_t hipStreamDestroy(hipStream_t s) { delete s; return hipSuccess; }
inline hipError_t hipStreamSynchronize(hipStream_t s) { emu::touch(s); return hipSuccess; }
inline hipError_t hipStreamWaitEvent(hipStream_t s, hipEvent_t e, unsigned) { emu::touch(e); emu::touch(s); return hipSuccess; }
inline hipError_t hipStreamBeginCapture(hipStream_t, hipStreamCaptureMode) { return hipErrorNotSupported; }
inline hipError_t hipStreamEndCapture(hipStream_t, hipGraph_t* g) { *g = nullptr; return hipErrorNotSupported; }
inline hipError_t hipGraphInstantiate(hipGraphExec_t*, hipGraph_t, void*, void*, size_t) { return hipErrorNotSupported; }
inline hipError_t hipGraphLaunch(hipGraphExec_t, hipStream_t) { return hipErrorNotSupported; }
inline hipError_t hipGraphDestroy(hipGraph_t) { return hipSuccess; }
inline hipError_t hipGraphExecDestroy(hipGraphExec_t) { return hipSuccess; }
inline hipError_t hipEventCreate(hipEvent_t* e) { *e = new ihipEvent_t; return hipSuccess; }
inline hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { *e = new ihipEvent_t; return hipSuccess; }
inline hipError_t hipEventDestroy(hipEvent_t e) { delete e; return hipSuccess; }
inline hipError_t hipEventRecord(hipEvent_t e, hipStream_t s) { emu::touch(s); emu::touch(e); return hipSuccess; }
inline hipError_t hipEventSynchronize(hipEvent_t e) { emu::touch(e); return hipSuccess; }
inline hipError_t hipEventElapsedTime(float* ms, hipEvent_t, hipEvent_t) { *ms = 0.f; return hipSuccess; }
// (exactly `bytes`, 256-byte aligned like the device allocator: a sanitizer build sees every overrun)
inline void* emu_alloc(size_t bytes) { void* p = nullptr; return posix_memalign(&p, 256, bytes ? bytes : 1) == 0 ? p : nullptr; }
template <typename T> inline hipError_t hipMalloc(T** p, size_t bytes) { *p = (T*)emu_alloc(bytes); return *p ? hipSuccess : hipErrorOutOfMemory; }
template <typename T> inline hipError_t hipHostMalloc(T** p, size_t bytes, unsigned = 0) { *p = (T*)emu_alloc(bytes); return *p ? hipSuccess : hipErrorOutOfMemory; }
inline hipError_t hipFree(void* p) { free(p); return hipSuccess; }
inline hipError_t hipHostFree(void* p) { free(p); return hipSuccess; }
inline hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) { emu::touch((ihipStream_t*)nullptr); memmove(d, s, n); emu::touch((ihipStream_t*)nullptr); return hipSuccess; }
inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t st = nullptr) { emu::touch(st); memmove(d, s, n); emu::touch(st); return hipSuccess; }
inline hipError_t hipMemset(void* d, int v, size_t n) { emu::touch((ihipStream_t*)nullptr); memset(d, v, n); emu::touch((ihipStream_t*)nullptr); return hipSuccess; }
inline hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t st = nullptr) { emu::touch(st); memset(d, v, n); emu::touch(st); return hipSuccess; }
typedef void* hipDeviceptr_t;
inline hipError_t hipMemsetD32Async(hipDeviceptr_t d, int v, size_t count, hipStream_t st = nullptr) {
  emu::touch(st);
  for (size_t i = 0; i < count; ++i) ((int*)d)[i] = v;
  emu::touch(st);
  return hipSuccess;
}
template <typename S> inline hipError_t hipMemcpyFromSymbol(void* d, const S& sym, size_t n) { memcpy(d, &sym, n); return hipSuccess; }
template <typename S> inline hipError_t hipMemcpyToSymbol(S& sym, const void* s, size_t n) { memcpy(&sym, s, n); return hipSuccess; }
inline hipError_t hipPointerGetAttributes(hipPointerAttribute_t* a, const void* p) {   // every allocation is "pinned host memory the device can write"
  a->type = hipMemoryTypeHost; a->device = 0; a->devicePointer = (void*)p; a->hostPointer = (void*)p; a->isManaged = 0; a->allocationFlags = 0;
  return hipSuccess;
}
#define hipLaunchKernelGGL(kernel, grid, block, lds, stream, ...) \
  (emu::touch(stream), emu::launch_named(#kernel, dim3(grid), dim3(block), [&] { kernel(__VA_ARGS__); }), emu::touch(stream))
#define hipExtLaunchKernelGGL(kernel, grid, block, lds, stream, ev0, ev1, flags, ...) \
  (emu::touch(stream), emu::launch_named(#kernel, dim3(grid), dim3(block), [&] { kernel(__VA_ARGS__); }), emu::touch(stream))
