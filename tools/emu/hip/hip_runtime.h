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
#include <ucontext.h>

#include <functional>
#include <vector>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __shared__
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
struct uint2 { unsigned x, y; };
inline uint2 make_uint2(unsigned x, unsigned y) { return uint2{x, y}; }

namespace emu {
inline dim3 g_threadIdx, g_blockIdx, g_blockDim, g_gridDim;

struct Wave {
  int alive = 0, arrived = 0, line = 0;
  unsigned gen = 0;
  uint64_t slot[2][64];
  uint64_t contrib[2] = {0, 0};
};
struct Block {
  std::vector<ucontext_t> ctx;
  std::vector<char*> stacks;
  std::vector<char> done;
  std::vector<Wave> waves;
  int alive = 0, arrived = 0, line = 0;
  unsigned gen = 0;
  ucontext_t sched;
  int cur = 0;
  std::function<void()> body;
  unsigned long long switches = 0;
};
inline Block* B = nullptr;
constexpr size_t kStack = 256 * 1024;

inline void yield() {
  Block* b = B;
  const int me = b->cur;
  ++b->switches;
  swapcontext(&b->ctx[me], &b->sched);
}
inline void fail(const char* what, int line) {
  fprintf(stderr, "emu: %s (source line %d, block %u,%u thread %u)\n", what, line, g_blockIdx.x, g_blockIdx.y, g_threadIdx.x);
  abort();
}
// every live lane of the calling wavefront contributes v; returns the parity of the exchange buffer to read
inline int wave_exchange(uint64_t v, int line) {
  Block* b = B;
  Wave& w = b->waves[b->cur >> 6];
  const int lane = b->cur & 63;
  const unsigned g = w.gen;
  const int par = (int)(g & 1u);
  if (w.arrived == 0) {
    w.line = line;
    w.contrib[par] = 0;
  } else if (w.line != line) {
    fail("wave collective reached from different source lines (divergent control flow)", line);
  }
  w.slot[par][lane] = v;
  w.contrib[par] |= 1ull << lane;
  ++w.arrived;
  while (w.gen == g) {
    if (w.arrived == w.alive) {
      ++w.gen;
      w.arrived = 0;
      break;
    }
    yield();
  }
  return par;
}
inline unsigned long long ballot(bool p, int line) {
  const int par = wave_exchange(p ? 1u : 0u, line);
  const Wave& w = B->waves[B->cur >> 6];
  unsigned long long m = 0;
  for (int l = 0; l < 64; ++l)
    if (((w.contrib[par] >> l) & 1ull) && (w.slot[par][l] & 1ull)) m |= 1ull << l;
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
inline T shfl_from(T x, int src, int line) {   // src = absolute lane; out of range / not participating: own value
  const int par = wave_exchange(to_bits(x), line);
  const Wave& w = B->waves[B->cur >> 6];
  if (src < 0 || src > 63 || !((w.contrib[par] >> src) & 1ull)) return x;
  return from_bits<T>(w.slot[par][src]);
}
inline unsigned lane() { return g_threadIdx.x & 63u; }
inline void wave_sync(int line) { (void)wave_exchange(0, line); }
inline void block_sync(int line) {
  Block* b = B;
  const unsigned g = b->gen;
  if (b->arrived == 0) b->line = line;
  else if (b->line != line) fail("__syncthreads reached from different source lines", line);
  ++b->arrived;
  while (b->gen == g) {
    if (b->arrived == b->alive) {
      ++b->gen;
      b->arrived = 0;
      break;
    }
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
  // returns to the scheduler through uc_link
}

// runs body() for every work-item of a grid; blocks one after the other, x fastest
inline void launch(dim3 grid, dim3 block, const std::function<void()>& body) {
  static Block blk;
  Block* b = &blk;
  B = b;
  const unsigned nt = block.x;
  if (block.y != 1 || block.z != 1 || (nt & 63u)) fail("block shape not supported", 0);
  while (b->stacks.size() < nt) b->stacks.push_back((char*)malloc(kStack));
  b->ctx.resize(nt);
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
          b->waves[w] = Wave();
          b->waves[w].alive = 64;
        }
        for (unsigned t = 0; t < nt; ++t) {
          b->done[t] = 0;
          getcontext(&b->ctx[t]);
          b->ctx[t].uc_stack.ss_sp = b->stacks[t];
          b->ctx[t].uc_stack.ss_size = kStack;
          b->ctx[t].uc_link = &b->sched;
          makecontext(&b->ctx[t], (void (*)())fiber_main, 0);
        }
        unsigned long long idle_passes = 0;
        while (b->alive > 0) {
          const unsigned long long before = b->switches;
          const int alive_before = b->alive;
          for (unsigned t = 0; t < nt; ++t) {
            if (b->done[t]) continue;
            b->cur = (int)t;
            g_threadIdx = dim3(t, 0, 0);
            swapcontext(&b->sched, &b->ctx[t]);
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
}  // namespace emu

#define threadIdx emu::g_threadIdx
#define blockIdx emu::g_blockIdx
#define blockDim emu::g_blockDim
#define gridDim emu::g_gridDim

#define __syncthreads() emu::block_sync(__LINE__)
#define __ballot(emu_p) emu::ballot((emu_p), __LINE__)
#define __shfl(emu_v, emu_src) emu::shfl_from((emu_v), (int)(emu_src), __LINE__)
#define __shfl_xor(emu_v, emu_m) emu::shfl_from((emu_v), (int)(emu::lane() ^ (unsigned)(emu_m)), __LINE__)
#define __shfl_up(emu_v, emu_d) emu::shfl_from((emu_v), (int)emu::lane() - (int)(emu_d), __LINE__)
#define __shfl_down(emu_v, emu_d) emu::shfl_from((emu_v), (int)emu::lane() + (int)(emu_d), __LINE__)
#define __builtin_amdgcn_readlane(emu_v, emu_k) emu::shfl_from((emu_v), (int)(emu_k), __LINE__)
#define KS_WAVE_LDS_ORDER() emu::wave_sync(__LINE__)
#define KS_WAIT_VMEM()

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
inline int max(int a, int b) { return a < b ? b : a; }
inline int min(int a, int b) { return b < a ? b : a; }
inline unsigned max(unsigned a, unsigned b) { return a < b ? b : a; }
inline unsigned min(unsigned a, unsigned b) { return b < a ? b : a; }
#define __hip_atomic_load(p, order, scope) (*(p))
#define __hip_atomic_store(p, v, order, scope) (*(p) = (v))
#define __hip_atomic_fetch_max(p, v, order, scope) emu_fetch_max((p), (v))
template <typename P, typename T> inline T emu_fetch_max(P p, T v) { const T o = *p; if (v > o) *p = v; return o; }
