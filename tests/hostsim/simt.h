// SIMT on fibers — part of the host stand-in for the HIP runtime (hip/hip_runtime.h: TEST INFRASTRUCTURE ONLY).  Units compiled
// with -DSIM_FIBERS run every workgroup as a set of ucontext fibers, one per thread, so that kernels that talk through LDS
// (__syncthreads) and through the wave (shuffles, votes, readlane) can run on the CPU:
//   __syncthreads()            the fiber yields until every live thread of the workgroup has arrived
//   __shfl* / __ballot / __any / __all / readlane / readfirstlane / wave_barrier
//                              a WAVE-UNIFORM exchange: every live lane of the 64-lane wave must arrive at the same call; a wave
//                              whose live lanes wait in different calls is reported and the process aborts — collectives under
//                              divergent control flow have no faithful emulation without the compiler's reconvergence points
//                              (plane.hip's divergent appends use the wave-of-one build instead)
//   atomics                    plain read-modify-write (one fiber runs at a time)
// LDS: static arrays (-D__shared__=static) and, for `extern __shared__` declarations, a per-workgroup buffer (sim_dynamic_lds;
// tests/hostsim/simdevice.py rewrites those declarations, the only change made to a source).
#pragma once
#include <stdio.h>
#include <ucontext.h>

#include <functional>
#include <vector>

struct SimFiber {
  ucontext_t ctx;
  char* stack = nullptr;
  int state = 0;  // 0 runnable, 1 at the workgroup barrier, 2 at a wave exchange, 3 done
  SimIdx tid{0, 0, 0};
  const void* site = nullptr;
  int orgen = 0;  // calls of __syncthreads_or so far (selects one of two flags)
};

struct SimBlockState {
  std::vector<SimFiber> fibers;
  ucontext_t sched;
  int cur = -1;
  std::vector<uint64_t> slots;  // [wave][64] exchange values
  std::function<void()> entry;
  unsigned char* dyn_lds = nullptr;
  int or_acc[2] = {0, 0};
  std::vector<uint32_t> pub[4];  // registers a wave made readable for v_readlane from divergent code (sim_publish / sim_peek)
};
extern thread_local SimBlockState g_sim;

static inline void sim_yield(int state, const void* site) {
  SimFiber& f = g_sim.fibers[g_sim.cur];
  f.state = state;
  f.site = site;
  swapcontext(&f.ctx, &g_sim.sched);
}
static inline unsigned sim_lane() { return (unsigned)g_sim.cur & 63u; }
static inline uint64_t* sim_wave_slots() { return g_sim.slots.data() + ((size_t)g_sim.cur >> 6) * 64; }
static inline unsigned char* sim_dynamic_lds() { return g_sim.dyn_lds; }

// v_readlane from DIVERGENT code (hardware: a scalar read of another lane's register, whatever the execution mask): the register
// is published once at a point where the wave is together, then peeked without an exchange (select4.hip's hs_qdm_pick, rewritten
// by simdevice.py)
static inline void sim_publish(int slot, uint32_t v) {
  if (g_sim.pub[slot].size() < g_sim.fibers.size()) g_sim.pub[slot].resize(g_sim.fibers.size());
  g_sim.pub[slot][g_sim.cur] = v;
}
static inline uint32_t sim_peek(int slot, int lane) { return g_sim.pub[slot][((size_t)g_sim.cur & ~(size_t)63) + (size_t)lane]; }

static inline void __syncthreads() { sim_yield(1, nullptr); }
// the workgroup barrier that also ORs a predicate: deposit, barrier, read, barrier; two flags used alternately (a thread that leaves
// clears the flag of THIS call while the quick ones already deposit into the other one for the next call)
static inline int __syncthreads_or(int pred) {
  SimFiber& f = g_sim.fibers[g_sim.cur];
  const int g = f.orgen++ & 1;
  if (pred) g_sim.or_acc[g] = 1;
  sim_yield(1, nullptr);
  const int r = g_sim.or_acc[g];
  sim_yield(1, nullptr);
  g_sim.or_acc[g] = 0;
  return r;
}

// every live lane deposits `v`, all wait, every lane reads what it needs, all wait again (the slots are reused by the next call)
template <typename T, typename Pick>
static inline T sim_exchange(T v, const void* site, Pick pick) {
  static_assert(sizeof(T) <= 8, "exchange of at most 8 bytes");
  uint64_t raw = 0;
  memcpy(&raw, &v, sizeof(T));
  sim_wave_slots()[sim_lane()] = raw;
  sim_yield(2, site);
  const uint64_t got = pick(sim_wave_slots());
  sim_yield(2, (const char*)site + 1);
  T r;
  memcpy(&r, &got, sizeof(T));
  return r;
}
// the lanes of my wave that are still alive (a finished lane deposits nothing)
static inline uint64_t sim_live_mask() {
  uint64_t m = 0;
  const size_t base = ((size_t)g_sim.cur >> 6) * 64;
  for (unsigned l = 0; l < 64 && base + l < g_sim.fibers.size(); ++l)
    if (g_sim.fibers[base + l].state != 3) m |= 1ull << l;
  return m;
}
#define SIM_SITE ([]() -> const void* { static const char here[2] = {0, 0}; return here; }())

// (HIP's segment rule for width < 64: the source stays inside the caller's aligned group of `width` lanes, else the caller's own value)
template <typename T> static inline T sim_shfl(T v, int src, const void* s, int width = 64) {
  const unsigned me = sim_lane(), w = (unsigned)width;
  return sim_exchange(v, s, [=](uint64_t* x) { return x[(((unsigned)src & (w - 1u)) + (me & ~(w - 1u))) & 63u]; });
}
template <typename T> static inline T sim_shfl_xor(T v, int mask, const void* s, int width = 64) {
  const unsigned me = sim_lane(), w = (unsigned)width;
  return sim_exchange(v, s, [=](uint64_t* x) { const unsigned i = me ^ (unsigned)mask; return x[(i >= ((me + w) & ~(w - 1u)) ? me : i) & 63u]; });
}
template <typename T> static inline T sim_shfl_up(T v, unsigned d, const void* s, int width = 64) {
  const unsigned me = sim_lane(), w = (unsigned)width;
  return sim_exchange(v, s, [=](uint64_t* x) { return x[(me >= d && me - d >= (me & ~(w - 1u))) ? me - d : me]; });
}
template <typename T> static inline T sim_shfl_down(T v, unsigned d, const void* s, int width = 64) {
  const unsigned me = sim_lane(), w = (unsigned)width;
  return sim_exchange(v, s, [=](uint64_t* x) { return x[(me + d < ((me + w) & ~(w - 1u))) ? me + d : me]; });
}
static inline uint64_t sim_ballot(int pred, const void* s) {
  const uint64_t live = sim_live_mask();
  return sim_exchange<uint64_t>(pred ? 1 : 0, s, [=](uint64_t* w) { uint64_t m = 0; for (unsigned l = 0; l < 64; ++l) if (((live >> l) & 1) && w[l]) m |= 1ull << l; return m; });
}
template <typename T> static inline T sim_readfirstlane(T v, const void* s) {
  const uint64_t live = sim_live_mask();
  const unsigned first = live ? (unsigned)__builtin_ctzll(live) : 0u;
  return sim_exchange(v, s, [=](uint64_t* w) { return w[first]; });
}
// DPP moves (v_mov_b32_dpp through __builtin_amdgcn_update_dpp): quad_perm selections — the lane pair swap [1,0,3,2] of the register
// sorting networks among them; anything else aborts rather than guess
static inline int sim_update_dpp(int src, int ctrl, const void* s) {
  if (ctrl < 0 || ctrl > 0xFF) { fprintf(stderr, "host simulation: DPP control 0x%x is not emulated\n", ctrl); abort(); }
  const unsigned me = sim_lane(), sel = ((unsigned)ctrl >> (2u * (me & 3u))) & 3u;
  return sim_exchange(src, s, [=](uint64_t* x) { return x[(me & ~3u) | sel]; });
}
#define __builtin_amdgcn_update_dpp(old, src, ctrl, rowmask, bankmask, boundctrl) sim_update_dpp((src), (ctrl), SIM_SITE)
#define __shfl(v, src, ...) sim_shfl((v), (src), SIM_SITE, ##__VA_ARGS__)
#define __shfl_xor(v, mask, ...) sim_shfl_xor((v), (mask), SIM_SITE, ##__VA_ARGS__)
#define __shfl_up(v, d, ...) sim_shfl_up((v), (d), SIM_SITE, ##__VA_ARGS__)
#define __shfl_down(v, d, ...) sim_shfl_down((v), (d), SIM_SITE, ##__VA_ARGS__)
#define __ballot(p) sim_ballot((p) ? 1 : 0, SIM_SITE)
#define __any(p) (sim_ballot((p) ? 1 : 0, SIM_SITE) != 0)
#define __all(p) (sim_ballot((p) ? 0 : 1, SIM_SITE) == 0)
#define __builtin_amdgcn_wave_barrier() ((void)sim_ballot(0, SIM_SITE))
#define __builtin_amdgcn_readfirstlane(v) sim_readfirstlane((v), SIM_SITE)
#define __builtin_amdgcn_readlane(v, l) sim_shfl((v), (l), SIM_SITE)
#define __builtin_amdgcn_sqrtf(x) sqrtf(x)
#define __popcll(x) __builtin_popcountll(x)
#define __popc(x) __builtin_popcount(x)
#define __ffsll(x) __builtin_ffsll(x)
#define __ffs(x) __builtin_ffs(x)
#define __threadfence_block() ((void)0)
#define __clz(x) ((x) ? __builtin_clz(x) : 32)

template <typename T> static inline T atomicAdd(T* p, T v) { const T old = *p; *p = old + v; return old; }
template <typename T> static inline T atomicMax(T* p, T v) { const T old = *p; *p = old > v ? old : v; return old; }
template <typename T> static inline T atomicMin(T* p, T v) { const T old = *p; *p = old < v ? old : v; return old; }
template <typename T> static inline T atomicOr(T* p, T v) { const T old = *p; *p = old | v; return old; }
// v_med3_f32: the median of three; with a NaN among them the hardware returns min3, which skips NaN
static inline float sim_fmed3f(float a, float b, float c) {
  if (a != a || b != b || c != c) return fminf(fminf(a, b), c);
  return fmaxf(fminf(a, b), fminf(fmaxf(a, b), c));
}
#define __builtin_amdgcn_fmed3f(a, b, c) sim_fmed3f((a), (b), (c))
#define __builtin_amdgcn_sched_barrier(mask) ((void)0)

void sim_run_block(size_t nthreads, const SimIdx& bdim);

template <typename F>
static inline void sim_launch_fibers(F body, dim3 grid, dim3 block) {
  gridDim = {grid.x, grid.y, grid.z};
  blockDim = {block.x, block.y, block.z};
  g_sim.entry = body;
  for (unsigned bz = 0; bz < grid.z; ++bz)
    for (unsigned by = 0; by < grid.y; ++by)
      for (unsigned bx = 0; bx < grid.x; ++bx) {
        blockIdx = {bx, by, bz};
        sim_run_block((size_t)block.x * block.y * block.z, blockDim);
      }
}
