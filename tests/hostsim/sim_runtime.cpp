// The few services of core.hip the simulated translation units need (see hip/hip_runtime.h: test infrastructure only):
// a context, error text, the table scratch, and stubs for the entry points whose kernels use LDS / wave intrinsics.
#include <stdarg.h>
#include <stdio.h>

#include <sys/mman.h>

#include <vector>

#define SIM_FIBERS 1
#include "common.h"

thread_local SimIdx blockIdx, threadIdx, gridDim, blockDim;

// ---- SIMT on fibers (simt.h) -------------------------------------------------------------------------------------------------
thread_local SimBlockState g_sim;
static constexpr size_t SIM_STACK = (size_t)1 << 20;   // per fiber; reserved, committed on touch

static void sim_trampoline() {
  g_sim.entry();
  g_sim.fibers[g_sim.cur].state = 3;
  swapcontext(&g_sim.fibers[g_sim.cur].ctx, &g_sim.sched);
}

void sim_run_block(size_t nthreads, const SimIdx& bdim) {
  if (g_sim.fibers.size() < nthreads) g_sim.fibers.resize(nthreads);
  if (!g_sim.dyn_lds) g_sim.dyn_lds = (unsigned char*)calloc(1, 160 * 1024);
  g_sim.slots.assign(((nthreads + 63) / 64) * 64, 0);
  for (size_t i = 0; i < nthreads; ++i) {
    SimFiber& f = g_sim.fibers[i];
    if (!f.stack) {
      f.stack = (char*)mmap(nullptr, SIM_STACK, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
      if (f.stack == MAP_FAILED) { fprintf(stderr, "host simulation: no stack for fiber %zu\n", i); abort(); }
    }
    getcontext(&f.ctx);
    f.ctx.uc_stack.ss_sp = f.stack;
    f.ctx.uc_stack.ss_size = SIM_STACK;
    f.ctx.uc_link = nullptr;
    makecontext(&f.ctx, sim_trampoline, 0);
    f.state = 0;
    f.site = nullptr;
    f.orgen = 0;
    f.tid = {(unsigned)(i % bdim.x), (unsigned)((i / bdim.x) % bdim.y), (unsigned)(i / ((size_t)bdim.x * bdim.y))};
  }
  const size_t nwaves = (nthreads + 63) / 64;
  for (;;) {
    bool ran = false;
    size_t done = 0;
    for (size_t i = 0; i < nthreads; ++i) {
      SimFiber& f = g_sim.fibers[i];
      if (f.state == 3) { ++done; continue; }
      if (f.state != 0) continue;
      g_sim.cur = (int)i;
      threadIdx = f.tid;
      swapcontext(&g_sim.sched, &f.ctx);
      ran = true;
      if (f.state == 3) ++done;
    }
    if (done == nthreads) break;
    bool released = false;
    // wave exchanges: all live lanes of a wave at the SAME call
    for (size_t w = 0; w < nwaves; ++w) {
      const size_t a = w * 64, b = a + 64 < nthreads ? a + 64 : nthreads;
      const void* site = nullptr;
      bool all = true, any = false, mixed = false;
      for (size_t i = a; i < b; ++i) {
        const SimFiber& f = g_sim.fibers[i];
        if (f.state == 3) continue;
        if (f.state != 2) { all = false; continue; }
        if (any && f.site != site) mixed = true;
        site = f.site;
        any = true;
      }
      if (any && all && !mixed) {
        for (size_t i = a; i < b; ++i)
          if (g_sim.fibers[i].state == 2) g_sim.fibers[i].state = 0;
        released = true;
      } else if (any && all && mixed) {
        fprintf(stderr, "host simulation: the live lanes of wave %zu wait in different wave exchanges (a collective under divergent "
                        "control flow): this kernel cannot be simulated on fibers\n", w);
        abort();
      }
    }
    // workgroup barrier: every live thread arrived
    bool all_bar = true, any_bar = false;
    for (size_t i = 0; i < nthreads; ++i) {
      const int st = g_sim.fibers[i].state;
      if (st == 3) continue;
      if (st == 1) any_bar = true; else all_bar = false;
    }
    if (any_bar && all_bar) {
      for (size_t i = 0; i < nthreads; ++i)
        if (g_sim.fibers[i].state == 1) g_sim.fibers[i].state = 0;
      released = true;
    }
    if (!ran && !released) {
      fprintf(stderr, "host simulation: deadlock (threads wait at a barrier / exchange that the others never reach)\n");
      abort();
    }
  }
}
static char g_err[1024];

void xh_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof g_err, fmt, ap);
  va_end(ap);
}
const char* xh_diag_env(const char* name) {   // (as core.hip: diagnostic switches count only under XH_DIAGNOSTICS=1)
  const char* on = getenv("XH_DIAGNOSTICS");
  if (!on || on[0] != '1') return nullptr;
  return getenv(name);
}

int xh_scratch_upload(xh_ctx* ctx, size_t* cursor, const void* host, size_t bytes, void** dptr) {
  const size_t off = (*cursor + 255) & ~(size_t)255;
  if (off + bytes > ctx->scratch_bytes) { xh_set_error("host simulation: table scratch exhausted"); return XH_ERR_LIMIT; }
  memcpy((char*)ctx->scratch + off, host, bytes);
  *dptr = (char*)ctx->scratch + off;
  *cursor = off + bytes;
  return XH_OK;
}
int xh_big_scratch(xh_ctx* ctx, size_t bytes, void** dptr) {
  if (bytes > ctx->big_bytes) { free(ctx->big); ctx->big = malloc(bytes); ctx->big_bytes = bytes; }
  *dptr = ctx->big;
  return XH_OK;
}
int xh_const_rows(xh_ctx* ctx, int64_t elems, const float** nan_row, const float** ninf_row, const float** pinf_row) {
  const size_t bytes = ((size_t)elems * 4 + 255) & ~(size_t)255;
  if (bytes > ctx->nanrow_bytes) {
    free(ctx->nanrow);
    ctx->nanrow = malloc(3 * bytes);
    ctx->nanrow_bytes = bytes;
    float* p = (float*)ctx->nanrow;
    const size_t n = bytes / 4;
    for (size_t i = 0; i < n; ++i) { p[i] = NAN; p[n + i] = -INFINITY; p[2 * n + i] = INFINITY; }
  }
  const char* p = (const char*)ctx->nanrow;
  if (nan_row) *nan_row = (const float*)p;
  if (ninf_row) *ninf_row = (const float*)(p + ctx->nanrow_bytes);
  if (pinf_row) *pinf_row = (const float*)(p + 2 * ctx->nanrow_bytes);
  return XH_OK;
}
// qdm3.hip (rocPRIM's segmented sort + the rank kernels behind it) is the one compute unit that is not simulated: its two entry
// points exist for the linker and refuse
struct QTab;
int xh_qdm_sorted_ws(int64_t, int64_t, size_t* bytes) { *bytes = 0; return XH_ERR_NOTIMPL; }
int xh_qdm_sorted(xh_ctx*, const float*, int64_t, int64_t, int64_t, const float*, int64_t, const double*, int, int, int, int, float*, int64_t, void*) {
  xh_set_error("host simulation: the global-sort rank kernels (rocPRIM) are not simulated");
  return XH_ERR_LIMIT;
}


extern "C" {
const char* xh_last_error(void) { return g_err; }
int xh_create(int device, xh_ctx** out) {
  xh_ctx* c = (xh_ctx*)calloc(1, sizeof(xh_ctx));
  c->device = device;
  c->num_cu = 2;   // (small grids: every thread is a loop iteration here)
  c->scratch_bytes = (size_t)64 << 20;
  c->scratch = malloc(c->scratch_bytes);
  *out = c;
  return XH_OK;
}
int xh_destroy(xh_ctx* c) { if (c) { free(c->scratch); free(c->big); free(c); } return XH_OK; }
int xh_sync(xh_ctx*) { return XH_OK; }
int xh_host_alloc(xh_ctx*, size_t bytes, void** out) { *out = malloc(bytes ? bytes : 1); return *out ? XH_OK : XH_ERR_HIP; }
int xh_host_free(xh_ctx*, void* p) { free(p); return XH_OK; }
}
