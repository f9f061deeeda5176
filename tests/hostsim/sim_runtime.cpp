// The few services of core.hip the simulated translation units need (see hip/hip_runtime.h: test infrastructure only):
// a context, error text, the table scratch, and stubs for the entry points whose kernels use LDS / wave intrinsics.
#include <stdarg.h>
#include <stdio.h>

#include <vector>

#include "common.h"

thread_local SimIdx blockIdx, threadIdx, gridDim, blockDim;
static char g_err[1024];

void xh_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof g_err, fmt, ap);
  va_end(ap);
}
const char* xh_diag_env(const char*) { return nullptr; }

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
// kernels with LDS / wave intrinsics are not simulated: the callers' documented fall-backs take over
int xh_launch_tcount_doy(xh_ctx*, const float*, int64_t, int64_t, int64_t, int, const double*, int64_t, const int32_t*, const int64_t*,
                         const int64_t*, int, int, int32_t*, int32_t*) { return XH_ERR_NOTIMPL; }

int xh_launch_doy_stats_sets(xh_ctx*, const float*, int64_t, int64_t, int64_t, const int32_t*, int, int, int, const uint8_t*, float*, float*, int64_t) { return XH_ERR_NOTIMPL; }
int xh_const_rows(xh_ctx*, int64_t, const float**, const float**, const float**) { return XH_ERR_NOTIMPL; }
// eqm.hip's quantile dispatch ends in the selection kernels (LDS, wave intrinsics): xh_eqm_train / xh_quantile_series are refused by
// the simulated device; these only satisfy the linker
int xh_select_time_major(xh_ctx*, const float*, int64_t, int64_t, int64_t, const double*, int, float*, int64_t, int64_t) { return XH_ERR_NOTIMPL; }
int xh_select_hist(xh_ctx*, const float*, int64_t, int64_t, int64_t, const double*, int, float*, int64_t, int64_t) { return XH_ERR_NOTIMPL; }
int xh_select_columns(xh_ctx*, const float*, int64_t, int64_t, int64_t, const double*, int, float*, int64_t, int64_t) {
  xh_set_error("host simulation: the selection kernels are not simulated");
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
int xh_transpose_f32(xh_ctx*, const float*, int64_t, int64_t, int64_t, float*, int64_t) {
  xh_set_error("host simulation: xh_transpose_f32 is not simulated");
  return XH_ERR_LIMIT;
}
}
