// core.hip — context, device memory, timers, synthetic input generator, tiled transpose.
#include <stdarg.h>

#include <stdlib.h>
#include <string.h>

#include "common.h"

static thread_local char g_err[512] = "";

void xh_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

const char* xh_diag_env(const char* name) {
  const char* on = getenv("XH_DIAGNOSTICS");
  if (!on || on[0] != '1') return nullptr;
  return getenv(name);
}

extern "C" {

int xh_abi_version(void) { return XH_ABI_VERSION; }
const char* xh_last_error(void) { return g_err; }

int xh_device_count(int* n) {
  XH_REQUIRE(n != nullptr, XH_ERR_ARG, "xh_device_count: n is NULL");
  int c = 0;
  hipError_t e = hipGetDeviceCount(&c);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    c = 0;
  }
  *n = c;
  return XH_OK;
}

int xh_create(int device, xh_ctx** out) {
  XH_REQUIRE(out != nullptr, XH_ERR_ARG, "xh_create: out is NULL");
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n <= 0) {
    (void)hipGetLastError();
    xh_set_error("xh_create: no HIP device visible (%s)", e == hipSuccess ? "count=0" : hipGetErrorString(e));
    return XH_ERR_NODEVICE;
  }
  XH_REQUIRE(device >= 0 && device < n, XH_ERR_ARG, "xh_create: device %d out of range [0,%d)", device, n);
  XH_CHECK_HIP(hipSetDevice(device));
  xh_ctx* ctx = new xh_ctx();
  memset(ctx, 0, sizeof(*ctx));
  ctx->device = device;
  XH_CHECK_HIP(hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking));
  XH_CHECK_HIP(hipEventCreate(&ctx->ev0));
  XH_CHECK_HIP(hipEventCreate(&ctx->ev1));
  {
    // the helper stream of the transposed-batch pipeline (eqm.hip): at the highest priority, so that its (memory-bound)
    // transposes are dispatched first into the slots the (VALU-bound) selection workgroups free up — since the selection
    // kernel got faster than the transposes next to it, they are the long pole (config-4 train 71.4 -> 67.6 ms);
    // XH_STREAM2_PRIO=0 (diagnostics) restores the default priority
    const char* pr = xh_diag_env("XH_STREAM2_PRIO");
    int least = 0, greatest = 0;
    if (!(pr && !atoi(pr)) && hipDeviceGetStreamPriorityRange(&least, &greatest) == hipSuccess)
      XH_CHECK_HIP(hipStreamCreateWithPriority(&ctx->stream2, hipStreamNonBlocking, greatest));
    else
      XH_CHECK_HIP(hipStreamCreateWithFlags(&ctx->stream2, hipStreamNonBlocking));
  }
  for (int i = 0; i < 2; ++i) {
    XH_CHECK_HIP(hipEventCreateWithFlags(&ctx->ev_ready[i], hipEventDisableTiming));
    XH_CHECK_HIP(hipEventCreateWithFlags(&ctx->ev_done[i], hipEventDisableTiming));
  }
  XH_CHECK_HIP(hipStreamCreateWithFlags(&ctx->copy_in, hipStreamNonBlocking));
  XH_CHECK_HIP(hipStreamCreateWithFlags(&ctx->copy_out, hipStreamNonBlocking));
  XH_CHECK_HIP(hipEventCreateWithFlags(&ctx->ev_lane, hipEventDisableTiming));
  ctx->scratch_bytes = 8u << 20;
  XH_CHECK_HIP(hipMalloc(&ctx->scratch, ctx->scratch_bytes));
  XH_CHECK_HIP(hipHostMalloc((void**)&ctx->scratch_host, ctx->scratch_bytes, hipHostMallocDefault));
  ctx->scratch_head = 0;
  hipDeviceProp_t prop;
  XH_CHECK_HIP(hipGetDeviceProperties(&prop, device));
  ctx->num_cu = prop.multiProcessorCount;
  *out = ctx;
  return XH_OK;
}

int xh_destroy(xh_ctx* ctx) {
  if (!ctx) return XH_OK;
  (void)hipSetDevice(ctx->device);
  (void)hipStreamSynchronize(ctx->stream);
  if (ctx->scratch) (void)hipFree(ctx->scratch);
  if (ctx->scratch_host) (void)hipHostFree(ctx->scratch_host);
  for (int i = 0; i < ctx->nretired; ++i) {
    (void)hipFree(ctx->retired[i]);
    (void)hipHostFree(ctx->retired_host[i]);
  }
  if (ctx->big) (void)hipFree(ctx->big);
  if (ctx->nanrow) (void)hipFree(ctx->nanrow);
  (void)hipEventDestroy(ctx->ev0);
  (void)hipEventDestroy(ctx->ev1);
  (void)hipStreamSynchronize(ctx->stream2);
  for (int i = 0; i < 2; ++i) {
    (void)hipEventDestroy(ctx->ev_ready[i]);
    (void)hipEventDestroy(ctx->ev_done[i]);
  }
  (void)hipStreamDestroy(ctx->stream2);
  (void)hipStreamSynchronize(ctx->copy_in);
  (void)hipStreamSynchronize(ctx->copy_out);
  (void)hipStreamDestroy(ctx->copy_in);
  (void)hipStreamDestroy(ctx->copy_out);
  (void)hipEventDestroy(ctx->ev_lane);
  (void)hipStreamDestroy(ctx->stream);
  delete ctx;
  return XH_OK;
}

int xh_sync(xh_ctx* ctx) {
  XH_REQUIRE(ctx, XH_ERR_ARG, "xh_sync: ctx is NULL");
  XH_CHECK_HIP(hipStreamSynchronize(ctx->stream));
  return XH_OK;
}

int xh_device_name(xh_ctx* ctx, char* buf, size_t buflen) {
  XH_REQUIRE(ctx && buf && buflen > 0, XH_ERR_ARG, "xh_device_name: bad args");
  hipDeviceProp_t prop;
  XH_CHECK_HIP(hipGetDeviceProperties(&prop, ctx->device));
  snprintf(buf, buflen, "%s|%s|cu=%d", prop.name, prop.gcnArchName, prop.multiProcessorCount);
  return XH_OK;
}

int xh_mem_info(xh_ctx* ctx, size_t* free_bytes, size_t* total_bytes) {
  XH_REQUIRE(ctx && free_bytes && total_bytes, XH_ERR_ARG, "xh_mem_info: bad args");
  XH_CHECK_HIP(hipSetDevice(ctx->device));
  XH_CHECK_HIP(hipMemGetInfo(free_bytes, total_bytes));
  return XH_OK;
}

int xh_malloc(xh_ctx* ctx, size_t bytes, void** dptr) {
  XH_REQUIRE(ctx && dptr, XH_ERR_ARG, "xh_malloc: bad args");
  XH_CHECK_HIP(hipSetDevice(ctx->device));
  if (bytes == 0) bytes = 16;
  XH_CHECK_HIP(hipMalloc(dptr, bytes));
  return XH_OK;
}

int xh_free(xh_ctx* ctx, void* dptr) {
  XH_REQUIRE(ctx, XH_ERR_ARG, "xh_free: ctx is NULL");
  if (!dptr) return XH_OK;
  XH_CHECK_HIP(hipStreamSynchronize(ctx->stream));
  XH_CHECK_HIP(hipFree(dptr));
  return XH_OK;
}

int xh_memset(xh_ctx* ctx, void* dptr, int value, size_t bytes) {
  XH_REQUIRE(ctx && dptr, XH_ERR_ARG, "xh_memset: bad args");
  XH_CHECK_HIP(hipMemsetAsync(dptr, value, bytes, ctx->stream));
  return XH_OK;
}

int xh_memcpy_h2d(xh_ctx* ctx, void* dst, const void* src, size_t bytes) {
  XH_REQUIRE(ctx && (bytes == 0 || (dst && src)), XH_ERR_ARG, "xh_memcpy_h2d: bad args");
  if (bytes == 0) return XH_OK;
  XH_CHECK_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, ctx->stream));
  // the host buffer is pageable and caller owned: do not return before it has been consumed
  XH_CHECK_HIP(hipStreamSynchronize(ctx->stream));
  return XH_OK;
}

int xh_memcpy_d2h(xh_ctx* ctx, void* dst, const void* src, size_t bytes) {
  XH_REQUIRE(ctx && (bytes == 0 || (dst && src)), XH_ERR_ARG, "xh_memcpy_d2h: bad args");
  if (bytes == 0) return XH_OK;
  XH_CHECK_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, ctx->stream));
  XH_CHECK_HIP(hipStreamSynchronize(ctx->stream));
  return XH_OK;
}

int xh_memcpy_d2d(xh_ctx* ctx, void* dst, const void* src, size_t bytes) {
  XH_REQUIRE(ctx && (bytes == 0 || (dst && src)), XH_ERR_ARG, "xh_memcpy_d2d: bad args");
  if (bytes == 0) return XH_OK;
  XH_CHECK_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, ctx->stream));
  return XH_OK;
}

// ---- host memory + strided copies on the copy lanes (block adapter, xclim_amd/blocks.py) --------------------------------
static hipStream_t lane_stream(xh_ctx* ctx, int lane) { return lane == 1 ? ctx->copy_in : lane == 2 ? ctx->copy_out : ctx->stream; }

int xh_host_alloc(xh_ctx* ctx, size_t bytes, void** hptr) {
  XH_REQUIRE(ctx && hptr, XH_ERR_ARG, "xh_host_alloc: bad args");
  *hptr = nullptr;
  if (bytes == 0) return XH_OK;
  XH_CHECK_HIP(hipHostMalloc(hptr, bytes, hipHostMallocDefault));
  return XH_OK;
}

int xh_host_free(xh_ctx* ctx, void* hptr) {
  XH_REQUIRE(ctx, XH_ERR_ARG, "xh_host_free: ctx is NULL");
  if (hptr) XH_CHECK_HIP(hipHostFree(hptr));
  return XH_OK;
}

int xh_host_register(xh_ctx* ctx, void* hptr, size_t bytes) {
  XH_REQUIRE(ctx && hptr && bytes > 0, XH_ERR_ARG, "xh_host_register: bad args");
  XH_CHECK_HIP(hipHostRegister(hptr, bytes, hipHostRegisterDefault));
  return XH_OK;
}

int xh_host_unregister(xh_ctx* ctx, void* hptr) {
  XH_REQUIRE(ctx && hptr, XH_ERR_ARG, "xh_host_unregister: bad args");
  XH_CHECK_HIP(hipHostUnregister(hptr));
  return XH_OK;
}

int xh_memcpy2d(xh_ctx* ctx, void* dst, size_t dpitch, const void* src, size_t spitch, size_t width, size_t height, int kind,
                int lane, int blocking) {
  XH_REQUIRE(ctx, XH_ERR_ARG, "xh_memcpy2d: ctx is NULL");
  XH_REQUIRE(kind >= 0 && kind <= 2, XH_ERR_ARG,
             "xh_memcpy2d: kind must be 0 (host -> device), 1 (device -> host) or 2 (device -> device)");
  XH_REQUIRE(lane >= 0 && lane <= 2, XH_ERR_ARG, "xh_memcpy2d: lane must be 0, 1 or 2");
  if (width == 0 || height == 0) return XH_OK;
  XH_REQUIRE(dst && src && dpitch >= width && spitch >= width, XH_ERR_ARG, "xh_memcpy2d: NULL pointer or pitch < width");
  hipStream_t s = lane_stream(ctx, lane);
  XH_CHECK_HIP(hipMemcpy2DAsync(dst, dpitch, src, spitch, width, height, kind == 0 ? hipMemcpyHostToDevice : (kind == 1 ? hipMemcpyDeviceToHost : hipMemcpyDeviceToDevice), s));
  if (blocking) XH_CHECK_HIP(hipStreamSynchronize(s));  // pageable host memory: do not return before it has been consumed
  return XH_OK;
}

int xh_lane_fence(xh_ctx* ctx, int from_lane, int to_lane) {
  XH_REQUIRE(ctx && from_lane >= 0 && from_lane <= 2 && to_lane >= 0 && to_lane <= 2, XH_ERR_ARG, "xh_lane_fence: bad args");
  if (from_lane == to_lane) return XH_OK;
  XH_CHECK_HIP(hipEventRecord(ctx->ev_lane, lane_stream(ctx, from_lane)));
  XH_CHECK_HIP(hipStreamWaitEvent(lane_stream(ctx, to_lane), ctx->ev_lane, 0));
  return XH_OK;
}

int xh_lane_sync(xh_ctx* ctx, int lane) {
  XH_REQUIRE(ctx && lane >= 0 && lane <= 2, XH_ERR_ARG, "xh_lane_sync: bad args");
  XH_CHECK_HIP(hipStreamSynchronize(lane_stream(ctx, lane)));
  return XH_OK;
}

int xh_timer_start(xh_ctx* ctx) {
  XH_REQUIRE(ctx, XH_ERR_ARG, "xh_timer_start: ctx is NULL");
  XH_CHECK_HIP(hipEventRecord(ctx->ev0, ctx->stream));
  return XH_OK;
}

int xh_timer_stop(xh_ctx* ctx, float* elapsed_ms) {
  XH_REQUIRE(ctx && elapsed_ms, XH_ERR_ARG, "xh_timer_stop: bad args");
  XH_CHECK_HIP(hipEventRecord(ctx->ev1, ctx->stream));
  XH_CHECK_HIP(hipEventSynchronize(ctx->ev1));
  XH_CHECK_HIP(hipEventElapsedTime(elapsed_ms, ctx->ev0, ctx->ev1));
  return XH_OK;
}

int xh_stream(xh_ctx* ctx, void** stream) {
  XH_REQUIRE(ctx && stream, XH_ERR_ARG, "xh_stream: bad args");
  *stream = (void*)ctx->stream;
  return XH_OK;
}

}  // extern "C"

// Host tables are tiny, caller-owned and possibly temporaries.  They are copied into a pinned ring that mirrors the device
// ring and sent with an asynchronous copy on the context's stream: the call does not wait for earlier kernels (the old
// synchronous upload drained the stream at every entry point).  The ring wraps with ONE stream synchronisation when it is
// exhausted (8 MiB: hundreds of calls), so a slot is never overwritten while an earlier kernel may still read it.
int xh_scratch_upload(xh_ctx* ctx, size_t* cursor, const void* host, size_t bytes, void** dptr) {
  size_t off = (ctx->scratch_head + 255) & ~(size_t)255;
  if (bytes > ctx->scratch_bytes / 4) {
    // an unusually large table: grow both rings (wait for in-flight users of the old one first)
    XH_CHECK_HIP(hipStreamSynchronize(ctx->stream));
    size_t nb = ctx->scratch_bytes;
    while (bytes > nb / 4) nb *= 2;
    void* n = nullptr;
    char* nh = nullptr;
    XH_CHECK_HIP(hipMalloc(&n, nb));
    XH_CHECK_HIP(hipHostMalloc((void**)&nh, nb, hipHostMallocDefault));
    if (ctx->nretired < 16) {
      ctx->retired[ctx->nretired] = ctx->scratch;
      ctx->retired_host[ctx->nretired] = ctx->scratch_host;
      ctx->nretired++;
    }  // (else: leaked on purpose — never reached with doubling sizes)
    ctx->scratch = n;
    ctx->scratch_host = nh;
    ctx->scratch_bytes = nb;
    off = 0;
  } else if (off + bytes > ctx->scratch_bytes) {
    XH_CHECK_HIP(hipStreamSynchronize(ctx->stream));  // wrap: every earlier consumer has finished
    off = 0;
  }
  char* d = (char*)ctx->scratch + off;
  memcpy(ctx->scratch_host + off, host, bytes);
  XH_CHECK_HIP(hipMemcpyAsync(d, ctx->scratch_host + off, bytes, hipMemcpyHostToDevice, ctx->stream));
  ctx->scratch_head = off + bytes;
  *dptr = d;
  *cursor += bytes;
  return XH_OK;
}

int xh_big_scratch(xh_ctx* ctx, size_t bytes, void** dptr) {
  if (bytes > ctx->big_bytes) {
    XH_CHECK_HIP(hipStreamSynchronize(ctx->stream));
    if (ctx->big) XH_CHECK_HIP(hipFree(ctx->big));
    ctx->big = nullptr;
    ctx->big_bytes = 0;
    XH_CHECK_HIP(hipMalloc(&ctx->big, bytes));
    ctx->big_bytes = bytes;
  }
  *dptr = ctx->big;
  return XH_OK;
}

int xh_const_rows(xh_ctx* ctx, int64_t elems, const float** nan_row, const float** ninf_row, const float** pinf_row) {
  const size_t bytes = ((size_t)elems * sizeof(float) + 255) & ~(size_t)255;
  if (bytes > ctx->nanrow_bytes) {
    XH_CHECK_HIP(hipStreamSynchronize(ctx->stream));  // (a kernel in flight may still read the old rows)
    if (ctx->nanrow) XH_CHECK_HIP(hipFree(ctx->nanrow));
    ctx->nanrow = nullptr;
    ctx->nanrow_bytes = 0;
    XH_CHECK_HIP(hipMalloc(&ctx->nanrow, 3 * bytes));
    ctx->nanrow_bytes = bytes;
    char* p = (char*)ctx->nanrow;
    XH_CHECK_HIP(hipMemsetAsync(p, 0xFF, bytes, ctx->stream));  // 0xFFFFFFFF is a NaN
    XH_CHECK_HIP(hipMemsetD32Async((hipDeviceptr_t)(p + bytes), (int)0xFF800000u, bytes / 4, ctx->stream));
    XH_CHECK_HIP(hipMemsetD32Async((hipDeviceptr_t)(p + 2 * bytes), (int)0x7F800000u, bytes / 4, ctx->stream));
  }
  const char* p = (const char*)ctx->nanrow;
  if (nan_row) *nan_row = (const float*)p;
  if (ninf_row) *ninf_row = (const float*)(p + ctx->nanrow_bytes);
  if (pinf_row) *pinf_row = (const float*)(p + 2 * ctx->nanrow_bytes);
  return XH_OK;
}

// ---- synthetic generator --------------------------------------------------------------------
__device__ __forceinline__ uint64_t xh_mix64(uint64_t z) {
  z ^= z >> 30;
  z *= 0xBF58476D1CE4E5B9ULL;
  z ^= z >> 27;
  z *= 0x94D049BB133111EBULL;
  z ^= z >> 31;
  return z;
}

__global__ void __launch_bounds__(XH_BLOCK)
k_fill_synthetic(float* __restrict__ out, int64_t T, int64_t C, int64_t st, int kind, uint64_t seed, int64_t cell0,
                 const float* __restrict__ base, float amp, float p_wet, uint32_t nan_ppm) {
  int64_t c = (int64_t)blockIdx.x * XH_BLOCK + threadIdx.x;
  if (c >= C) return;
  uint64_t cell = (uint64_t)(cell0 + c);
  for (int64_t t = blockIdx.y; t < T; t += gridDim.y) {
    uint64_t key = seed * 0xD1342543DE82EF95ULL + (uint64_t)t * 0x9E3779B97F4A7C15ULL + cell * 0xC2B2AE3D27D4EB4FULL;
    uint64_t z = xh_mix64(key);
    uint64_t w = xh_mix64(z + 0x9E3779B97F4A7C15ULL);
    float val;
    if (kind == 0) {
      float u0 = (float)(z & 0xFFFF) * (1.0f / 65536.0f);
      float u1 = (float)((z >> 16) & 0xFFFF) * (1.0f / 65536.0f);
      float u2 = (float)((z >> 32) & 0xFFFF) * (1.0f / 65536.0f);
      float u3 = (float)((z >> 48) & 0xFFFF) * (1.0f / 65536.0f);
      float z4 = ((u0 + u1) + (u2 + u3)) - 2.0f;
      val = base[t] + amp * z4;
    } else {
      float uw = (float)(w & 0xFFFFFF) * (1.0f / 16777216.0f);
      float ua = (float)(z >> 40) * (1.0f / 16777216.0f);
      float amount = ((ua * ua) * ua) * amp;
      val = (uw < p_wet) ? (base[t] + amount) : 0.0f;
    }
    uint32_t r = (uint32_t)((((w >> 32) & 0xFFFFFFFFULL) * 1000000ULL) >> 32);
    if (r < nan_ppm) val = xh_nan32();
    out[t * st + c] = val;
  }
}

// ---- tiled transpose ------------------------------------------------------------------------
__global__ void __launch_bounds__(XH_BLOCK)
k_transpose_f32(const float* __restrict__ in, int64_t rows, int64_t cols, int64_t in_stride, float* __restrict__ out,
                int64_t out_stride) {
  __shared__ float tile[64][65];
  int64_t r0 = (int64_t)blockIdx.y * 64, c0 = (int64_t)blockIdx.x * 64;
  int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;  // 4 row groups
  // unconditional, clamped loads first (a load under a condition is followed by s_waitcnt vmcnt(0): the 16 loads of a
  // thread would be serialised); out-of-range elements are never stored
  float v[16];
  const int64_t cc = c0 + tx < cols ? c0 + tx : cols - 1;
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    const int64_t r = r0 + ty + 4 * k;
    v[k] = in[(r < rows ? r : rows - 1) * in_stride + cc];
  }
#pragma unroll
  for (int k = 0; k < 16; ++k) tile[ty + 4 * k][tx] = v[k];
  __syncthreads();
  for (int i = ty; i < 64; i += 4) {
    int64_t c = c0 + i, r = r0 + tx;
    if (r < rows && c < cols) out[c * out_stride + r] = tile[tx][i];
  }
}

// 128 x 128 tile, 16-byte accesses on both sides: a wave reads two 512-byte row pieces per instruction and writes two
// 512-byte column pieces (the 64 x 64 tile above moves 256-byte pieces: twice as many DRAM page switches per byte).
// Interior tiles only (rows / cols multiples of 128 are not required: the edges fall back to clamped scalar stores).
__global__ void __launch_bounds__(XH_BLOCK)
k_transpose128_f32(const float* __restrict__ in, int64_t rows, int64_t cols, int64_t in_stride, float* __restrict__ out,
                   int64_t out_stride) {
  __shared__ float tile[128][129];
  const int64_t r0 = (int64_t)blockIdx.y * 128, c0 = (int64_t)blockIdx.x * 128;
  const int q = threadIdx.x & 31, h = threadIdx.x >> 5;  // 32 lanes x 16 B = one 512-byte row piece; 8 pieces per pass
  const bool full = r0 + 128 <= rows && c0 + 128 <= cols;
  if (full) {
    float4 v[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) v[k] = *reinterpret_cast<const float4*>(in + (r0 + h + 8 * k) * in_stride + c0 + 4 * q);
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      float* t = &tile[h + 8 * k][4 * q];
      t[0] = v[k].x; t[1] = v[k].y; t[2] = v[k].z; t[3] = v[k].w;
    }
  } else {  // edge tile: element-wise, clamped (values beyond the edge are never stored)
    const int tx = threadIdx.x & 127, ty = threadIdx.x >> 7;
    const int64_t cc = c0 + tx < cols ? c0 + tx : cols - 1;
    for (int k = 0; k < 64; ++k) {
      const int64_t r = r0 + ty + 2 * k;
      tile[ty + 2 * k][tx] = in[(r < rows ? r : rows - 1) * in_stride + cc];
    }
  }
  __syncthreads();
#pragma unroll 4
  for (int k = 0; k < 16; ++k) {
    const int c = h + 8 * k;  // output row = input column
    const int r = 4 * q;
    const float4 o = make_float4(tile[r][c], tile[r + 1][c], tile[r + 2][c], tile[r + 3][c]);
    float* dst = out + (c0 + c) * out_stride + r0 + r;
    if (full) *reinterpret_cast<float4*>(dst) = o;
    else if (c0 + c < cols) {
      if (r0 + r < rows) dst[0] = o.x;
      if (r0 + r + 1 < rows) dst[1] = o.y;
      if (r0 + r + 2 < rows) dst[2] = o.z;
      if (r0 + r + 3 < rows) dst[3] = o.w;
    }
  }
}

extern "C" {

int xh_fill_synthetic(xh_ctx* ctx, float* out, int64_t T, int64_t C, int64_t st, int kind, uint64_t seed, int64_t cell0,
                      const float* base, float amp, float p_wet, uint32_t nan_per_million) {
  XH_REQUIRE(ctx && out && base, XH_ERR_ARG, "xh_fill_synthetic: NULL argument");
  XH_REQUIRE(T >= 0 && C >= 0 && st >= C, XH_ERR_ARG, "xh_fill_synthetic: bad shape");
  XH_REQUIRE(kind == 0 || kind == 1, XH_ERR_ARG, "xh_fill_synthetic: kind must be 0 or 1");
  if (T == 0 || C == 0) return XH_OK;
  dim3 grid((unsigned)cdiv64(C, XH_BLOCK), (unsigned)(T < 64 ? T : 64));
  hipLaunchKernelGGL(k_fill_synthetic, grid, dim3(XH_BLOCK), 0, ctx->stream, out, T, C, st, kind, seed, cell0, base, amp,
                     p_wet, nan_per_million);
  XH_LAUNCH_CHECK();
  return XH_OK;
}

int xh_transpose_f32(xh_ctx* ctx, const float* in, int64_t rows, int64_t cols, int64_t in_stride, float* out,
                     int64_t out_stride) {
  XH_REQUIRE(ctx && in && out, XH_ERR_ARG, "xh_transpose_f32: NULL argument");
  XH_REQUIRE(rows >= 0 && cols >= 0 && in_stride >= cols && out_stride >= rows, XH_ERR_ARG,
             "xh_transpose_f32: bad shape/strides");
  if (rows == 0 || cols == 0) return XH_OK;
  const bool al16 = ((reinterpret_cast<uintptr_t>(in) | reinterpret_cast<uintptr_t>(out)) & 15) == 0 && in_stride % 4 == 0 &&
                    out_stride % 4 == 0 && cols >= 4;
  const char* e64 = xh_diag_env("XH_TRANSPOSE_64");  // diagnostics: force the 64 x 64 kernel
  if (al16 && !(e64 && atoi(e64))) {
    dim3 grid((unsigned)cdiv64(cols, 128), (unsigned)cdiv64(rows, 128));
    XH_REQUIRE(grid.y <= 65535u, XH_ERR_LIMIT, "xh_transpose_f32: too many row tiles");
    hipLaunchKernelGGL(k_transpose128_f32, grid, dim3(XH_BLOCK), 0, ctx->stream, in, rows, cols, in_stride, out, out_stride);
    XH_LAUNCH_CHECK();
    return XH_OK;
  }
  dim3 grid((unsigned)cdiv64(cols, 64), (unsigned)cdiv64(rows, 64));
  XH_REQUIRE(grid.y <= 65535u, XH_ERR_LIMIT, "xh_transpose_f32: too many row tiles");
  hipLaunchKernelGGL(k_transpose_f32, grid, dim3(XH_BLOCK), 0, ctx->stream, in, rows, cols, in_stride, out, out_stride);
  XH_LAUNCH_CHECK();
  return XH_OK;
}

}  // extern "C"
