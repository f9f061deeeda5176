// window.h — register-ring rolling-window kernels (window.hip); both return XH_ERR_NOTIMPL for windows > 8 steps so
// that the callers fall back to the generic kernels.
#pragma once
#include "common.h"

int xh_launch_rolling_ring(xh_ctx* ctx, const float* x, int64_t T, int64_t C, int64_t st, int window, int left, int right,
                           int reducer, float* out, int64_t out_st);
int xh_launch_spell_ring(xh_ctx* ctx, const float* x, int64_t T, int64_t C, int64_t st, int window, int win_red, int op,
                         float thr, const float* d_weights, float* out, int64_t out_st);
// spell mask + run statistics of the mask in one pass, runs cut at the segments (device segment table d_seg[P + 1])
int xh_launch_spell_runs(xh_ctx* ctx, const float* x, int64_t T, int64_t C, int64_t st, int window, int win_red, int op, float thr,
                         const float* d_weights, int stat, const int64_t* d_seg, int P, float* out, int32_t* valid_out);
