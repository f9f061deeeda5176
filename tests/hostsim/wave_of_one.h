// Force-included (-include) when plane.hip is compiled for the host simulation: ONE thread at a time is a wave whose only active
// lane is that thread.  The work-list appends of plane.hip (ballot -> one atomic per wave -> shuffle -> slot by the lanes below)
// then append one entry per call with the same code; lane-private LDS columns ([..][threadIdx.x]) become static arrays.
#pragma once
#include <hip/hip_runtime.h>
#define __shared__ static
#define __ballot(pred) ((pred) ? (1ull << (threadIdx.x & 63u)) : 0ull)
#define __shfl(value, srclane) (value)
#define __ffsll(x) __builtin_ffsll(x)
#define __popcll(x) __builtin_popcountll(x)
#define atomicAdd(ptr, val) sim_atomic_add(ptr, val)
