// TEST INFRASTRUCTURE ONLY -- a host stand-in for <hip/hip_runtime.h>, just large enough to compile the HBM-bound kernels of
// declip_amd/csrc (no MFMA, no cross-lane operations) as plain C++ and run them on the CPU: every HIP thread of a block is a
// ucontext fiber, __syncthreads() yields to the block scheduler (tests/hipemu/emu.cpp).  What this checks is index arithmetic,
// bounds, reductions and the host-side launch logic of the C-ABI entry points -- not performance, not wave-level behaviour.
#pragma once
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)
#define __shared__ static

struct dim3 {
  unsigned x, y, z;
  dim3(unsigned a = 1, unsigned b = 1, unsigned c = 1) : x(a), y(b), z(c) {}
};
struct emu_idx { unsigned x, y, z; };
extern emu_idx threadIdx, blockIdx;
extern dim3 blockDim, gridDim;

struct uint4 { uint32_t x, y, z, w; };
struct float4 { float x, y, z, w; };
static inline float4 make_float4(float a, float b, float c, float d) { return float4{a, b, c, d}; }

static inline float __uint_as_float(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
static inline uint32_t __float_as_uint(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
#define __expf expf
static inline float rsqrtf(float x) { return 1.f / sqrtf(x); }
static inline float __shfl_xor(float, int, int) { abort(); }   // cross-lane operations are not emulated

void emu_syncthreads();
#define __syncthreads() emu_syncthreads()

typedef void* hipStream_t;
typedef int hipError_t;
#define hipSuccess 0
static inline hipError_t hipGetLastError() { return hipSuccess; }
static inline const char* hipGetErrorString(hipError_t) { return "emu"; }
static inline hipError_t hipMemsetAsync(void* p, int v, size_t n, hipStream_t) { memset(p, v, n); return hipSuccess; }

#include <functional>
void emu_launch(dim3 grid, dim3 block, const std::function<void()>& body);
#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...) \
  emu_launch((grid), (block), [&]() { kernel(__VA_ARGS__); })
