// TEST INFRASTRUCTURE ONLY -- a host stand-in for <hip/hip_runtime.h>, just large enough to compile the kernels of
// declip_amd/csrc that are written in plain HIP C++ (no inline ISA) as ordinary C++ and run them on the CPU (-DDH_HOST_EMU):
// every HIP thread of a block is a ucontext fiber; __syncthreads() is a counted block barrier; the cross-lane operations the
// kernels use (__shfl_xor, the MFMA builtins, the LDS transpose read) are wave collectives: each lane deposits its operands,
// the last lane to arrive computes, every lane picks up its share (tests/hipemu/emu.cpp).  What this checks is index
// arithmetic, bounds, LDS layouts (dynamic LDS is NaN-poisoned per block), reductions, fragment layouts and the host-side
// launch logic of the C-ABI entry points -- not performance, not memory ordering, not occupancy.
#pragma once
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <functional>

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
#define warpSize 64

struct uint4 { uint32_t x, y, z, w; };
struct int4 { int32_t x, y, z, w; };
struct uint2 { uint32_t x, y; };
struct float4 { float x, y, z, w; };
struct float2 { float x, y; };
static inline float4 make_float4(float a, float b, float c, float d) { return float4{a, b, c, d}; }
static inline float2 make_float2(float a, float b) { return float2{a, b}; }
static inline uint4 make_uint4(uint32_t a, uint32_t b, uint32_t c, uint32_t d) { return uint4{a, b, c, d}; }
static inline int4 make_int4(int32_t a, int32_t b, int32_t c, int32_t d) { return int4{a, b, c, d}; }
static inline uint2 make_uint2(uint32_t a, uint32_t b) { return uint2{a, b}; }

static inline float __uint_as_float(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
static inline uint32_t __float_as_uint(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline float __int_as_float(int u) { float f; memcpy(&f, &u, 4); return f; }
static inline int __float_as_int(float f) { int u; memcpy(&u, &f, 4); return u; }
#define __expf expf
#define __logf logf
static inline float rsqrtf(float x) { return 1.f / sqrtf(x); }
static inline float __frcp_rn(float x) { return 1.f / x; }
static inline float __fdividef(float a, float b) { return a / b; }
using std::max;
using std::min;

// ---- block / wave scheduling (emu.cpp)
void emu_syncthreads();
#define __syncthreads() emu_syncthreads()
void* emu_dyn_lds();
// wave collective: deposit `n` bytes, returns the 64 lanes' deposits (valid until this lane's next collective)
const unsigned char* emu_wave_gather(const void* mine, size_t n);
unsigned emu_lane();

template <typename T>
static inline T __shfl_xor(T v, int mask, int width = 64) {
  (void)width;
  const unsigned char* all = emu_wave_gather(&v, sizeof(T));
  T r;
  memcpy(&r, all + (size_t)((emu_lane() ^ (unsigned)mask) & 63) * sizeof(T), sizeof(T));
  return r;
}
template <typename T>
static inline T __shfl(T v, int src, int width = 64) {
  (void)width;
  const unsigned char* all = emu_wave_gather(&v, sizeof(T));
  T r;
  memcpy(&r, all + (size_t)(src & 63) * sizeof(T), sizeof(T));
  return r;
}
// lanes of a wave run in lockstep on the hardware; here they are independent fibers, so a wave barrier must really wait
static inline void emu_wave_barrier() { const char c = 0; (void)emu_wave_gather(&c, 1); }
#define __builtin_amdgcn_readlane(v, l) __shfl((v), (l))
#define __builtin_amdgcn_wave_barrier() emu_wave_barrier()
#define __builtin_amdgcn_fence(...) ((void)0)
#define __builtin_amdgcn_s_barrier() emu_syncthreads()
#define __builtin_amdgcn_sched_barrier(m) ((void)0)
#define __builtin_amdgcn_s_setprio(p) ((void)0)
#define __threadfence() ((void)0)
#define __threadfence_block() ((void)0)

// atomics: the fibers of a grid run one after the other
template <typename T> static inline T atomicAdd(T* p, T v) { T o = *p; *p = o + v; return o; }
static inline float atomicAdd(float* p, double v) { float o = *p; *p = o + (float)v; return o; }
template <typename T> static inline T atomicMax(T* p, T v) { T o = *p; if (v > o) *p = v; return o; }
static inline float unsafeAtomicAdd(float* p, float v) { return atomicAdd(p, v); }

// ---- MFMA builtins as wave collectives (operand / accumulator layouts of the CDNA3/4 ISA)
typedef float emu_f32x16 __attribute__((ext_vector_type(16)));
typedef float emu_f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 emu_bf16x8 __attribute__((ext_vector_type(8)));
typedef short emu_i16x4 __attribute__((ext_vector_type(4)));
emu_f32x16 emu_mfma_f32_32x32x2f32(float a, float b, emu_f32x16 c);
emu_f32x16 emu_mfma_f32_32x32x16_bf16(emu_bf16x8 a, emu_bf16x8 b, emu_f32x16 c);
emu_f32x4 emu_mfma_f32_16x16x32_bf16(emu_bf16x8 a, emu_bf16x8 b, emu_f32x4 c);
emu_i16x4 emu_ds_read_tr16_b64(const void* lds_addr);
#define __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, x, y, z) emu_mfma_f32_32x32x2f32((a), (b), (c))
#define __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, x, y, z) emu_mfma_f32_32x32x16_bf16((a), (b), (c))
#define __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, x, y, z) emu_mfma_f32_16x16x32_bf16((a), (b), (c))
#define __builtin_amdgcn_ds_read_tr16_b64_v4i16(p) emu_ds_read_tr16_b64((const void*)(p))

// ---- runtime API used by the host side of the library
typedef void* hipStream_t;
typedef int hipError_t;
#define hipSuccess 0
static inline hipError_t hipGetLastError() { return hipSuccess; }
static inline const char* hipGetErrorString(hipError_t) { return "emu"; }
static inline hipError_t hipMemsetAsync(void* p, int v, size_t n, hipStream_t) { memset(p, v, n); return hipSuccess; }
static inline hipError_t hipMemset(void* p, int v, size_t n) { memset(p, v, n); return hipSuccess; }
static inline hipError_t hipMemset2DAsync(void* p, size_t pitch, int v, size_t width, size_t height, hipStream_t) { for (size_t r = 0; r < height; ++r) memset((char*)p + r * pitch, v, width); return hipSuccess; }
template <typename T> static inline hipError_t hipMalloc(T** p, size_t n) { *p = (T*)malloc(n); return *p ? hipSuccess : 2; }
enum hipMemcpyKind { hipMemcpyDeviceToDevice = 3, hipMemcpyDefault = 4 };
static inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t) { memmove(d, s, n); return hipSuccess; }
enum hipFuncAttribute { hipFuncAttributeMaxDynamicSharedMemorySize = 8 };
static inline hipError_t hipFuncSetAttribute(const void*, hipFuncAttribute, int) { return hipSuccess; }
struct hipDeviceProp_t { int multiProcessorCount; int clockRate; size_t sharedMemPerBlock; size_t maxSharedMemoryPerMultiProcessor; char gcnArchName[32]; };
static inline hipError_t hipGetDevice(int* d) { *d = 0; return hipSuccess; }
static inline hipError_t hipGetDeviceProperties(hipDeviceProp_t* p, int) {
  memset(p, 0, sizeof *p);
  p->multiProcessorCount = 4;       // a small "chip": persistent kernels loop over several work items per block
  p->clockRate = 2400000;
  p->sharedMemPerBlock = p->maxSharedMemoryPerMultiProcessor = 160 * 1024;
  strcpy(p->gcnArchName, "gfx950-emu");
  return hipSuccess;
}
template <typename F>
static inline hipError_t hipOccupancyMaxActiveBlocksPerMultiprocessor(int* n, F, int, size_t) { *n = 2; return hipSuccess; }

void emu_launch(dim3 grid, dim3 block, size_t dyn_lds, const std::function<void()>& body);
#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...) \
  emu_launch((grid), (block), (size_t)(shmem), [&]() { kernel(__VA_ARGS__); })
