// Common device/host helpers for libdeclip_hip.so (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>

#include "../../include/declip_hip.h"

typedef uint16_t bf16_t;  // raw bf16 bits
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;

#define DH_WAVE 64

// dynamic LDS of a kernel.  The host emulation build of the HBM-/LDS-level kernels (tests/hipemu, -DDH_HOST_EMU: HIP threads as
// fibers, test infrastructure only) hands out one buffer per running block instead.
#ifndef DH_HOST_EMU
#define DH_DYN_LDS(T, name) extern __shared__ T name[]
#define DH_DYN_LDS_A16(T, name) extern __shared__ __attribute__((aligned(16))) T name[]
#else
#define DH_DYN_LDS(T, name) T* name = (T*)emu_dyn_lds()
#define DH_DYN_LDS_A16(T, name) T* name = (T*)emu_dyn_lds()
#endif

// ----------------------------------------------------------------------------- errors
void dh_set_error(const char* fmt, ...);
#define DH_FAIL(code, ...)        \
  do {                            \
    dh_set_error(__VA_ARGS__);    \
    return (code);                \
  } while (0)
#define DH_CHECK_LAUNCH()                                              \
  do {                                                                 \
    hipError_t e__ = hipGetLastError();                                \
    if (e__ != hipSuccess) DH_FAIL(DH_ERR_LAUNCH, "%s:%d launch failed: %s", __FILE__, __LINE__, hipGetErrorString(e__)); \
  } while (0)
// Launch helpers that return bool ("took the problem": gemm_v4.hip's dh_*_try_v4) report a failed runtime call of their own -- a
// clearing memset in front of their launch -- through this thread-local slot; the C-ABI entry point that called them turns it into
// its return code (DH_HELPER_FAILED).  Entry points that return int check their runtime calls with DH_RT.
inline int& dh_helper_error() { static thread_local int e = 0; return e; }
#define DH_RT_NOTE(expr, what)                                                       \
  do {                                                                               \
    const hipError_t rt__ = (expr);                                                  \
    if (rt__ != hipSuccess) {                                                        \
      dh_set_error("%s failed: %s", what, hipGetErrorString(rt__));                  \
      dh_helper_error() = DH_ERR_LAUNCH;                                             \
    }                                                                                \
  } while (0)
#define DH_HELPER_FAILED()                                        \
  do {                                                            \
    if (const int he__ = dh_helper_error()) {                     \
      dh_helper_error() = 0;                                      \
      return he__;                                                \
    }                                                             \
  } while (0)
#define DH_RT(expr, what)                                                                                   \
  do {                                                                                                      \
    const hipError_t rt__ = (expr);                                                                         \
    if (rt__ != hipSuccess) DH_FAIL(DH_ERR_LAUNCH, "%s failed: %s", what, hipGetErrorString(rt__));         \
  } while (0)
#define DH_REQUIRE(cond, ...)                       \
  do {                                              \
    if (!(cond)) DH_FAIL(DH_ERR_ARG, __VA_ARGS__);  \
  } while (0)

// ----------------------------------------------------------------------------- bf16
__device__ __forceinline__ float bf2f(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }
__device__ __forceinline__ bf16_t f2bf(float f) {  // round-to-nearest-even, NaN preserved
  uint32_t u = __float_as_uint(f);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (bf16_t)((u >> 16) | 0x40);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (bf16_t)(u >> 16);
}
__device__ __forceinline__ uint32_t pack2bf(float lo, float hi) {
  return (uint32_t)f2bf(lo) | ((uint32_t)f2bf(hi) << 16);
}
// hardware RNE pack (v_cvt_pk_bf16_f32, gfx950): same rounding as f2bf for finite values, 1 instruction per pair
typedef __attribute__((ext_vector_type(2))) float dh_f32x2_t;
typedef __attribute__((ext_vector_type(2))) __bf16 dh_bf16x2_t;
__device__ __forceinline__ uint32_t pack2bf_hw(float lo, float hi) {
  dh_f32x2_t v = {lo, hi};
  dh_bf16x2_t b = __builtin_convertvector(v, dh_bf16x2_t);
  return *reinterpret_cast<uint32_t*>(&b);
}
__device__ __forceinline__ void st8_hw(bf16_t* p, const float* o) {
  uint4 v;
  v.x = pack2bf_hw(o[0], o[1]); v.y = pack2bf_hw(o[2], o[3]); v.z = pack2bf_hw(o[4], o[5]); v.w = pack2bf_hw(o[6], o[7]);
  *reinterpret_cast<uint4*>(p) = v;
}

// typed load/store of activation element types (T = float or bf16_t)
template <typename T> __device__ __forceinline__ float ld(const T* p);
template <> __device__ __forceinline__ float ld<float>(const float* p) { return *p; }
template <> __device__ __forceinline__ float ld<bf16_t>(const bf16_t* p) { return bf2f(*p); }
template <typename T> __device__ __forceinline__ void st(T* p, float v);
template <> __device__ __forceinline__ void st<float>(float* p, float v) { *p = v; }
template <> __device__ __forceinline__ void st<bf16_t>(bf16_t* p, float v) { *p = f2bf(v); }

// load 8 consecutive elements as floats (p must be 16B aligned for bf16, 32B for float)
__device__ __forceinline__ void ld8(const bf16_t* p, float* o) {
  uint4 v = *reinterpret_cast<const uint4*>(p);
  uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    o[2 * i] = __uint_as_float(w[i] << 16);
    o[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u);
  }
}
__device__ __forceinline__ void ld8(const float* p, float* o) {
  float4 a = *reinterpret_cast<const float4*>(p);
  float4 b = *reinterpret_cast<const float4*>(p + 4);
  o[0] = a.x; o[1] = a.y; o[2] = a.z; o[3] = a.w; o[4] = b.x; o[5] = b.y; o[6] = b.z; o[7] = b.w;
}
__device__ __forceinline__ void st8(bf16_t* p, const float* o) {
  uint4 v;
  v.x = pack2bf(o[0], o[1]); v.y = pack2bf(o[2], o[3]); v.z = pack2bf(o[4], o[5]); v.w = pack2bf(o[6], o[7]);
  *reinterpret_cast<uint4*>(p) = v;
}
__device__ __forceinline__ void st8(float* p, const float* o) {
  *reinterpret_cast<float4*>(p) = make_float4(o[0], o[1], o[2], o[3]);
  *reinterpret_cast<float4*>(p + 4) = make_float4(o[4], o[5], o[6], o[7]);
}

// st8 with the hardware bf16 pack where the output is bf16
__device__ __forceinline__ void st8_fast(bf16_t* p, const float* o) { st8_hw(p, o); }
__device__ __forceinline__ void st8_fast(float* p, const float* o) { st8(p, o); }

// ----------------------------------------------------------------------------- reductions
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
// Wave reductions on the DPP cross-lane path (round 6).  __shfl_xor compiles to ds_bpermute_b32 -- a round trip through the LDS
// crossbar, ~100 cycles each, six of them in a dependent chain per reduction: the LayerNorm kernels (two / four dependent
// reductions per row) spent most of a row's time there.  Here: quad_perm x2, row_half_mirror, row_mirror (every lane of a 16-lane
// row then holds the row's value), row_bcast15 / row_bcast31 (lane 63 holds the wave's), one v_readlane -- 6 VALU + 1 readlane, no LDS.
// The summation ORDER differs from the xor butterfly (same value up to fp32 rounding); index-producing kernels keep wave_sum / wave_max.
#ifndef DH_HOST_EMU
#define DH_DPP(old, v, ctrl, rm) __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, (float)(old)), __builtin_bit_cast(int, (float)(v)), (ctrl), (rm), 0xF, false))
__device__ __forceinline__ float wave_sum_fast(float v) {
  v += DH_DPP(0.f, v, 0xB1, 0xF);      // quad_perm [1,0,3,2]
  v += DH_DPP(0.f, v, 0x4E, 0xF);      // quad_perm [2,3,0,1]
  v += DH_DPP(0.f, v, 0x141, 0xF);     // row_half_mirror
  v += DH_DPP(0.f, v, 0x140, 0xF);     // row_mirror: every lane of a row holds the row sum
  v += DH_DPP(0.f, v, 0x142, 0xA);     // row_bcast15 into rows 1, 3
  v += DH_DPP(0.f, v, 0x143, 0xC);     // row_bcast31 into rows 2, 3: lane 63 holds the total
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
__device__ __forceinline__ float wave_max_fast(float v) {
  v = fmaxf(v, DH_DPP(v, v, 0xB1, 0xF));
  v = fmaxf(v, DH_DPP(v, v, 0x4E, 0xF));
  v = fmaxf(v, DH_DPP(v, v, 0x141, 0xF));
  v = fmaxf(v, DH_DPP(v, v, 0x140, 0xF));
  v = fmaxf(v, DH_DPP(v, v, 0x142, 0xA));
  v = fmaxf(v, DH_DPP(v, v, 0x143, 0xC));
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
#else
__device__ __forceinline__ float wave_sum_fast(float v) { return wave_sum(v); }
__device__ __forceinline__ float wave_max_fast(float v) { return wave_max(v); }
#endif
// block reductions for blockDim.x == 256 (4 waves); `red` is >= 8 floats of LDS
__device__ __forceinline__ float block_sum256(float v, float* red) {
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return red[0] + red[1] + red[2] + red[3];
}
__device__ __forceinline__ float block_max256(float v, float* red) {
  v = wave_max(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}

// QuickGELU and its derivative from ONE sigmoid (round 5): the forward epilogue (DH_EPI_GELU) stores the derivative as its aux
// output, so the backward epilogue (DH_EPI_DGELU) is a plain multiply -- the two transcendentals per element that the dX GEMM of
// c_proj used to spend on re-deriving QuickGELU'(u) from the stored pre-activation are gone, the forward pays three more VALU.
#ifdef DH_GELU_ABL   // ablation builds only (tools/build_lib_variant.sh abl -DDH_GELU_ABL): what the activation's arithmetic costs the epilogues
__device__ __forceinline__ float quick_gelu_f(float x) { return x * 0.5f; }
__device__ __forceinline__ float quick_gelu_grad_f(float x) { return 0.5f + 0.001f * x; }
__device__ __forceinline__ void quick_gelu_both_f(float x, float& g, float& d) { g = x * 0.5f; d = 0.5f + 0.001f * x; }
#else
__device__ __forceinline__ float quick_gelu_f(float x) { return x / (1.f + __expf(-1.702f * x)); }
__device__ __forceinline__ float quick_gelu_grad_f(float x) {
  float s = 1.f / (1.f + __expf(-1.702f * x));
  return s * (1.f + 1.702f * x * (1.f - s));
}
__device__ __forceinline__ void quick_gelu_both_f(float x, float& g, float& d) {
  const float s = 1.f / (1.f + __expf(-1.702f * x));
  g = x * s;
  d = s + 1.702f * g * (1.f - s);            // = s (1 + 1.702 x (1 - s))
}
#endif

static inline int dh_cdiv(long a, long b) { return (int)((a + b - 1) / b); }
