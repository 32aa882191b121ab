// Zero a LIST of ranges of one fp32 buffer in one launch (round 5: the flat gradient buffer minus the slots that the first weight-
// gradient GEMM of the step overwrites -- engine.FlatParams.begin_backward).  The reference has no counterpart: torch's
// optimizer.zero_grad(set_to_none=True) drops every .grad and autograd assigns the first contribution; the flat store keeps its
// gradients in ONE buffer for the bucketed all-reduce and the fused AdamW, so "assign the first contribution" is a flag on the GEMM
// (dh_gemm_args.accumulate = 2) and everything that GEMMs do not write first is cleared here.
#include "dh_common.h"

// table: [n][2] int64 (lo, hi) element offsets, lo % 4 == 0 and hi % 4 == 0 (slots are 64-element aligned); grid (X, n)
__global__ __launch_bounds__(256) void zero_ranges_kernel(float* __restrict__ base, const long long* __restrict__ table) {
  const long long lo = table[2 * blockIdx.y], hi = table[2 * blockIdx.y + 1];
  const long long stride = (long long)gridDim.x * blockDim.x * 4;
  const f32x4_t z = {0.f, 0.f, 0.f, 0.f};
  for (long long i = lo + ((long long)blockIdx.x * blockDim.x + threadIdx.x) * 4; i < hi; i += stride)
    *reinterpret_cast<f32x4_t*>(base + i) = z;
}

extern "C" int dh_zero_ranges(float* base, const int64_t* table_dev, int n, int64_t max_len, dh_stream_t stream) {
  DH_REQUIRE(base && table_dev && n >= 0 && (((uintptr_t)base) & 15) == 0, "dh_zero_ranges: bad arguments");
  if (n == 0) return DH_OK;
  // enough blocks in x for the longest range to be streamed by a few thousand threads; short ranges leave most of them idle at once
  long long bx = (max_len / 4 + 255) / 256;
  if (bx > 256) bx = 256;
  if (bx < 1) bx = 1;
  hipLaunchKernelGGL(zero_ranges_kernel, dim3((unsigned)bx, (unsigned)n), dim3(256), 0, (hipStream_t)stream, base, (const long long*)table_dev);
  DH_CHECK_LAUNCH();
  return DH_OK;
}
