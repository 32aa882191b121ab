// FILIP token-wise late interaction (model/filip.py:71-106):
//   * token selection: per sample, the 16 image tokens with the largest summed similarity to the caption's tokens and
//     vice versa.  sum_t <a_j, b_t> = <a_j, sum_t b_t>, so the [49,77] cross matrix is never formed.
//   * max-sim reduction: logits[i,l] = scale * mean_j max_m S[(i,j),(l,m)] from the token-similarity GEMM output
//     S [b*J, B*16]; the arg-max m is kept (uint8) for the backward.
//   * backward scatter: G[(i,j),(l,m)] = scale * dlogits[i,l] / J at m = argmax, 0 elsewhere (feeds two MFMA GEMMs).
#include "dh_common.h"

namespace {

constexpr int TOPK = 16;

// block per sample; a [J][D], b [T][D] fp32 (already L2-normalised); J,T <= 128, D <= 1024
__global__ __launch_bounds__(256) void filip_select_kernel(const float* __restrict__ A, const float* __restrict__ Bt, int J, int T, int D,
                                                           int64_t* __restrict__ idx_a, int64_t* __restrict__ idx_b) {
  DH_DYN_LDS(float, sm);
  float* sA = sm;            // [D] sum over image tokens
  float* sB = sA + D;        // [D] sum over text tokens
  float* sc = sB + D;        // [256] scores (image then text)
  const int s = blockIdx.x;
  const float* a = A + (long)s * J * D;
  const float* b = Bt + (long)s * T * D;
  for (int k = threadIdx.x; k < D; k += 256) {
    float x = 0.f, y = 0.f;
    for (int j = 0; j < J; ++j) x += a[(long)j * D + k];
    for (int t = 0; t < T; ++t) y += b[(long)t * D + k];
    sA[k] = x; sB[k] = y;
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int r = wave; r < J + T; r += 4) {
    const float* row = r < J ? a + (long)r * D : b + (long)(r - J) * D;
    const float* other = r < J ? sB : sA;
    float d = 0.f;
    for (int k = lane; k < D; k += 64) d = fmaf(row[k], other[k], d);
    d = wave_sum(d);
    if (lane == 0) sc[r] = d;
  }
  __syncthreads();
  // top-16 by repeated arg-max (first maximum wins), wave 0 -> image tokens, wave 1 -> text tokens
  if (wave < 2) {
    float* s0 = wave == 0 ? sc : sc + J;
    const int n = wave == 0 ? J : T;
    int64_t* out = (wave == 0 ? idx_a : idx_b) + (long)s * TOPK;
    for (int it = 0; it < TOPK; ++it) {
      float bv = -INFINITY; int bi = 0x7fffffff;
      for (int i = lane; i < n; i += 64) { float v = s0[i]; if (v > bv) { bv = v; bi = i; } }
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) {
        const float v2 = __shfl_xor(bv, o, 64); const int i2 = __shfl_xor(bi, o, 64);
        if (v2 > bv || (v2 == bv && i2 < bi)) { bv = v2; bi = i2; }
      }
      if (lane == 0) { out[it] = bi; s0[bi] = -INFINITY; }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      __builtin_amdgcn_wave_barrier();
    }
  }
}

// S [b*J][B*16] (ld = lds) -> raw[i,l] = mean_j max_m S, arg [b*J][B] (uint8).  grid (B/64, b); block 256 = 64 l x 4 j-lanes
template <typename T>
__global__ __launch_bounds__(256) void maxsim_reduce_kernel(const T* __restrict__ S, long lds, int b, int B, int J,
                                                            const float* __restrict__ scale_p, float* __restrict__ logits,
                                                            float* __restrict__ raw, unsigned char* __restrict__ arg) {
  __shared__ float red[4][64];
  const int l = blockIdx.x * 64 + (threadIdx.x & 63);
  const int jl = threadIdx.x >> 6;
  const int i = blockIdx.y;
  float acc = 0.f;
  if (l < B)
    for (int j = jl; j < J; j += 4) {
      const T* p = S + ((long)i * J + j) * lds + (long)l * TOPK;
      float v[16];
      ld8(p, v); ld8(p + 8, v + 8);
      float bv = v[0]; int bi = 0;
#pragma unroll
      for (int m = 1; m < 16; ++m) if (v[m] > bv) { bv = v[m]; bi = m; }
      acc += bv;
      arg[((long)i * J + j) * B + l] = (unsigned char)bi;
    }
  red[jl][threadIdx.x & 63] = acc;
  __syncthreads();
  if (jl == 0 && l < B) {
    const float r = (red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x]) / J;
    raw[(long)i * B + l] = r;
    logits[(long)i * B + l] = r * (*scale_p);
  }
}

// G [(i,j)][(l,m)] = scale * g[i,l] / J at m == arg, else 0; 16-B vector stores (8 elements)
template <typename T>
__global__ __launch_bounds__(256) void maxsim_scatter_kernel(const float* __restrict__ g, const unsigned char* __restrict__ arg,
                                                             const float* __restrict__ scale_p, T* __restrict__ G, long ldg, int b, int B, int J) {
  const float sc = (*scale_p) / J;
  const long total = (long)b * J * B * 2;       // (row, l, half)
  for (long t = (long)blockIdx.x * 256 + threadIdx.x; t < total; t += (long)gridDim.x * 256) {
    const int half = (int)(t & 1);
    const long rl = t >> 1;
    const int l = (int)(rl % B);
    const long row = rl / B;
    const int i = (int)(row / J);
    const int m = arg[row * B + l];
    float v[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if ((m >> 3) == half) v[m & 7] = sc * g[(long)i * B + l];
    st8(G + row * ldg + (long)l * TOPK + half * 8, v);
  }
}

__global__ __launch_bounds__(256) void maxsim_scale_kernel(const float* __restrict__ raw, const float* __restrict__ scale_p, float* __restrict__ logits, long n) {
  const float sc = *scale_p;
  for (long t = (long)blockIdx.x * 256 + threadIdx.x; t < n; t += (long)gridDim.x * 256) logits[t] = raw[t] * sc;
}

// rows [r0, r0 + nrows) of G only (the backward walks G in row chunks through a small rotating buffer): Gc [nrows][B*16]
template <typename T>
__global__ __launch_bounds__(256) void maxsim_scatter_rows_kernel(const float* __restrict__ g, const unsigned char* __restrict__ arg,
                                                                  const float* __restrict__ scale_p, T* __restrict__ G, long ldg, int b, int B, int J,
                                                                  long r0, long nrows) {
  const float sc = (*scale_p) / J;
  const long total = nrows * B * 2;       // (row, l, half)
  for (long t = (long)blockIdx.x * 256 + threadIdx.x; t < total; t += (long)gridDim.x * 256) {
    const int half = (int)(t & 1);
    const long rl = t >> 1;
    const int l = (int)(rl % B);
    const long rr = rl / B;                // row inside the chunk
    const long row = r0 + rr;
    const int i = (int)(row / J);
    float v[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (i < b) {                           // padding rows beyond b*J: zero rows
      const int m = arg[row * B + l];
      if ((m >> 3) == half) v[m & 7] = sc * g[(long)i * B + l];
    }
    st8(G + rr * ldg + (long)l * TOPK + half * 8, v);
  }
}

}  // namespace

bool dh_maxsim_try_v4(const void* Q, const void* Ksel, int rows_pad, int b, int B, int J, int D, float* raw, uint8_t* arg, hipStream_t st);   // gemm_v4.hip

// Fused forward (bf16 token features): token-similarity GEMM on the persistent MFMA kernel with the max-over-m / mean-over-j
// reduction in its epilogue (gemm_v4.hip MODE_MAXSIM); the [b*J, B*16] similarity matrix is never written.
extern "C" int dh_maxsim_fused_fwd(const void* Q_bf16, const void* K_bf16, int rows_pad, int b, int B, int J, int D, const float* scale_dev,
                                   float* logits, float* raw, uint8_t* argmax, dh_stream_t stream) {
  hipStream_t st = (hipStream_t)stream;
  DH_REQUIRE(Q_bf16 && K_bf16 && scale_dev && logits && raw && argmax && b > 0 && B > 0 && J > 0 && rows_pad >= b * J, "dh_maxsim_fused_fwd: bad args");
  DH_REQUIRE(dh_maxsim_try_v4(Q_bf16, K_bf16, rows_pad, b, B, J, D, raw, argmax, st),
             "dh_maxsim_fused_fwd: needs rows_pad %% 256 == 0, B %% 16 == 0, D %% 64 == 0, D >= 128, J >= 19, 16-byte aligned operands (got rows_pad %d B %d D %d J %d)",
             rows_pad, B, D, J);
  DH_HELPER_FAILED();
  DH_CHECK_LAUNCH();
  const long n = (long)b * B;
  hipLaunchKernelGGL(maxsim_scale_kernel, dim3((int)((n + 255) / 256 > 1024 ? 1024 : (n + 255) / 256)), dim3(256), 0, st, raw, scale_dev, logits, n);
  DH_CHECK_LAUNCH();
  return DH_OK;
}

// G rows [r0, r0 + nrows) into a chunk buffer Gc [nrows][ldg >= B*16] (see maxsim_scatter_rows_kernel)
extern "C" int dh_maxsim_scatter_rows(int g_dtype, const float* dlogits, const uint8_t* argmax, const float* scale_dev, void* Gc,
                                      int64_t ldg, int b, int B, int J, int64_t r0, int64_t nrows, dh_stream_t stream) {
  hipStream_t st = (hipStream_t)stream;
  DH_REQUIRE(dlogits && argmax && scale_dev && Gc && b > 0 && B > 0 && J > 0 && r0 >= 0 && nrows > 0 && ldg >= (int64_t)B * TOPK, "dh_maxsim_scatter_rows: bad args");
  long total = (long)nrows * B * 2;
  long gsz = (total + 255) / 256;
  if (gsz > 65536) gsz = 65536;
  if (g_dtype == DH_BF16) hipLaunchKernelGGL(maxsim_scatter_rows_kernel<bf16_t>, dim3((int)gsz), dim3(256), 0, st, dlogits, argmax, scale_dev, (bf16_t*)Gc, (long)ldg, b, B, J, (long)r0, (long)nrows);
  else hipLaunchKernelGGL(maxsim_scatter_rows_kernel<float>, dim3((int)gsz), dim3(256), 0, st, dlogits, argmax, scale_dev, (float*)Gc, (long)ldg, b, B, J, (long)r0, (long)nrows);
  DH_CHECK_LAUNCH();
  return DH_OK;
}

extern "C" int dh_filip_select(const float* img_tok, const float* txt_tok, int b, int J, int T, int D, int64_t* idx_img,
                               int64_t* idx_txt, dh_stream_t stream) {
  hipStream_t st = (hipStream_t)stream;
  DH_REQUIRE(img_tok && txt_tok && idx_img && idx_txt && b > 0 && J >= TOPK && T >= TOPK && J + T <= 256 && D <= 2048,
             "dh_filip_select: bad args (16 <= J,T; J+T <= 256)");
  size_t lds = (size_t)(2 * D + 256) * sizeof(float);
  hipLaunchKernelGGL(filip_select_kernel, dim3(b), dim3(256), lds, st, img_tok, txt_tok, J, T, D, idx_img, idx_txt);
  DH_CHECK_LAUNCH();
  return DH_OK;
}

extern "C" int dh_maxsim_reduce(int s_dtype, const void* S, int64_t lds, int b, int B, int J, const float* scale_dev,
                                float* logits, float* raw, uint8_t* argmax, dh_stream_t stream) {
  hipStream_t st = (hipStream_t)stream;
  DH_REQUIRE(S && scale_dev && logits && raw && argmax && b > 0 && B > 0 && J > 0 && lds >= (int64_t)B * TOPK,
             "dh_maxsim_reduce: bad args");
  dim3 grid(dh_cdiv(B, 64), b);
  if (s_dtype == DH_BF16) hipLaunchKernelGGL(maxsim_reduce_kernel<bf16_t>, grid, dim3(256), 0, st, (const bf16_t*)S, (long)lds, b, B, J, scale_dev, logits, raw, argmax);
  else hipLaunchKernelGGL(maxsim_reduce_kernel<float>, grid, dim3(256), 0, st, (const float*)S, (long)lds, b, B, J, scale_dev, logits, raw, argmax);
  DH_CHECK_LAUNCH();
  return DH_OK;
}

extern "C" int dh_maxsim_scatter(int g_dtype, const float* dlogits, const uint8_t* argmax, const float* scale_dev, void* G,
                                 int64_t ldg, int b, int B, int J, dh_stream_t stream) {
  hipStream_t st = (hipStream_t)stream;
  DH_REQUIRE(dlogits && argmax && scale_dev && G && b > 0 && B > 0 && J > 0 && ldg >= (int64_t)B * TOPK, "dh_maxsim_scatter: bad args");
  long total = (long)b * J * B * 2;
  long gsz = (total + 255) / 256;
  if (gsz > 65536) gsz = 65536;
  if (g_dtype == DH_BF16) hipLaunchKernelGGL(maxsim_scatter_kernel<bf16_t>, dim3((int)gsz), dim3(256), 0, st, dlogits, argmax, scale_dev, (bf16_t*)G, (long)ldg, b, B, J);
  else hipLaunchKernelGGL(maxsim_scatter_kernel<float>, dim3((int)gsz), dim3(256), 0, st, dlogits, argmax, scale_dev, (float*)G, (long)ldg, b, B, J);
  DH_CHECK_LAUNCH();
  return DH_OK;
}
