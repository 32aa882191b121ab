// ModifiedResNet tower kernels (reference: prototype/model/image_encoder/modified_resnet.py).
//
// Layout: every activation of the tower is NHWC, i.e. a row-major matrix [N*H*W, C] of "pixel rows" -- the same
// token-major layout the transformer towers use.  With it
//   * a 1x1 convolution IS a GEMM on the activation as stored (no copies),
//   * a 3x3 convolution is conv_rows (below) + a GEMM against conv.weight.view(Cout, Cin*9) as stored,
//   * BatchNorm statistics are column reductions, and the attention pool's token matrix [b, HW, C] is the last
//     feature map as stored (the reference permutes NCHW -> (HW)NC, modified_resnet.py:70).
// Everything here is HBM-bound: 16-byte accesses, one pass per tensor, deterministic two-level reductions (no atomics).
#include "dh_common.h"

#ifndef DH_GRID_CAP
#define DH_GRID_CAP 4096      // grid-stride loops cover the rest (a host test build sets a small cap to exercise them)
#endif
namespace {
int grid_for(long work_items) {
  long g = (work_items + 255) / 256;
  if (g > DH_GRID_CAP) g = DH_GRID_CAP;
  if (g < 1) g = 1;
  return (int)g;
}
}  // namespace

// ------------------------------------------------------------------------------------------------ conv patches
// rows[(n, oy, ox)][c*9 + ky*3 + kx] = x[n, oy*stride - pad + ky, ox*stride - pad + kx, c]  (0 outside the image).
// The inner order (c, ky, kx) is the one of conv.weight.view(Cout, Cin*9): the weight (and its gradient in the flat
// buffer) is used as stored.  One task = one output pixel x 8 channels: 9 coalesced 16-byte loads, transposed in
// registers, 72 consecutive outputs.
template <typename T>
__global__ __launch_bounds__(256) void conv3_rows_nhwc_kernel(const T* __restrict__ x, T* __restrict__ rows, int N, int H,
                                                              int W, int C, int Ho, int Wo, int stride, int pad) {
  const int cgs = C >> 3;
  const long ntask = (long)N * Ho * Wo * cgs;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < ntask; i += (long)gridDim.x * 256) {
    const int cg = (int)(i % cgs);
    const long row = i / cgs;
    long r = row;
    const int ox = (int)(r % Wo); r /= Wo;
    const int oy = (int)(r % Ho);
    const int n = (int)(r / Ho);
    float v[9][8];
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        const int iy = oy * stride - pad + ky, ix = ox * stride - pad + kx;
        if (iy >= 0 && iy < H && ix >= 0 && ix < W) {
          ld8(x + (((long)n * H + iy) * W + ix) * C + cg * 8, v[ky * 3 + kx]);
        } else {
#pragma unroll
          for (int j = 0; j < 8; ++j) v[ky * 3 + kx][j] = 0.f;
        }
      }
    T* dst = rows + row * (9L * C) + (long)cg * 72;
#pragma unroll
    for (int q = 0; q < 9; ++q) {
      float o[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int e = q * 8 + u;          // e = j * 9 + t  (channel j of the group, tap t)
        o[u] = v[e % 9][e / 9];
      }
      st8(dst + q * 8, o);
    }
  }
}

// Stem conv1 (modified_resnet.py:144): patches of the fp32 NCHW image batch (c_total channels, view at c0), 3 channels,
// K = 27 zero-padded to Kpad (a multiple of 8).  One task = one output pixel.
template <typename T, int KPAD>
__global__ __launch_bounds__(256) void conv3_rows_image_kernel(const float* __restrict__ img, int c_total, int c0,
                                                               T* __restrict__ rows, int N, int H, int W, int Ho, int Wo,
                                                               int stride, int pad) {
  const long ntask = (long)N * Ho * Wo;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < ntask; i += (long)gridDim.x * 256) {
    long r = i;
    const int ox = (int)(r % Wo); r /= Wo;
    const int oy = (int)(r % Ho);
    const int n = (int)(r / Ho);
    float o[KPAD];
#pragma unroll
    for (int e = 0; e < KPAD; ++e) o[e] = 0.f;
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
      for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
          const int iy = oy * stride - pad + ky, ix = ox * stride - pad + kx;
          if (iy >= 0 && iy < H && ix >= 0 && ix < W) o[c * 9 + ky * 3 + kx] = img[(((long)n * c_total + c0 + c) * H + iy) * W + ix];
        }
    T* dst = rows + i * KPAD;
#pragma unroll
    for (int q = 0; q < KPAD / 8; ++q) st8(dst + q * 8, o + q * 8);
  }
}

extern "C" int dh_conv_rows(int dtype, const void* src, int src_layout, int c_total, int c0, void* rows, int N, int H, int W,
                            int C, int k, int stride, int pad, int Kpad, dh_stream_t stream) {
  hipStream_t st = (hipStream_t)stream;
  DH_REQUIRE(src && rows && N > 0 && H > 0 && W > 0 && C > 0, "dh_conv_rows: bad args");
  DH_REQUIRE(k == 3 && stride >= 1 && pad >= 0, "dh_conv_rows: only 3x3 windows are implemented (k=%d)", k);
  DH_REQUIRE(dtype == DH_F32 || dtype == DH_BF16, "dh_conv_rows: bad dtype");
  const int Ho = (H + 2 * pad - 3) / stride + 1, Wo = (W + 2 * pad - 3) / stride + 1;
  DH_REQUIRE(Ho > 0 && Wo > 0, "dh_conv_rows: empty output");
  if (src_layout == 0) {
    DH_REQUIRE(C % 8 == 0 && Kpad == 9 * C, "dh_conv_rows: NHWC source needs C %% 8 == 0 and Kpad == 9*C (C=%d Kpad=%d)", C, Kpad);
    const long ntask = (long)N * Ho * Wo * (C / 8);
    if (dtype == DH_BF16)
      hipLaunchKernelGGL(conv3_rows_nhwc_kernel<bf16_t>, dim3(grid_for(ntask)), dim3(256), 0, st, (const bf16_t*)src, (bf16_t*)rows, N, H, W, C, Ho, Wo, stride, pad);
    else
      hipLaunchKernelGGL(conv3_rows_nhwc_kernel<float>, dim3(grid_for(ntask)), dim3(256), 0, st, (const float*)src, (float*)rows, N, H, W, C, Ho, Wo, stride, pad);
  } else {
    DH_REQUIRE(src_layout == 1 && C == 3 && Kpad == 32 && c0 >= 0 && c0 + 3 <= c_total,
               "dh_conv_rows: image source needs C == 3, Kpad == 32 and a valid channel window");
    const long ntask = (long)N * Ho * Wo;
    if (dtype == DH_BF16)
      hipLaunchKernelGGL((conv3_rows_image_kernel<bf16_t, 32>), dim3(grid_for(ntask)), dim3(256), 0, st, (const float*)src, c_total, c0, (bf16_t*)rows, N, H, W, Ho, Wo, stride, pad);
    else
      hipLaunchKernelGGL((conv3_rows_image_kernel<float, 32>), dim3(grid_for(ntask)), dim3(256), 0, st, (const float*)src, c_total, c0, (float*)rows, N, H, W, Ho, Wo, stride, pad);
  }
  DH_CHECK_LAUNCH();
  return DH_OK;
}

// ------------------------------------------------------------------------------------------------ BatchNorm2d
// Column reductions over R pixel rows, two levels: (1) every block reduces a slab of rows for a group of channels into
// partial[blk][2][C] (8 channels per thread, the row offsets of a block combined through LDS), (2) one thread per channel
// adds the partials in double.  Deterministic (no atomics), one pass over the data per level-1 kernel.
#define BN_MAX_SLABS 1024

static inline int bn_slabs(long R) {
  long s = (R + 255) / 256;
  if (s > BN_MAX_SLABS) s = BN_MAX_SLABS;
  if (s < 1) s = 1;
  return (int)s;
}

extern "C" int64_t dh_bn2d_ws_bytes(int rows, int C) {
  return (int64_t)sizeof(float) * ((int64_t)bn_slabs(rows) * 2 * C + 2 * (int64_t)C);
}

// MODE 0: (sum x, sum x^2);  MODE 1: (sum dyr, sum dyr * xhat), dyr = dy masked by (y > 0) when relu
template <typename T, int MODE>
__global__ __launch_bounds__(256) void bn2d_reduce_kernel(const T* __restrict__ x, const T* __restrict__ dy, const T* __restrict__ y,
                                                          const float* __restrict__ mean, const float* __restrict__ invstd,
                                                          float* __restrict__ partial, long R, int C, int cgb, int rows_per_slab,
                                                          int relu) {
  __shared__ float red[16][256];
  const int cgs = C >> 3;
  const int rpi = 256 / cgb;                       // row offsets per block iteration
  const int cgl = threadIdx.x % cgb, ro = threadIdx.x / cgb;
  const int cg = blockIdx.x * cgb + cgl;
  const bool active = ro < rpi && cg < cgs;
  float a0[8], a1[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) a0[j] = a1[j] = 0.f;
  float mu[8], is[8];
  if (MODE == 1 && active) { ld8(mean + cg * 8, mu); ld8(invstd + cg * 8, is); }
  const long r0 = (long)blockIdx.y * rows_per_slab;
  long r1 = r0 + rows_per_slab;
  if (r1 > R) r1 = R;
  if (active)
    for (long r = r0 + ro; r < r1; r += rpi) {
      float v[8];
      ld8(x + r * C + cg * 8, v);
      if (MODE == 0) {
#pragma unroll
        for (int j = 0; j < 8; ++j) { a0[j] += v[j]; a1[j] += v[j] * v[j]; }
      } else {
        float d[8];
        ld8(dy + r * C + cg * 8, d);
        if (relu) {
          float yy[8];
          ld8(y + r * C + cg * 8, yy);
#pragma unroll
          for (int j = 0; j < 8; ++j) if (yy[j] <= 0.f) d[j] = 0.f;
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) { a0[j] += d[j]; a1[j] += d[j] * ((v[j] - mu[j]) * is[j]); }
      }
    }
#pragma unroll
  for (int j = 0; j < 8; ++j) { red[j][threadIdx.x] = a0[j]; red[8 + j][threadIdx.x] = a1[j]; }
  __syncthreads();
  if (ro == 0 && cg < cgs) {
    float s0[8], s1[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { s0[j] = 0.f; s1[j] = 0.f; }
    for (int q = 0; q < rpi; ++q) {
#pragma unroll
      for (int j = 0; j < 8; ++j) { s0[j] += red[j][q * cgb + cgl]; s1[j] += red[8 + j][q * cgb + cgl]; }
    }
    float* p = partial + (long)blockIdx.y * 2 * C;
    st8(p + cg * 8, s0);
    st8(p + C + cg * 8, s1);
  }
}

// level 2 of the forward statistics: mean, biased variance -> invstd; running buffers move by `momentum` towards the
// batch mean / UNBIASED variance (torch.nn.BatchNorm2d semantics)
__global__ __launch_bounds__(256) void bn2d_stats_finalize_kernel(const float* __restrict__ partial, int slabs, long R, int C,
                                                                  float eps, float momentum, float* __restrict__ save_mean,
                                                                  float* __restrict__ save_invstd, float* __restrict__ run_mean,
                                                                  float* __restrict__ run_var) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= C) return;
  double s = 0.0, q = 0.0;
  for (int b = 0; b < slabs; ++b) { s += (double)partial[(long)b * 2 * C + c]; q += (double)partial[(long)b * 2 * C + C + c]; }
  const double m = s / (double)R;
  double var = q / (double)R - m * m;
  if (var < 0.0) var = 0.0;
  save_mean[c] = (float)m;
  save_invstd[c] = (float)(1.0 / sqrt(var + (double)eps));
  if (run_mean) {
    const double unbiased = R > 1 ? var * (double)R / (double)(R - 1) : var;
    run_mean[c] = (float)((1.0 - (double)momentum) * (double)run_mean[c] + (double)momentum * m);
    run_var[c] = (float)((1.0 - (double)momentum) * (double)run_var[c] + (double)momentum * unbiased);
  }
}

__global__ __launch_bounds__(256) void bn2d_eval_stats_kernel(const float* __restrict__ run_mean, const float* __restrict__ run_var,
                                                              int C, float eps, float* __restrict__ save_mean,
                                                              float* __restrict__ save_invstd) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= C) return;
  save_mean[c] = run_mean[c];
  save_invstd[c] = 1.f / sqrtf(run_var[c] + eps);
}

// y = relu?((x - mean) * invstd * w + b (+ residual))
template <typename T>
__global__ __launch_bounds__(256) void bn2d_apply_kernel(const T* __restrict__ x, const T* __restrict__ res, const float* __restrict__ w,
                                                         const float* __restrict__ b, const float* __restrict__ mean,
                                                         const float* __restrict__ invstd, T* __restrict__ y, long R, int C, int relu) {
  const int cgs = C >> 3;
  const long ntask = R * cgs;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < ntask; i += (long)gridDim.x * 256) {
    const int cg = (int)(i % cgs);
    float v[8], mu[8], is[8], ww[8], bb[8];
    ld8(x + i * 8, v);
    ld8(mean + cg * 8, mu); ld8(invstd + cg * 8, is); ld8(w + cg * 8, ww); ld8(b + cg * 8, bb);
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = (v[j] - mu[j]) * is[j] * ww[j] + bb[j];
    if (res) {
      float rr[8];
      ld8(res + i * 8, rr);
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] += rr[j];
    }
    if (relu) {
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = fmaxf(v[j], 0.f);
    }
    st8(y + i * 8, v);
  }
}

// level 2 of the backward reductions: dw += sum dyr*xhat, db += sum dyr (accumulate-into contract of the flat gradient
// buffer), and the two per-channel means the dx formula needs
__global__ __launch_bounds__(256) void bn2d_bwd_finalize_kernel(const float* __restrict__ partial, int slabs, long R, int C,
                                                                float* __restrict__ dw, float* __restrict__ db,
                                                                float* __restrict__ m12) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= C) return;
  double s1 = 0.0, s2 = 0.0;
  for (int b = 0; b < slabs; ++b) { s1 += (double)partial[(long)b * 2 * C + c]; s2 += (double)partial[(long)b * 2 * C + C + c]; }
  db[c] += (float)s1;
  dw[c] += (float)s2;
  m12[c] = (float)(s1 / (double)R);
  m12[C + c] = (float)(s2 / (double)R);
}

// dx = w * invstd * (dyr - mean(dyr) - xhat * mean(dyr * xhat));  dres = dyr (gradient of the residual branch)
template <typename T>
__global__ __launch_bounds__(256) void bn2d_bwd_apply_kernel(const T* __restrict__ dy, const T* __restrict__ x, const T* __restrict__ y,
                                                             const float* __restrict__ w, const float* __restrict__ mean,
                                                             const float* __restrict__ invstd, const float* __restrict__ m12,
                                                             T* __restrict__ dx, T* __restrict__ dres, long R, int C, int relu) {
  const int cgs = C >> 3;
  const long ntask = R * cgs;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < ntask; i += (long)gridDim.x * 256) {
    const int cg = (int)(i % cgs);
    float d[8], v[8], mu[8], is[8], ww[8], m1[8], m2[8];
    ld8(dy + i * 8, d);
    ld8(x + i * 8, v);
    ld8(mean + cg * 8, mu); ld8(invstd + cg * 8, is); ld8(w + cg * 8, ww); ld8(m12 + cg * 8, m1); ld8(m12 + C + cg * 8, m2);
    if (relu) {
      float yy[8];
      ld8(y + i * 8, yy);
#pragma unroll
      for (int j = 0; j < 8; ++j) if (yy[j] <= 0.f) d[j] = 0.f;
    }
    if (dres) st8(dres + i * 8, d);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float xh = (v[j] - mu[j]) * is[j];
      v[j] = ww[j] * is[j] * (d[j] - m1[j] - xh * m2[j]);
    }
    st8(dx + i * 8, v);
  }
}

static inline void bn_geometry(long R, int C, int* cgb, int* gx, int* slabs, int* rps) {
  const int cgs = C / 8;
  *cgb = cgs < 256 ? cgs : 256;
  *gx = (cgs + *cgb - 1) / *cgb;
  *slabs = bn_slabs(R);
  *rps = (int)((R + *slabs - 1) / *slabs);
  *slabs = (int)((R + *rps - 1) / *rps);
}

extern "C" int dh_bn2d_fwd(int dtype, const void* x, const void* residual, const float* w, const float* b, void* y, float* save_mean,
                           float* save_invstd, float* running_mean, float* running_var, int rows, int C, float eps, float momentum,
                           int relu, int training, void* ws, int64_t ws_bytes, dh_stream_t stream) {
  hipStream_t st = (hipStream_t)stream;
  DH_REQUIRE(x && w && b && y && save_mean && save_invstd && rows > 0 && C > 0, "dh_bn2d_fwd: bad args");
  DH_REQUIRE(C % 8 == 0, "dh_bn2d_fwd: C must be a multiple of 8 (C=%d)", C);
  DH_REQUIRE(dtype == DH_F32 || dtype == DH_BF16, "dh_bn2d_fwd: bad dtype");
  DH_REQUIRE(training || (running_mean && running_var), "dh_bn2d_fwd: eval needs running stats");
  DH_REQUIRE((running_mean == nullptr) == (running_var == nullptr), "dh_bn2d_fwd: running_mean / running_var go together");
  const long R = rows;
  if (training) {
    DH_REQUIRE(ws && ws_bytes >= dh_bn2d_ws_bytes(rows, C), "dh_bn2d_fwd: workspace too small");
    int cgb, gx, slabs, rps;
    bn_geometry(R, C, &cgb, &gx, &slabs, &rps);
    float* partial = (float*)ws;
    if (dtype == DH_BF16)
      hipLaunchKernelGGL((bn2d_reduce_kernel<bf16_t, 0>), dim3(gx, slabs), dim3(256), 0, st, (const bf16_t*)x, (const bf16_t*)nullptr,
                         (const bf16_t*)nullptr, (const float*)nullptr, (const float*)nullptr, partial, R, C, cgb, rps, 0);
    else
      hipLaunchKernelGGL((bn2d_reduce_kernel<float, 0>), dim3(gx, slabs), dim3(256), 0, st, (const float*)x, (const float*)nullptr,
                         (const float*)nullptr, (const float*)nullptr, (const float*)nullptr, partial, R, C, cgb, rps, 0);
    DH_CHECK_LAUNCH();
    hipLaunchKernelGGL(bn2d_stats_finalize_kernel, dim3(dh_cdiv(C, 256)), dim3(256), 0, st, (const float*)partial, slabs, R, C, eps,
                       momentum, save_mean, save_invstd, running_mean, running_var);
  } else {
    hipLaunchKernelGGL(bn2d_eval_stats_kernel, dim3(dh_cdiv(C, 256)), dim3(256), 0, st, (const float*)running_mean,
                       (const float*)running_var, C, eps, save_mean, save_invstd);
  }
  DH_CHECK_LAUNCH();
  const long ntask = R * (C / 8);
  if (dtype == DH_BF16)
    hipLaunchKernelGGL(bn2d_apply_kernel<bf16_t>, dim3(grid_for(ntask)), dim3(256), 0, st, (const bf16_t*)x, (const bf16_t*)residual, w, b,
                       (const float*)save_mean, (const float*)save_invstd, (bf16_t*)y, R, C, relu);
  else
    hipLaunchKernelGGL(bn2d_apply_kernel<float>, dim3(grid_for(ntask)), dim3(256), 0, st, (const float*)x, (const float*)residual, w, b,
                       (const float*)save_mean, (const float*)save_invstd, (float*)y, R, C, relu);
  DH_CHECK_LAUNCH();
  return DH_OK;
}

extern "C" int dh_bn2d_bwd(int dtype, const void* dy, const void* x, const void* y, const float* w, const float* save_mean,
                           const float* save_invstd, void* dx, void* dres, float* dw, float* db, int rows, int C, int relu, void* ws,
                           int64_t ws_bytes, dh_stream_t stream) {
  hipStream_t st = (hipStream_t)stream;
  DH_REQUIRE(dy && x && w && save_mean && save_invstd && dx && dw && db && (!relu || y) && rows > 0 && C > 0, "dh_bn2d_bwd: bad args");
  DH_REQUIRE(C % 8 == 0, "dh_bn2d_bwd: C must be a multiple of 8 (C=%d)", C);
  DH_REQUIRE(dtype == DH_F32 || dtype == DH_BF16, "dh_bn2d_bwd: bad dtype");
  DH_REQUIRE(ws && ws_bytes >= dh_bn2d_ws_bytes(rows, C), "dh_bn2d_bwd: workspace too small");
  const long R = rows;
  int cgb, gx, slabs, rps;
  bn_geometry(R, C, &cgb, &gx, &slabs, &rps);
  float* partial = (float*)ws;
  float* m12 = partial + (long)bn_slabs(R) * 2 * C;
  if (dtype == DH_BF16)
    hipLaunchKernelGGL((bn2d_reduce_kernel<bf16_t, 1>), dim3(gx, slabs), dim3(256), 0, st, (const bf16_t*)x, (const bf16_t*)dy, (const bf16_t*)y,
                       save_mean, save_invstd, partial, R, C, cgb, rps, relu);
  else
    hipLaunchKernelGGL((bn2d_reduce_kernel<float, 1>), dim3(gx, slabs), dim3(256), 0, st, (const float*)x, (const float*)dy, (const float*)y,
                       save_mean, save_invstd, partial, R, C, cgb, rps, relu);
  DH_CHECK_LAUNCH();
  hipLaunchKernelGGL(bn2d_bwd_finalize_kernel, dim3(dh_cdiv(C, 256)), dim3(256), 0, st, (const float*)partial, slabs, R, C, dw, db, m12);
  DH_CHECK_LAUNCH();
  const long ntask = R * (C / 8);
  if (dtype == DH_BF16)
    hipLaunchKernelGGL(bn2d_bwd_apply_kernel<bf16_t>, dim3(grid_for(ntask)), dim3(256), 0, st, (const bf16_t*)dy, (const bf16_t*)x, (const bf16_t*)y,
                       w, save_mean, save_invstd, (const float*)m12, (bf16_t*)dx, (bf16_t*)dres, R, C, relu);
  else
    hipLaunchKernelGGL(bn2d_bwd_apply_kernel<float>, dim3(grid_for(ntask)), dim3(256), 0, st, (const float*)dy, (const float*)x, (const float*)y,
                       w, save_mean, save_invstd, (const float*)m12, (float*)dx, (float*)dres, R, C, relu);
  DH_CHECK_LAUNCH();
  return DH_OK;
}

// ------------------------------------------------------------------------------------------------ BatchNorm2d across ranks
// Synchronised BatchNorm (the `use_sync_bn: True` branch of modified_resnet.py:118-140; semantics of
// torch.nn.SyncBatchNorm): the statistics of a layer are per-channel SUMS, so the exchange between ranks is one SUM
// all-reduce of [2C + 1] doubles (sums, sums of squares / of dy*xhat, row count) between the reduction and the apply pass.
// The three stages of dh_bn2d_fwd / dh_bn2d_bwd as separate entry points; the row count travels in DEVICE memory (no host
// round trip between the collective and the apply pass).
__global__ __launch_bounds__(256) void bn2d_sums_finalize_kernel(const float* __restrict__ partial, int slabs, long R, int C,
                                                                 double* __restrict__ sums) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c == 0) sums[2 * C] = (double)R;
  if (c >= C) return;
  double s = 0.0, q = 0.0;
  for (int b = 0; b < slabs; ++b) { s += (double)partial[(long)b * 2 * C + c]; q += (double)partial[(long)b * 2 * C + C + c]; }
  sums[c] = s;
  sums[C + c] = q;
}

__global__ __launch_bounds__(256) void bn2d_stats_from_sums_kernel(const double* __restrict__ sums, int C, float eps, float momentum,
                                                                   float* __restrict__ save_mean, float* __restrict__ save_invstd,
                                                                   float* __restrict__ run_mean, float* __restrict__ run_var) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= C) return;
  const double n = sums[2 * C];
  const double m = sums[c] / n;
  double var = sums[C + c] / n - m * m;
  if (var < 0.0) var = 0.0;
  save_mean[c] = (float)m;
  save_invstd[c] = (float)(1.0 / sqrt(var + (double)eps));
  if (run_mean) {
    const double unbiased = n > 1.0 ? var * n / (n - 1.0) : var;
    run_mean[c] = (float)((1.0 - (double)momentum) * (double)run_mean[c] + (double)momentum * m);
    run_var[c] = (float)((1.0 - (double)momentum) * (double)run_var[c] + (double)momentum * unbiased);
  }
}

// dw / db take the LOCAL sums (the parameter gradients are summed over ranks by the gradient all-reduce); the two means of
// the dx formula are GLOBAL sums over the global row count
__global__ __launch_bounds__(256) void bn2d_bwd_from_sums_kernel(const double* __restrict__ local, const double* __restrict__ global,
                                                                 int C, float* __restrict__ dw, float* __restrict__ db,
                                                                 float* __restrict__ m12) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= C) return;
  db[c] += (float)local[c];
  dw[c] += (float)local[C + c];
  const double n = global[2 * C];
  m12[c] = (float)(global[c] / n);
  m12[C + c] = (float)(global[C + c] / n);
}

extern "C" int dh_bn2d_sums(int dtype, int mode, const void* x, const void* dy, const void* y, const float* mean, const float* invstd,
                            int relu, int rows, int C, double* sums, void* ws, int64_t ws_bytes, dh_stream_t stream) {
  hipStream_t st = (hipStream_t)stream;
  DH_REQUIRE(x && sums && rows > 0 && C > 0 && (mode == 0 || mode == 1), "dh_bn2d_sums: bad args");
  DH_REQUIRE(mode == 0 || (dy && mean && invstd && (!relu || y)), "dh_bn2d_sums: mode 1 needs dy, mean, invstd (and y with relu)");
  DH_REQUIRE(C % 8 == 0, "dh_bn2d_sums: C must be a multiple of 8 (C=%d)", C);
  DH_REQUIRE(dtype == DH_F32 || dtype == DH_BF16, "dh_bn2d_sums: bad dtype");
  DH_REQUIRE(ws && ws_bytes >= dh_bn2d_ws_bytes(rows, C), "dh_bn2d_sums: workspace too small");
  const long R = rows;
  int cgb, gx, slabs, rps;
  bn_geometry(R, C, &cgb, &gx, &slabs, &rps);
  float* partial = (float*)ws;
  if (mode == 0) {
    if (dtype == DH_BF16)
      hipLaunchKernelGGL((bn2d_reduce_kernel<bf16_t, 0>), dim3(gx, slabs), dim3(256), 0, st, (const bf16_t*)x, (const bf16_t*)nullptr,
                         (const bf16_t*)nullptr, (const float*)nullptr, (const float*)nullptr, partial, R, C, cgb, rps, 0);
    else
      hipLaunchKernelGGL((bn2d_reduce_kernel<float, 0>), dim3(gx, slabs), dim3(256), 0, st, (const float*)x, (const float*)nullptr,
                         (const float*)nullptr, (const float*)nullptr, (const float*)nullptr, partial, R, C, cgb, rps, 0);
  } else {
    if (dtype == DH_BF16)
      hipLaunchKernelGGL((bn2d_reduce_kernel<bf16_t, 1>), dim3(gx, slabs), dim3(256), 0, st, (const bf16_t*)x, (const bf16_t*)dy, (const bf16_t*)y,
                         mean, invstd, partial, R, C, cgb, rps, relu);
    else
      hipLaunchKernelGGL((bn2d_reduce_kernel<float, 1>), dim3(gx, slabs), dim3(256), 0, st, (const float*)x, (const float*)dy, (const float*)y,
                         mean, invstd, partial, R, C, cgb, rps, relu);
  }
  DH_CHECK_LAUNCH();
  hipLaunchKernelGGL(bn2d_sums_finalize_kernel, dim3(dh_cdiv(C + 1, 256)), dim3(256), 0, st, (const float*)partial, slabs, R, C, sums);
  DH_CHECK_LAUNCH();
  return DH_OK;
}

extern "C" int dh_bn2d_fwd_apply(int dtype, const void* x, const void* residual, const float* w, const float* b, const double* sums,
                                 void* y, float* save_mean, float* save_invstd, float* running_mean, float* running_var, int rows, int C,
                                 float eps, float momentum, int relu, dh_stream_t stream) {
  hipStream_t st = (hipStream_t)stream;
  DH_REQUIRE(x && w && b && sums && y && save_mean && save_invstd && rows > 0 && C > 0, "dh_bn2d_fwd_apply: bad args");
  DH_REQUIRE(C % 8 == 0, "dh_bn2d_fwd_apply: C must be a multiple of 8 (C=%d)", C);
  DH_REQUIRE(dtype == DH_F32 || dtype == DH_BF16, "dh_bn2d_fwd_apply: bad dtype");
  DH_REQUIRE((running_mean == nullptr) == (running_var == nullptr), "dh_bn2d_fwd_apply: running_mean / running_var go together");
  hipLaunchKernelGGL(bn2d_stats_from_sums_kernel, dim3(dh_cdiv(C, 256)), dim3(256), 0, st, sums, C, eps, momentum, save_mean, save_invstd,
                     running_mean, running_var);
  DH_CHECK_LAUNCH();
  const long R = rows, ntask = R * (C / 8);
  if (dtype == DH_BF16)
    hipLaunchKernelGGL(bn2d_apply_kernel<bf16_t>, dim3(grid_for(ntask)), dim3(256), 0, st, (const bf16_t*)x, (const bf16_t*)residual, w, b,
                       (const float*)save_mean, (const float*)save_invstd, (bf16_t*)y, R, C, relu);
  else
    hipLaunchKernelGGL(bn2d_apply_kernel<float>, dim3(grid_for(ntask)), dim3(256), 0, st, (const float*)x, (const float*)residual, w, b,
                       (const float*)save_mean, (const float*)save_invstd, (float*)y, R, C, relu);
  DH_CHECK_LAUNCH();
  return DH_OK;
}

extern "C" int dh_bn2d_bwd_apply(int dtype, const void* dy, const void* x, const void* y, const float* w, const float* save_mean,
                                 const float* save_invstd, const double* sums_local, const double* sums_global, void* dx, void* dres,
                                 float* dw, float* db, int rows, int C, int relu, void* ws, int64_t ws_bytes, dh_stream_t stream) {
  hipStream_t st = (hipStream_t)stream;
  DH_REQUIRE(dy && x && w && save_mean && save_invstd && sums_local && sums_global && dx && dw && db && (!relu || y) && rows > 0 && C > 0,
             "dh_bn2d_bwd_apply: bad args");
  DH_REQUIRE(C % 8 == 0, "dh_bn2d_bwd_apply: C must be a multiple of 8 (C=%d)", C);
  DH_REQUIRE(dtype == DH_F32 || dtype == DH_BF16, "dh_bn2d_bwd_apply: bad dtype");
  DH_REQUIRE(ws && ws_bytes >= (int64_t)sizeof(float) * 2 * C, "dh_bn2d_bwd_apply: workspace too small (2*C floats)");
  float* m12 = (float*)ws;
  hipLaunchKernelGGL(bn2d_bwd_from_sums_kernel, dim3(dh_cdiv(C, 256)), dim3(256), 0, st, sums_local, sums_global, C, dw, db, m12);
  DH_CHECK_LAUNCH();
  const long R = rows, ntask = R * (C / 8);
  if (dtype == DH_BF16)
    hipLaunchKernelGGL(bn2d_bwd_apply_kernel<bf16_t>, dim3(grid_for(ntask)), dim3(256), 0, st, (const bf16_t*)dy, (const bf16_t*)x, (const bf16_t*)y,
                       w, save_mean, save_invstd, (const float*)m12, (bf16_t*)dx, (bf16_t*)dres, R, C, relu);
  else
    hipLaunchKernelGGL(bn2d_bwd_apply_kernel<float>, dim3(grid_for(ntask)), dim3(256), 0, st, (const float*)dy, (const float*)x, (const float*)y,
                       w, save_mean, save_invstd, (const float*)m12, (float*)dx, (float*)dres, R, C, relu);
  DH_CHECK_LAUNCH();
  return DH_OK;
}

// ------------------------------------------------------------------------------------------------ average pool
// nn.AvgPool2d(k) on NHWC (modified_resnet.py:26,36,149): y[n, oy, ox, :] = mean of the k x k window.
template <typename T>
__global__ __launch_bounds__(256) void avgpool_fwd_kernel(const T* __restrict__ x, T* __restrict__ y, int N, int H, int W, int C, int k) {
  const int cgs = C >> 3, Ho = H / k, Wo = W / k;
  const long ntask = (long)N * Ho * Wo * cgs;
  const float inv = 1.f / (float)(k * k);
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < ntask; i += (long)gridDim.x * 256) {
    const int cg = (int)(i % cgs);
    long r = i / cgs;
    const int ox = (int)(r % Wo); r /= Wo;
    const int oy = (int)(r % Ho);
    const int n = (int)(r / Ho);
    float acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = 0.f;
    for (int dy = 0; dy < k; ++dy)
      for (int dx = 0; dx < k; ++dx) {
        float v[8];
        ld8(x + (((long)n * H + oy * k + dy) * W + ox * k + dx) * C + cg * 8, v);
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] += v[j];
      }
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] *= inv;
    st8(y + i * 8, acc);
  }
}

template <typename T>
__global__ __launch_bounds__(256) void avgpool_bwd_kernel(const T* __restrict__ dy, T* __restrict__ dx, int N, int H, int W, int C, int k) {
  const int cgs = C >> 3, Ho = H / k, Wo = W / k;
  const long ntask = (long)N * H * W * cgs;
  const float inv = 1.f / (float)(k * k);
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < ntask; i += (long)gridDim.x * 256) {
    const int cg = (int)(i % cgs);
    long r = i / cgs;
    const int ix = (int)(r % W); r /= W;
    const int iy = (int)(r % H);
    const int n = (int)(r / H);
    float v[8];
    ld8(dy + (((long)n * Ho + iy / k) * Wo + ix / k) * C + cg * 8, v);
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] *= inv;
    st8(dx + i * 8, v);
  }
}

extern "C" int dh_avgpool_fwd(int dtype, const void* x, void* y, int N, int H, int W, int C, int k, dh_stream_t stream) {
  hipStream_t st = (hipStream_t)stream;
  DH_REQUIRE(x && y && N > 0 && k >= 1 && H % k == 0 && W % k == 0 && C % 8 == 0 && C > 0 && H > 0 && W > 0,
             "dh_avgpool_fwd: bad args (H, W must be multiples of k, C of 8)");
  const long ntask = (long)N * (H / k) * (W / k) * (C / 8);
  if (dtype == DH_BF16) hipLaunchKernelGGL(avgpool_fwd_kernel<bf16_t>, dim3(grid_for(ntask)), dim3(256), 0, st, (const bf16_t*)x, (bf16_t*)y, N, H, W, C, k);
  else if (dtype == DH_F32) hipLaunchKernelGGL(avgpool_fwd_kernel<float>, dim3(grid_for(ntask)), dim3(256), 0, st, (const float*)x, (float*)y, N, H, W, C, k);
  else DH_FAIL(DH_ERR_ARG, "dh_avgpool_fwd: bad dtype");
  DH_CHECK_LAUNCH();
  return DH_OK;
}

extern "C" int dh_avgpool_bwd(int dtype, const void* dy, void* dx, int N, int H, int W, int C, int k, dh_stream_t stream) {
  hipStream_t st = (hipStream_t)stream;
  DH_REQUIRE(dy && dx && N > 0 && k >= 1 && H % k == 0 && W % k == 0 && C % 8 == 0 && C > 0 && H > 0 && W > 0,
             "dh_avgpool_bwd: bad args (H, W must be multiples of k, C of 8)");
  const long ntask = (long)N * H * W * (C / 8);
  if (dtype == DH_BF16) hipLaunchKernelGGL(avgpool_bwd_kernel<bf16_t>, dim3(grid_for(ntask)), dim3(256), 0, st, (const bf16_t*)dy, (bf16_t*)dx, N, H, W, C, k);
  else if (dtype == DH_F32) hipLaunchKernelGGL(avgpool_bwd_kernel<float>, dim3(grid_for(ntask)), dim3(256), 0, st, (const float*)dy, (float*)dx, N, H, W, C, k);
  else DH_FAIL(DH_ERR_ARG, "dh_avgpool_bwd: bad dtype");
  DH_CHECK_LAUNCH();
  return DH_OK;
}

// ------------------------------------------------------------------------------------------------ attention-pool tokens
// AttentionPool2d (modified_resnet.py:70-72): tok[b, 0] = mean_l x[b, l] + pos[0];  tok[b, l + 1] = x[b, l] + pos[l + 1].
template <typename T>
__global__ __launch_bounds__(256) void attnpool_tokens_fwd_kernel(const T* __restrict__ x, const float* __restrict__ pos, T* __restrict__ tok,
                                                                  int b, int HW, int C) {
  const int cgs = C >> 3, L = HW + 1;
  const long ntask = (long)b * L * cgs;
  const float inv = 1.f / (float)HW;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < ntask; i += (long)gridDim.x * 256) {
    const int cg = (int)(i % cgs);
    const long row = i / cgs;
    const int l = (int)(row % L), bi = (int)(row / L);
    float a[8], p[8];
    if (l == 0) {
#pragma unroll
      for (int j = 0; j < 8; ++j) a[j] = 0.f;
      for (int q = 0; q < HW; ++q) {
        float v[8];
        ld8(x + ((long)bi * HW + q) * C + cg * 8, v);
#pragma unroll
        for (int j = 0; j < 8; ++j) a[j] += v[j];
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) a[j] *= inv;
    } else {
      ld8(x + ((long)bi * HW + l - 1) * C + cg * 8, a);
    }
    ld8(pos + (long)l * C + cg * 8, p);
#pragma unroll
    for (int j = 0; j < 8; ++j) a[j] += p[j];
    st8(tok + i * 8, a);
  }
}

// dx[b, l] = dtok[b, l + 1] + dtok[b, 0] / HW
template <typename T>
__global__ __launch_bounds__(256) void attnpool_tokens_bwd_kernel(const T* __restrict__ dtok, T* __restrict__ dx, int b, int HW, int C) {
  const int cgs = C >> 3, L = HW + 1;
  const long ntask = (long)b * HW * cgs;
  const float inv = 1.f / (float)HW;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < ntask; i += (long)gridDim.x * 256) {
    const int cg = (int)(i % cgs);
    const long row = i / cgs;
    const int l = (int)(row % HW), bi = (int)(row / HW);
    float a[8], m[8];
    ld8(dtok + ((long)bi * L + l + 1) * C + cg * 8, a);
    ld8(dtok + ((long)bi * L) * C + cg * 8, m);
#pragma unroll
    for (int j = 0; j < 8; ++j) a[j] += m[j] * inv;
    st8(dx + i * 8, a);
  }
}

// dpos[l] += sum_b dtok[b, l]
template <typename T>
__global__ __launch_bounds__(256) void attnpool_dpos_kernel(const T* __restrict__ dtok, float* __restrict__ dpos, int b, int HW, int C) {
  const int cgs = C >> 3, L = HW + 1;
  const long ntask = (long)L * cgs;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < ntask; i += (long)gridDim.x * 256) {
    const int cg = (int)(i % cgs);
    const int l = (int)(i / cgs);
    float acc[8];
    ld8(dpos + (long)l * C + cg * 8, acc);
    for (int bi = 0; bi < b; ++bi) {
      float v[8];
      ld8(dtok + ((long)bi * L + l) * C + cg * 8, v);
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[j] += v[j];
    }
    st8(dpos + (long)l * C + cg * 8, acc);
  }
}

extern "C" int dh_attnpool_tokens_fwd(int dtype, const void* x, const float* pos, void* tok, int b, int HW, int C, dh_stream_t stream) {
  hipStream_t st = (hipStream_t)stream;
  DH_REQUIRE(x && pos && tok && b > 0 && HW > 0 && C > 0 && C % 8 == 0, "dh_attnpool_tokens_fwd: bad args");
  const long ntask = (long)b * (HW + 1) * (C / 8);
  if (dtype == DH_BF16) hipLaunchKernelGGL(attnpool_tokens_fwd_kernel<bf16_t>, dim3(grid_for(ntask)), dim3(256), 0, st, (const bf16_t*)x, pos, (bf16_t*)tok, b, HW, C);
  else if (dtype == DH_F32) hipLaunchKernelGGL(attnpool_tokens_fwd_kernel<float>, dim3(grid_for(ntask)), dim3(256), 0, st, (const float*)x, pos, (float*)tok, b, HW, C);
  else DH_FAIL(DH_ERR_ARG, "dh_attnpool_tokens_fwd: bad dtype");
  DH_CHECK_LAUNCH();
  return DH_OK;
}

extern "C" int dh_attnpool_tokens_bwd(int dtype, const void* dtok, void* dx, float* dpos, int b, int HW, int C, dh_stream_t stream) {
  hipStream_t st = (hipStream_t)stream;
  DH_REQUIRE(dtok && dx && b > 0 && HW > 0 && C > 0 && C % 8 == 0, "dh_attnpool_tokens_bwd: bad args");
  DH_REQUIRE(dtype == DH_F32 || dtype == DH_BF16, "dh_attnpool_tokens_bwd: bad dtype");
  const long ntask = (long)b * HW * (C / 8);
  if (dtype == DH_BF16) hipLaunchKernelGGL(attnpool_tokens_bwd_kernel<bf16_t>, dim3(grid_for(ntask)), dim3(256), 0, st, (const bf16_t*)dtok, (bf16_t*)dx, b, HW, C);
  else hipLaunchKernelGGL(attnpool_tokens_bwd_kernel<float>, dim3(grid_for(ntask)), dim3(256), 0, st, (const float*)dtok, (float*)dx, b, HW, C);
  DH_CHECK_LAUNCH();
  if (dpos) {
    const long nt2 = (long)(HW + 1) * (C / 8);
    if (dtype == DH_BF16) hipLaunchKernelGGL(attnpool_dpos_kernel<bf16_t>, dim3(grid_for(nt2)), dim3(256), 0, st, (const bf16_t*)dtok, dpos, b, HW, C);
    else hipLaunchKernelGGL(attnpool_dpos_kernel<float>, dim3(grid_for(nt2)), dim3(256), 0, st, (const float*)dtok, dpos, b, HW, C);
    DH_CHECK_LAUNCH();
  }
  return DH_OK;
}
