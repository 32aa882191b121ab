// Kernels specific to the DeCLIP / SLIP heads:
//   * BatchNorm1d (+ReLU) forward/backward with per-group batch statistics (model/declip.py:33-130: the SimSiam
//     projector / predictor apply plain nn.BatchNorm1d per view, per rank)
//   * negative-cosine (SimSiam D(p, stopgrad z), loss_functions/loss.py:49-55)
//   * nearest-neighbour search in the feature bank (model/utils/nnclr_modules/nn_memory_bank.py:42-65)
//   * row gather by index list (masked-LM rows, model/declip.py:326-334)
#include "dh_common.h"

namespace {

// ---------------------------------------------------------------------------------- BatchNorm1d
// x [G*R, C] (G groups of R rows; statistics per group), y = relu?((x - mean) * invstd * w + b).
// One block (256 threads = 64 columns x 4 row lanes) per (column tile, group).
template <typename T>
__global__ __launch_bounds__(256) void bn1d_fwd_kernel(const T* __restrict__ x, const float* __restrict__ w,
                                                       const float* __restrict__ b, T* __restrict__ y,
                                                       float* __restrict__ save_mean, float* __restrict__ save_invstd,
                                                       float* __restrict__ run_mean, float* __restrict__ run_var, int R,
                                                       int C, float eps, float momentum, int relu, int training) {
  __shared__ float red[2][4][64];
  const int c = blockIdx.x * 64 + (threadIdx.x & 63);
  const int rl = threadIdx.x >> 6;
  const int g = blockIdx.y;
  const T* xg = x + (long)g * R * C;
  T* yg = y + (long)g * R * C;
  float mean, invstd;
  if (training) {
    float s = 0.f, q = 0.f;
    if (c < C)
      for (int r = rl; r < R; r += 4) { float v = ld<T>(xg + (long)r * C + c); s += v; q += v * v; }
    red[0][rl][threadIdx.x & 63] = s;
    red[1][rl][threadIdx.x & 63] = q;
    __syncthreads();
    const int cl = threadIdx.x & 63;
    s = red[0][0][cl] + red[0][1][cl] + red[0][2][cl] + red[0][3][cl];
    q = red[1][0][cl] + red[1][1][cl] + red[1][2][cl] + red[1][3][cl];
    mean = s / R;
    float var = fmaxf(q / R - mean * mean, 0.f);            // biased variance normalises (torch semantics)
    invstd = rsqrtf(var + eps);
    if (rl == 0 && c < C) {
      save_mean[(long)g * C + c] = mean;
      save_invstd[(long)g * C + c] = invstd;
      if (run_mean) {                                        // groups update sequentially: done on the host side order
        const float unbiased = R > 1 ? var * R / (R - 1) : var;
        run_mean[c] = (1.f - momentum) * run_mean[c] + momentum * mean;
        run_var[c] = (1.f - momentum) * run_var[c] + momentum * unbiased;
      }
    }
  } else {
    mean = c < C ? run_mean[c] : 0.f;
    invstd = c < C ? rsqrtf(run_var[c] + eps) : 0.f;
  }
  if (c < C) {
    const float wc = w[c], bc = b[c];
    for (int r = rl; r < R; r += 4) {
      float v = (ld<T>(xg + (long)r * C + c) - mean) * invstd * wc + bc;
      if (relu) v = fmaxf(v, 0.f);
      st<T>(yg + (long)r * C + c, v);
    }
  }
}

// dx = w * invstd * (dyr - mean(dyr) - xhat * mean(dyr * xhat)), dyr = dy * (y > 0) when relu;
// dw += sum(dyr * xhat), db += sum(dyr)
template <typename T>
__global__ __launch_bounds__(256) void bn1d_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ x,
                                                       const T* __restrict__ y, const float* __restrict__ w,
                                                       const float* __restrict__ save_mean,
                                                       const float* __restrict__ save_invstd, T* __restrict__ dx,
                                                       float* __restrict__ dw, float* __restrict__ db, int R, int C, int relu) {
  __shared__ float red[2][4][64];
  const int c = blockIdx.x * 64 + (threadIdx.x & 63);
  const int rl = threadIdx.x >> 6;
  const int g = blockIdx.y;
  const long base = (long)g * R * C;
  const float mean = c < C ? save_mean[(long)g * C + c] : 0.f, invstd = c < C ? save_invstd[(long)g * C + c] : 0.f;
  float s1 = 0.f, s2 = 0.f;
  if (c < C)
    for (int r = rl; r < R; r += 4) {
      float d = ld<T>(dy + base + (long)r * C + c);
      if (relu && ld<T>(y + base + (long)r * C + c) <= 0.f) d = 0.f;
      const float xh = (ld<T>(x + base + (long)r * C + c) - mean) * invstd;
      s1 += d; s2 += d * xh;
    }
  red[0][rl][threadIdx.x & 63] = s1;
  red[1][rl][threadIdx.x & 63] = s2;
  __syncthreads();
  const int cl = threadIdx.x & 63;
  s1 = red[0][0][cl] + red[0][1][cl] + red[0][2][cl] + red[0][3][cl];
  s2 = red[1][0][cl] + red[1][1][cl] + red[1][2][cl] + red[1][3][cl];
  if (c < C) {
    if (rl == 0) { atomicAdd(dw + c, s2); atomicAdd(db + c, s1); }
    const float wc = w[c], m1 = s1 / R, m2 = s2 / R;
    for (int r = rl; r < R; r += 4) {
      float d = ld<T>(dy + base + (long)r * C + c);
      if (relu && ld<T>(y + base + (long)r * C + c) <= 0.f) d = 0.f;
      const float xh = (ld<T>(x + base + (long)r * C + c) - mean) * invstd;
      st<T>(dx + base + (long)r * C + c, wc * invstd * (d - m1 - xh * m2));
    }
  }
}

// The same two kernels for C % 8 == 0 with 16-byte accesses: block = 8 column chunks (8 columns each) x 32 row lanes, so a thread
// walks R / 32 rows with independent 16-byte loads instead of R / 4 rows with 2-byte loads (the DeCLIP projector / predictor,
// R = 512, C = 1024..4096: 171 -> ~20 us for the backward; the 2-byte version was latency-bound at 1-2 % of HBM bandwidth).
template <typename T>
__global__ __launch_bounds__(256) void bn1d_fwd_vec_kernel(const T* __restrict__ x, const float* __restrict__ w,
                                                           const float* __restrict__ b, T* __restrict__ y,
                                                           float* __restrict__ save_mean, float* __restrict__ save_invstd,
                                                           float* __restrict__ run_mean, float* __restrict__ run_var, int R,
                                                           int C, float eps, float momentum, int relu, int training) {
  __shared__ float red[2][32][65];
  __shared__ float stat[2][64];
  const int cl = threadIdx.x & 7, rl = threadIdx.x >> 3;
  const int c0 = blockIdx.x * 64 + cl * 8;
  const int g = blockIdx.y;
  const T* xg = x + (long)g * R * C;
  T* yg = y + (long)g * R * C;
  const bool live = c0 < C;                  // C % 8 == 0: a chunk is inside or outside as a whole
  if (training) {
    float s[8], q[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) { s[k] = 0.f; q[k] = 0.f; }
    if (live)
      for (int r = rl; r < R; r += 32) {
        float v[8];
        ld8(xg + (long)r * C + c0, v);
#pragma unroll
        for (int k = 0; k < 8; ++k) { s[k] += v[k]; q[k] += v[k] * v[k]; }
      }
#pragma unroll
    for (int k = 0; k < 8; ++k) { red[0][rl][cl * 8 + k] = s[k]; red[1][rl][cl * 8 + k] = q[k]; }
    __syncthreads();
    if (threadIdx.x < 64) {
      const int c = blockIdx.x * 64 + threadIdx.x;
      float ss = 0.f, qq = 0.f;
#pragma unroll 8
      for (int l = 0; l < 32; ++l) { ss += red[0][l][threadIdx.x]; qq += red[1][l][threadIdx.x]; }
      const float mean = ss / R;
      const float var = fmaxf(qq / R - mean * mean, 0.f);
      const float invstd = rsqrtf(var + eps);
      stat[0][threadIdx.x] = mean; stat[1][threadIdx.x] = invstd;
      if (c < C) {
        save_mean[(long)g * C + c] = mean;
        save_invstd[(long)g * C + c] = invstd;
        if (run_mean) {
          const float unbiased = R > 1 ? var * R / (R - 1) : var;
          run_mean[c] = (1.f - momentum) * run_mean[c] + momentum * mean;
          run_var[c] = (1.f - momentum) * run_var[c] + momentum * unbiased;
        }
      }
    }
  } else if (threadIdx.x < 64) {
    const int c = blockIdx.x * 64 + threadIdx.x;
    stat[0][threadIdx.x] = c < C ? run_mean[c] : 0.f;
    stat[1][threadIdx.x] = c < C ? rsqrtf(run_var[c] + eps) : 0.f;
  }
  __syncthreads();
  if (live) {
    float sc[8], sh[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const float a = stat[1][cl * 8 + k] * w[c0 + k];
      sc[k] = a; sh[k] = b[c0 + k] - stat[0][cl * 8 + k] * a;
    }
    for (int r = rl; r < R; r += 32) {
      float v[8];
      ld8(xg + (long)r * C + c0, v);
#pragma unroll
      for (int k = 0; k < 8; ++k) { v[k] = v[k] * sc[k] + sh[k]; if (relu) v[k] = fmaxf(v[k], 0.f); }
      st8(yg + (long)r * C + c0, v);
    }
  }
}
template <typename T>
__global__ __launch_bounds__(256) void bn1d_bwd_vec_kernel(const T* __restrict__ dy, const T* __restrict__ x,
                                                           const T* __restrict__ y, const float* __restrict__ w,
                                                           const float* __restrict__ save_mean,
                                                           const float* __restrict__ save_invstd, T* __restrict__ dx,
                                                           float* __restrict__ dw, float* __restrict__ db, int R, int C, int relu) {
  __shared__ float red[2][32][65];
  __shared__ float stat[2][64];
  const int cl = threadIdx.x & 7, rl = threadIdx.x >> 3;
  const int c0 = blockIdx.x * 64 + cl * 8;
  const int g = blockIdx.y;
  const long base = (long)g * R * C;
  const bool live = c0 < C;
  float mean[8], invstd[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) { mean[k] = live ? save_mean[(long)g * C + c0 + k] : 0.f; invstd[k] = live ? save_invstd[(long)g * C + c0 + k] : 0.f; }
  float s1[8], s2[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) { s1[k] = 0.f; s2[k] = 0.f; }
  if (live)
    for (int r = rl; r < R; r += 32) {
      float d[8], xv[8], yv[8];
      ld8(dy + base + (long)r * C + c0, d);
      ld8(x + base + (long)r * C + c0, xv);
      if (relu) ld8(y + base + (long)r * C + c0, yv);
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const float dd = (relu && yv[k] <= 0.f) ? 0.f : d[k];
        s1[k] += dd; s2[k] += dd * (xv[k] - mean[k]) * invstd[k];
      }
    }
#pragma unroll
  for (int k = 0; k < 8; ++k) { red[0][rl][cl * 8 + k] = s1[k]; red[1][rl][cl * 8 + k] = s2[k]; }
  __syncthreads();
  if (threadIdx.x < 64) {
    const int c = blockIdx.x * 64 + threadIdx.x;
    float a = 0.f, bsum = 0.f;
#pragma unroll 8
    for (int l = 0; l < 32; ++l) { a += red[0][l][threadIdx.x]; bsum += red[1][l][threadIdx.x]; }
    stat[0][threadIdx.x] = a; stat[1][threadIdx.x] = bsum;
    if (c < C) { atomicAdd(dw + c, bsum); atomicAdd(db + c, a); }
  }
  __syncthreads();
  if (live) {
    float m1[8], m2[8], wi[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) { m1[k] = stat[0][cl * 8 + k] / R; m2[k] = stat[1][cl * 8 + k] / R; wi[k] = w[c0 + k] * invstd[k]; }
    for (int r = rl; r < R; r += 32) {
      float d[8], xv[8], yv[8], o[8];
      ld8(dy + base + (long)r * C + c0, d);
      ld8(x + base + (long)r * C + c0, xv);
      if (relu) ld8(y + base + (long)r * C + c0, yv);
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const float dd = (relu && yv[k] <= 0.f) ? 0.f : d[k];
        o[k] = wi[k] * (dd - m1[k] - (xv[k] - mean[k]) * invstd[k] * m2[k]);
      }
      st8(dx + base + (long)r * C + c0, o);
    }
  }
}

// ---------------------------------------------------------------------------------- negative cosine
// cos[r] = <p_r, z_r> / (|p_r| |z_r|); one wave per row.  bwd (w.r.t. p only, z is stop-grad):
// dp = g_r * ( z/(|p||z|) - cos * p / |p|^2 )
template <typename T>
__global__ __launch_bounds__(256) void cos_rows_fwd_kernel(const T* __restrict__ p, const T* __restrict__ z,
                                                           float* __restrict__ cosv, int rows, int d) {
  const int lane = threadIdx.x & 63;
  for (int r = blockIdx.x * 4 + (threadIdx.x >> 6); r < rows; r += gridDim.x * 4) {
    float pz = 0.f, pp = 0.f, zz = 0.f;
    for (int i = lane; i < d; i += 64) {
      const float a = ld<T>(p + (long)r * d + i), b = ld<T>(z + (long)r * d + i);
      pz += a * b; pp += a * a; zz += b * b;
    }
    pz = wave_sum(pz); pp = wave_sum(pp); zz = wave_sum(zz);
    if (lane == 0) cosv[r] = pz / (sqrtf(pp) * sqrtf(zz));
  }
}
template <typename T>
__global__ __launch_bounds__(256) void cos_rows_bwd_kernel(const T* __restrict__ p, const T* __restrict__ z,
                                                           const float* __restrict__ g_row, T* __restrict__ dp, int rows, int d) {
  const int lane = threadIdx.x & 63;
  for (int r = blockIdx.x * 4 + (threadIdx.x >> 6); r < rows; r += gridDim.x * 4) {
    float pz = 0.f, pp = 0.f, zz = 0.f;
    for (int i = lane; i < d; i += 64) {
      const float a = ld<T>(p + (long)r * d + i), b = ld<T>(z + (long)r * d + i);
      pz += a * b; pp += a * a; zz += b * b;
    }
    pz = wave_sum(pz); pp = wave_sum(pp); zz = wave_sum(zz);
    const float np = sqrtf(pp), nz = sqrtf(zz), cosv = pz / (np * nz), g = g_row[r];
    for (int i = lane; i < d; i += 64) {
      const float a = ld<T>(p + (long)r * d + i), b = ld<T>(z + (long)r * d + i);
      st<T>(dp + (long)r * d + i, g * (b / (np * nz) - cosv * a / pp));
    }
  }
}

// ---------------------------------------------------------------------------------- NN bank search
// idx[r] = argmax_j <q_r, bank_j> over the bank rows [size][D] (fp32, rows are unit vectors);
// exact fp32 (the neighbour choice must match the reference's torch.topk).  Two kernels: per
// (row tile, bank chunk) partial (best value, best index), then a merge + gather.
constexpr int NN_RT = 32, NN_CT = 64, NN_KC = 32;
__global__ __launch_bounds__(256) void nn_search_kernel(const float* __restrict__ Q, const float* __restrict__ bank, int rows,
                                                        int size, int D, int chunk, float* __restrict__ pval, int* __restrict__ pidx) {
  DH_DYN_LDS(float, sm);
  float* Xs = sm;                          // [32][D+1]
  float* Ys = Xs + NN_RT * (D + 1);        // [64][33]
  const int t = threadIdx.x, tx = t & 15, ty = t >> 4;
  const int r0 = blockIdx.x * NN_RT;
  const int cbeg = blockIdx.y * chunk, cend = min(size, cbeg + chunk);
  for (int i = t; i < NN_RT * D; i += 256) {
    int r = i / D, k = i % D;
    Xs[r * (D + 1) + k] = (r0 + r < rows) ? Q[(long)(r0 + r) * D + k] : 0.f;
  }
  float best[2] = {-INFINITY, -INFINITY};
  int bidx[2] = {0, 0};
  for (int c0 = cbeg; c0 < cend; c0 += NN_CT) {
    float acc[2][4];
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
      for (int c = 0; c < 4; ++c) acc[r][c] = 0.f;
    for (int k0 = 0; k0 < D; k0 += NN_KC) {
      __syncthreads();
#pragma unroll
      for (int p = 0; p < 8; ++p) {
        int idx = t + p * 256;
        int kk = idx & 31, yy = idx >> 5;
        int y = c0 + yy, k = k0 + kk;
        Ys[yy * (NN_KC + 1) + kk] = (y < cend && k < D) ? bank[(long)y * D + k] : 0.f;
      }
      __syncthreads();
      const int kmax = min(NN_KC, D - k0);
      for (int kk = 0; kk < kmax; ++kk) {
        float xv[2], yv[4];
#pragma unroll
        for (int r = 0; r < 2; ++r) xv[r] = Xs[(ty * 2 + r) * (D + 1) + k0 + kk];
#pragma unroll
        for (int c = 0; c < 4; ++c) yv[c] = Ys[(tx * 4 + c) * (NN_KC + 1) + kk];
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
          for (int c = 0; c < 4; ++c) acc[r][c] = fmaf(xv[r], yv[c], acc[r][c]);
      }
    }
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const int col = c0 + tx * 4 + c;
        if (col < cend && acc[r][c] > best[r]) { best[r] = acc[r][c]; bidx[r] = col; }   // first maximum wins (ascending col)
      }
  }
#pragma unroll
  for (int r = 0; r < 2; ++r) {
#pragma unroll
    for (int o = 1; o < 16; o <<= 1) {
      const float v2 = __shfl_xor(best[r], o, 64);
      const int i2 = __shfl_xor(bidx[r], o, 64);
      if (v2 > best[r] || (v2 == best[r] && i2 < bidx[r])) { best[r] = v2; bidx[r] = i2; }
    }
    const int row = r0 + ty * 2 + r;
    if (tx == 0 && row < rows) { pval[(long)blockIdx.y * rows + row] = best[r]; pidx[(long)blockIdx.y * rows + row] = bidx[r]; }
  }
}
// The same search on the matrix pipe: v_mfma_f32_32x32x2_f32 is an exact fp32 FMA chain at the fp32 vector peak (157 TF),
// ~12x what the VALU tile kernel above reaches.  Block = 4 waves: 64 query rows (whole rows in LDS) x one chunk of the bank,
// streamed as tiles of 128 bank rows x 32 k (register-prefetched, one LDS stage).  Wave w multiplies bank rows w*32.. of
// the tile with both 32-row query blocks; operands are issued swapped (A = bank, B = queries) so a LANE owns one query row
// and its 16 accumulator registers are 16 bank rows: the running arg-max is in-lane, ascending bank index, strict '>'
// (first maximum wins, as torch.topk's tie order on ascending indices).  k pairing: lane (r, h) feeds k = 8g + 4h + j to
// MFMA j of group g for BOTH operands, so one 16-byte LDS read per operand feeds 4 MFMAs.
constexpr int NM_QT = 64, NM_BT = 128, NM_KC = 32;
__global__ __launch_bounds__(256) void nn_search_mfma_kernel(const float* __restrict__ Q, const float* __restrict__ bank, int rows,
                                                             int size, int D, int chunk, float* __restrict__ pval, int* __restrict__ pidx) {
  DH_DYN_LDS_A16(float, sm);
  const int QS = D + 4, BS = NM_KC + 4;                      // row strides: = 4 (mod 32) words -> conflict-free 16-byte reads
  float* Xs = sm;                                            // [64][QS]
  float* Ys = Xs + NM_QT * QS;                               // [128][BS]
  float* redv = Ys + NM_BT * BS;                             // [4 waves][64]
  int* redi = reinterpret_cast<int*>(redv + 4 * NM_QT);
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int r0 = blockIdx.x * NM_QT;
  const int cbeg = blockIdx.y * chunk, cend = min(size, cbeg + chunk);
  for (int i = t; i < NM_QT * (D / 4); i += 256) {
    const int rr = i / (D / 4), k4 = i % (D / 4);
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (r0 + rr < rows) v = *reinterpret_cast<const float4*>(Q + (long)(r0 + rr) * D + k4 * 4);
    *reinterpret_cast<float4*>(Xs + rr * QS + k4 * 4) = v;
  }
  float best[2] = {-INFINITY, -INFINITY};
  int bidx[2] = {0x7fffffff, 0x7fffffff};
  // stage loader: thread -> bank row t >> 1 of the tile, 16 consecutive k (t & 1) * 16 .. +15  (4 float4)
  const int srow = t >> 1, sk = (t & 1) * 16;
  float4 pre[4];
  auto fetch = [&](int c0, int k0) {
    const int y = c0 + srow;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      pre[q] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (y < cend) pre[q] = *reinterpret_cast<const float4*>(bank + (long)y * D + k0 + sk + 4 * q);
    }
  };
  const int nk = D / NM_KC;
  fetch(cbeg, 0);
  for (int c0 = cbeg; c0 < cend; c0 += NM_BT) {
    f32x16_t acc[2];
#pragma unroll
    for (int qb = 0; qb < 2; ++qb)
#pragma unroll
      for (int x = 0; x < 16; ++x) acc[qb][x] = 0.f;
    for (int kc = 0; kc < nk; ++kc) {
      __syncthreads();                                       // the previous stage has been consumed
#pragma unroll
      for (int q = 0; q < 4; ++q) *reinterpret_cast<float4*>(Ys + srow * BS + sk + 4 * q) = pre[q];
      __syncthreads();
      // next stage in flight during this stage's MFMAs
      if (kc + 1 < nk) fetch(c0, (kc + 1) * NM_KC);
      else if (c0 + NM_BT < cend) fetch(c0 + NM_BT, 0);
#pragma unroll
      for (int g = 0; g < NM_KC / 8; ++g) {
        const float4 a = *reinterpret_cast<const float4*>(Ys + (wave * 32 + r) * BS + 8 * g + 4 * h);
        const float av[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
        for (int qb = 0; qb < 2; ++qb) {
          const float4 b = *reinterpret_cast<const float4*>(Xs + (qb * 32 + r) * QS + kc * NM_KC + 8 * g + 4 * h);
          const float bv[4] = {b.x, b.y, b.z, b.w};
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[qb] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[j], bv[j], acc[qb], 0, 0, 0);
        }
      }
    }
    // acc[qb][x] of lane (r, h): query row qb*32 + r, bank row c0 + wave*32 + (x/4)*8 + h*4 + x%4  (ascending in x)
#pragma unroll
    for (int qb = 0; qb < 2; ++qb)
#pragma unroll
      for (int x = 0; x < 16; ++x) {
        const int col = c0 + wave * 32 + (x >> 2) * 8 + h * 4 + (x & 3);
        if (col < cend && acc[qb][x] > best[qb]) { best[qb] = acc[qb][x]; bidx[qb] = col; }
      }
  }
  // lanes r and r + 32 hold the same query row; then the four waves (different bank rows) through LDS
#pragma unroll
  for (int qb = 0; qb < 2; ++qb) {
    const float v2 = __shfl_xor(best[qb], 32, 64);
    const int i2 = __shfl_xor(bidx[qb], 32, 64);
    if (v2 > best[qb] || (v2 == best[qb] && i2 < bidx[qb])) { best[qb] = v2; bidx[qb] = i2; }
    if (h == 0) { redv[wave * NM_QT + qb * 32 + r] = best[qb]; redi[wave * NM_QT + qb * 32 + r] = bidx[qb]; }
  }
  __syncthreads();
  if (t < NM_QT && r0 + t < rows) {
    float bv = redv[t]; int bi = redi[t];
#pragma unroll
    for (int w = 1; w < 4; ++w) {
      const float v = redv[w * NM_QT + t]; const int i = redi[w * NM_QT + t];
      if (v > bv || (v == bv && i < bi)) { bv = v; bi = i; }
    }
    pval[(long)blockIdx.y * rows + r0 + t] = bv;
    pidx[(long)blockIdx.y * rows + r0 + t] = bi;
  }
}

__global__ void nn_merge_gather_kernel(const float* __restrict__ pval, const int* __restrict__ pidx, int nchunk, int rows,
                                       const float* __restrict__ bank, int D, int64_t* __restrict__ idx_out, float* __restrict__ feat_out) {
  const int row = blockIdx.x;
  __shared__ int s_idx;
  if (threadIdx.x == 0) {
    float bv = -INFINITY; int bi = 0;
    for (int c = 0; c < nchunk; ++c) {
      const float v = pval[(long)c * rows + row];
      const int i = pidx[(long)c * rows + row];
      if (v > bv || (v == bv && i < bi)) { bv = v; bi = i; }
    }
    s_idx = bi;
    idx_out[row] = bi;
  }
  __syncthreads();
  const int bi = s_idx;
  for (int k = threadIdx.x; k < D; k += blockDim.x) feat_out[(long)row * D + k] = bank[(long)bi * D + k];
}

// ---------------------------------------------------------------------------------- row gather / scatter
template <typename T>
__global__ __launch_bounds__(256) void gather_rows_kernel(const T* __restrict__ x, const int64_t* __restrict__ idx,
                                                          T* __restrict__ out, int n, int n_pad, int d) {
  const int nchunk = d >> 3;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < (long)n_pad * nchunk; i += (long)gridDim.x * 256) {
    const int r = (int)(i / nchunk), ch = (int)(i % nchunk);
    float a[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (r < n) ld8(x + idx[r] * (long)d + ch * 8, a);
    st8(out + (long)r * d + ch * 8, a);
  }
}
// dx[idx[r], :] += dout[r, :]  (indices are unique: masked positions) ; dx pre-zeroed by the caller
template <typename T>
__global__ __launch_bounds__(256) void scatter_rows_kernel(const T* __restrict__ dout, const int64_t* __restrict__ idx,
                                                           T* __restrict__ dx, int n, int d) {
  const int nchunk = d >> 3;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < (long)n * nchunk; i += (long)gridDim.x * 256) {
    const int r = (int)(i / nchunk), ch = (int)(i % nchunk);
    float a[8], b[8];
    ld8(dout + (long)r * d + ch * 8, a);
    ld8(dx + idx[r] * (long)d + ch * 8, b);
#pragma unroll
    for (int k = 0; k < 8; ++k) a[k] += b[k];
    st8(dx + idx[r] * (long)d + ch * 8, a);
  }
}

int grid_for(long work_items) {
  long g = (work_items + 255) / 256;
  if (g > 4096) g = 4096;
  if (g < 1) g = 1;
  return (int)g;
}

}  // namespace

extern "C" int dh_bn1d_fwd(int dtype, const void* x, const float* w, const float* b, void* y, float* save_mean,
                           float* save_invstd, float* running_mean, float* running_var, int groups, int rows_per_group, int C,
                           float eps, float momentum, int relu, int training, dh_stream_t stream) {
  hipStream_t st = (hipStream_t)stream;
  DH_REQUIRE(x && w && b && y && groups >= 1 && rows_per_group >= 1 && C >= 1, "dh_bn1d_fwd: bad args");
  DH_REQUIRE(!training || (save_mean && save_invstd), "dh_bn1d_fwd: training needs save buffers");
  DH_REQUIRE(training || (running_mean && running_var), "dh_bn1d_fwd: eval needs running stats");
  // groups update the running statistics one after the other (reference applies the module once per view)
  for (int g = 0; g < (training && running_mean ? groups : 1); ++g) {
    const int ng = (training && running_mean) ? 1 : groups;
    const long off = (long)g * rows_per_group * C;
    dim3 grid(dh_cdiv(C, 64), ng);
    const bool vec = C % 8 == 0 && ((uintptr_t)x & 15) == 0 && ((uintptr_t)y & 15) == 0;
    if (vec && dtype == DH_BF16)
      hipLaunchKernelGGL(bn1d_fwd_vec_kernel<bf16_t>, grid, dim3(256), 0, st, (const bf16_t*)x + off, w, b, (bf16_t*)y + off,
                         save_mean ? save_mean + (long)g * C : nullptr, save_invstd ? save_invstd + (long)g * C : nullptr,
                         running_mean, running_var, rows_per_group, C, eps, momentum, relu, training);
    else if (vec && dtype == DH_F32)
      hipLaunchKernelGGL(bn1d_fwd_vec_kernel<float>, grid, dim3(256), 0, st, (const float*)x + off, w, b, (float*)y + off,
                         save_mean ? save_mean + (long)g * C : nullptr, save_invstd ? save_invstd + (long)g * C : nullptr,
                         running_mean, running_var, rows_per_group, C, eps, momentum, relu, training);
    else if (dtype == DH_BF16)
      hipLaunchKernelGGL(bn1d_fwd_kernel<bf16_t>, grid, dim3(256), 0, st, (const bf16_t*)x + off, w, b, (bf16_t*)y + off,
                         save_mean ? save_mean + (long)g * C : nullptr, save_invstd ? save_invstd + (long)g * C : nullptr,
                         running_mean, running_var, rows_per_group, C, eps, momentum, relu, training);
    else
      hipLaunchKernelGGL(bn1d_fwd_kernel<float>, grid, dim3(256), 0, st, (const float*)x + off, w, b, (float*)y + off,
                         save_mean ? save_mean + (long)g * C : nullptr, save_invstd ? save_invstd + (long)g * C : nullptr,
                         running_mean, running_var, rows_per_group, C, eps, momentum, relu, training);
    DH_CHECK_LAUNCH();
  }
  return DH_OK;
}

extern "C" int dh_bn1d_bwd(int dtype, const void* dy, const void* x, const void* y, const float* w, const float* save_mean,
                           const float* save_invstd, void* dx, float* dw, float* db, int groups, int rows_per_group, int C,
                           int relu, dh_stream_t stream) {
  hipStream_t st = (hipStream_t)stream;
  DH_REQUIRE(dy && x && w && save_mean && save_invstd && dx && dw && db && (!relu || y), "dh_bn1d_bwd: bad args");
  dim3 grid(dh_cdiv(C, 64), groups);
  const bool vec = C % 8 == 0 && ((uintptr_t)dy & 15) == 0 && ((uintptr_t)x & 15) == 0 && ((uintptr_t)dx & 15) == 0 && (!relu || ((uintptr_t)y & 15) == 0);
  if (vec && dtype == DH_BF16)
    hipLaunchKernelGGL(bn1d_bwd_vec_kernel<bf16_t>, grid, dim3(256), 0, st, (const bf16_t*)dy, (const bf16_t*)x, (const bf16_t*)y, w,
                       save_mean, save_invstd, (bf16_t*)dx, dw, db, rows_per_group, C, relu);
  else if (vec && dtype == DH_F32)
    hipLaunchKernelGGL(bn1d_bwd_vec_kernel<float>, grid, dim3(256), 0, st, (const float*)dy, (const float*)x, (const float*)y, w,
                       save_mean, save_invstd, (float*)dx, dw, db, rows_per_group, C, relu);
  else if (dtype == DH_BF16)
    hipLaunchKernelGGL(bn1d_bwd_kernel<bf16_t>, grid, dim3(256), 0, st, (const bf16_t*)dy, (const bf16_t*)x, (const bf16_t*)y, w,
                       save_mean, save_invstd, (bf16_t*)dx, dw, db, rows_per_group, C, relu);
  else
    hipLaunchKernelGGL(bn1d_bwd_kernel<float>, grid, dim3(256), 0, st, (const float*)dy, (const float*)x, (const float*)y, w,
                       save_mean, save_invstd, (float*)dx, dw, db, rows_per_group, C, relu);
  DH_CHECK_LAUNCH();
  return DH_OK;
}

extern "C" int dh_cos_rows_fwd(int dtype, const void* p, const void* z, float* cosv, int rows, int d, dh_stream_t stream) {
  hipStream_t st = (hipStream_t)stream;
  DH_REQUIRE(p && z && cosv && rows > 0 && d > 0, "dh_cos_rows_fwd: bad args");
  dim3 grid(dh_cdiv(rows, 4) > 1024 ? 1024 : dh_cdiv(rows, 4));
  if (dtype == DH_BF16) hipLaunchKernelGGL(cos_rows_fwd_kernel<bf16_t>, grid, dim3(256), 0, st, (const bf16_t*)p, (const bf16_t*)z, cosv, rows, d);
  else hipLaunchKernelGGL(cos_rows_fwd_kernel<float>, grid, dim3(256), 0, st, (const float*)p, (const float*)z, cosv, rows, d);
  DH_CHECK_LAUNCH();
  return DH_OK;
}
extern "C" int dh_cos_rows_bwd(int dtype, const void* p, const void* z, const float* g_row, void* dp, int rows, int d,
                               dh_stream_t stream) {
  hipStream_t st = (hipStream_t)stream;
  DH_REQUIRE(p && z && g_row && dp && rows > 0 && d > 0, "dh_cos_rows_bwd: bad args");
  dim3 grid(dh_cdiv(rows, 4) > 1024 ? 1024 : dh_cdiv(rows, 4));
  if (dtype == DH_BF16) hipLaunchKernelGGL(cos_rows_bwd_kernel<bf16_t>, grid, dim3(256), 0, st, (const bf16_t*)p, (const bf16_t*)z, g_row, (bf16_t*)dp, rows, d);
  else hipLaunchKernelGGL(cos_rows_bwd_kernel<float>, grid, dim3(256), 0, st, (const float*)p, (const float*)z, g_row, (float*)dp, rows, d);
  DH_CHECK_LAUNCH();
  return DH_OK;
}

extern "C" int64_t dh_nn_bank_ws_bytes(int rows, int size) {
  const int chunk = 2048;
  return (int64_t)dh_cdiv(size, chunk) * rows * 8;
}
extern "C" int dh_nn_bank_query(const float* q, const float* bank, int rows, int size, int D, int64_t* idx_out,
                                float* feat_out, void* ws, int64_t ws_bytes, dh_stream_t stream) {
  hipStream_t st = (hipStream_t)stream;
  DH_REQUIRE(q && bank && idx_out && feat_out && rows > 0 && size > 0 && D > 0 && D <= 1024, "dh_nn_bank_query: bad args");
  const int chunk = 2048;
  const int nchunk = dh_cdiv(size, chunk);
  DH_REQUIRE(ws && ws_bytes >= (int64_t)nchunk * rows * 8, "dh_nn_bank_query: workspace too small");
  float* pval = (float*)ws;
  int* pidx = (int*)(pval + (long)nchunk * rows);
  const size_t lds_m = (size_t)(NM_QT * (D + 4) + NM_BT * (NM_KC + 4) + 8 * NM_QT) * sizeof(float);
  if (D % NM_KC == 0 && lds_m <= 160 * 1024) {                 // matrix-pipe search (D <= 512: whole query rows fit LDS)
    hipFuncSetAttribute((const void*)nn_search_mfma_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_m);
    hipLaunchKernelGGL(nn_search_mfma_kernel, dim3(dh_cdiv(rows, NM_QT), nchunk), dim3(256), lds_m, st, q, bank, rows, size, D, chunk, pval, pidx);
  } else {
    size_t lds = (size_t)(NN_RT * (D + 1) + NN_CT * (NN_KC + 1)) * sizeof(float);
    hipFuncSetAttribute((const void*)nn_search_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(nn_search_kernel, dim3(dh_cdiv(rows, NN_RT), nchunk), dim3(256), lds, st, q, bank, rows, size, D, chunk, pval, pidx);
  }
  DH_CHECK_LAUNCH();
  hipLaunchKernelGGL(nn_merge_gather_kernel, dim3(rows), dim3(128), 0, st, (const float*)pval, (const int*)pidx, nchunk, rows, bank, D, idx_out, feat_out);
  DH_CHECK_LAUNCH();
  return DH_OK;
}

extern "C" int dh_gather_rows(int dtype, const void* x, const int64_t* idx, void* out, int n, int n_pad, int d, dh_stream_t stream) {
  hipStream_t st = (hipStream_t)stream;
  DH_REQUIRE(x && idx && out && n >= 0 && n_pad >= n && n_pad > 0 && d % 8 == 0, "dh_gather_rows: bad args");
  if (dtype == DH_BF16) hipLaunchKernelGGL(gather_rows_kernel<bf16_t>, dim3(grid_for((long)n_pad * d / 8)), dim3(256), 0, st, (const bf16_t*)x, idx, (bf16_t*)out, n, n_pad, d);
  else hipLaunchKernelGGL(gather_rows_kernel<float>, dim3(grid_for((long)n_pad * d / 8)), dim3(256), 0, st, (const float*)x, idx, (float*)out, n, n_pad, d);
  DH_CHECK_LAUNCH();
  return DH_OK;
}
extern "C" int dh_scatter_rows_add(int dtype, const void* dout, const int64_t* idx, void* dx, int n, int d, dh_stream_t stream) {
  hipStream_t st = (hipStream_t)stream;
  DH_REQUIRE(dout && idx && dx && n >= 0 && d % 8 == 0, "dh_scatter_rows_add: bad args");
  if (n == 0) return DH_OK;
  if (dtype == DH_BF16) hipLaunchKernelGGL(scatter_rows_kernel<bf16_t>, dim3(grid_for((long)n * d / 8)), dim3(256), 0, st, (const bf16_t*)dout, idx, (bf16_t*)dx, n, d);
  else hipLaunchKernelGGL(scatter_rows_kernel<float>, dim3(grid_for((long)n * d / 8)), dim3(256), 0, st, (const float*)dout, idx, (float*)dx, n, d);
  DH_CHECK_LAUNCH();
  return DH_OK;
}

// ---- NN memory bank: FIFO enqueue (nnclr_modules/memory_bank.py:71-87) with the write pointer in DEVICE memory ----------------
// rows ptr .. ptr + b - 1 of `store` ([size + spill][D]: the bank followed by a spill region that is never searched) receive the
// batch -- the reference drops the part of a batch that would run over the end and resets the pointer; here that tail lands in the
// spill rows -- then ptr = 0 if ptr + b >= size else ptr + b.  No host value in either launch: a captured step advances the queue on
// every replay.  Two launches (copy, pointer update) instead of five torch ops.
__global__ __launch_bounds__(256) void nn_enqueue_kernel(float* __restrict__ store, const long long* __restrict__ ptr,
                                                         const float* __restrict__ batch, int b, int D4) {
  const long long p0 = *ptr;
  const float4* src = reinterpret_cast<const float4*>(batch);
  float4* dst = reinterpret_cast<float4*>(store) + p0 * D4;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < (long)b * D4; i += (long)gridDim.x * 256) dst[i] = src[i];
}
__global__ void nn_ptr_advance_kernel(long long* ptr, int b, int size) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    const long long n = *ptr + b;
    *ptr = n >= size ? 0 : n;
  }
}
extern "C" int dh_nn_bank_enqueue(float* store, int64_t* ptr, const float* batch, int b, int size, int spill, int D, dh_stream_t stream) {
  hipStream_t st = (hipStream_t)stream;
  DH_REQUIRE(store && ptr && batch && b > 0 && size > 0 && D > 0 && D % 4 == 0, "dh_nn_bank_enqueue: bad args");
  DH_REQUIRE(spill >= b, "dh_nn_bank_enqueue: spill region of %d rows for a batch of %d", spill, b);
  DH_REQUIRE((((uintptr_t)store | (uintptr_t)batch) & 15) == 0, "dh_nn_bank_enqueue: store / batch must be 16-byte aligned");
  int blocks = dh_cdiv((long)b * (D / 4), 256);
  if (blocks > 1024) blocks = 1024;
  hipLaunchKernelGGL(nn_enqueue_kernel, dim3(blocks), dim3(256), 0, st, store, (const long long*)ptr, batch, b, D / 4);
  hipLaunchKernelGGL(nn_ptr_advance_kernel, dim3(1), dim3(64), 0, st, (long long*)ptr, b, size);
  DH_CHECK_LAUNCH();
  return DH_OK;
}
