// LayerNorm forward/backward (base_transformer.py:10-18; nn.LayerNorm eps=1e-5).
// HBM-bound: one wave per row, 16-byte vector loads, fp32 statistics in registers.
// Backward fuses the residual-branch gradient add (dx = LN'(dy) + dres) and produces
// dgamma/dbeta through per-block partials + a second tiny reduce kernel (no atomics storm).
#include "dh_common.h"
#include <string.h>
#include <stdlib.h>

namespace {

constexpr int LN_MAXC = 4;  // up to 4 chunks of 8 per lane -> d <= 2048 on the vector path

// 8 consecutive elements as they come from memory (bf16: one 16-byte register quad) -- rows are PREFETCHED in this form
template <typename T> struct Raw8;
template <> struct Raw8<bf16_t> {
  uint4 v;
  __device__ __forceinline__ void load(const bf16_t* p) { v = *reinterpret_cast<const uint4*>(p); }
  __device__ __forceinline__ void unpack(float* o) const {
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) { o[2 * i] = __uint_as_float(w[i] << 16); o[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u); }
  }
};
template <> struct Raw8<float> {
  float4 a, b;
  __device__ __forceinline__ void load(const float* p) { a = *reinterpret_cast<const float4*>(p); b = *reinterpret_cast<const float4*>(p + 4); }
  __device__ __forceinline__ void unpack(float* o) const { o[0] = a.x; o[1] = a.y; o[2] = a.z; o[3] = a.w; o[4] = b.x; o[5] = b.y; o[6] = b.z; o[7] = b.w; }
};

// Row PAIRS: when d is not a multiple of 512 elements but 2 d is (d = 768: the ViT-B width; d = 256), two consecutive rows are one
// contiguous run of NC * 64 16-byte chunks -- every load / store of the wave is a full 1 KB request and no lane idles (the per-row
// layout above leaves half of the wave masked in its last chunk: 4 requests and 32 elements of arithmetic per lane and pair instead
// of 3 and 24).  Lane l holds chunks g = l + 64 k of the pair; chunk g belongs to row g / (d/8), column chunk g % (d/8).
template <typename T, int NC>
__global__ __launch_bounds__(256) void ln_fwd_pair_kernel(const T* __restrict__ x, const float* __restrict__ w,
                                                          const float* __restrict__ b, T* __restrict__ y,
                                                          float* __restrict__ mean, float* __restrict__ rstd, int rows,
                                                          int d, float eps) {
  const int lane = threadIdx.x & 63;
  const int wave_global = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int nwaves = gridDim.x * 4;
  const int nchunk = d >> 3;                    // chunks per row; 2 * nchunk == 64 * NC
  float wv[NC][8], bv[NC][8];
  bool second[NC];                              // chunk k of this lane lies in the pair's second row
#pragma unroll
  for (int k = 0; k < NC; ++k) {
    const int g = lane + 64 * k;
    second[k] = g >= nchunk;
    const int col = g - (second[k] ? nchunk : 0);
    ld8(w + col * 8, wv[k]); ld8(b + col * 8, bv[k]);
  }
  Raw8<T> nxt[NC];
  auto fetch = [&](int row0) {
    const bool both = row0 + 1 < rows;          // an odd row count: the last pair has one row (its second-row chunks re-read the first row)
#pragma unroll
    for (int k = 0; k < NC; ++k) nxt[k].load(x + (long)row0 * d + (long)(lane + 64 * k) * 8 - ((second[k] && !both) ? d : 0));
  };
  int row0 = wave_global * 2;
  if (row0 < rows) fetch(row0);
  for (; row0 < rows; row0 += nwaves * 2) {
    float v[NC][8];
    float s0 = 0.f, s1 = 0.f;
#pragma unroll
    for (int k = 0; k < NC; ++k) {
      nxt[k].unpack(v[k]);
      float t = 0.f;
#pragma unroll
      for (int i = 0; i < 8; ++i) t += v[k][i];
      if (second[k]) s1 += t; else s0 += t;
    }
    if (row0 + nwaves * 2 < rows) fetch(row0 + nwaves * 2);
    const float mu0 = wave_sum_fast(s0) / d, mu1 = wave_sum_fast(s1) / d;
    float q0 = 0.f, q1 = 0.f;
#pragma unroll
    for (int k = 0; k < NC; ++k) {
      const float mu = second[k] ? mu1 : mu0;
      float t = 0.f;
#pragma unroll
      for (int i = 0; i < 8; ++i) { const float u = v[k][i] - mu; t += u * u; }
      if (second[k]) q1 += t; else q0 += t;
    }
    const float rs0 = rsqrtf(wave_sum_fast(q0) / d + eps), rs1 = rsqrtf(wave_sum_fast(q1) / d + eps);
    const bool both = row0 + 1 < rows;
    if (lane == 0) {
      if (mean) { mean[row0] = mu0; if (both) mean[row0 + 1] = mu1; }
      if (rstd) { rstd[row0] = rs0; if (both) rstd[row0 + 1] = rs1; }
    }
#pragma unroll
    for (int k = 0; k < NC; ++k) {
      if (second[k] && !both) continue;
      const float mu = second[k] ? mu1 : mu0, rs = second[k] ? rs1 : rs0;
      float o[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) o[i] = (v[k][i] - mu) * rs * wv[k][i] + bv[k][i];
      st8_fast(y + (long)row0 * d + (long)(lane + 64 * k) * 8, o);
    }
  }
}

// NCH = 16-byte chunks per lane (d <= 512 * NCH); two rows per wave per iteration, and the NEXT iteration's two rows are
// requested (packed, 4 registers per chunk) before the current ones are reduced: the kernel is latency-bound (load ->
// two dependent wave reductions -> store), a wave that handles one iteration at a time leaves HBM idle most of the time
template <typename T, int NCH, int R = 2>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const T* __restrict__ x, const float* __restrict__ w,
                                                     const float* __restrict__ b, T* __restrict__ y,
                                                     float* __restrict__ mean, float* __restrict__ rstd, int rows,
                                                     int d, float eps) {
  const int lane = threadIdx.x & 63;
  const int wave_global = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int nwaves = gridDim.x * 4;
  const int nchunk = d >> 3;
  float wv[NCH][8], bv[NCH][8];
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const int ch = lane + 64 * c;
    if (ch < nchunk) { ld8(w + ch * 8, wv[c]); ld8(b + ch * 8, bv[c]); }
  }
  Raw8<T> nxt[R][NCH];
  auto fetch = [&](int row0) {
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int row = row0 + r < rows ? row0 + r : rows - 1;
#pragma unroll
      for (int c = 0; c < NCH; ++c) {
        const int ch = lane + 64 * c;
        if (ch < nchunk) nxt[r][c].load(x + (long)row * d + ch * 8);
      }
    }
  };
  int row0 = wave_global * R;
  if (row0 < rows) fetch(row0);
  for (; row0 < rows; row0 += nwaves * R) {
    float v[R][NCH][8];
    float s[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
      s[r] = 0.f;
#pragma unroll
      for (int c = 0; c < NCH; ++c)
        if (lane + 64 * c < nchunk) {
          nxt[r][c].unpack(v[r][c]);
#pragma unroll
          for (int i = 0; i < 8; ++i) s[r] += v[r][c][i];
        }
    }
    if (row0 + nwaves * R < rows) fetch(row0 + nwaves * R);
    float mu[R], rs[R];
#pragma unroll
    for (int r = 0; r < R; ++r) mu[r] = wave_sum_fast(s[r]) / d;
#pragma unroll
    for (int r = 0; r < R; ++r) {
      float q = 0.f;
#pragma unroll
      for (int c = 0; c < NCH; ++c)
        if (lane + 64 * c < nchunk) {
#pragma unroll
          for (int i = 0; i < 8; ++i) { float t = v[r][c][i] - mu[r]; q += t * t; }
        }
      rs[r] = rsqrtf(wave_sum_fast(q) / d + eps);
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int row = row0 + r;
      if (row < rows) {
        if (lane == 0) { if (mean) mean[row] = mu[r]; if (rstd) rstd[row] = rs[r]; }
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
          const int ch = lane + 64 * c;
          if (ch < nchunk) {
            float o[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) o[i] = (v[r][c][i] - mu[r]) * rs[r] * wv[c][i] + bv[c][i];
            st8_fast(y + (long)row * d + ch * 8, o);
          }
        }
      }
    }
  }
}

// scalar fallback (any d): one wave per row, three passes over the row.
template <typename T>
__global__ __launch_bounds__(256) void ln_fwd_scalar_kernel(const T* __restrict__ x, const float* __restrict__ w,
                                                            const float* __restrict__ b, T* __restrict__ y,
                                                            float* __restrict__ mean, float* __restrict__ rstd,
                                                            int rows, int d, float eps) {
  const int lane = threadIdx.x & 63;
  const int nwaves = gridDim.x * 4;
  for (int row = blockIdx.x * 4 + (threadIdx.x >> 6); row < rows; row += nwaves) {
    const T* xr = x + (long)row * d;
    float s = 0.f;
    for (int i = lane; i < d; i += 64) s += ld<T>(xr + i);
    const float mu = wave_sum(s) / d;
    float q = 0.f;
    for (int i = lane; i < d; i += 64) { float t = ld<T>(xr + i) - mu; q += t * t; }
    const float rs = rsqrtf(wave_sum(q) / d + eps);
    if (lane == 0) { if (mean) mean[row] = mu; if (rstd) rstd[row] = rs; }
    for (int i = lane; i < d; i += 64) st<T>(y + (long)row * d + i, (ld<T>(xr + i) - mu) * rs * w[i] + b[i]);
  }
}

// backward: wave per row (grid-stride); per-lane register partials for dw/db of the
// lane's own columns; block partial written to part[block][2][d].  NCH as in the forward kernel (register budget ->
// occupancy: the d = 768 / 512 towers need 2 / 1 chunk slots, not 4).
template <typename T, int NCH>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ x,
                                                     const float* __restrict__ w, const float* __restrict__ mean,
                                                     const float* __restrict__ rstd, const T* __restrict__ dres,
                                                     T* __restrict__ dx, float* __restrict__ part, int rows, int d) {
  DH_DYN_LDS(float, sm);  // [4 waves][2][d]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nwaves = gridDim.x * 4;
  const int nchunk = d >> 3;
  float aw[NCH][8], ab[NCH][8], wv[NCH][8];
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
#pragma unroll
    for (int i = 0; i < 8; ++i) { aw[c][i] = 0.f; ab[c][i] = 0.f; wv[c][i] = 0.f; }
    if (lane + 64 * c < nchunk) ld8(w + (lane + 64 * c) * 8, wv[c]);
  }

  for (int row = blockIdx.x * 4 + wave; row < rows; row += nwaves) {
    float xv[NCH][8], dv[NCH][8], rv[NCH][8];
    // all loads of the row first (x, dy, residual-branch gradient), then the statistics
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const int ch = lane + 64 * c;
      if (ch < nchunk) {
        ld8(x + (long)row * d + ch * 8, xv[c]);
        ld8(dy + (long)row * d + ch * 8, dv[c]);
        if (dres) ld8(dres + (long)row * d + ch * 8, rv[c]);
      }
    }
    const float mu = mean[row], rs = rstd[row];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      if (lane + 64 * c < nchunk) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const float xh = (xv[c][i] - mu) * rs, g = dv[c][i] * wv[c][i];
          s1 += g;
          s2 += g * xh;
          aw[c][i] += dv[c][i] * xh;
          ab[c][i] += dv[c][i];
          xv[c][i] = xh;
          dv[c][i] = g;
        }
      }
    }
    const float c1 = wave_sum(s1) / d, c2 = wave_sum(s2) / d;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const int ch = lane + 64 * c;
      if (ch < nchunk) {
        float o[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) o[i] = rs * (dv[c][i] - c1 - xv[c][i] * c2);
        if (dres) {
#pragma unroll
          for (int i = 0; i < 8; ++i) o[i] += rv[c][i];
        }
        st8_fast(dx + (long)row * d + ch * 8, o);
      }
    }
  }
  // block reduce of the 4 waves' partials through LDS
  float* my = sm + wave * 2 * d;
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    int ch = lane + 64 * c;
    if (ch < nchunk) {
#pragma unroll
      for (int i = 0; i < 8; ++i) { my[ch * 8 + i] = aw[c][i]; my[d + ch * 8 + i] = ab[c][i]; }
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 2 * d; i += 256)
    part[(long)blockIdx.x * 2 * d + i] = sm[i] + sm[2 * d + i] + sm[4 * d + i] + sm[6 * d + i];
}

// bf16 specialisation of the backward: the three row operands stay PACKED in registers (4 instead of 8 registers per chunk
// each) and are unpacked twice (statistics pass, output pass).  The generic kernel above keeps them as floats: 148 VGPRs at
// d = 768 -> 3 waves per SIMD, and its 1024-block grid then runs as 768 + 256 blocks (a second, third-full round).  This
// one fits 4 waves per SIMD; the grid is sized from the occupancy the runtime reports so every block is resident at once.
__device__ __forceinline__ void unpack8(const uint4& v, float* o) {
  const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    o[2 * i] = __uint_as_float(w[i] << 16);
    o[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u);
  }
}

template <int NCH>
__global__ __launch_bounds__(256, NCH <= 2 ? 4 : 2) void ln_bwd_bf16_kernel(const bf16_t* __restrict__ dy, const bf16_t* __restrict__ x,
                                                             const float* __restrict__ w, const float* __restrict__ mean,
                                                             const float* __restrict__ rstd, const bf16_t* __restrict__ dres,
                                                             bf16_t* __restrict__ dx, float* __restrict__ part, int rows, int d) {
  DH_DYN_LDS(float, sm);  // [4 waves][2][d]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nwaves = gridDim.x * 4;
  const int nchunk = d >> 3;
  float aw[NCH][8], ab[NCH][8], wv[NCH][8];
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
#pragma unroll
    for (int i = 0; i < 8; ++i) { aw[c][i] = 0.f; ab[c][i] = 0.f; wv[c][i] = 0.f; }
    if (lane + 64 * c < nchunk) ld8(w + (lane + 64 * c) * 8, wv[c]);
  }
  for (int row = blockIdx.x * 4 + wave; row < rows; row += nwaves) {
    uint4 xr[NCH], dr[NCH], rr[NCH];
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const int ch = lane + 64 * c;
      if (ch < nchunk) {
        xr[c] = *reinterpret_cast<const uint4*>(x + (long)row * d + ch * 8);
        dr[c] = *reinterpret_cast<const uint4*>(dy + (long)row * d + ch * 8);
        if (dres) rr[c] = *reinterpret_cast<const uint4*>(dres + (long)row * d + ch * 8);
      }
    }
    const float mu = mean[row], rs = rstd[row];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      if (lane + 64 * c < nchunk) {
        float xv[8], dv[8];
        unpack8(xr[c], xv);
        unpack8(dr[c], dv);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const float xh = (xv[i] - mu) * rs, g = dv[i] * wv[c][i];
          s1 += g;
          s2 += g * xh;
          aw[c][i] += dv[i] * xh;
          ab[c][i] += dv[i];
        }
      }
    }
    const float c1 = wave_sum_fast(s1) / d, c2 = wave_sum_fast(s2) / d;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      // opaque to the optimiser: the unpacked floats of the statistics pass must die there, not be carried to this pass
#ifndef DH_HOST_EMU   // (a VGPR constraint: meaningless on the host build of tests/hipemu)
      asm volatile("" : "+v"(xr[c].x), "+v"(xr[c].y), "+v"(xr[c].z), "+v"(xr[c].w), "+v"(dr[c].x), "+v"(dr[c].y), "+v"(dr[c].z), "+v"(dr[c].w));
#endif
    }
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const int ch = lane + 64 * c;
      if (ch < nchunk) {
        float xv[8], dv[8], o[8];
        unpack8(xr[c], xv);
        unpack8(dr[c], dv);
#pragma unroll
        for (int i = 0; i < 8; ++i) o[i] = rs * (dv[i] * wv[c][i] - c1 - (xv[i] - mu) * rs * c2);
        if (dres) {
          float rv[8];
          unpack8(rr[c], rv);
#pragma unroll
          for (int i = 0; i < 8; ++i) o[i] += rv[i];
        }
        st8_hw(dx + (long)row * d + ch * 8, o);
      }
    }
  }
  float* my = sm + wave * 2 * d;
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    int ch = lane + 64 * c;
    if (ch < nchunk) {
#pragma unroll
      for (int i = 0; i < 8; ++i) { my[ch * 8 + i] = aw[c][i]; my[d + ch * 8 + i] = ab[c][i]; }
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 2 * d; i += 256)
    part[(long)blockIdx.x * 2 * d + i] = sm[i] + sm[2 * d + i] + sm[4 * d + i] + sm[6 * d + i];
}

template <typename T>
__global__ __launch_bounds__(256) void ln_bwd_scalar_kernel(const T* __restrict__ dy, const T* __restrict__ x,
                                                            const float* __restrict__ w, const float* __restrict__ mean,
                                                            const float* __restrict__ rstd, const T* __restrict__ dres,
                                                            T* __restrict__ dx, float* __restrict__ dw,
                                                            float* __restrict__ db, int rows, int d) {
  const int lane = threadIdx.x & 63;
  const int nwaves = gridDim.x * 4;
  for (int row = blockIdx.x * 4 + (threadIdx.x >> 6); row < rows; row += nwaves) {
    const float mu = mean[row], rs = rstd[row];
    float s1 = 0.f, s2 = 0.f;
    for (int i = lane; i < d; i += 64) {
      float xh = (ld<T>(x + (long)row * d + i) - mu) * rs, dv = ld<T>(dy + (long)row * d + i), g = dv * w[i];
      s1 += g; s2 += g * xh;
      atomicAdd(dw + i, dv * xh);
      atomicAdd(db + i, dv);
    }
    const float c1 = wave_sum(s1) / d, c2 = wave_sum(s2) / d;
    for (int i = lane; i < d; i += 64) {
      float xh = (ld<T>(x + (long)row * d + i) - mu) * rs, g = ld<T>(dy + (long)row * d + i) * w[i];
      float o = rs * (g - c1 - xh * c2);
      if (dres) o += ld<T>(dres + (long)row * d + i);
      st<T>(dx + (long)row * d + i, o);
    }
  }
}

// part[nblocks][2d] -> dw/db: block = 64 columns x 4 row-lanes, coalesced rows of the partial matrix
__global__ __launch_bounds__(256) void ln_reduce_kernel(const float* __restrict__ part, int nblocks, int d,
                                                        float* __restrict__ dw, float* __restrict__ db) {
  __shared__ float red[4][64];
  const int i = blockIdx.x * 64 + (threadIdx.x & 63);  // over 2*d
  const int rl = threadIdx.x >> 6;
  const int per = (nblocks + gridDim.y - 1) / gridDim.y;
  const int b0 = blockIdx.y * per, b1 = min(nblocks, b0 + per);
  float s = 0.f;
  if (i < 2 * d)
    for (int b = b0 + rl; b < b1; b += 4) s += part[(long)b * 2 * d + i];
  red[rl][threadIdx.x & 63] = s;
  __syncthreads();
  if (rl == 0 && i < 2 * d) {
    s = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
    atomicAdd(i < d ? dw + i : db + (i - d), s);
  }
}

// The same for up to LN_MANY LayerNorms in ONE launch (blockIdx.z = which): a tower's backward leaves the per-block partials of
// every LayerNorm in place and reduces them together at its end -- 51 reduce launches per CLIP step become 2.
constexpr int LN_MANY = 32;
struct LnMany {
  const float* part[LN_MANY];
  float* dw[LN_MANY];
  float* db[LN_MANY];
  int nb[LN_MANY];
  int d[LN_MANY];
};
__global__ __launch_bounds__(256) void ln_reduce_many_kernel(LnMany m) {
  __shared__ float red[4][64];
  const int z = blockIdx.z;
  const int d = m.d[z], nblocks = m.nb[z];
  const float* __restrict__ part = m.part[z];
  const int i = blockIdx.x * 64 + (threadIdx.x & 63);  // over 2*d
  if (blockIdx.x * 64 >= 2 * d) return;                // (block-uniform: grid.x is sized for the widest item)
  const int rl = threadIdx.x >> 6;
  const int per = (nblocks + gridDim.y - 1) / gridDim.y;
  const int b0 = blockIdx.y * per, b1 = min(nblocks, b0 + per);
  float s = 0.f;
  if (i < 2 * d)
    for (int b = b0 + rl; b < b1; b += 4) s += part[(long)b * 2 * d + i];
  red[rl][threadIdx.x & 63] = s;
  __syncthreads();
  if (rl == 0 && i < 2 * d) {
    s = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
    atomicAdd(i < d ? m.dw[z] + i : m.db[z] + (i - d), s);
  }
}

}  // namespace

static int ln_grid(int rows) {
  int g = dh_cdiv(rows, 4);
  return g > 1024 ? 1024 : g;
}

// grid <= cap whose waves all run the same number of loop iterations (rows_per_block rows per block and iteration): 25600
// rows on 1024 blocks of 8 rows is 3.1 iterations -> a quarter of the waves run a 4th one while the rest idle
static int ln_balanced_grid(int rows, int rows_per_block, int cap) {
  const int iters = dh_cdiv(rows, rows_per_block * cap);
  int g = dh_cdiv(rows, rows_per_block * iters);
  return g < 1 ? 1 : g;
}

extern "C" int dh_layernorm_fwd(int dtype, const void* x, const float* w, const float* b, void* y, float* mean,
                                float* rstd, int rows, int d, float eps, dh_stream_t stream) {
  hipStream_t st = (hipStream_t)stream;
  DH_REQUIRE(x && w && b && y && rows > 0 && d > 0, "dh_layernorm_fwd: bad args");
  const bool vec = (d % 8 == 0) && d <= 8 * 64 * LN_MAXC;
  // tuning knobs (read once): rows per wave and iteration (2 / 4), cap of the grid
  static int ln_r = 0, ln_cap = 0;
  if (!ln_r) { const char* ev = getenv("DH_LN_FWD_R"); ln_r = ev ? atoi(ev) : 2; if (ln_r != 4) ln_r = 2; }
  if (!ln_cap) { const char* ev = getenv("DH_LN_FWD_CAP"); ln_cap = ev ? atoi(ev) : 1024; if (ln_cap < 64) ln_cap = 64; }
  dim3 grid(vec ? ln_balanced_grid(rows, 4 * ln_r, ln_cap) : ln_grid(rows));      // vector kernel: 4 waves x R rows per iteration
  const int nch = dh_cdiv(d / 8, 64);
#define LN_FWD(TT)                                                                                                                       \
  {                                                                                                                                      \
    if (ln_r == 4 && nch <= 1) hipLaunchKernelGGL((ln_fwd_kernel<TT, 1, 4>), grid, dim3(256), 0, st, (const TT*)x, w, b, (TT*)y, mean, rstd, rows, d, eps);       \
    else if (ln_r == 4 && nch == 2) hipLaunchKernelGGL((ln_fwd_kernel<TT, 2, 4>), grid, dim3(256), 0, st, (const TT*)x, w, b, (TT*)y, mean, rstd, rows, d, eps);  \
    else if (nch <= 1) hipLaunchKernelGGL((ln_fwd_kernel<TT, 1>), grid, dim3(256), 0, st, (const TT*)x, w, b, (TT*)y, mean, rstd, rows, d, eps);       \
    else if (nch == 2) hipLaunchKernelGGL((ln_fwd_kernel<TT, 2>), grid, dim3(256), 0, st, (const TT*)x, w, b, (TT*)y, mean, rstd, rows, d, eps);  \
    else hipLaunchKernelGGL((ln_fwd_kernel<TT, LN_MAXC>), grid, dim3(256), 0, st, (const TT*)x, w, b, (TT*)y, mean, rstd, rows, d, eps);          \
  }
  // row pairs (d = 768, 256): see ln_fwd_pair_kernel.  DH_LN_FWD_PAIR=0 (read once) keeps the per-row kernel for A/B runs.
  static int ln_pair = -1;
  if (ln_pair < 0) { const char* ev = getenv("DH_LN_FWD_PAIR"); ln_pair = ev ? (atoi(ev) != 0) : 1; }
  const int pair_nc = (vec && ln_pair && ln_r == 2 && (d / 8) % 64 != 0 && (2 * (d / 8)) % 64 == 0) ? 2 * (d / 8) / 64 : 0;
  if (pair_nc == 1 || pair_nc == 3) {
    dim3 pgrid(ln_balanced_grid(rows, 8, ln_cap));            // 4 waves x one row pair per iteration
#define LN_PAIR(TT)                                                                                                                                 \
    {                                                                                                                                               \
      if (pair_nc == 1) hipLaunchKernelGGL((ln_fwd_pair_kernel<TT, 1>), pgrid, dim3(256), 0, st, (const TT*)x, w, b, (TT*)y, mean, rstd, rows, d, eps); \
      else hipLaunchKernelGGL((ln_fwd_pair_kernel<TT, 3>), pgrid, dim3(256), 0, st, (const TT*)x, w, b, (TT*)y, mean, rstd, rows, d, eps);             \
    }
    if (dtype == DH_BF16) LN_PAIR(bf16_t)
    else LN_PAIR(float)
#undef LN_PAIR
    DH_CHECK_LAUNCH();
    return DH_OK;
  }
  if (dtype == DH_BF16) {
    if (vec) LN_FWD(bf16_t)
    else hipLaunchKernelGGL(ln_fwd_scalar_kernel<bf16_t>, grid, dim3(256), 0, st, (const bf16_t*)x, w, b, (bf16_t*)y, mean, rstd, rows, d, eps);
  } else {
    if (vec) LN_FWD(float)
    else hipLaunchKernelGGL(ln_fwd_scalar_kernel<float>, grid, dim3(256), 0, st, (const float*)x, w, b, (float*)y, mean, rstd, rows, d, eps);
  }
  DH_CHECK_LAUNCH();
  return DH_OK;
}

static int ln_bwd_blocks(int rows) {
  int nb = dh_cdiv(rows, 8);     // >= 2 rows per wave so the register partials amortise
  if (nb > 1024) nb = 1024;
  if (nb < 1) nb = 1;
  return nb;
}

extern "C" int64_t dh_layernorm_bwd_ws_bytes(int rows, int d) {
  int nb = ln_bwd_blocks(rows);
  return (int64_t)nb * 2 * d * sizeof(float);
}

// nb_out == nullptr: the whole backward (partials + reduce into dw / db).  Otherwise the reduce is left to the caller
// (dh_ln_reduce_many on the partials kept in `ws`): *nb_out = partial rows written, 0 if this shape took the scalar kernel
// (which accumulates into dw / db directly).
static int ln_bwd_impl(int dtype, const void* dy, const void* x, const float* w, const float* mean,
                       const float* rstd, const void* dres, void* dx, float* dw, float* db, int rows, int d,
                       void* ws, int64_t ws_bytes, dh_stream_t stream, int* nb_out) {
  hipStream_t st = (hipStream_t)stream;
  DH_REQUIRE(dy && x && w && mean && rstd && dx && dw && db && rows > 0 && d > 0, "dh_layernorm_bwd: bad args");
  const bool vec = (d % 8 == 0) && d <= 8 * 64 * LN_MAXC && (size_t)(8 * d * sizeof(float)) <= 64 * 1024;
  if (vec) {
    int nb = ln_bwd_blocks(rows);
    DH_REQUIRE(ws && ws_bytes >= (int64_t)nb * 2 * d * (int64_t)sizeof(float), "dh_layernorm_bwd: workspace too small");
    size_t lds = 8 * d * sizeof(float);
    const int nch = dh_cdiv(d / 8, 64);
    if (dtype == DH_BF16) {
      // all blocks resident at once: blocks per CU from the runtime's occupancy query (registers + the LDS of this d)
      static int cus = 0;
      if (!cus) {
        hipDeviceProp_t prop;
        int dev = 0;
        cus = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) ? prop.multiProcessorCount : 256;
      }
      int per_cu = 0;
#define LN_OCC(N) hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, ln_bwd_bf16_kernel<N>, 256, lds)
      hipError_t oe = nch <= 1 ? LN_OCC(1) : nch == 2 ? LN_OCC(2) : LN_OCC(LN_MAXC);
#undef LN_OCC
      int cap = (oe == hipSuccess && per_cu > 0 && per_cu * cus < 1024) ? per_cu * cus : 1024;
      if (cap > nb) cap = nb;                                  // never more blocks than the workspace was sized for
      nb = ln_balanced_grid(rows, 4, cap);
#define LN_BWD16(N) hipLaunchKernelGGL((ln_bwd_bf16_kernel<N>), dim3(nb), dim3(256), lds, st, (const bf16_t*)dy, (const bf16_t*)x, w, mean, rstd, (const bf16_t*)dres, (bf16_t*)dx, (float*)ws, rows, d)
      if (nch <= 1) LN_BWD16(1); else if (nch == 2) LN_BWD16(2); else LN_BWD16(LN_MAXC);
#undef LN_BWD16
    } else {
#define LN_BWD32(N) hipLaunchKernelGGL((ln_bwd_kernel<float, N>), dim3(nb), dim3(256), lds, st, (const float*)dy, (const float*)x, w, mean, rstd, (const float*)dres, (float*)dx, (float*)ws, rows, d)
      if (nch <= 1) LN_BWD32(1); else if (nch == 2) LN_BWD32(2); else LN_BWD32(LN_MAXC);
#undef LN_BWD32
    }
    DH_CHECK_LAUNCH();
    if (nb_out) *nb_out = nb;
    else hipLaunchKernelGGL(ln_reduce_kernel, dim3(dh_cdiv(2 * d, 64), nb >= 64 ? 16 : 1), dim3(256), 0, st, (const float*)ws, nb, d, dw, db);
  } else {
    if (nb_out) *nb_out = 0;
    dim3 grid(ln_grid(rows) > 256 ? 256 : ln_grid(rows));
    if (dtype == DH_BF16)
      hipLaunchKernelGGL(ln_bwd_scalar_kernel<bf16_t>, grid, dim3(256), 0, st, (const bf16_t*)dy, (const bf16_t*)x, w, mean, rstd, (const bf16_t*)dres, (bf16_t*)dx, dw, db, rows, d);
    else
      hipLaunchKernelGGL(ln_bwd_scalar_kernel<float>, grid, dim3(256), 0, st, (const float*)dy, (const float*)x, w, mean, rstd, (const float*)dres, (float*)dx, dw, db, rows, d);
  }
  DH_CHECK_LAUNCH();
  return DH_OK;
}

extern "C" int dh_layernorm_bwd(int dtype, const void* dy, const void* x, const float* w, const float* mean,
                                const float* rstd, const void* dres, void* dx, float* dw, float* db, int rows, int d,
                                void* ws, int64_t ws_bytes, dh_stream_t stream) {
  return ln_bwd_impl(dtype, dy, x, w, mean, rstd, dres, dx, dw, db, rows, d, ws, ws_bytes, stream, nullptr);
}

// LayerNorm backward WITHOUT the reduction of the weight / bias gradient partials: they stay in `part` ([*nb_out][2 d] fp32, the
// size dh_layernorm_bwd_ws_bytes reports) until dh_ln_reduce_many adds them into dw / db.
extern "C" int dh_layernorm_bwd_part(int dtype, const void* dy, const void* x, const float* w, const float* mean,
                                     const float* rstd, const void* dres, void* dx, float* dw, float* db, int rows, int d,
                                     void* part, int64_t part_bytes, int* nb_out, dh_stream_t stream) {
  DH_REQUIRE(nb_out, "dh_layernorm_bwd_part: nb_out is NULL");
  return ln_bwd_impl(dtype, dy, x, w, mean, rstd, dres, dx, dw, db, rows, d, part, part_bytes, stream, nb_out);
}

// dw[i] += column sums of part[i][nb[i]][0 .. d[i]), db[i] += those of columns d[i] .. 2 d[i]; ONE launch per 32 items.
extern "C" int dh_ln_reduce_many(const dh_ln_part* items, int n, dh_stream_t stream) {
  hipStream_t st = (hipStream_t)stream;
  DH_REQUIRE(items && n >= 0, "dh_ln_reduce_many: bad args");
  for (int i0 = 0; i0 < n; i0 += LN_MANY) {
    LnMany m;
    memset(&m, 0, sizeof(m));
    int cnt = 0, dmax = 0, nbmax = 0;
    for (int i = i0; i < n && cnt < LN_MANY; ++i) {
      const dh_ln_part& it = items[i];
      if (it.nb <= 0) continue;                      // (the scalar kernel accumulated directly)
      DH_REQUIRE(it.part && it.dw && it.db && it.d > 0, "dh_ln_reduce_many: item %d: bad pointers / width", i);
      m.part[cnt] = it.part; m.dw[cnt] = it.dw; m.db[cnt] = it.db; m.nb[cnt] = it.nb; m.d[cnt] = it.d;
      dmax = it.d > dmax ? it.d : dmax;
      nbmax = it.nb > nbmax ? it.nb : nbmax;
      ++cnt;
    }
    // slices of the partial rows per item: 1 while every item is short (< 64 partial rows: ONE add per element, in a fixed
    // order -- the pooled ln_post / ln_final and small batches stay run-to-run deterministic, and 15 of 16 blocks would add zeros),
    // 16 for the long ones (1024 partial rows of a full tower LayerNorm: the float atomics' order is the only non-determinism)
    const int gy = nbmax < 64 ? 1 : 16;
    if (cnt) hipLaunchKernelGGL(ln_reduce_many_kernel, dim3(dh_cdiv(2 * dmax, 64), gy, cnt), dim3(256), 0, st, m);
  }
  DH_CHECK_LAUNCH();
  return DH_OK;
}
