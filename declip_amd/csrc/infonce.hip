// Fused contrastive (InfoNCE) loss: logits = scale * Q K^T are produced tile by tile in LDS/registers,
// consumed by an online log-sum-exp and never written to HBM (unless the caller asks for them).
//   forward : model/clip.py:140-141 + loss_functions/loss.py:37-47 + utils/misc.py:415-428
//   backward: autograd of the above; two passes with swapped roles (rows of Q, then rows of K) so
//             neither needs atomics on the [B,D] gradient.
// All arithmetic fp32 (features are fp32): the loss is what parity is judged on.
#include "dh_common.h"
#include <type_traits>

#define DH_MAX_PAIRS 16

namespace {

struct PairTable {
  const float* Q[DH_MAX_PAIRS];
  const float* K[DH_MAX_PAIRS];
  float* dQ[DH_MAX_PAIRS];
  float* dK[DH_MAX_PAIRS];
  int label0[DH_MAX_PAIRS];   // positive column of local row i is label0 + i
  int excl0[DH_MAX_PAIRS];    // column excl0 + i is removed from row i's softmax (NT-Xent self-pair); < 0: none
};

constexpr int CT = 64;   // columns per tile
constexpr int KC = 32;   // contraction chunk

// acc[rr][cc] for rows ty*RPT+rr, cols tx*4+cc of the (RT x 64) tile: X rows from LDS Xs[RT][D+1],
// Y tile streamed from global through Ys[64][KC+1].
template <int RT>
__device__ __forceinline__ void tile_dots(const float* Xs, int D, const float* __restrict__ Y, int y0, int ny, float* Ys,
                                          float (&acc)[RT / 16][4]) {
  constexpr int RPT = RT / 16;
  const int t = threadIdx.x, tx = t & 15, ty = t >> 4;
#pragma unroll
  for (int r = 0; r < RPT; ++r)
#pragma unroll
    for (int c = 0; c < 4; ++c) acc[r][c] = 0.f;
  for (int k0 = 0; k0 < D; k0 += KC) {
    __syncthreads();
    // load Y[y0..y0+64)[k0..k0+32) : 2048 floats, 8 per thread
#pragma unroll
    for (int p = 0; p < 8; ++p) {
      int idx = t + p * 256;
      int kk = idx & 31, yy = idx >> 5;
      int y = y0 + yy, k = k0 + kk;
      Ys[yy * (KC + 1) + kk] = (y < ny && k < D) ? Y[(long)y * D + k] : 0.f;
    }
    __syncthreads();
    const int kmax = min(KC, D - k0);
    for (int kk = 0; kk < kmax; ++kk) {
      float x[RPT], y[4];
#pragma unroll
      for (int r = 0; r < RPT; ++r) x[r] = Xs[(ty * RPT + r) * (D + 1) + k0 + kk];
#pragma unroll
      for (int c = 0; c < 4; ++c) y[c] = Ys[(tx * 4 + c) * (KC + 1) + kk];
#pragma unroll
      for (int r = 0; r < RPT; ++r)
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[r][c] = fmaf(x[r], y[c], acc[r][c]);
    }
  }
}

template <int RT>
__device__ __forceinline__ void load_rows(float* Xs, const float* __restrict__ X, int x0, int nx, int D) {
  for (int i = threadIdx.x; i < RT * D; i += 256) {
    int r = i / D, k = i % D;
    Xs[r * (D + 1) + k] = (x0 + r < nx) ? X[(long)(x0 + r) * D + k] : 0.f;
  }
}

// merge (m, s) online-softmax states
__device__ __forceinline__ void lse_merge(float& m, float& s, float m2, float s2) {
  float mm = fmaxf(m, m2);
  if (mm == -INFINITY) { m = mm; s = 0.f; return; }
  s = s * __expf(m - mm) + s2 * __expf(m2 - mm);
  m = mm;
}

// forward partials: block (row tile, pair, column chunk) streams its chunk of K and writes, per row,
// the online-softmax state (max, sum), the label logit and the count of logits above it; a tiny
// second kernel merges the chunks.  Column chunking is what gives B = 4096 (8 GPUs) enough blocks.
// columns per block: chosen on the host so that (row tiles x pairs x chunks) fills the 256 CUs
static int nce_chunk_cols(int rows, int cols, int n_pairs, int rt) {
  const int tiles = dh_cdiv(rows, rt) * n_pairs;
  int want = dh_cdiv(512, tiles);                 // ~2 blocks per CU
  int chunk = dh_cdiv(dh_cdiv(cols, want), 64) * 64;
  if (chunk < 64) chunk = 64;
  return chunk;
}

template <int RT>
__global__ __launch_bounds__(256) void nce_fwd_kernel(PairTable pt, int b, int B, int D, const float* __restrict__ scale_p, int label0,
                                                      float* __restrict__ part, float* __restrict__ logits_out, int chunk_cols) {
  constexpr int RPT = RT / 16;
  const float scale = *scale_p;
  DH_DYN_LDS(float, sm);
  float* Xs = sm;                      // [RT][D+1]
  float* Ys = Xs + RT * (D + 1);       // [64][KC+1]
  float* lab = Ys + CT * (KC + 1);     // [RT] label logits
  const int pair = blockIdx.y;
  label0 = pt.label0[pair];
  const int excl0 = pt.excl0[pair];
  const int nchunk = gridDim.z, chunk = blockIdx.z;
  const int cbeg = chunk * chunk_cols, cend = min(B, cbeg + chunk_cols);
  const float* Q = pt.Q[pair];
  const float* K = pt.K[pair];
  const int r0 = blockIdx.x * RT;
  const int t = threadIdx.x, tx = t & 15, ty = t >> 4;
  load_rows<RT>(Xs, Q, r0, b, D);
  __syncthreads();
  // label logit: 8 threads per row (RT <= 32)
  {
    const int r = t >> 3, part_i = t & 7;
    float a = 0.f;
    if (r < RT && r0 + r < b) {
      const float* kr = K + (long)(label0 + r0 + r) * D;
      for (int k = part_i; k < D; k += 8) a = fmaf(Xs[r * (D + 1) + k], kr[k], a);
    }
    a += __shfl_xor(a, 1, 64); a += __shfl_xor(a, 2, 64); a += __shfl_xor(a, 4, 64);
    if (r < RT && part_i == 0) lab[r] = a * scale;
  }
  __syncthreads();
  float m[RPT], s[RPT], cnt[RPT], ll[RPT];
#pragma unroll
  for (int r = 0; r < RPT; ++r) { m[r] = -INFINITY; s[r] = 0.f; cnt[r] = 0.f; ll[r] = lab[ty * RPT + r]; }
  for (int c0 = cbeg; c0 < cend; c0 += CT) {
    float acc[RPT][4];
    tile_dots<RT>(Xs, D, K, c0, cend, Ys, acc);
#pragma unroll
    for (int r = 0; r < RPT; ++r) {
      const int row = r0 + ty * RPT + r;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const int col = c0 + tx * 4 + c;
        if (col < cend) {
          const float v = acc[r][c] * scale;
          if (logits_out && row < b) logits_out[((long)pair * b + row) * B + col] = v;
          if (excl0 >= 0 && col == excl0 + row) continue;             // self-pair removed from the softmax
          if (col != label0 + row && v > ll[r]) cnt[r] += 1.f;
          if (v > m[r]) { s[r] = s[r] * __expf(m[r] - v) + 1.f; m[r] = v; } else s[r] += __expf(v - m[r]);
        }
      }
    }
  }
  // combine the 16 threads (tx) that share rows: lanes differing in bits 0..3
#pragma unroll
  for (int r = 0; r < RPT; ++r) {
#pragma unroll
    for (int o = 1; o < 16; o <<= 1) {
      float m2 = __shfl_xor(m[r], o, 64), s2 = __shfl_xor(s[r], o, 64);
      lse_merge(m[r], s[r], m2, s2);
      cnt[r] += __shfl_xor(cnt[r], o, 64);
    }
    const int row = r0 + ty * RPT + r;
    if (tx == 0 && row < b) {
      float* o = part + (((long)pair * nchunk + chunk) * b + row) * 4;
      o[0] = m[r]; o[1] = s[r]; o[2] = cnt[r]; o[3] = ll[r];
    }
  }
}

__global__ void nce_finalize_kernel(const float* __restrict__ part, int n_pairs, int nchunk, int b,
                                    float* __restrict__ row_loss, float* __restrict__ row_lse,
                                    float* __restrict__ correct1, float* __restrict__ correct5) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;  // over pairs * b
  if (i >= n_pairs * b) return;
  const int pair = i / b, row = i % b;
  float m = -INFINITY, s = 0.f, cnt = 0.f, ll = 0.f;
  for (int c = 0; c < nchunk; ++c) {
    const float* o = part + (((long)pair * nchunk + c) * b + row) * 4;
    lse_merge(m, s, o[0], o[1]);
    cnt += o[2];
    ll = o[3];
  }
  const float lse = m + __logf(s);
  row_lse[i] = lse;
  row_loss[i] = lse - ll;
  if (correct1) correct1[i] = cnt < 0.5f ? 1.f : 0.f;
  if (correct5) correct5[i] = cnt < 4.5f ? 1.f : 0.f;
}

// dX_x = scale * sum_y G(x,y) Y_y for a tile of RT rows of X against all rows of Y.
//   mode 0 (X = Q rows i, Y = K rows j): G = g_i (exp(scale<Q_i,K_j> - lse_i) - [j == label0 + i]); also dscale.
//   mode 1 (X = K rows j, Y = Q rows i): same G, statistics indexed by the Y row.
template <int RT>
__global__ __launch_bounds__(256) void nce_bwd_kernel(PairTable pt, int mode, int b, int B, int D, const float* __restrict__ scale_p,
                                                      int label0, const float* __restrict__ row_lse,
                                                      const float* __restrict__ g_row, float* __restrict__ dscale, int chunk_cols) {
  constexpr int RPT = RT / 16;
  const float scale = *scale_p;
  DH_DYN_LDS(float, sm);
  float* Xs = sm;                        // [RT][D+1]
  float* dXs = Xs + RT * (D + 1);        // [RT][D+1]
  float* Ys = dXs + RT * (D + 1);        // [64][KC+1]
  float* Gs = Ys + CT * (KC + 1);        // [RT][65]
  float* red = Gs + RT * (CT + 1);       // [8]
  const int pair = blockIdx.y;
  label0 = pt.label0[pair];
  const int excl0 = pt.excl0[pair];
  const float* X = mode == 0 ? pt.Q[pair] : pt.K[pair];
  const float* Y = mode == 0 ? pt.K[pair] : pt.Q[pair];
  float* dX = mode == 0 ? pt.dQ[pair] : pt.dK[pair];
  if (dX == nullptr && !(mode == 0 && dscale != nullptr)) return;     // gradient not wanted for this operand
  const int nx = mode == 0 ? b : B, ny_all = mode == 0 ? B : b;
  const int ybeg = blockIdx.z * chunk_cols, ny = min(ny_all, ybeg + chunk_cols);   // column chunk of this block
  const bool chunked = gridDim.z > 1;
  const int r0 = blockIdx.x * RT;
  const int t = threadIdx.x, tx = t & 15, ty = t >> 4;
  const float* lse_p = row_lse + (long)pair * b;
  const float* g_p = g_row + (long)pair * b;
  load_rows<RT>(Xs, X, r0, nx, D);
  for (int i = t; i < RT * (D + 1); i += 256) dXs[i] = 0.f;
  float ds_acc = 0.f;
  for (int c0 = ybeg; c0 < ny; c0 += CT) {
    float acc[RPT][4];
    tile_dots<RT>(Xs, D, Y, c0, ny, Ys, acc);
#pragma unroll
    for (int r = 0; r < RPT; ++r) {
      const int xr = r0 + ty * RPT + r;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const int yc = c0 + tx * 4 + c;
        float gval = 0.f;
        if (xr < nx && yc < ny) {
          const int i = mode == 0 ? xr : yc;   // Q row (local)
          const int j = mode == 0 ? yc : xr;   // K row (global)
          const float logit = acc[r][c] * scale;
          const float p = (excl0 >= 0 && j == excl0 + i) ? 0.f : __expf(logit - lse_p[i]);
          gval = g_p[i] * (p - (j == label0 + i ? 1.f : 0.f));
          ds_acc += gval * acc[r][c];
        }
        Gs[(ty * RPT + r) * (CT + 1) + tx * 4 + c] = gval;
      }
    }
    // dX[RT][D] += G[RT][64] . Y[64][D], streamed over D chunks (Y chunk re-read through Ys)
    for (int k0 = 0; k0 < D; k0 += KC) {
      __syncthreads();
#pragma unroll
      for (int p = 0; p < 8; ++p) {
        int idx = t + p * 256;
        int kk = idx & 31, yy = idx >> 5;
        int y = c0 + yy, k = k0 + kk;
        Ys[yy * (KC + 1) + kk] = (y < ny && k < D) ? Y[(long)y * D + k] : 0.f;
      }
      __syncthreads();
      // thread owns rows ty*RPT+r, chunk columns tx and tx+16
#pragma unroll
      for (int r = 0; r < RPT; ++r) {
        float a0 = 0.f, a1 = 0.f;
        const float* grow = Gs + (ty * RPT + r) * (CT + 1);
#pragma unroll 8
        for (int yy = 0; yy < CT; ++yy) {
          const float gv = grow[yy];
          a0 = fmaf(gv, Ys[yy * (KC + 1) + tx], a0);
          a1 = fmaf(gv, Ys[yy * (KC + 1) + tx + 16], a1);
        }
        float* drow = dXs + (ty * RPT + r) * (D + 1) + k0;
        if (k0 + tx < D) drow[tx] += a0;
        if (k0 + tx + 16 < D) drow[tx + 16] += a1;
      }
    }
  }
  __syncthreads();
  for (int i = t; i < RT * D; i += 256) {
    int r = i / D, k = i % D;
    if (dX != nullptr && r0 + r < nx) {
      if (chunked) atomicAdd(dX + (long)(r0 + r) * D + k, dXs[r * (D + 1) + k] * scale);
      else dX[(long)(r0 + r) * D + k] = dXs[r * (D + 1) + k] * scale;
    }
  }
  if (mode == 0 && dscale) {
    float tot = block_sum256(ds_acc, red);
    if (t == 0) atomicAdd(dscale, tot);
  }
}

// ------------------------------------------------------------------------------------------------------------------
// The same two kernels on the matrix pipe.  v_mfma_f32_32x32x2_f32 is an exact fp32 FMA chain at the fp32 vector peak; the
// VALU tile loops above reach ~4 % of it, and their cost grows with the number of gathered columns (B = 512 * world): at
// 8 ranks the loss took 2.9 ms of a ~33 ms step (tools/bench_nce.py).  Block = 4 waves, 32 rows of X (whole rows in LDS),
// Y streamed as tiles of 128 rows x 128 contraction columns, register-prefetched.  Operands are issued swapped for the
// logits (A = Y, B = X): a LANE owns one X row and its 16 accumulator registers are 16 Y rows, so the online softmax /
// the G factor are in-lane.  k pairing as in nn_search_mfma_kernel: lane (r, h) feeds k = 8g + 4h + j to MFMA j of group g
// for both operands (one 16-byte LDS read per operand per 4 MFMAs).
constexpr int MX = 32, MY = 128, MW = 128;

struct YStage { float4 v[16]; };
// thread t stages row t >> 1 of the tile, 64 consecutive columns (t & 1) * 64 .. of the pass
__device__ __forceinline__ void ystage_fetch(YStage& st, const float* __restrict__ Y, int y0, int ny, int D, int d0) {
  const int t = threadIdx.x, y = y0 + (t >> 1), dd = d0 + (t & 1) * 64;
#pragma unroll
  for (int q = 0; q < 16; ++q) {
    st.v[q] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (y < ny && dd + 4 * q < D) st.v[q] = *reinterpret_cast<const float4*>(Y + (long)y * D + dd + 4 * q);
  }
}
__device__ __forceinline__ void ystage_commit(const YStage& st, float* Ys) {
  const int t = threadIdx.x;
  float* dst = Ys + (t >> 1) * (MW + 4) + (t & 1) * 64;
#pragma unroll
  for (int q = 0; q < 16; ++q) *reinterpret_cast<float4*>(dst + 4 * q) = st.v[q];
}
// D > 512: the 32 X rows no longer fit in LDS next to the Y tile -- only the 128 columns of the current pass are staged
// ([32][132], re-read from L2 for every Y tile: + 25 % traffic on a kernel that is bound by the fp32 matrix rate)
__device__ __forceinline__ void xpass_load(float* Xs, const float* __restrict__ X, int r0, int nx, int D, int d0) {
  const int t = threadIdx.x;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int idx = t + 256 * q, rr = idx >> 5, c = (idx & 31) * 4;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (r0 + rr < nx && d0 + c < D) v = *reinterpret_cast<const float4*>(X + (long)(r0 + rr) * D + d0 + c);
    *reinterpret_cast<float4*>(Xs + rr * (MW + 4) + c) = v;
  }
}
// Y row (inside the wave's 32) of accumulator register x for a lane in half h
__device__ __forceinline__ int acc_row(int x, int h) { return (x >> 2) * 8 + h * 4 + (x & 3); }

// S^T-fragment of one pass: sacc[x] += sum over the 128 staged columns of Y[w*32 + acc_row][k] * X[lane & 31][d0 + k]
// (only the groups of 8 columns that exist: for D not a multiple of 128 the rest of the pass would read the uninitialised pad
// of the X rows -- NaN bit patterns left in LDS by an earlier kernel times the staged zeros are NaN, not 0)
// Xp: column d0 of row 0 of the X rows in LDS (row stride XS)
__device__ __forceinline__ void logits_pass(f32x16_t& sacc, const float* Xp, int XS, const float* Ys, int d0, int D, int wave, int r, int h) {
  const int ng = min(MW / 8, (D - d0) / 8);
#pragma unroll 4
  for (int g = 0; g < ng; ++g) {
    const float4 a = *reinterpret_cast<const float4*>(Ys + (wave * 32 + r) * (MW + 4) + 8 * g + 4 * h);
    const float4 b = *reinterpret_cast<const float4*>(Xp + r * XS + 8 * g + 4 * h);
    sacc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, sacc, 0, 0, 0);
    sacc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, sacc, 0, 0, 0);
    sacc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b.z, sacc, 0, 0, 0);
    sacc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b.w, sacc, 0, 0, 0);
  }
}

template <bool XRES>
__global__ __launch_bounds__(256) void nce_fwd_mfma_kernel(PairTable pt, int b, int B, int D, const float* __restrict__ scale_p,
                                                           float* __restrict__ part, float* __restrict__ logits_out, int chunk_cols) {
  const float scale = *scale_p;
  DH_DYN_LDS_A16(float, sm);
  const int XS = XRES ? D + 4 : MW + 4;
  float* Xs = sm;                       // XRES: [32][D + 4] whole rows; else [32][132] the columns of the current pass
  float* Ys = Xs + MX * XS;             // [128][132]
  float* lab = Ys + MY * (MW + 4);      // [32] label logits
  float* red = lab + MX;                // [4 waves][32][3]
  const int pair = blockIdx.y;
  const int label0 = pt.label0[pair], excl0 = pt.excl0[pair];
  const int nchunk = gridDim.z, chunk = blockIdx.z;
  const int cbeg = chunk * chunk_cols, cend = min(B, cbeg + chunk_cols);
  const float* Q = pt.Q[pair];
  const float* K = pt.K[pair];
  const int r0 = blockIdx.x * MX;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, r = lane & 31, h = lane >> 5;
  if (XRES) {
    for (int i = t; i < MX * (D / 4); i += 256) {
      const int rr = i / (D / 4), k4 = i % (D / 4);
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (r0 + rr < b) v = *reinterpret_cast<const float4*>(Q + (long)(r0 + rr) * D + k4 * 4);
      *reinterpret_cast<float4*>(Xs + rr * XS + k4 * 4) = v;
    }
    __syncthreads();
  }
  {  // label logit: 8 threads per row
    const int rr = t >> 3, part_i = t & 7;
    float a = 0.f;
    if (r0 + rr < b) {
      const float* kr = K + (long)(label0 + r0 + rr) * D;
      const float* xr = XRES ? Xs + rr * XS : Q + (long)(r0 + rr) * D;
      for (int k = part_i; k < D; k += 8) a = fmaf(xr[k], kr[k], a);
    }
    a += __shfl_xor(a, 1, 64); a += __shfl_xor(a, 2, 64); a += __shfl_xor(a, 4, 64);
    if (part_i == 0) lab[rr] = a * scale;
  }
  __syncthreads();
  const int row = r0 + r;
  const float ll = lab[r];
  float m = -INFINITY, ssum = 0.f, cnt = 0.f;
  const int npass = (D + MW - 1) / MW;
  YStage st;
  ystage_fetch(st, K, cbeg, cend, D, 0);
  for (int c0 = cbeg; c0 < cend; c0 += MY) {
    f32x16_t sacc;
#pragma unroll
    for (int x = 0; x < 16; ++x) sacc[x] = 0.f;
    for (int p = 0; p < npass; ++p) {
      __syncthreads();
      ystage_commit(st, Ys);
      if (!XRES) xpass_load(Xs, Q, r0, b, D, p * MW);
      __syncthreads();
      if (p + 1 < npass) ystage_fetch(st, K, c0, cend, D, (p + 1) * MW);
      else if (c0 + MY < cend) ystage_fetch(st, K, c0 + MY, cend, D, 0);
      logits_pass(sacc, XRES ? Xs + p * MW : Xs, XS, Ys, p * MW, D, wave, r, h);
    }
#pragma unroll
    for (int x = 0; x < 16; ++x) {
      const int col = c0 + wave * 32 + acc_row(x, h);
      if (col < cend && row < b) {
        const float v = sacc[x] * scale;
        if (logits_out) logits_out[((long)pair * b + row) * B + col] = v;
        if (excl0 >= 0 && col == excl0 + row) continue;               // self-pair removed from the softmax
        if (col != label0 + row && v > ll) cnt += 1.f;
        if (v > m) { ssum = ssum * __expf(m - v) + 1.f; m = v; } else ssum += __expf(v - m);
      }
    }
  }
  {  // the two half-wave lanes of a row, then the four waves
    const float m2 = __shfl_xor(m, 32, 64), s2 = __shfl_xor(ssum, 32, 64);
    lse_merge(m, ssum, m2, s2);
    cnt += __shfl_xor(cnt, 32, 64);
    if (h == 0) { red[(wave * MX + r) * 3 + 0] = m; red[(wave * MX + r) * 3 + 1] = ssum; red[(wave * MX + r) * 3 + 2] = cnt; }
  }
  __syncthreads();
  if (t < MX && r0 + t < b) {
    float mm = red[t * 3], ss = red[t * 3 + 1], cc = red[t * 3 + 2];
#pragma unroll
    for (int w = 1; w < 4; ++w) {
      lse_merge(mm, ss, red[(w * MX + t) * 3], red[(w * MX + t) * 3 + 1]);
      cc += red[(w * MX + t) * 3 + 2];
    }
    float* o = part + (((long)pair * nchunk + chunk) * b + r0 + t) * 4;
    o[0] = mm; o[1] = ss; o[2] = cc; o[3] = lab[t];
  }
}

// backward on the matrix pipe: per Y tile, 4 passes build the logits fragment, G goes to LDS TRANSPOSED ([y][x]: the lanes of
// a wave write consecutive x), 4 more passes over the same Y columns accumulate dX[32][D] += G . Y with wave w owning the
// w-th 32-column chunk of every pass (16 accumulator tiles of 32 x 32 over the 4 waves = 512 columns of dX per block).  D > 512
// (!XRES): blockIdx.x = (512-column chunk of dX, row block) -- every chunk's block rebuilds the logits over the whole of D and
// sweeps its own columns, so the cost is (D / 512 + 1) / 2 of the ideal; still the matrix pipe instead of the VALU tile loop.
template <bool XRES>
__global__ __launch_bounds__(256) void nce_bwd_mfma_kernel(PairTable pt, int mode, int b, int B, int D, const float* __restrict__ scale_p,
                                                           const float* __restrict__ row_lse, const float* __restrict__ g_row,
                                                           float* __restrict__ dscale, int chunk_cols) {
  const float scale = *scale_p;
  DH_DYN_LDS_A16(float, sm);
  const int XS = XRES ? D + 4 : MW + 4;
  constexpr int GS = MX + 4;
  float* Xs = sm;                        // XRES: [32][D + 4]; else [32][132] the columns of the current logits pass
  float* Ys = Xs + MX * XS;              // [128][132]
  float* Gt = Ys + MY * (MW + 4);        // [128][36]  G transposed
  float* red = Gt + MY * GS;             // [8]
  // mode 2: BOTH directions in one launch (b == B: the two have the same shape) -- blockIdx.y = direction * pairs + pair
  const int npairs = mode == 2 ? gridDim.y / 2 : gridDim.y;
  const int pair = blockIdx.y % npairs;
  if (mode == 2) mode = blockIdx.y / npairs;
  const int label0 = pt.label0[pair], excl0 = pt.excl0[pair];
  const float* X = mode == 0 ? pt.Q[pair] : pt.K[pair];
  const float* Y = mode == 0 ? pt.K[pair] : pt.Q[pair];
  float* dX = mode == 0 ? pt.dQ[pair] : pt.dK[pair];
  const int nx = mode == 0 ? b : B, ny_all = mode == 0 ? B : b;
  const int npass = (D + MW - 1) / MW;
  const int nrb = (nx + MX - 1) / MX;      // gridDim.x = nrb * (number of 512-column chunks of dX)
  const int dchunk = XRES ? 0 : blockIdx.x / nrb;
  const int gp0 = dchunk * 4, np_local = min(4, npass - gp0);       // this block's dX passes: gp0 .. gp0 + np_local
  if (dX == nullptr && !(mode == 0 && dscale != nullptr && dchunk == 0)) return;     // gradient not wanted for this operand
  const int ybeg = blockIdx.z * chunk_cols, ny = min(ny_all, ybeg + chunk_cols);
  const bool chunked = gridDim.z > 1;
  const int r0 = (XRES ? blockIdx.x : blockIdx.x % nrb) * MX;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, r = lane & 31, h = lane >> 5;
  const float* lse_p = row_lse + (long)pair * b;
  const float* g_p = g_row + (long)pair * b;
  if (XRES)
    for (int i = t; i < MX * (D / 4); i += 256) {
      const int rr = i / (D / 4), k4 = i % (D / 4);
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (r0 + rr < nx) v = *reinterpret_cast<const float4*>(X + (long)(r0 + rr) * D + k4 * 4);
      *reinterpret_cast<float4*>(Xs + rr * XS + k4 * 4) = v;
    }
  f32x16_t dacc[4];
#pragma unroll
  for (int p = 0; p < 4; ++p)
#pragma unroll
    for (int x = 0; x < 16; ++x) dacc[p][x] = 0.f;
  const int xr = r0 + r;                  // this lane's X row (logits fragment)
  float lse_x = 0.f, g_x = 0.f;
  if (mode == 0 && xr < nx) { lse_x = lse_p[xr]; g_x = g_p[xr]; }
  float ds_acc = 0.f;
  YStage st;
  ystage_fetch(st, Y, ybeg, ny, D, 0);
  for (int c0 = ybeg; c0 < ny; c0 += MY) {
    f32x16_t sacc;
#pragma unroll
    for (int x = 0; x < 16; ++x) sacc[x] = 0.f;
    for (int p = 0; p < npass; ++p) {
      __syncthreads();
      ystage_commit(st, Ys);
      if (!XRES) xpass_load(Xs, X, r0, nx, D, p * MW);
      __syncthreads();
      ystage_fetch(st, Y, c0, ny, D, (p + 1 < npass ? p + 1 : gp0) * MW);      // next logits pass, or the first pass of the dX sweep
      logits_pass(sacc, XRES ? Xs + p * MW : Xs, XS, Ys, p * MW, D, wave, r, h);
    }
    // G for this lane's X row against its 16 Y rows; stored transposed
#pragma unroll
    for (int x = 0; x < 16; ++x) {
      const int yl = wave * 32 + acc_row(x, h), yc = c0 + yl;
      float gval = 0.f;
      if (xr < nx && yc < ny) {
        const int i = mode == 0 ? xr : yc;   // Q row (local)
        const int j = mode == 0 ? yc : xr;   // K row (global)
        const float lse_i = mode == 0 ? lse_x : lse_p[i], g_i = mode == 0 ? g_x : g_p[i];
        const float pr = (excl0 >= 0 && j == excl0 + i) ? 0.f : __expf(sacc[x] * scale - lse_i);
        gval = g_i * (pr - (j == label0 + i ? 1.f : 0.f));
        ds_acc += gval * sacc[x];
      }
      Gt[yl * GS + r] = gval;
    }
    // dX[32][D] += G[32][128] . Y[128][D]: pass p re-stages columns 128p.., wave w multiplies its 32-column chunk
#pragma unroll
    for (int p = 0; p < 4; ++p) {          // unrolled: dacc[] must stay in registers (no dynamic indexing)
      if (p < np_local) {
        __syncthreads();                  // (p == 0: also publishes Gt)
        ystage_commit(st, Ys);
        __syncthreads();
        if (p + 1 < np_local) ystage_fetch(st, Y, c0, ny, D, (gp0 + p + 1) * MW);
        else if (c0 + MY < ny) ystage_fetch(st, Y, c0 + MY, ny, D, 0);
        if ((gp0 + p) * MW + wave * 32 < D) {
#pragma unroll 4
          for (int k0 = 0; k0 < MY; k0 += 8) {
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
              const float a = Gt[(k0 + 4 * h + jj) * GS + r];
              const float bb = Ys[(k0 + 4 * h + jj) * (MW + 4) + wave * 32 + r];
              dacc[p] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bb, dacc[p], 0, 0, 0);
            }
          }
        }
      }
    }
  }
  // dacc[p][x] of lane (r, h): X row acc_row(x, h), column 128 (gp0 + p) + 32*wave + r
  if (dX != nullptr) {
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const int col = (gp0 + p) * MW + wave * 32 + r;
      if (p < np_local && col < D) {
#pragma unroll
        for (int x = 0; x < 16; ++x) {
          const int rr = r0 + acc_row(x, h);
          if (rr < nx) {
            if (chunked) atomicAdd(dX + (long)rr * D + col, dacc[p][x] * scale);
            else dX[(long)rr * D + col] = dacc[p][x] * scale;
          }
        }
      }
    }
  }
  if (mode == 0 && dscale && dchunk == 0) {
    __syncthreads();
    float tot = block_sum256(ds_acc, red);
    if (t == 0) atomicAdd(dscale, tot);
  }
}

// column chunk for the matrix-pipe kernels: multiples of the 128-row Y tile, ~1 block per CU (their LDS footprint allows one)
static int nce_chunk_cols_mfma(int rows, int cols, int n_pairs) {
  const int tiles = dh_cdiv(rows, MX) * n_pairs;
  const int want = dh_cdiv(256, tiles);
  int chunk = dh_cdiv(dh_cdiv(cols, want), MY) * MY;
  if (chunk < MY) chunk = MY;
  return chunk;
}
// D <= 512: X rows resident in LDS; above (FILIP's embed_dim 768, the 3072 of the DeCLIP-88M configs): X staged per pass
static bool nce_mfma_ok(int D) { return D % 32 == 0 && D >= 32 && D <= 8192; }
static bool nce_xres(int D) { return D <= 512; }
static int nce_xs(int D) { return nce_xres(D) ? D + 4 : MW + 4; }
static size_t nce_fwd_mfma_lds(int D) { return (size_t)(MX * nce_xs(D) + MY * (MW + 4) + MX + 4 * MX * 3) * sizeof(float); }
static size_t nce_bwd_mfma_lds(int D) { return (size_t)(MX * nce_xs(D) + MY * (MW + 4) + MY * (MX + 4) + 8) * sizeof(float); }

// ---- plain row-wise softmax cross-entropy on materialised logits [rows, C] (fp32):
// used for logits handed in as tensors (loss_functions/loss.py:44-45) and for the MLM head
// (model/declip.py:326-334).  label < 0 (e.g. -100) rows are ignored (loss 0, zero grad).
__global__ __launch_bounds__(256) void ce_rows_fwd_kernel(const float* __restrict__ logits, long ld, const int64_t* __restrict__ labels,
                                                          int rows, int C, float* __restrict__ row_loss, float* __restrict__ row_lse,
                                                          float* __restrict__ correct1, float* __restrict__ correct5) {
  __shared__ float red[8];
  for (int row = blockIdx.x; row < rows; row += gridDim.x) {
    const float* lr = logits + (long)row * ld;
    const long lab = labels[row];
    float m = -INFINITY;
    for (int c = threadIdx.x; c < C; c += 256) m = fmaxf(m, lr[c]);
    m = block_max256(m, red);
    const float ll = (lab >= 0 && lab < C) ? lr[lab] : 0.f;
    float s = 0.f, cnt = 0.f;
    for (int c = threadIdx.x; c < C; c += 256) { float v = lr[c]; s += __expf(v - m); if (c != lab && v > ll) cnt += 1.f; }
    s = block_sum256(s, red);
    cnt = block_sum256(cnt, red);
    if (threadIdx.x == 0) {
      const float lse = m + __logf(s);
      const bool valid = lab >= 0 && lab < C;
      row_lse[row] = lse;
      row_loss[row] = valid ? lse - ll : 0.f;
      if (correct1) correct1[row] = (valid && cnt < 0.5f) ? 1.f : 0.f;
      if (correct5) correct5[row] = (valid && cnt < 4.5f) ? 1.f : 0.f;
    }
    __syncthreads();
  }
}
__global__ __launch_bounds__(256) void ce_rows_bwd_kernel(const float* __restrict__ logits, long ld, const int64_t* __restrict__ labels,
                                                          int rows, int C, const float* __restrict__ row_lse, const float* __restrict__ g_row,
                                                          float* __restrict__ dlogits, long ldd) {
  for (int row = blockIdx.x; row < rows; row += gridDim.x) {
    const long lab = labels[row];
    const bool valid = lab >= 0 && lab < C;
    const float g = valid ? g_row[row] : 0.f, lse = row_lse[row];
    for (int c = threadIdx.x; c < C; c += 256)
      dlogits[(long)row * ldd + c] = g * (__expf(logits[(long)row * ld + c] - lse) - (c == lab ? 1.f : 0.f));
  }
}

// padded / typed variant for the masked-LM head: dlogits [rows_pad][ldd] of type T, zero outside [rows) x [C)
template <typename T>
__global__ __launch_bounds__(256) void ce_rows_bwd_pad_kernel(const float* __restrict__ logits, long ld, const int64_t* __restrict__ labels,
                                                              int rows, int C, const float* __restrict__ row_lse,
                                                              const float* __restrict__ g_row, T* __restrict__ dlogits, long ldd,
                                                              int rows_pad, int C_pad) {
  for (int row = blockIdx.x; row < rows_pad; row += gridDim.x) {
    const bool live = row < rows;
    const long lab = live ? labels[row] : -1;
    const bool valid = lab >= 0 && lab < C;
    const float g = valid ? g_row[row] : 0.f, lse = live ? row_lse[row] : 0.f;
    for (int c = threadIdx.x; c < C_pad; c += 256) {
      float v = 0.f;
      if (live && c < C && valid) v = g * (__expf(logits[(long)row * ld + c] - lse) - (c == lab ? 1.f : 0.f));
      st<T>(dlogits + (long)row * ldd + c, v);
    }
  }
}

int fill_table(PairTable& pt, const dh_nce_pair* pairs, int n, int label0, const int* label0s, const int* excl0s) {
  for (int i = 0; i < DH_MAX_PAIRS; ++i) { pt.Q[i] = nullptr; pt.K[i] = nullptr; pt.dQ[i] = nullptr; pt.dK[i] = nullptr; pt.label0[i] = label0; pt.excl0[i] = -1; }
  for (int i = 0; i < n; ++i) {
    pt.Q[i] = pairs[i].Q; pt.K[i] = pairs[i].K; pt.dQ[i] = pairs[i].dQ; pt.dK[i] = pairs[i].dK;
    if (label0s) pt.label0[i] = label0s[i];
    if (excl0s) pt.excl0[i] = excl0s[i];
  }
  return 0;
}

}  // namespace

extern "C" int64_t dh_infonce_ws_bytes(int n_pairs, int b, int B) {
  const int nchunk = dh_cdiv(B, nce_chunk_cols(b, B, n_pairs, 32));
  return (int64_t)n_pairs * nchunk * b * 4 * sizeof(float);
}

extern "C" int dh_infonce_fwd(const dh_nce_pair* pairs, int n_pairs, int b, int B, int D, const float* scale, int label0,
                              const int* label0s, const int* excl0s, float* row_loss, float* row_lse, float* correct1,
                              float* correct5, float* logits_out, void* ws, int64_t ws_bytes, dh_stream_t stream) {
  hipStream_t st = (hipStream_t)stream;
  DH_REQUIRE(pairs && n_pairs >= 1 && n_pairs <= DH_MAX_PAIRS, "dh_infonce_fwd: 1..%d pairs", DH_MAX_PAIRS);
  DH_REQUIRE(b > 0 && B >= b && D > 0 && row_loss && row_lse && scale, "dh_infonce_fwd: bad args");
  DH_REQUIRE(ws && ws_bytes >= dh_infonce_ws_bytes(n_pairs, b, B), "dh_infonce_fwd: workspace too small");
  PairTable pt;
  fill_table(pt, pairs, n_pairs, label0, label0s, excl0s);
  for (int i = 0; i < n_pairs; ++i) DH_REQUIRE(pt.label0[i] >= 0 && pt.label0[i] + b <= B, "dh_infonce_fwd: labels out of range");
  for (int i = 0; i < n_pairs; ++i) DH_REQUIRE(pt.Q[i] && pt.K[i], "dh_infonce_fwd: null feature pointer");
  DH_REQUIRE(nce_mfma_ok(D) || D <= 1024, "dh_infonce_fwd: D=%d (multiples of 32 up to 8192, anything up to 1024)", D);
  constexpr int RT = 32;
  int nchunk;
  if (nce_mfma_ok(D)) {                  // matrix-pipe kernel (fewer, larger column chunks: the workspace bound still holds)
    const int chunk_cols = nce_chunk_cols_mfma(b, B, n_pairs);
    nchunk = dh_cdiv(B, chunk_cols);
    DH_REQUIRE(ws_bytes >= (int64_t)n_pairs * nchunk * b * 4 * (int64_t)sizeof(float), "dh_infonce_fwd: workspace too small");
    const size_t lds = nce_fwd_mfma_lds(D);
    auto kern = nce_xres(D) ? nce_fwd_mfma_kernel<true> : nce_fwd_mfma_kernel<false>;
    hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(kern, dim3(dh_cdiv(b, MX), n_pairs, nchunk), dim3(256), lds, st, pt, b, B, D, scale, (float*)ws, logits_out,
                       chunk_cols);
  } else {
    const int chunk_cols = nce_chunk_cols(b, B, n_pairs, RT);
    nchunk = dh_cdiv(B, chunk_cols);
    size_t lds = (size_t)(RT * (D + 1) + CT * (KC + 1) + RT) * sizeof(float);
    hipFuncSetAttribute((const void*)nce_fwd_kernel<RT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(nce_fwd_kernel<RT>, dim3(dh_cdiv(b, RT), n_pairs, nchunk), dim3(256), lds, st, pt, b, B, D, scale, label0,
                       (float*)ws, logits_out, chunk_cols);
  }
  DH_CHECK_LAUNCH();
  hipLaunchKernelGGL(nce_finalize_kernel, dim3(dh_cdiv(n_pairs * b, 256)), dim3(256), 0, st, (const float*)ws, n_pairs, nchunk, b,
                     row_loss, row_lse, correct1, correct5);
  DH_CHECK_LAUNCH();
  return DH_OK;
}

extern "C" int dh_infonce_bwd(const dh_nce_pair* pairs, int n_pairs, int b, int B, int D, const float* scale, int label0,
                              const int* label0s, const int* excl0s, const float* row_lse, const float* g_row, float* dscale,
                              dh_stream_t stream) {
  hipStream_t st = (hipStream_t)stream;
  DH_REQUIRE(pairs && n_pairs >= 1 && n_pairs <= DH_MAX_PAIRS, "dh_infonce_bwd: 1..%d pairs", DH_MAX_PAIRS);
  DH_REQUIRE(b > 0 && B >= b && D > 0 && row_lse && g_row && scale, "dh_infonce_bwd: bad args");
  PairTable pt;
  fill_table(pt, pairs, n_pairs, label0, label0s, excl0s);
  for (int i = 0; i < n_pairs; ++i) DH_REQUIRE(pt.Q[i] && pt.K[i], "dh_infonce_bwd: null feature pointer");  // dQ/dK may be NULL: skipped
  auto launch = [&](auto rt_tag, int mode) {
    constexpr int RT = decltype(rt_tag)::value;
    size_t lds = (size_t)(2 * RT * (D + 1) + CT * (KC + 1) + RT * (CT + 1) + 8) * sizeof(float);
    hipFuncSetAttribute((const void*)nce_bwd_kernel<RT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    const int nx = mode == 0 ? b : B, ny = mode == 0 ? B : b;
    const int chunk_cols = nce_chunk_cols(nx, ny, n_pairs, RT);
    const int nz = dh_cdiv(ny, chunk_cols);
    if (nz > 1)
      for (int i = 0; i < n_pairs; ++i) {
        void* dst = mode == 0 ? (void*)pt.dQ[i] : (void*)pt.dK[i];
        if (dst) DH_RT_NOTE(hipMemsetAsync(dst, 0, sizeof(float) * (size_t)nx * D, st), "dh_infonce_bwd: clearing a chunked gradient");
      }
    hipLaunchKernelGGL(nce_bwd_kernel<RT>, dim3(dh_cdiv(nx, RT), n_pairs, nz), dim3(256), lds, st, pt, mode, b, B, D, scale,
                       label0, row_lse, g_row, dscale, chunk_cols);
  };
  auto launch_mfma = [&](int mode) {       // mode 2: both directions in ONE launch (b == B only: same shape, twice the blocks)
    const size_t lds = nce_bwd_mfma_lds(D);
    auto kern = nce_xres(D) ? nce_bwd_mfma_kernel<true> : nce_bwd_mfma_kernel<false>;
    hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    const int nx = mode == 1 ? B : b, ny = mode == 1 ? b : B;
    const int ndc = nce_xres(D) ? 1 : dh_cdiv(dh_cdiv(D, MW), 4);          // 512-column chunks of dX, one block each
    const int ndir = mode == 2 ? 2 : 1;
    const int chunk_cols = nce_chunk_cols_mfma(nx * ndc, ny, n_pairs * ndir);
    const int nz = dh_cdiv(ny, chunk_cols);
    if (nz > 1)
      for (int i = 0; i < n_pairs; ++i) {
        void* dq = (void*)pt.dQ[i];
        void* dk = (void*)pt.dK[i];
        if (mode != 1 && dq) DH_RT_NOTE(hipMemsetAsync(dq, 0, sizeof(float) * (size_t)nx * D, st), "dh_infonce_bwd: clearing a chunked gradient");
        if (mode != 0 && dk) DH_RT_NOTE(hipMemsetAsync(dk, 0, sizeof(float) * (size_t)nx * D, st), "dh_infonce_bwd: clearing a chunked gradient");
      }
    hipLaunchKernelGGL(kern, dim3(dh_cdiv(nx, MX) * ndc, n_pairs * ndir, nz), dim3(256), lds, st, pt, mode, b, B, D, scale, row_lse, g_row,
                       dscale, chunk_cols);
  };
  static int merged = -1;      // DH_NCE_BWD_MERGED (read once): 1 (default) = one launch for both directions when b == B; 0 = two launches
  if (merged < 0) { const char* ev = getenv("DH_NCE_BWD_MERGED"); merged = ev ? (atoi(ev) != 0) : 1; }
  if (nce_mfma_ok(D)) {
    if (merged && b == B) {
      launch_mfma(2);
    } else {
      launch_mfma(0);
      launch_mfma(1);
    }
  } else if (D <= 512) {
    launch(std::integral_constant<int, 32>{}, 0);
    launch(std::integral_constant<int, 32>{}, 1);
  } else if (D <= 1024) {
    launch(std::integral_constant<int, 16>{}, 0);
    launch(std::integral_constant<int, 16>{}, 1);
  } else {
    DH_FAIL(DH_ERR_UNSUPPORTED, "dh_infonce_bwd: D=%d > 1024", D);
  }
  DH_HELPER_FAILED();
  DH_CHECK_LAUNCH();
  return DH_OK;
}

bool dh_ce_try_v4_fwd(const void* X, const void* W, const float* bias, const long long* labels, int n, int n_pad, int V, int K,
                      float* row_loss, float* row_lse, float* ws, int64_t ws_bytes, hipStream_t st);                      // gemm_v4.hip
bool dh_ce_try_v4_bwd(const void* X, const void* W, const float* bias, const long long* labels, const float* row_lse, const float* g_row,
                      int n, int n_pad, int V, int K, void* dl, int64_t ldd, hipStream_t st);

extern "C" int64_t dh_ce_fused_ws_bytes(int n_pad, int V) { return (int64_t)n_pad * (dh_cdiv(V, 256) * 2 + 1) * (int64_t)sizeof(float); }

// Linear(K -> V) + cross-entropy in one pass over the vocabulary, bf16 operands (the masked-LM head, declip.py:326-334): the
// logits exist only as tiles of the persistent MFMA GEMM; see gemm_v4.hip MODE_CE_FWD / MODE_CE_BWD.
extern "C" int dh_ce_fused_fwd(const void* X_bf16, const void* W_bf16, const float* bias, const int64_t* labels, int n, int n_pad, int V,
                               int K, float* row_loss, float* row_lse, void* ws, int64_t ws_bytes, dh_stream_t stream) {
  hipStream_t st = (hipStream_t)stream;
  DH_REQUIRE(X_bf16 && W_bf16 && labels && row_loss && row_lse && ws && n > 0 && n <= n_pad && V > 0 && K > 0, "dh_ce_fused_fwd: bad args");
  DH_REQUIRE(dh_ce_try_v4_fwd(X_bf16, W_bf16, bias, (const long long*)labels, n, n_pad, V, K, row_loss, row_lse, (float*)ws, ws_bytes, st),
             "dh_ce_fused_fwd: needs n_pad %% 256 == 0, V >= 256, K %% 64 == 0, K >= 128, 16-byte aligned operands, ws >= dh_ce_fused_ws_bytes (n_pad %d V %d K %d)",
             n_pad, V, K);
  DH_CHECK_LAUNCH();
  return DH_OK;
}
extern "C" int dh_ce_fused_bwd(const void* X_bf16, const void* W_bf16, const float* bias, const int64_t* labels, const float* row_lse,
                               const float* g_row, int n, int n_pad, int V, int K, void* dl_bf16, int64_t ldd, dh_stream_t stream) {
  hipStream_t st = (hipStream_t)stream;
  DH_REQUIRE(X_bf16 && W_bf16 && labels && row_lse && g_row && dl_bf16 && n > 0 && n <= n_pad && V > 0 && K > 0, "dh_ce_fused_bwd: bad args");
  DH_REQUIRE(dh_ce_try_v4_bwd(X_bf16, W_bf16, bias, (const long long*)labels, row_lse, g_row, n, n_pad, V, K, dl_bf16, ldd, st),
             "dh_ce_fused_bwd: needs n_pad %% 256 == 0, V >= 256, K %% 64 == 0, ldd %% 8 == 0, ldd >= V, 16-byte aligned operands (n_pad %d V %d K %d ldd %ld)",
             n_pad, V, K, (long)ldd);
  DH_CHECK_LAUNCH();
  return DH_OK;
}

extern "C" int dh_ce_rows_fwd(const float* logits, int64_t ld, const int64_t* labels, int rows, int C, float* row_loss,
                              float* row_lse, float* correct1, float* correct5, dh_stream_t stream) {
  hipStream_t st = (hipStream_t)stream;
  DH_REQUIRE(logits && labels && rows > 0 && C > 0 && row_loss && row_lse, "dh_ce_rows_fwd: bad args");
  hipLaunchKernelGGL(ce_rows_fwd_kernel, dim3(rows > 4096 ? 4096 : rows), dim3(256), 0, st, logits, (long)ld, labels, rows, C,
                     row_loss, row_lse, correct1, correct5);
  DH_CHECK_LAUNCH();
  return DH_OK;
}
extern "C" int dh_ce_rows_bwd(const float* logits, int64_t ld, const int64_t* labels, int rows, int C, const float* row_lse,
                              const float* g_row, float* dlogits, int64_t ldd, dh_stream_t stream) {
  hipStream_t st = (hipStream_t)stream;
  DH_REQUIRE(logits && labels && rows > 0 && C > 0 && row_lse && g_row && dlogits, "dh_ce_rows_bwd: bad args");
  hipLaunchKernelGGL(ce_rows_bwd_kernel, dim3(rows > 4096 ? 4096 : rows), dim3(256), 0, st, logits, (long)ld, labels, rows, C,
                     row_lse, g_row, dlogits, (long)ldd);
  DH_CHECK_LAUNCH();
  return DH_OK;
}

extern "C" int dh_ce_rows_bwd_padded(const float* logits, int64_t ld, const int64_t* labels, int rows, int C, const float* row_lse,
                                     const float* g_row, void* dlogits, int out_dtype, int64_t ldd, int rows_pad, int C_pad,
                                     dh_stream_t stream) {
  hipStream_t st = (hipStream_t)stream;
  DH_REQUIRE(logits && labels && rows > 0 && C > 0 && row_lse && g_row && dlogits && rows_pad >= rows && C_pad >= C && ldd >= C_pad,
             "dh_ce_rows_bwd_padded: bad args");
  dim3 grid(rows_pad > 4096 ? 4096 : rows_pad);
  if (out_dtype == DH_BF16)
    hipLaunchKernelGGL(ce_rows_bwd_pad_kernel<bf16_t>, grid, dim3(256), 0, st, logits, (long)ld, labels, rows, C, row_lse, g_row,
                       (bf16_t*)dlogits, (long)ldd, rows_pad, C_pad);
  else
    hipLaunchKernelGGL(ce_rows_bwd_pad_kernel<float>, grid, dim3(256), 0, st, logits, (long)ld, labels, rows, C, row_lse, g_row,
                       (float*)dlogits, (long)ldd, rows_pad, C_pad);
  DH_CHECK_LAUNCH();
  return DH_OK;
}
