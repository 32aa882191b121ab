// Token / patch embedding, pooling and L2-normalisation kernels (all HBM-bound, 16-byte vectors).
//   text : text_encoder/text_transformer.py:188-190,203     vision: image_encoder/visual_transformer.py:56-66
//   norm : model/clip.py:129-130
#include "dh_common.h"

namespace {

template <typename T>
__global__ __launch_bounds__(256) void text_embed_fwd_kernel(const int64_t* __restrict__ ids, const float* __restrict__ table,
                                                             const float* __restrict__ pos, T* __restrict__ x, int rows,
                                                             int L, int d) {
  const int nchunk = d >> 3;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < (long)rows * nchunk; i += (long)gridDim.x * 256) {
    const int row = (int)(i / nchunk), ch = (int)(i % nchunk);
    const long id = ids[row];
    float a[8], p[8];
    ld8(table + id * d + ch * 8, a);
    ld8(pos + (long)(row % L) * d + ch * 8, p);
#pragma unroll
    for (int k = 0; k < 8; ++k) a[k] += p[k];
    st8(x + (long)row * d + ch * 8, a);
  }
}

struct HotIds { long id[4]; int n; };

template <typename T>
__global__ __launch_bounds__(256) void text_embed_bwd_kernel(const int64_t* __restrict__ ids, const T* __restrict__ dx,
                                                             float* __restrict__ dtable, float* __restrict__ dpos,
                                                             int rows, int L, int d, HotIds hot) {
  // dtable: scatter-add rows (atomic) for all ids except the "hot" ones (pad / SOT / EOT occur in every
  // caption: thousands of rows would serialise on the same table row) -- those go through
  // embed_hot_reduce_kernel.  dpos is a batch reduction (batch_reduce_kernel).
  const int nchunk = d >> 3;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < (long)rows * nchunk; i += (long)gridDim.x * 256) {
    const int row = (int)(i / nchunk), ch = (int)(i % nchunk);
    const long id = ids[row];
    bool is_hot = false;
    for (int h = 0; h < hot.n; ++h) is_hot |= (id == hot.id[h]);
    if (is_hot) continue;
    float g[8];
    ld8(dx + (long)row * d + ch * 8, g);
#pragma unroll
    for (int k = 0; k < 8; ++k) atomicAdd(dtable + id * d + ch * 8 + k, g[k]);
  }
}

// dtable[hot.id[h], :] += sum over rows with ids[row] == hot.id[h] of dx[row, :]
// grid (ceil(d/64), row chunks, n hot); block 256 = 64 columns x 4 row lanes; one atomic per block column
template <typename T>
__global__ __launch_bounds__(256) void embed_hot_reduce_kernel(const int64_t* __restrict__ ids, const T* __restrict__ dx,
                                                               float* __restrict__ dtable, int rows, int d,
                                                               int rows_per_block, HotIds hot) {
  __shared__ float red[4][64];
  const long id = hot.id[blockIdx.z];
  const int c = blockIdx.x * 64 + (threadIdx.x & 63);
  const int rl = threadIdx.x >> 6;
  const int r0 = blockIdx.y * rows_per_block, r1 = min(rows, r0 + rows_per_block);
  float s = 0.f;
  if (c < d)
    for (int r = r0 + rl; r < r1; r += 4)
      if (ids[r] == id) s += ld<T>(dx + (long)r * d + c);
  red[rl][threadIdx.x & 63] = s;
  __syncthreads();
  if (rl == 0 && c < d) {
    s = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
    if (s != 0.f) atomicAdd(dtable + id * d + c, s);
  }
}

// out[l, :] += sum_b dx[b, l, :]  for l in [0, Lx); used for positional-embedding grads.
// grid (ceil(d/64), Lx, bsplit); block 256 = 64 columns x 4 batch lanes
template <typename T>
__global__ __launch_bounds__(256) void batch_reduce_kernel(const T* __restrict__ dx, float* __restrict__ out, int b,
                                                           int Lx, int d, int b_per_block) {
  __shared__ float red[4][64];
  const int c = blockIdx.x * 64 + (threadIdx.x & 63);
  const int l = blockIdx.y;
  const int bl = threadIdx.x >> 6;
  const int b0 = blockIdx.z * b_per_block, b1 = min(b, b0 + b_per_block);
  float s = 0.f;
  if (c < d)
    for (int bi = b0 + bl; bi < b1; bi += 4) s += ld<T>(dx + ((long)bi * Lx + l) * d + c);
  red[bl][threadIdx.x & 63] = s;
  __syncthreads();
  if (bl == 0 && c < d) atomicAdd(out + (long)l * d + c, red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x]);
}

template <typename T>
__global__ __launch_bounds__(256) void im2row_kernel(const float* __restrict__ img, int c_total, int c0, T* __restrict__ rows,
                                                     int b, int H, int W, int P) {
  // one task = 8 consecutive pw of one (b, gy, gx, c, ph)
  const int gh = H / P, gw = W / P;
  const int pc = P >> 3;
  const long ntask = (long)b * gh * gw * 3 * P * pc;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < ntask; i += (long)gridDim.x * 256) {
    long r = i;
    const int pw8 = (int)(r % pc); r /= pc;
    const int ph = (int)(r % P); r /= P;
    const int c = (int)(r % 3); r /= 3;
    const int gx = (int)(r % gw); r /= gw;
    const int gy = (int)(r % gh); r /= gh;
    const int bi = (int)r;
    const float* src = img + (((long)bi * c_total + c0 + c) * H + gy * P + ph) * W + gx * P + pw8 * 8;
    float v[8];
    ld8(src, v);
    T* dst = rows + (((long)bi * gh + gy) * gw + gx) * (3L * P * P) + ((long)c * P + ph) * P + pw8 * 8;
    st8(dst, v);
  }
}

template <typename T>
__global__ __launch_bounds__(256) void vit_assemble_fwd_kernel(const T* __restrict__ patches, const float* __restrict__ cls,
                                                               const float* __restrict__ pos, T* __restrict__ x, int b,
                                                               int np, int d) {
  const int nchunk = d >> 3, Lx = np + 1;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < (long)b * Lx * nchunk; i += (long)gridDim.x * 256) {
    const int ch = (int)(i % nchunk);
    const long row = i / nchunk;
    const int l = (int)(row % Lx), bi = (int)(row / Lx);
    float a[8], p[8];
    if (l == 0) ld8(cls + ch * 8, a); else ld8(patches + ((long)bi * np + l - 1) * d + ch * 8, a);
    ld8(pos + (long)l * d + ch * 8, p);
#pragma unroll
    for (int k = 0; k < 8; ++k) a[k] += p[k];
    st8(x + row * d + ch * 8, a);
  }
}

template <typename T>
__global__ __launch_bounds__(256) void pool_rows_fwd_kernel(const T* __restrict__ x, const int64_t* __restrict__ idx,
                                                            T* __restrict__ out, int b, int L, int d) {
  const int nchunk = d >> 3;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < (long)b * nchunk; i += (long)gridDim.x * 256) {
    const int bi = (int)(i / nchunk), ch = (int)(i % nchunk);
    const long l = idx ? idx[bi] : 0;
    float a[8];
    ld8(x + ((long)bi * L + l) * d + ch * 8, a);
    st8(out + (long)bi * d + ch * 8, a);
  }
}
template <typename T>
__global__ __launch_bounds__(256) void pool_rows_bwd_kernel(const T* __restrict__ dout, const int64_t* __restrict__ idx,
                                                            T* __restrict__ dx, int b, int L, int d) {
  const int nchunk = d >> 3;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < (long)b * L * nchunk; i += (long)gridDim.x * 256) {
    const int ch = (int)(i % nchunk);
    const long row = i / nchunk;
    const int l = (int)(row % L), bi = (int)(row / L);
    const long tgt = idx ? idx[bi] : 0;
    float a[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (l == tgt) ld8(dout + (long)bi * d + ch * 8, a);
    st8(dx + row * d + ch * 8, a);
  }
}

// y = x / (||x|| + eps): one wave per row
template <typename T>
__global__ __launch_bounds__(256) void l2norm_fwd_kernel(const T* __restrict__ x, float* __restrict__ y,
                                                         float* __restrict__ norm, int rows, int d, float eps) {
  const int lane = threadIdx.x & 63;
  for (int row = blockIdx.x * 4 + (threadIdx.x >> 6); row < rows; row += gridDim.x * 4) {
    float s = 0.f;
    for (int i = lane; i < d; i += 64) { float v = ld<T>(x + (long)row * d + i); s += v * v; }
    const float n = sqrtf(wave_sum(s));
    if (lane == 0 && norm) norm[row] = n;
    const float inv = 1.f / (n + eps);
    for (int i = lane; i < d; i += 64) y[(long)row * d + i] = ld<T>(x + (long)row * d + i) * inv;
  }
}
// y = x / (n + eps), n = ||x||:  dx = dy/(n+eps) - x * <dy,x> / (n * (n+eps)^2)
template <typename T>
__global__ __launch_bounds__(256) void l2norm_bwd_kernel(const T* __restrict__ x, const float* __restrict__ norm,
                                                         const float* __restrict__ dy, T* __restrict__ dx, int rows,
                                                         int d, float eps) {
  const int lane = threadIdx.x & 63;
  for (int row = blockIdx.x * 4 + (threadIdx.x >> 6); row < rows; row += gridDim.x * 4) {
    float s = 0.f;
    for (int i = lane; i < d; i += 64) s += dy[(long)row * d + i] * ld<T>(x + (long)row * d + i);
    s = wave_sum(s);
    const float n = norm[row];
    const float inv = 1.f / (n + eps);
    const float coef = s * inv * inv / fmaxf(n, 1e-30f);
    for (int i = lane; i < d; i += 64)
      st<T>(dx + (long)row * d + i, dy[(long)row * d + i] * inv - ld<T>(x + (long)row * d + i) * coef);
  }
}

__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                    float* __restrict__ v, bf16_t* __restrict__ pb, long n, float lr,
                                                    float b1, float b2, float eps, float wd, float bc1, float bc2_sqrt,
                                                    float gscale) {
  // torch.optim.AdamW (single-tensor path): p *= 1 - lr*wd; m,v update; p -= lr/bc1 * m / (sqrt(v)/sqrt(bc2) + eps)
  const long n4 = n >> 2;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
    float4 pv = reinterpret_cast<float4*>(p)[i], gv = reinterpret_cast<const float4*>(g)[i];
    float4 mv = reinterpret_cast<float4*>(m)[i], vv = reinterpret_cast<float4*>(v)[i];
    float pp[4] = {pv.x, pv.y, pv.z, pv.w}, gg[4] = {gv.x, gv.y, gv.z, gv.w};
    float mm[4] = {mv.x, mv.y, mv.z, mv.w}, vq[4] = {vv.x, vv.y, vv.z, vv.w};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float gk = gg[k] * gscale;
      pp[k] *= (1.f - lr * wd);
      mm[k] = b1 * mm[k] + (1.f - b1) * gk;
      vq[k] = b2 * vq[k] + (1.f - b2) * gk * gk;
      pp[k] -= (lr / bc1) * mm[k] / (sqrtf(vq[k]) / bc2_sqrt + eps);
    }
    reinterpret_cast<float4*>(p)[i] = make_float4(pp[0], pp[1], pp[2], pp[3]);
    reinterpret_cast<float4*>(m)[i] = make_float4(mm[0], mm[1], mm[2], mm[3]);
    reinterpret_cast<float4*>(v)[i] = make_float4(vq[0], vq[1], vq[2], vq[3]);
    if (pb) reinterpret_cast<uint2*>(pb)[i] = make_uint2(pack2bf(pp[0], pp[1]), pack2bf(pp[2], pp[3]));
  }
  // tail (n % 4)
  for (long i = n4 * 4 + (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const float gk = g[i] * gscale;
    float pp = p[i] * (1.f - lr * wd);
    const float mm = b1 * m[i] + (1.f - b1) * gk, vq = b2 * v[i] + (1.f - b2) * gk * gk;
    pp -= (lr / bc1) * mm / (sqrtf(vq) / bc2_sqrt + eps);
    p[i] = pp; m[i] = mm; v[i] = vq;
    if (pb) pb[i] = f2bf(pp);
  }
}

// Segmented variant: hyper-parameters (lr, weight decay) vary per contiguous segment of the flat
// buffer (param groups of utils/misc.py:267-412 `param_group_all`); ONE launch for the whole model.
// seg_start[nseg+1] (element offsets, multiples of 4, ascending), seg_lr[nseg], seg_wd[nseg].  seg_lr < 0 marks a segment the
// optimizer does not own or whose gradient is None (frozen / off-path parameters): values and moments untouched, as
// torch.optim.AdamW skips them -- an lr that a schedule legitimately drives to 0 still updates the moments.
__global__ __launch_bounds__(256) void adamw_seg_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                        float* __restrict__ v, bf16_t* __restrict__ pb, long n4,
                                                        const long* __restrict__ seg_start, const float* __restrict__ seg_lr,
                                                        const float* __restrict__ seg_wd, int nseg, float b1, float b2,
                                                        float eps, float bc1, float bc2_sqrt, float gscale) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
    const long e = i * 4;
    int lo = 0, hi = nseg - 1;
    while (lo < hi) { int mid = (lo + hi + 1) >> 1; if (seg_start[mid] <= e) lo = mid; else hi = mid - 1; }
    const float lr = seg_lr[lo], wd = seg_wd[lo];
    if (lr < 0.f) { if (pb) { float4 q = reinterpret_cast<float4*>(p)[i]; reinterpret_cast<uint2*>(pb)[i] = make_uint2(pack2bf(q.x, q.y), pack2bf(q.z, q.w)); } continue; }
    float4 pv = reinterpret_cast<float4*>(p)[i], gv = reinterpret_cast<const float4*>(g)[i];
    float4 mv = reinterpret_cast<float4*>(m)[i], vv = reinterpret_cast<float4*>(v)[i];
    float pp[4] = {pv.x, pv.y, pv.z, pv.w}, gg[4] = {gv.x, gv.y, gv.z, gv.w};
    float mm[4] = {mv.x, mv.y, mv.z, mv.w}, vq[4] = {vv.x, vv.y, vv.z, vv.w};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float gk = gg[k] * gscale;
      pp[k] *= (1.f - lr * wd);
      mm[k] = b1 * mm[k] + (1.f - b1) * gk;
      vq[k] = b2 * vq[k] + (1.f - b2) * gk * gk;
      pp[k] -= (lr / bc1) * mm[k] / (sqrtf(vq[k]) / bc2_sqrt + eps);
    }
    reinterpret_cast<float4*>(p)[i] = make_float4(pp[0], pp[1], pp[2], pp[3]);
    reinterpret_cast<float4*>(m)[i] = make_float4(mm[0], mm[1], mm[2], mm[3]);
    reinterpret_cast<float4*>(v)[i] = make_float4(vq[0], vq[1], vq[2], vq[3]);
    if (pb) reinterpret_cast<uint2*>(pb)[i] = make_uint2(pack2bf(pp[0], pp[1]), pack2bf(pp[2], pp[3]));
  }
}

template <typename TS, typename TD>
__global__ __launch_bounds__(256) void cast_kernel(const TS* __restrict__ s, TD* __restrict__ d, long n) {
  const long n8 = n >> 3;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n8; i += (long)gridDim.x * 256) {
    float v[8];
    ld8(s + i * 8, v);
    st8(d + i * 8, v);
  }
  for (long i = n8 * 8 + (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) st<TD>(d + i, ld<TS>(s + i));
}

int grid_for(long work_items) {
  long g = (work_items + 255) / 256;
  if (g > 4096) g = 4096;
  if (g < 1) g = 1;
  return (int)g;
}

}  // namespace

#define DH_DISPATCH_T(dtype, KERNEL, grid, block, lds, st, ...)                                   \
  do {                                                                                             \
    if ((dtype) == DH_BF16) hipLaunchKernelGGL(KERNEL<bf16_t>, grid, block, lds, st, __VA_ARGS__); \
    else hipLaunchKernelGGL(KERNEL<float>, grid, block, lds, st, __VA_ARGS__);                     \
  } while (0)

extern "C" int dh_text_embed_fwd(int dtype, const int64_t* ids, const float* table, const float* pos, void* x, int b,
                                 int L, int d, dh_stream_t stream) {
  hipStream_t st = (hipStream_t)stream;
  DH_REQUIRE(ids && table && pos && x && d % 8 == 0, "dh_text_embed_fwd: bad args (d %% 8 == 0 required)");
  const int rows = b * L;
  if (dtype == DH_BF16) hipLaunchKernelGGL(text_embed_fwd_kernel<bf16_t>, dim3(grid_for((long)rows * d / 8)), dim3(256), 0, st, ids, table, pos, (bf16_t*)x, rows, L, d);
  else hipLaunchKernelGGL(text_embed_fwd_kernel<float>, dim3(grid_for((long)rows * d / 8)), dim3(256), 0, st, ids, table, pos, (float*)x, rows, L, d);
  DH_CHECK_LAUNCH();
  return DH_OK;
}

// ---- packed (variable-length) captions: only the tokens up to and including <|endoftext|> of every caption are rows.
// Under the causal mask a token never sees a later one and only the EOT row is pooled (text_transformer.py:136-142,203), so
// the padding rows of the reference's [b, 77] layout are dead work: ~45 % of the text tower on the synthetic captions of
// SURVEY.md s8(d) (lengths U{6..75}), more on real ones.  x[r] = table[ids_p[r]] + pos[pos_idx[r]] for r < rows, 0 up to rows_pad.
template <typename T>
__global__ __launch_bounds__(256) void text_embed_packed_fwd_kernel(const int64_t* __restrict__ ids_p, const int* __restrict__ pos_idx,
                                                                    const float* __restrict__ table, const float* __restrict__ pos,
                                                                    T* __restrict__ x, int rows, int rows_pad, int d) {
  const int nchunk = d >> 3;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < (long)rows_pad * nchunk; i += (long)gridDim.x * 256) {
    const int row = (int)(i / nchunk), ch = (int)(i % nchunk);
    float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (row < rows && pos_idx[row] >= 0) {       // (pos_idx < 0: a padding row of a bookkeeping sized by rows_pad alone)
      float p[8];
      ld8(table + ids_p[row] * d + ch * 8, a);
      ld8(pos + (long)pos_idx[row] * d + ch * 8, p);
#pragma unroll
      for (int k = 0; k < 8; ++k) a[k] += p[k];
    }
    st8(x + (long)row * d + ch * 8, a);
  }
}

// dpos[p, :] += sum over the captions i longer than p of dx[cu[i] + p, :]   (no atomics: one block owns (p, 64 columns))
template <typename T>
__global__ __launch_bounds__(256) void packed_pos_grad_kernel(const T* __restrict__ dx, const int* __restrict__ cu, int b, int d,
                                                              float* __restrict__ dpos) {
  __shared__ float red[4][64];
  const int p = blockIdx.y;
  const int c = blockIdx.x * 64 + (threadIdx.x & 63);
  const int rl = threadIdx.x >> 6;
  float s = 0.f;
  if (c < d)
    for (int i = rl; i < b; i += 4) {
      const int r0 = cu[i], len = cu[i + 1] - r0;
      if (p < len) s += ld<T>(dx + (long)(r0 + p) * d + c);
    }
  red[rl][threadIdx.x & 63] = s;
  __syncthreads();
  if (rl == 0 && c < d) dpos[(long)p * d + c] += red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
}

extern "C" int dh_text_embed_packed_fwd(int dtype, const int64_t* ids_p, const int* pos_idx, const float* table, const float* pos,
                                        void* x, int rows, int rows_pad, int d, dh_stream_t stream) {
  hipStream_t st = (hipStream_t)stream;
  DH_REQUIRE(ids_p && pos_idx && table && pos && x && rows > 0 && rows_pad >= rows && d % 8 == 0, "dh_text_embed_packed_fwd: bad args");
  if (dtype == DH_BF16) hipLaunchKernelGGL(text_embed_packed_fwd_kernel<bf16_t>, dim3(grid_for((long)rows_pad * d / 8)), dim3(256), 0, st, ids_p, pos_idx, table, pos, (bf16_t*)x, rows, rows_pad, d);
  else if (dtype == DH_F32) hipLaunchKernelGGL(text_embed_packed_fwd_kernel<float>, dim3(grid_for((long)rows_pad * d / 8)), dim3(256), 0, st, ids_p, pos_idx, table, pos, (float*)x, rows, rows_pad, d);
  else DH_FAIL(DH_ERR_ARG, "dh_text_embed_packed_fwd: bad dtype");
  DH_CHECK_LAUNCH();
  return DH_OK;
}

// bf16, d % 8 == 0 (round 6): the same sum with 16-byte loads and the caption offsets in LDS -- one 1024-thread block owns (p, 512 columns):
// 64 column lanes x 8 columns, 16 row lanes x b/16 captions each, every load independent of the others (the scalar version walked 128
// captions per thread with a dependent cu -> dx chain of 2-byte loads: 112 us for 22 MB at the step's shape, on the tail of the text backward).
// No atomics, fixed summation order: deterministic like the version above.
__global__ __launch_bounds__(1024) void packed_pos_grad_vec_kernel(const bf16_t* __restrict__ dx, const int* __restrict__ cu, int b, int d,
                                                                   float* __restrict__ dpos) {
  __shared__ float red[16][64][8];
  __shared__ int scu[1025];
  const int p = blockIdx.y;
  const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
  const int c = blockIdx.x * 512 + cl * 8;
  float s[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) s[k] = 0.f;
  for (int i0 = 0; i0 < b; i0 += 1024) {
    const int nb = min(1024, b - i0);
    __syncthreads();
    for (int i = threadIdx.x; i <= nb; i += 1024) scu[i] = cu[i0 + i];
    __syncthreads();
    if (c < d) {
      for (int i = rl; i < nb; i += 16) {
        const int r0 = scu[i], len = scu[i + 1] - r0;
        if (p < len) {
          float v[8];
          ld8(dx + (long)(r0 + p) * d + c, v);
#pragma unroll
          for (int k = 0; k < 8; ++k) s[k] += v[k];
        }
      }
    }
  }
#pragma unroll
  for (int k = 0; k < 8; ++k) red[rl][cl][k] = s[k];
  __syncthreads();
  if (rl == 0 && c < d) {
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      float t = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) t += red[r][cl][k];
      dpos[(long)p * d + c + k] += t;
    }
  }
}

extern "C" int dh_packed_pos_grad(int dtype, const void* dx, const int* cu_seqlens, int b, int Lmax, int d, float* dpos, dh_stream_t stream) {
  hipStream_t st = (hipStream_t)stream;
  DH_REQUIRE(dx && cu_seqlens && dpos && b > 0 && Lmax > 0 && d > 0, "dh_packed_pos_grad: bad args");
  if (dtype == DH_BF16 && d % 8 == 0 && (((uintptr_t)dx) & 15) == 0) {
    hipLaunchKernelGGL(packed_pos_grad_vec_kernel, dim3(dh_cdiv(d, 512), Lmax), dim3(1024), 0, st, (const bf16_t*)dx, cu_seqlens, b, d, dpos);
    DH_CHECK_LAUNCH();
    return DH_OK;
  }
  dim3 grid(dh_cdiv(d, 64), Lmax);
  if (dtype == DH_BF16) hipLaunchKernelGGL(packed_pos_grad_kernel<bf16_t>, grid, dim3(256), 0, st, (const bf16_t*)dx, cu_seqlens, b, d, dpos);
  else if (dtype == DH_F32) hipLaunchKernelGGL(packed_pos_grad_kernel<float>, grid, dim3(256), 0, st, (const float*)dx, cu_seqlens, b, d, dpos);
  else DH_FAIL(DH_ERR_ARG, "dh_packed_pos_grad: bad dtype");
  DH_CHECK_LAUNCH();
  return DH_OK;
}

static int launch_batch_reduce(int dtype, const void* dx, float* out, int b, int Lx, int d, hipStream_t st) {
  int bsplit = b >= 64 ? 8 : 1;
  int bpb = dh_cdiv(b, bsplit);
  dim3 grid(dh_cdiv(d, 64), Lx, dh_cdiv(b, bpb));
  if (dtype == DH_BF16) hipLaunchKernelGGL(batch_reduce_kernel<bf16_t>, grid, dim3(256), 0, st, (const bf16_t*)dx, out, b, Lx, d, bpb);
  else hipLaunchKernelGGL(batch_reduce_kernel<float>, grid, dim3(256), 0, st, (const float*)dx, out, b, Lx, d, bpb);
  return 0;
}

extern "C" int dh_text_embed_bwd(int dtype, const int64_t* ids, const void* dx, float* dtable, float* dpos, int b,
                                 int L, int d, const int64_t* hot_ids, int n_hot, dh_stream_t stream) {
  hipStream_t st = (hipStream_t)stream;
  DH_REQUIRE(ids && dx && d % 8 == 0, "dh_text_embed_bwd: bad args");
  const int rows = b * L;
  if (dtable) {
    HotIds hot;
    hot.n = 0;
    for (int h = 0; h < n_hot && h < 4; ++h) hot.id[hot.n++] = hot_ids[h];
    if (dtype == DH_BF16) hipLaunchKernelGGL(text_embed_bwd_kernel<bf16_t>, dim3(grid_for((long)rows * d / 8)), dim3(256), 0, st, ids, (const bf16_t*)dx, dtable, dpos, rows, L, d, hot);
    else hipLaunchKernelGGL(text_embed_bwd_kernel<float>, dim3(grid_for((long)rows * d / 8)), dim3(256), 0, st, ids, (const float*)dx, dtable, dpos, rows, L, d, hot);
    DH_CHECK_LAUNCH();
    if (hot.n > 0) {
      int chunks = dh_cdiv(rows, 256);
      if (chunks > 64) chunks = 64;
      const int rpb = dh_cdiv(rows, chunks);
      dim3 grid(dh_cdiv(d, 64), dh_cdiv(rows, rpb), hot.n);
      if (dtype == DH_BF16) hipLaunchKernelGGL(embed_hot_reduce_kernel<bf16_t>, grid, dim3(256), 0, st, ids, (const bf16_t*)dx, dtable, rows, d, rpb, hot);
      else hipLaunchKernelGGL(embed_hot_reduce_kernel<float>, grid, dim3(256), 0, st, ids, (const float*)dx, dtable, rows, d, rpb, hot);
      DH_CHECK_LAUNCH();
    }
  }
  if (dpos) launch_batch_reduce(dtype, dx, dpos, b, L, d, st);
  DH_CHECK_LAUNCH();
  return DH_OK;
}

// ------------------------------------------------------------------------------------------------------------------
// Token-table gradient as a sort-by-id segmented reduction (nn.Embedding backward, text_transformer.py:188-190).
// The scatter-add above issues one fp32 atomic per gradient ELEMENT (rows x d: 11 M at b = 512, 0.3 ms) and serialises on
// the table rows of frequent tokens (real captions are Zipf-distributed; SOT / EOT occur in every caption).  Here the rows are
// counting-sorted by token id with integer atomics only (one per ROW), and one wave per 16 sorted rows adds runs of equal ids
// in registers: a run that lies inside its chunk is written with plain 16-byte read-modify-writes (nobody else touches that
// table row), only runs that cross a chunk boundary (long runs: frequent tokens) flush with float atomics, one per 16 rows
// instead of one per row.  ws: count[V] | cursor[V] | perm[rows] | sorted_id[rows] (int32).
__global__ __launch_bounds__(256) void embed_hist_kernel(const int64_t* __restrict__ ids, int rows, int V, int* __restrict__ count) {
  for (int r = blockIdx.x * 256 + threadIdx.x; r < rows; r += gridDim.x * 256) {
    const long id = ids[r];
    if (id >= 0 && id < V) atomicAdd(count + id, 1);
  }
}
// exclusive prefix sum of count[V] -> cursor[V]; one block of 1024 threads, thread t owns 64 consecutive bins of every round of
// 65536 bins (a CLIP vocabulary is one round), read as 16 int4 loads issued back to back (one memory latency, not 64)
__global__ __launch_bounds__(1024) void embed_scan_kernel(const int* __restrict__ count, int* __restrict__ cursor, int V) {
  __shared__ int part[1024];
  __shared__ int carry_s;
  const int t = threadIdx.x;
  if (t == 0) carry_s = 0;
  for (int base = 0; base < V; base += 65536) {
    const int lo = base + t * 64;
    int4 c[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const int i0 = lo + 4 * q;
      if (i0 + 3 < V) c[q] = *reinterpret_cast<const int4*>(count + i0);       // (V * 4 bytes after a 16-byte aligned base)
      else c[q] = make_int4(i0 < V ? count[i0] : 0, i0 + 1 < V ? count[i0 + 1] : 0, i0 + 2 < V ? count[i0 + 2] : 0, 0);
    }
    int s = 0;
#pragma unroll
    for (int q = 0; q < 16; ++q) s += c[q].x + c[q].y + c[q].z + c[q].w;
    part[t] = s;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {          // Hillis-Steele inclusive scan
      const int v = t >= off ? part[t - off] : 0;
      __syncthreads();
      part[t] += v;
      __syncthreads();
    }
    int run = carry_s + part[t] - s;                     // exclusive
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const int i0 = lo + 4 * q;
      int4 o;
      o.x = run; run += c[q].x;
      o.y = run; run += c[q].y;
      o.z = run; run += c[q].z;
      o.w = run; run += c[q].w;
      if (i0 + 3 < V) *reinterpret_cast<int4*>(cursor + i0) = o;
      else {
        if (i0 < V) cursor[i0] = o.x;
        if (i0 + 1 < V) cursor[i0 + 1] = o.y;
        if (i0 + 2 < V) cursor[i0 + 2] = o.z;
      }
    }
    __syncthreads();
    if (t == 1023) carry_s += part[1023];
    __syncthreads();
  }
}
__global__ __launch_bounds__(256) void embed_scatter_kernel(const int64_t* __restrict__ ids, int rows, int V, int* __restrict__ cursor,
                                                            int* __restrict__ perm, int* __restrict__ sid) {
  for (int r = blockIdx.x * 256 + threadIdx.x; r < rows; r += gridDim.x * 256) {
    const long id = ids[r];
    if (id >= 0 && id < V) {
      const int pos = atomicAdd(cursor + id, 1);
      perm[pos] = r;
      sid[pos] = (int)id;
    }
  }
}
constexpr int SEG_R = 16;      // sorted rows per wave
template <typename T>
__global__ __launch_bounds__(256) void embed_segreduce_kernel(const T* __restrict__ dx, const int* __restrict__ perm, const int* __restrict__ sid,
                                                              const int* __restrict__ count, int V, float* __restrict__ dtable, int d) {
  // the number of valid (in-range) rows is the sum of the histogram = the end of the last bin's run; cheaper: cursor after the
  // scatter holds run ENDS, so the total is cursor[V - 1] -- passed in as `count` = cursor
  const int n_valid = count[V - 1];
  const int wave = (blockIdx.x * 256 + threadIdx.x) >> 6, lane = threadIdx.x & 63;
  const int p0 = wave * SEG_R;
  if (p0 >= n_valid) return;
  const int n = min(SEG_R, n_valid - p0);
  // lanes 0..15: the chunk's rows; lane 16: the row before the chunk, lane 17: the row after it (run continues across the boundary?)
  int my_id = -1, my_row = 0;
  if (lane < n) { my_id = sid[p0 + lane]; my_row = perm[p0 + lane]; }
  else if (lane == 16 && p0 > 0) my_id = sid[p0 - 1];
  else if (lane == 17 && p0 + n < n_valid) my_id = sid[p0 + n];
  // wave-uniform copies in scalar registers (the column loop below runs with some lanes masked off: no cross-lane reads in it)
  const int id_before = __builtin_amdgcn_readlane(my_id, 16), id_after = __builtin_amdgcn_readlane(my_id, 17);
  int cid[SEG_R], crow[SEG_R];
#pragma unroll
  for (int i = 0; i < SEG_R; ++i) { cid[i] = __builtin_amdgcn_readlane(my_id, i); crow[i] = __builtin_amdgcn_readlane(my_row, i); }
  const int nch = d >> 3;
  for (int ch = lane; ch < nch; ch += 64) {
    float v[SEG_R][8];
#pragma unroll
    for (int i = 0; i < SEG_R; ++i)                    // all loads in flight before the first add
      if (i < n) ld8(dx + (long)crow[i] * d + ch * 8, v[i]);
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    int run_start = 0;
#pragma unroll
    for (int i = 0; i < SEG_R; ++i) {
      if (i < n) {
        const int id = cid[i];
        const int id_next = (i + 1 < SEG_R && i + 1 < n) ? cid[(i + 1) % SEG_R] : -2;
#pragma unroll
        for (int k = 0; k < 8; ++k) acc[k] += v[i][k];
        if (id_next != id) {                           // end of a run inside this chunk
          const bool shared = (run_start == 0 && id_before == id) || (i == n - 1 && id_after == id);
          float* dst = dtable + (long)id * d + ch * 8;
          if (shared) {
#pragma unroll
            for (int k = 0; k < 8; ++k)
              if (acc[k] != 0.f) atomicAdd(dst + k, acc[k]);
          } else {
            float4 a = *reinterpret_cast<float4*>(dst), c = *reinterpret_cast<float4*>(dst + 4);
            a.x += acc[0]; a.y += acc[1]; a.z += acc[2]; a.w += acc[3];
            c.x += acc[4]; c.y += acc[5]; c.z += acc[6]; c.w += acc[7];
            *reinterpret_cast<float4*>(dst) = a;
            *reinterpret_cast<float4*>(dst + 4) = c;
          }
#pragma unroll
          for (int k = 0; k < 8; ++k) acc[k] = 0.f;
          run_start = i + 1;
        }
      }
    }
  }
}

static inline int64_t vocab_slot(int vocab) { return ((int64_t)vocab + 3) / 4 * 4; }      // 16-byte aligned sections
extern "C" int64_t dh_embed_table_grad_ws_bytes(int rows, int vocab) { return (2 * vocab_slot(vocab) + 2 * (int64_t)rows) * 4; }

extern "C" int dh_embed_table_grad(int dtype, const int64_t* ids, const void* dx, float* dtable, int rows, int d, int vocab, void* ws,
                                   int64_t ws_bytes, dh_stream_t stream) {
  hipStream_t st = (hipStream_t)stream;
  DH_REQUIRE(ids && dx && dtable && rows > 0 && vocab > 0 && d > 0 && d % 8 == 0, "dh_embed_table_grad: bad args (d %% 8 == 0)");
  DH_REQUIRE(ws && ws_bytes >= dh_embed_table_grad_ws_bytes(rows, vocab), "dh_embed_table_grad: workspace too small");
  DH_REQUIRE(dtype == DH_BF16 || dtype == DH_F32, "dh_embed_table_grad: bad dtype");
  DH_REQUIRE(((uintptr_t)ws & 15) == 0, "dh_embed_table_grad: workspace not 16-byte aligned");
  int* count = (int*)ws;
  int* cursor = count + vocab_slot(vocab);
  int* perm = cursor + vocab_slot(vocab);
  int* sid = perm + rows;
  if (hipMemsetAsync(count, 0, sizeof(int) * (size_t)vocab, st) != hipSuccess) DH_FAIL(DH_ERR_LAUNCH, "dh_embed_table_grad: memset failed");
  const int g = grid_for(rows);
  hipLaunchKernelGGL(embed_hist_kernel, dim3(g), dim3(256), 0, st, ids, rows, vocab, count);
  hipLaunchKernelGGL(embed_scan_kernel, dim3(1), dim3(1024), 0, st, (const int*)count, cursor, vocab);
  hipLaunchKernelGGL(embed_scatter_kernel, dim3(g), dim3(256), 0, st, ids, rows, vocab, cursor, perm, sid);
  DH_CHECK_LAUNCH();
  const int waves = dh_cdiv(rows, SEG_R);
  const dim3 grid(dh_cdiv(waves, 4));
  if (dtype == DH_BF16)
    hipLaunchKernelGGL(embed_segreduce_kernel<bf16_t>, grid, dim3(256), 0, st, (const bf16_t*)dx, (const int*)perm, (const int*)sid,
                       (const int*)cursor, vocab, dtable, d);
  else
    hipLaunchKernelGGL(embed_segreduce_kernel<float>, grid, dim3(256), 0, st, (const float*)dx, (const int*)perm, (const int*)sid,
                       (const int*)cursor, vocab, dtable, d);
  DH_CHECK_LAUNCH();
  return DH_OK;
}

extern "C" int dh_im2row(int dtype, const float* images, int c_total, int c0, void* rows, int b, int H, int W, int P,
                         dh_stream_t stream) {
  hipStream_t st = (hipStream_t)stream;
  DH_REQUIRE(images && rows && P % 8 == 0 && H % P == 0 && W % P == 0 && W % 4 == 0, "dh_im2row: bad args");
  long ntask = (long)b * (H / P) * (W / P) * 3 * P * (P / 8);
  if (dtype == DH_BF16) hipLaunchKernelGGL(im2row_kernel<bf16_t>, dim3(grid_for(ntask)), dim3(256), 0, st, images, c_total, c0, (bf16_t*)rows, b, H, W, P);
  else hipLaunchKernelGGL(im2row_kernel<float>, dim3(grid_for(ntask)), dim3(256), 0, st, images, c_total, c0, (float*)rows, b, H, W, P);
  DH_CHECK_LAUNCH();
  return DH_OK;
}

// uint8 HWC images (what a decoder / the host hands over: 4x fewer PCIe and HBM bytes than fp32 CHW) -> the fp32 CHW
// batch contract of the towers: crop window, optional horizontal mirror, (x / 255 - mean[c]) / std[c]  (ToTensor + Normalize
// + crop + flip of the reference's pipelines, data/nvidia_dali_dataloader.py crop_mirror_normalize / data/transforms.py).
// One thread = 4 consecutive output pixels of a row, all 3 channels.
__global__ __launch_bounds__(256) void image_prep_u8_kernel(const uint8_t* __restrict__ src, int src_h, int src_w,
                                                            const int* __restrict__ crop_xy, const uint8_t* __restrict__ flip,
                                                            float m0, float m1, float m2, float s0, float s1, float s2,
                                                            float* __restrict__ dst, int c_total, int c0, int b, int H, int W) {
  const long n = (long)b * H * (W / 4);
  const float mean[3] = {m0, m1, m2}, inv[3] = {1.f / s0, 1.f / s1, 1.f / s2};
  for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (long)gridDim.x * blockDim.x) {
    const int xg = (int)(t % (W / 4));
    const int y = (int)((t / (W / 4)) % H);
    const int bi = (int)(t / ((long)(W / 4) * H));
    const int x0 = crop_xy ? crop_xy[2 * bi] : 0, y0 = crop_xy ? crop_xy[2 * bi + 1] : 0;
    const bool mir = flip && flip[bi];
    const uint8_t* row = src + ((long)bi * src_h + (y0 + y)) * src_w * 3;
    float o[3][4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int x = xg * 4 + i;
      const uint8_t* px = row + (long)(x0 + (mir ? W - 1 - x : x)) * 3;
#pragma unroll
      for (int c = 0; c < 3; ++c) o[c][i] = ((float)px[c] / 255.f - mean[c]) * inv[c];
    }
#pragma unroll
    for (int c = 0; c < 3; ++c)
      *reinterpret_cast<float4*>(dst + (((long)bi * c_total + c0 + c) * H + y) * W + xg * 4) = make_float4(o[c][0], o[c][1], o[c][2], o[c][3]);
  }
}

// RandomResizedCrop / Resize + CenterCrop of decoded uint8 HWC images on the GPU (the resize the reference's pipelines run on CPU
// workers or in DALI: data/imagenet_dataloader.py:36-47,105-111 -> torchvision TF.resized_crop / Resize on PIL images = PIL
// Image.crop + Image.resize(BILINEAR): an ANTIALIASED triangle filter -- support max(scale, 1) source pixels, taps clamped to the
// crop box, weights normalised per axis), fused with the optional mirror, ToTensor (/255) and Normalize.
// One thread = one output pixel, all 3 channels; x- and y-weights are recomputed on the fly (ALU is free next to the gather).
__device__ __forceinline__ float tri_w(int j, float center, float invscale) {
  const float t = ((float)j - center + 0.5f) * invscale;
  return fmaxf(0.f, 1.f - fabsf(t));
}

__global__ __launch_bounds__(256) void image_resized_crop_u8_kernel(const uint8_t* __restrict__ src, int src_h, int src_w,
                                                                    const int* __restrict__ params, const uint8_t* __restrict__ flip,
                                                                    float m0, float m1, float m2, float s0, float s1, float s2,
                                                                    float* __restrict__ dst, int c_total, int c0, int b, int H, int W,
                                                                    int round_u8) {
  const long n = (long)b * H * W;
  const float mean[3] = {m0, m1, m2}, inv[3] = {1.f / s0, 1.f / s1, 1.f / s2};
  for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (long)gridDim.x * blockDim.x) {
    const int x = (int)(t % W);
    const int y = (int)((t / W) % H);
    const int bi = (int)(t / ((long)W * H));
    const int* p = params + 8 * bi;
    const int bx = p[0], by = p[1], bw = p[2], bh = p[3], Wf = p[4], Hf = p[5], ox = p[6], oy = p[7];
    const bool mir = flip && flip[bi];
    // position of this output pixel inside the Wf x Hf image the crop box is resized to
    const int fx = ox + (mir ? W - 1 - x : x), fy = oy + y;
    // correctly rounded fp32 quotients (through fp64): this file is built with -ffast-math, whose v_rcp_f32 quotient is ~1 ulp
    // off, and (j - cx + 0.5) below cancels ~3 digits at source coordinates of several hundred pixels -- on the MI355X that was
    // 1.4e-4 of a normalised pixel against the oracle (torch's antialiased resize divides in IEEE fp32), i.e. rounding flips
    const float sx = (float)((double)bw / (double)Wf), sy = (float)((double)bh / (double)Hf);
    const float supx = sx >= 1.f ? sx : 1.f, supy = sy >= 1.f ? sy : 1.f;
    const float isx = sx >= 1.f ? (float)(1.0 / (double)sx) : 1.f, isy = sy >= 1.f ? (float)(1.0 / (double)sy) : 1.f;
    const float cx = ((float)fx + 0.5f) * sx, cy = ((float)fy + 0.5f) * sy;
    int x_lo = (int)(cx - supx + 0.5f), x_hi = (int)(cx + supx + 0.5f);
    int y_lo = (int)(cy - supy + 0.5f), y_hi = (int)(cy + supy + 0.5f);
    if (x_lo < 0) x_lo = 0;
    if (y_lo < 0) y_lo = 0;
    if (x_hi > bw) x_hi = bw;
    if (y_hi > bh) y_hi = bh;
    float wsx = 0.f, wsy = 0.f;
    for (int j = x_lo; j < x_hi; ++j) wsx += tri_w(j, cx, isx);
    for (int i = y_lo; i < y_hi; ++i) wsy += tri_w(i, cy, isy);
    float acc[3] = {0.f, 0.f, 0.f};
    const uint8_t* img = src + (long)bi * src_h * src_w * 3;
    for (int i = y_lo; i < y_hi; ++i) {
      const float wy = tri_w(i, cy, isy);
      const uint8_t* row = img + ((long)(by + i) * src_w + bx) * 3;
      float r[3] = {0.f, 0.f, 0.f};
      for (int j = x_lo; j < x_hi; ++j) {
        const float wx = tri_w(j, cx, isx);
        r[0] += wx * (float)row[3 * j]; r[1] += wx * (float)row[3 * j + 1]; r[2] += wx * (float)row[3 * j + 2];
      }
      acc[0] += wy * r[0]; acc[1] += wy * r[1]; acc[2] += wy * r[2];
    }
    const double norm = 1.0 / ((double)wsx * (double)wsy);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      float v = (float)((double)acc[c] * norm);
      if (round_u8) v = fminf(fmaxf(floorf(v + 0.5f), 0.f), 255.f);   // the PIL image between resize and ToTensor is uint8
      dst[(((long)bi * c_total + c0 + c) * H + y) * W + x] = (v / 255.f - mean[c]) * inv[c];
    }
  }
}

extern "C" int dh_image_resized_crop_u8(const uint8_t* src, int b, int src_h, int src_w, const int* params_dev, const uint8_t* flip_dev,
                                        const float* mean3, const float* std3, float* dst, int c_total, int c0, int H, int W,
                                        int round_u8, dh_stream_t stream) {
  hipStream_t st = (hipStream_t)stream;
  DH_REQUIRE(src && params_dev && dst && mean3 && std3 && b > 0 && H > 0 && W > 0 && src_h > 0 && src_w > 0 && c0 >= 0 && c0 + 3 <= c_total,
             "dh_image_resized_crop_u8: bad args");
  hipLaunchKernelGGL(image_resized_crop_u8_kernel, dim3(grid_for((long)b * H * W)), dim3(256), 0, st, src, src_h, src_w, params_dev, flip_dev,
                     mean3[0], mean3[1], mean3[2], std3[0], std3[1], std3[2], dst, c_total, c0, b, H, W, round_u8);
  DH_CHECK_LAUNCH();
  return DH_OK;
}

extern "C" int dh_image_prep_u8(const uint8_t* src, int b, int src_h, int src_w, const int* crop_xy_dev, const uint8_t* flip_dev,
                                const float* mean3, const float* std3, float* dst, int c_total, int c0, int H, int W,
                                dh_stream_t stream) {
  hipStream_t st = (hipStream_t)stream;
  DH_REQUIRE(src && dst && mean3 && std3 && b > 0 && H > 0 && W > 0 && W % 4 == 0 && src_h >= H && src_w >= W && c0 >= 0 && c0 + 3 <= c_total,
             "dh_image_prep_u8: bad args");
  DH_REQUIRE(crop_xy_dev || (src_h == H && src_w == W), "dh_image_prep_u8: a crop table is required when the source is larger than the output");
  DH_REQUIRE(std3[0] != 0.f && std3[1] != 0.f && std3[2] != 0.f, "dh_image_prep_u8: zero std");
  const long n = (long)b * H * (W / 4);
  hipLaunchKernelGGL(image_prep_u8_kernel, dim3(grid_for(n)), dim3(256), 0, st, src, src_h, src_w, crop_xy_dev, flip_dev, mean3[0], mean3[1],
                     mean3[2], std3[0], std3[1], std3[2], dst, c_total, c0, b, H, W);
  DH_CHECK_LAUNCH();
  return DH_OK;
}

extern "C" int dh_vit_assemble_fwd(int dtype, const void* patches, const float* cls, const float* pos, void* x, int b,
                                   int np, int d, dh_stream_t stream) {
  hipStream_t st = (hipStream_t)stream;
  DH_REQUIRE(patches && cls && pos && x && d % 8 == 0, "dh_vit_assemble_fwd: bad args");
  long n = (long)b * (np + 1) * d / 8;
  if (dtype == DH_BF16) hipLaunchKernelGGL(vit_assemble_fwd_kernel<bf16_t>, dim3(grid_for(n)), dim3(256), 0, st, (const bf16_t*)patches, cls, pos, (bf16_t*)x, b, np, d);
  else hipLaunchKernelGGL(vit_assemble_fwd_kernel<float>, dim3(grid_for(n)), dim3(256), 0, st, (const float*)patches, cls, pos, (float*)x, b, np, d);
  DH_CHECK_LAUNCH();
  return DH_OK;
}

extern "C" int dh_vit_assemble_bwd(int dtype, const void* dx, float* dcls, float* dpos, int b, int np, int d,
                                   dh_stream_t stream) {
  hipStream_t st = (hipStream_t)stream;
  DH_REQUIRE(dx && dpos, "dh_vit_assemble_bwd: bad args");
  // dpos[l] += sum_b dx[b,l]; dcls += sum_b dx[b,0]
  launch_batch_reduce(dtype, dx, dpos, b, np + 1, d, st);
  DH_CHECK_LAUNCH();
  if (dcls) {
    // the cls row: same reduction restricted to l = 0 (row stride (np+1)*d)
    int bsplit = b >= 64 ? 8 : 1;
    int bpb = dh_cdiv(b, bsplit);
    dim3 grid(dh_cdiv(d, 64), 1, dh_cdiv(b, bpb));
    if (dtype == DH_BF16) hipLaunchKernelGGL(batch_reduce_kernel<bf16_t>, grid, dim3(256), 0, st, (const bf16_t*)dx, dcls, b, np + 1, d, bpb);
    else hipLaunchKernelGGL(batch_reduce_kernel<float>, grid, dim3(256), 0, st, (const float*)dx, dcls, b, np + 1, d, bpb);
    DH_CHECK_LAUNCH();
  }
  return DH_OK;
}

extern "C" int dh_pool_rows_fwd(int dtype, const void* x, const int64_t* idx, void* out, int b, int L, int d,
                                dh_stream_t stream) {
  hipStream_t st = (hipStream_t)stream;
  DH_REQUIRE(x && out && d % 8 == 0, "dh_pool_rows_fwd: bad args");
  if (dtype == DH_BF16) hipLaunchKernelGGL(pool_rows_fwd_kernel<bf16_t>, dim3(grid_for((long)b * d / 8)), dim3(256), 0, st, (const bf16_t*)x, idx, (bf16_t*)out, b, L, d);
  else hipLaunchKernelGGL(pool_rows_fwd_kernel<float>, dim3(grid_for((long)b * d / 8)), dim3(256), 0, st, (const float*)x, idx, (float*)out, b, L, d);
  DH_CHECK_LAUNCH();
  return DH_OK;
}
extern "C" int dh_pool_rows_bwd(int dtype, const void* dout, const int64_t* idx, void* dx, int b, int L, int d,
                                dh_stream_t stream) {
  hipStream_t st = (hipStream_t)stream;
  DH_REQUIRE(dout && dx && d % 8 == 0, "dh_pool_rows_bwd: bad args");
  if (dtype == DH_BF16) hipLaunchKernelGGL(pool_rows_bwd_kernel<bf16_t>, dim3(grid_for((long)b * L * d / 8)), dim3(256), 0, st, (const bf16_t*)dout, idx, (bf16_t*)dx, b, L, d);
  else hipLaunchKernelGGL(pool_rows_bwd_kernel<float>, dim3(grid_for((long)b * L * d / 8)), dim3(256), 0, st, (const float*)dout, idx, (float*)dx, b, L, d);
  DH_CHECK_LAUNCH();
  return DH_OK;
}

extern "C" int dh_l2norm_fwd(int dtype, const void* x, float* y, float* norm, int rows, int d, float eps,
                             dh_stream_t stream) {
  hipStream_t st = (hipStream_t)stream;
  DH_REQUIRE(x && y && rows > 0, "dh_l2norm_fwd: bad args");
  dim3 grid(dh_cdiv(rows, 4) > 1024 ? 1024 : dh_cdiv(rows, 4));
  if (dtype == DH_BF16) hipLaunchKernelGGL(l2norm_fwd_kernel<bf16_t>, grid, dim3(256), 0, st, (const bf16_t*)x, y, norm, rows, d, eps);
  else hipLaunchKernelGGL(l2norm_fwd_kernel<float>, grid, dim3(256), 0, st, (const float*)x, y, norm, rows, d, eps);
  DH_CHECK_LAUNCH();
  return DH_OK;
}
extern "C" int dh_l2norm_bwd(int dtype, const void* x, const float* norm, const float* dy, void* dx, int rows, int d,
                             float eps, dh_stream_t stream) {
  hipStream_t st = (hipStream_t)stream;
  DH_REQUIRE(x && norm && dy && dx && rows > 0, "dh_l2norm_bwd: bad args");
  dim3 grid(dh_cdiv(rows, 4) > 1024 ? 1024 : dh_cdiv(rows, 4));
  if (dtype == DH_BF16) hipLaunchKernelGGL(l2norm_bwd_kernel<bf16_t>, grid, dim3(256), 0, st, (const bf16_t*)x, norm, dy, (bf16_t*)dx, rows, d, eps);
  else hipLaunchKernelGGL(l2norm_bwd_kernel<float>, grid, dim3(256), 0, st, (const float*)x, norm, dy, (float*)dx, rows, d, eps);
  DH_CHECK_LAUNCH();
  return DH_OK;
}

extern "C" int dh_adamw(float* p, const float* g, float* m, float* v, void* p_bf16, int64_t n, float lr, float beta1,
                        float beta2, float eps, float weight_decay, int step, float grad_scale, dh_stream_t stream) {
  hipStream_t st = (hipStream_t)stream;
  DH_REQUIRE(p && g && m && v && n > 0 && step >= 1, "dh_adamw: bad args");
  DH_REQUIRE((((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) == 0, "dh_adamw: 16-byte alignment required");
  const float bc1 = 1.f - powf(beta1, (float)step);
  const float bc2 = 1.f - powf(beta2, (float)step);
  hipLaunchKernelGGL(adamw_kernel, dim3(grid_for(n / 4 + 1)), dim3(256), 0, st, p, g, m, v, (bf16_t*)p_bf16, (long)n, lr,
                     beta1, beta2, eps, weight_decay, bc1, sqrtf(bc2), grad_scale);
  DH_CHECK_LAUNCH();
  return DH_OK;
}

extern "C" int dh_adamw_segmented(float* p, const float* g, float* m, float* v, void* p_bf16, int64_t n,
                                  const int64_t* seg_start_dev, const float* seg_lr_dev, const float* seg_wd_dev, int nseg,
                                  float beta1, float beta2, float eps, int step, float grad_scale, dh_stream_t stream) {
  hipStream_t st = (hipStream_t)stream;
  DH_REQUIRE(p && g && m && v && n > 0 && n % 4 == 0 && step >= 1 && nseg >= 1 && seg_start_dev && seg_lr_dev && seg_wd_dev,
             "dh_adamw_segmented: bad args (n must be a multiple of 4)");
  const float bc1 = 1.f - powf(beta1, (float)step);
  const float bc2 = 1.f - powf(beta2, (float)step);
  hipLaunchKernelGGL(adamw_seg_kernel, dim3(grid_for(n / 4)), dim3(256), 0, st, p, g, m, v, (bf16_t*)p_bf16, (long)(n / 4),
                     (const long*)seg_start_dev, seg_lr_dev, seg_wd_dev, nseg, beta1, beta2, eps, bc1, sqrtf(bc2), grad_scale);
  DH_CHECK_LAUNCH();
  return DH_OK;
}

extern "C" int dh_cast(int src_dtype, const void* src, int dst_dtype, void* dst, int64_t n, dh_stream_t stream) {
  hipStream_t st = (hipStream_t)stream;
  DH_REQUIRE(src && dst && n > 0, "dh_cast: bad args");
  dim3 grid(grid_for(n / 8 + 1));
  if (src_dtype == DH_F32 && dst_dtype == DH_BF16) hipLaunchKernelGGL((cast_kernel<float, bf16_t>), grid, dim3(256), 0, st, (const float*)src, (bf16_t*)dst, (long)n);
  else if (src_dtype == DH_BF16 && dst_dtype == DH_F32) hipLaunchKernelGGL((cast_kernel<bf16_t, float>), grid, dim3(256), 0, st, (const bf16_t*)src, (float*)dst, (long)n);
  else if (src_dtype == DH_F32 && dst_dtype == DH_F32) hipLaunchKernelGGL((cast_kernel<float, float>), grid, dim3(256), 0, st, (const float*)src, (float*)dst, (long)n);
  else hipLaunchKernelGGL((cast_kernel<bf16_t, bf16_t>), grid, dim3(256), 0, st, (const bf16_t*)src, (bf16_t*)dst, (long)n);
  DH_CHECK_LAUNCH();
  return DH_OK;
}
