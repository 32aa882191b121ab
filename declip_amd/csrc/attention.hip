// Multi-head self-attention core for short sequences (L <= 128: ViT-B/32 has 50 tokens, the
// text tower 77) -- nn.MultiheadAttention(x,x,x,attn_mask) in base_transformer.py:33,45-48.
//
// One workgroup per (batch, head); the whole sequence lives in LDS, the [L,L] score matrix
// never touches HBM.  bf16 path: v_mfma_f32_16x16x32_bf16, one wave per 16-query block.
//   fwd:  S^T = K Q^T (so every lane owns ONE query column: softmax is in-lane + 2 shuffles),
//         P written to LDS as the transpose of the C fragment (8-byte stores), O = P V with the V
//         operand fetched by the hardware transpose read (ds_read_b64_tr_b16) from the row-major tile.
//   bwd:  P recomputed from q,k,lse; dP = dO V^T; dS = P o (dP - rowsum(dO o O)) / sqrt(hd);
//         dV = P^T dO, dK = dS^T Q, dQ = dS K  (all five products on MFMA).
// fp32 path (validation precision): same math, scalar FMA, any head dim <= 64.
#include "dh_common.h"
#include <stdlib.h>

namespace {

constexpr int HD = 64;
constexpr int RS = HD + 8;  // row stride (elements) of row-major [L][64] bf16 tiles: 144 B, 16-B aligned
#ifndef ATTN_LINE_OUT
#define ATTN_LINE_OUT 1     // outputs leave as whole 128-byte rows through LDS (0: 8-byte pieces straight from the MFMA fragments)
#endif

__device__ __forceinline__ bf16x8_t lds_frag(const bf16_t* p) { return *reinterpret_cast<const bf16x8_t*>(p); }
__device__ __forceinline__ f32x4_t mfma16(bf16x8_t a, bf16x8_t b, f32x4_t c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}
// reduce across the 4 lane groups (lanes with equal lane&15)
__device__ __forceinline__ float quad_max(float v) {
  v = fmaxf(v, __shfl_xor(v, 16, 64));
  return fmaxf(v, __shfl_xor(v, 32, 64));
}
__device__ __forceinline__ float quad_sum(float v) {
  v += __shfl_xor(v, 16, 64);
  return v + __shfl_xor(v, 32, 64);
}

typedef __attribute__((ext_vector_type(4))) short s16x4_t;

// stage rows [0,L) x 64 of a q/k/v/do block (global row stride gs elements) into a row-major
// LDS tile rm[Lrows][RS] (rows >= L zero-filled); 16-byte global loads.
__device__ __forceinline__ void stage_tile(const bf16_t* __restrict__ g, long gs, int L, int Lrows, bf16_t* rm, int tid,
                                           int nthr) {
  for (int task = tid; task < Lrows * 8; task += nthr) {
    const int r = task >> 3, c = task & 7;
    uint4 v = make_uint4(0, 0, 0, 0);
    if (r < L) v = *reinterpret_cast<const uint4*>(g + (long)r * gs + c * 8);
    *reinterpret_cast<uint4*>(rm + r * RS + c * 8) = v;
  }
}

// The MFMA kernels are PERSISTENT: a workgroup walks (batch, head) pairs and fetches the tiles of the NEXT pair into registers
// (2 x 16 bytes per thread and tile) while it multiplies the current one -- with 2-4 workgroups per CU (LDS-bound) the
// one-shot version spent about half of its time waiting for its own loads.
struct TileRegs { uint4 v[2]; };
__device__ __forceinline__ TileRegs tile_load(const bf16_t* __restrict__ g, long gs, int L, int tid, int nthr) {
  TileRegs r;
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int task = tid + k * nthr, row = task >> 3, c = task & 7;
    r.v[k] = make_uint4(0, 0, 0, 0);
    if (row < L) r.v[k] = *reinterpret_cast<const uint4*>(g + (long)row * gs + c * 8);
  }
  return r;
}
__device__ __forceinline__ void tile_store(const TileRegs& r, bf16_t* rm, int tid, int nthr) {
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int task = tid + k * nthr, row = task >> 3, c = task & 7;
    *reinterpret_cast<uint4*>(rm + row * RS + c * 8) = r.v[k];
  }
}

// MFMA 16x16x32 operand whose contraction index is the ROW of a row-major LDS tile (stride ld
// elements): lane (t = lane&15, g = lane>>4) needs k = k0 + 8g .. +7 for column c0 + t.  Two hardware
// transpose reads (ds_read_b64_tr_b16; semantics measured in profiles/r01_hw_probe_trread_glds.txt):
// lane t supplies the address of row k + (t>>2), columns c0 + 4*(t&3) .. +3 and receives column c0 + t.
__device__ __forceinline__ bf16x8_t frag_tr(const bf16_t* tile, int ld, int k0, int c0, int lane) {
  const int t = lane & 15;
  const bf16_t* p = tile + (k0 + 8 * (lane >> 4) + (t >> 2)) * ld + c0 + 4 * (t & 3);
  s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(p));
  s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(p + 4 * ld));
  union { struct { s16x4_t a, b; } s; bf16x8_t v; } u;
  u.s.a = lo; u.s.b = hi;
  return u.v;
}
// zero the k-slots beyond the contraction length (last 32-wide step when L16 % 32 == 16): both
// operands are cleared so that stale LDS bits (possibly NaN patterns) never reach the MFMA.
__device__ __forceinline__ bf16x8_t kmask(bf16x8_t f, bool dead) {
  union { bf16x8_t v; uint4 u; } x;
  x.v = f;
  if (dead) x.u = make_uint4(0, 0, 0, 0);
  return x.v;
}

// ------------------------------------------------------------------------------------------
// forward, bf16 / MFMA.  blockDim = 64 * NKB; wave w owns queries [16w, 16w+16)
// ------------------------------------------------------------------------------------------
// VL: variable-length (packed) sequences -- pair (bi, h) owns rows cu[bi] .. cu[bi+1] of qkv / out instead of bi*L .. bi*L + L
// (lse stays [b][heads][L]); the dense instantiation is the code it was before the template parameter existed.
template <int NKB, bool VL, int PF = 1>  // NKB: number of 16-key blocks (L16/16), compile-time so scores stay in registers; PF: pairs prefetched ahead
__global__ void attn_fwd_mfma_kernel(const bf16_t* __restrict__ qkv, bf16_t* __restrict__ out, float* __restrict__ lse,
                                     int L, int heads, int causal, float scale, int nbh, const int* __restrict__ cu,
                                     int tail0, int tail1, const int* __restrict__ seq_list, const int* __restrict__ seq_range) {
  // seq_list / seq_range (packed sequences only): this launch takes the sequences seq_list[s0 .. s0 + n), (s0, n) = seq_range[0..1] read
  // HERE -- length-bucketed launches (dh_attn_bucketed_*): the short captions on an instantiation with fewer key blocks, without
  // a host-side count in the launch
  const int nb_total = nbh / heads;
  const int s0 = (VL && seq_list) ? seq_range[0] : 0;
  if (VL && seq_list) nbh = seq_range[1] * heads;
#define SEQ(i) ((VL && seq_list) ? seq_list[s0 + (i)] : (i))
  constexpr int L16 = NKB * 16;
  constexpr int NKS = (L16 + 31) / 32;          // 32-wide k-steps over the keys
  constexpr bool KTAIL = (L16 % 32) != 0;
  constexpr int TS = L16 + 8;
  DH_DYN_LDS_A16(unsigned char, smem_raw);
  bf16_t* Qs = reinterpret_cast<bf16_t*>(smem_raw);  // [L16][RS]
  bf16_t* Ks = Qs + L16 * RS;                        // [L16][RS]
  bf16_t* Vs = Ks + L16 * RS;                        // [L16][RS]
  bf16_t* Ps = Vs + L16 * RS;                        // [L16][TS]  (+ 32 elements slack for the masked tail read)

  const int d_model = heads * HD;
  const long gs = 3L * d_model;
  const int tid = threadIdx.x, nthr = blockDim.x;
  const int lane = tid & 63, wave = tid >> 6;
  // PF pairs ahead: the tiles of pair bh + PF * grid are requested while pair bh is multiplied (registers, 24 per stage).  PF = 1
  // (rounds 1-5) left the packed text tower latency-bound: 2-4 pairs per workgroup, each iteration waiting most of a memory round
  // trip for tiles requested one short iteration earlier (2.1-2.4 TB/s against 4.1 of the image tower, whose iterations are longer).
  // The loop is unrolled by PF so that every stage keeps its own registers (a rotating copy would have to wait for the loads).
  TileRegs rq[PF], rk[PF], rv[PF];
  auto fetch = [&](int bn, TileRegs& q_, TileRegs& k_, TileRegs& v_) {
    const int b1 = SEQ(bn / heads);
    const int L1 = VL ? cu[b1 + 1] - cu[b1] : L;
    const bf16_t* qn = qkv + (VL ? (long)cu[b1] * gs : (long)b1 * L * gs) + (bn % heads) * HD;
    q_ = tile_load(qn, gs, L1, tid, nthr); k_ = tile_load(qn + d_model, gs, L1, tid, nthr); v_ = tile_load(qn + 2 * d_model, gs, L1, tid, nthr);
  };
#pragma unroll
  for (int s = 0; s < PF; ++s)
    if ((int)blockIdx.x + s * (int)gridDim.x < nbh) fetch(blockIdx.x + s * gridDim.x, rq[s], rk[s], rv[s]);
  for (int bh0 = blockIdx.x; bh0 < nbh; bh0 += PF * gridDim.x) {
#pragma unroll
  for (int stage = 0; stage < PF; ++stage) {
  const int bh = bh0 + stage * (int)gridDim.x;
  if (bh >= nbh) break;
  const int bi = SEQ(bh / heads), h = bh % heads;
  const int Lp = VL ? cu[bi + 1] - cu[bi] : L;              // length of this sequence (rows cu[bi] .. of qkv / out when packed)
  tile_store(rq[stage], Qs, tid, nthr); tile_store(rk[stage], Ks, tid, nthr); tile_store(rv[stage], Vs, tid, nthr);
  __syncthreads();
  if (bh + PF * (int)gridDim.x < nbh) fetch(bh + PF * gridDim.x, rq[stage], rk[stage], rv[stage]);   // in flight during PF pairs' arithmetic

  const int qb = wave;
  const int q = qb * 16 + (lane & 15);  // this lane's query (column of S^T)
  // S^T[key][q]: A = K rows, B = Q rows
  f32x4_t s[NKB];
  bf16x8_t qf[2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) qf[ks] = lds_frag(Qs + q * RS + ks * 32 + 8 * (lane >> 4));
#pragma unroll
  for (int kb = 0; kb < NKB; ++kb) {
    f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
    // blocks the masks below turn into -inf entirely (keys beyond the sequence, keys after every query of this wave under the
    // causal mask) or whose queries are never stored: no MFMAs (wave-uniform test)
    if (!(kb * 16 >= Lp || qb * 16 >= Lp || (causal && kb > qb))) {
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        bf16x8_t kf = lds_frag(Ks + (kb * 16 + (lane & 15)) * RS + ks * 32 + 8 * (lane >> 4));
        acc = mfma16(kf, qf[ks], acc);
      }
    }
    s[kb] = acc;
  }
  // lane holds keys kb*16 + 4*(lane>>4) + r for its query q
  float mx = -INFINITY;
#pragma unroll
  for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int key = kb * 16 + 4 * (lane >> 4) + r;
      float v = s[kb][r] * scale;
      if (key >= Lp || (causal && key > q)) v = -INFINITY;
      s[kb][r] = v;
      mx = fmaxf(mx, v);
    }
  mx = quad_max(mx);
  float sum = 0.f;
#pragma unroll
  for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float p = __expf(s[kb][r] - mx);
      s[kb][r] = p;
      sum += p;
    }
  sum = quad_sum(sum);
  const float inv = 1.f / sum;
  if (q < Lp && (lane >> 4) == 0) lse[((long)bi * heads + h) * L + q] = mx + __logf(sum);
  // P[q][key..key+3] <- transpose of the C fragment: one 8-byte store per fragment
#pragma unroll
  for (int kb = 0; kb < NKB; ++kb) {
    uint2 w;
    w.x = pack2bf_hw(s[kb][0] * inv, s[kb][1] * inv);
    w.y = pack2bf_hw(s[kb][2] * inv, s[kb][3] * inv);
    *reinterpret_cast<uint2*>(Ps + q * TS + kb * 16 + 4 * (lane >> 4)) = w;
  }
  __syncthreads();
  // O[q][d] = sum_key P[q][key] V[key][d]: A = P rows (this wave's queries), B = V via transpose reads
  const int nks_live = qb * 16 >= Lp ? 0 : (causal ? min((Lp + 31) >> 5, ((qb + 1) * 16 + 31) >> 5) : (Lp + 31) >> 5);   // P is zero beyond
  // C^T layout: lane&15 = query row, registers = 4 consecutive head-dim columns 4*(lane>>4)+r.  Stored straight from there a wave writes
  // sixteen 32-byte pieces per instruction, every 128-byte row of the head in four visits; the store path is priced per line touched
  // (profiles/r06_epilogue_traces.txt).  LINE_OUT: the wave's 16 x 64 block goes through its OWN query rows of Qs (nobody reads them
  // after the first MFMAs) and leaves as whole 128-byte rows, 16 bytes per lane.
  bf16_t* Os = Qs + qb * 16 * RS;
#pragma unroll
  for (int db = 0; db < 4; ++db) {
    f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
      if (ks >= nks_live) continue;
      const bool dead = KTAIL && ks == NKS - 1 && (lane >> 4) >= 2;
      bf16x8_t pf = kmask(lds_frag(Ps + (qb * 16 + (lane & 15)) * TS + ks * 32 + 8 * (lane >> 4)), dead);
      bf16x8_t vf = kmask(frag_tr(Vs, RS, ks * 32, db * 16, dead ? (lane & 31) : lane), dead);
      acc = mfma16(vf, pf, acc);            // swapped operands: the fragment comes out transposed
    }
    uint2 w;
    w.x = pack2bf_hw(acc[0], acc[1]); w.y = pack2bf_hw(acc[2], acc[3]);
    if (ATTN_LINE_OUT) {
      *reinterpret_cast<uint2*>(Os + (lane & 15) * RS + db * 16 + 4 * (lane >> 4)) = w;
    } else {
      const int qq = qb * 16 + (lane & 15);
      if (qq < Lp) *reinterpret_cast<uint2*>(out + ((VL ? (long)cu[bi] : (long)bi * L) + qq) * d_model + h * HD + db * 16 + 4 * (lane >> 4)) = w;
    }
  }
  if (ATTN_LINE_OUT) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");   // LDS written by some lanes is read by others of the same wave
    __builtin_amdgcn_wave_barrier();
    bf16_t* og = out + ((VL ? (long)cu[bi] : (long)bi * L) + qb * 16) * d_model + h * HD;
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
      const int r = pass * 8 + (lane >> 3), c = lane & 7;
      const uint4 v = *reinterpret_cast<const uint4*>(Os + r * RS + c * 8);
      if (qb * 16 + r < Lp) *reinterpret_cast<uint4*>(og + (long)r * d_model + c * 8) = v;
    }
  }
  __syncthreads();                         // every wave is done with the tiles before the next pair overwrites them
  }
  }
  if (VL) tail0 = tail0 < 0 ? cu[nb_total] : tail0;        // rows = -1: the valid row count is cu_seqlens[b], read here (no host value in the launch: the captured step replays for any batch of this padded size)
  if (VL && tail1 > tail0) {
    // packed layout: the rows between the last caption and the whole-tile row count are ZERO (they meet the weight-gradient GEMMs as
    // contraction rows); written here instead of by a separate fill launch per attention call (24 launches per CLIP step)
    uint4* tp_ = reinterpret_cast<uint4*>(out + (long)tail0 * (d_model));
    const long n16 = (long)(tail1 - tail0) * (d_model) / 8;
    for (long i = (long)blockIdx.x * nthr + tid; i < n16; i += (long)gridDim.x * nthr) tp_[i] = make_uint4(0u, 0u, 0u, 0u);
  }
}

// ------------------------------------------------------------------------------------------
// backward, bf16 / MFMA
// ------------------------------------------------------------------------------------------
// __launch_bounds__: without it the compiler budgets registers for 1024-thread workgroups (128 VGPRs) and the packed-caption
// instantiation spilled 22 of them to scratch; 64 * NKB threads with >= 3 waves per SIMD leaves 168 (image tower: 80.6 vs 87.5 us
// per call on one box, text tower 78.1 vs 80.4; profiles/r03_attn_bwd_variants.txt)
template <int NKB, bool VL, int PF = 1>
__global__ __launch_bounds__(64 * NKB, PF == 2 ? 2 : 3) void attn_bwd_mfma_kernel(const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ out,
                                     const bf16_t* __restrict__ dout, const float* __restrict__ lse,
                                     bf16_t* __restrict__ dqkv, int L, int heads, int causal, float scale, int nbh,
                                     const int* __restrict__ cu, int tail0, int tail1, const int* __restrict__ seq_list,
                                     const int* __restrict__ seq_range) {
  const int nb_total = nbh / heads;
  const int s0 = (VL && seq_list) ? seq_range[0] : 0;       // (see attn_fwd_mfma_kernel)
  if (VL && seq_list) nbh = seq_range[1] * heads;
  constexpr int L16 = NKB * 16;
  constexpr int NKS = (L16 + 31) / 32;
  constexpr bool KTAIL = (L16 % 32) != 0;
  constexpr int TS = L16 + 8;
  DH_DYN_LDS_A16(unsigned char, smem_raw);
  bf16_t* Qs = reinterpret_cast<bf16_t*>(smem_raw);  // [L16][RS] row-major tiles
  bf16_t* Ks = Qs + L16 * RS;
  bf16_t* Vs = Ks + L16 * RS;
  bf16_t* Gs = Vs + L16 * RS;     // dO
  bf16_t* Pt = Gs + L16 * RS;     // P^T  [key][q]   [L16][TS]
  bf16_t* dSt = Pt + L16 * TS;    // dS^T [key][q]   (+ slack for masked tail reads)
  float* Dq = reinterpret_cast<float*>(dSt + L16 * TS + 64);  // [L16] rowsum(dO o O)

  const int d_model = heads * HD;
  const long gs = 3L * d_model;
  const int tid = threadIdx.x, nthr = blockDim.x;
  const int lane = tid & 63, wave = tid >> 6;
  // D[q] = sum_d dO[q][d] * O[q][d]: thread -> row tid >> 2, 16 columns (tid & 3) * 16 .. +15 (two 16-byte pieces each of O, dO)
  struct DRegs { uint4 o[2], g[2]; };
  auto d_load = [&](const bf16_t* og, const bf16_t* gg, int Lc) {
    DRegs r;
    const int row = tid >> 2, part = tid & 3;
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      r.o[c] = make_uint4(0, 0, 0, 0); r.g[c] = make_uint4(0, 0, 0, 0);
      if (row < Lc) {
        r.o[c] = *reinterpret_cast<const uint4*>(og + (long)row * d_model + part * 16 + c * 8);
        r.g[c] = *reinterpret_cast<const uint4*>(gg + (long)row * d_model + part * 16 + c * 8);
      }
    }
    return r;
  };
  // PF pairs ahead (see the forward kernel): 48 registers per stage
  TileRegs rq[PF], rk[PF], rv[PF], rg[PF];
  DRegs rd[PF];
  auto fetch = [&](int bn, TileRegs& q_, TileRegs& k_, TileRegs& v_, TileRegs& g_, DRegs& d_) {
    const int b1 = SEQ(bn / heads), h1 = bn % heads;
    const long r1 = VL ? (long)cu[b1] : (long)b1 * L;
    const int L1 = VL ? cu[b1 + 1] - cu[b1] : L;
    const bf16_t* qn = qkv + r1 * gs + h1 * HD;
    const bf16_t* gn = dout + r1 * d_model + h1 * HD;
    q_ = tile_load(qn, gs, L1, tid, nthr); k_ = tile_load(qn + d_model, gs, L1, tid, nthr); v_ = tile_load(qn + 2 * d_model, gs, L1, tid, nthr);
    g_ = tile_load(gn, d_model, L1, tid, nthr);
    d_ = d_load(out + r1 * d_model + h1 * HD, gn, L1);
  };
#pragma unroll
  for (int s = 0; s < PF; ++s)
    if ((int)blockIdx.x + s * (int)gridDim.x < nbh) fetch(blockIdx.x + s * gridDim.x, rq[s], rk[s], rv[s], rg[s], rd[s]);
  for (int bh0 = blockIdx.x; bh0 < nbh; bh0 += PF * gridDim.x) {
#pragma unroll
  for (int stage = 0; stage < PF; ++stage) {
  const int bh = bh0 + stage * (int)gridDim.x;
  if (bh >= nbh) break;
  const int bi = SEQ(bh / heads), h = bh % heads;
  const long row0 = VL ? (long)cu[bi] : (long)bi * L;
  const int Lp = VL ? cu[bi + 1] - cu[bi] : L;
  tile_store(rq[stage], Qs, tid, nthr); tile_store(rk[stage], Ks, tid, nthr); tile_store(rv[stage], Vs, tid, nthr); tile_store(rg[stage], Gs, tid, nthr);
  {
    const uint32_t* ow = reinterpret_cast<const uint32_t*>(rd[stage].o);
    const uint32_t* gw = reinterpret_cast<const uint32_t*>(rd[stage].g);
    float acc = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i)
      acc += __uint_as_float(ow[i] << 16) * __uint_as_float(gw[i] << 16) + __uint_as_float(ow[i] & 0xffff0000u) * __uint_as_float(gw[i] & 0xffff0000u);
    acc += __shfl_xor(acc, 1, 64);
    acc += __shfl_xor(acc, 2, 64);
    if ((tid & 3) == 0) Dq[tid >> 2] = acc;
  }
  __syncthreads();
  // this pair's log-sum-exp values are requested BEFORE the prefetch: a load issued behind the prefetch would have to wait
  // for all of it (vmcnt retires in order)
  float lse_r[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int qq = wave * 16 + 4 * (lane >> 4) + r;
    lse_r[r] = qq < Lp ? lse[((long)bi * heads + h) * L + qq] : 0.f;
  }
  if (bh + PF * (int)gridDim.x < nbh) fetch(bh + PF * gridDim.x, rq[stage], rk[stage], rv[stage], rg[stage], rd[stage]);   // in flight during PF pairs' arithmetic

  // ---- phase 1: wave owns query block qb: S[q][key] (rows q), dP[q][key]
  {
    const int qb = wave;
    bf16x8_t qf[2], gf[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      qf[ks] = lds_frag(Qs + (qb * 16 + (lane & 15)) * RS + ks * 32 + 8 * (lane >> 4));
      gf[ks] = lds_frag(Gs + (qb * 16 + (lane & 15)) * RS + ks * 32 + 8 * (lane >> 4));
    }
    float d_r[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) d_r[r] = Dq[qb * 16 + 4 * (lane >> 4) + r];
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb) {
      const int key = kb * 16 + (lane & 15);
      // a 16 x 16 block that the masks zero entirely -- keys or queries beyond the sequence (captions average 43 of 77 tokens),
      // keys after every query of the block under the causal mask (10 of 25 blocks at full length) -- is written as zeros
      // without its MFMAs and exponentials (wave-uniform test)
      if (kb * 16 >= Lp || qb * 16 >= Lp || (causal && kb > qb)) {
        const uint2 z = make_uint2(0u, 0u);
        *reinterpret_cast<uint2*>(Pt + key * TS + qb * 16 + 4 * (lane >> 4)) = z;
        *reinterpret_cast<uint2*>(dSt + key * TS + qb * 16 + 4 * (lane >> 4)) = z;
        continue;
      }
      f32x4_t sacc = {0.f, 0.f, 0.f, 0.f}, pacc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        bf16x8_t kf = lds_frag(Ks + (kb * 16 + (lane & 15)) * RS + ks * 32 + 8 * (lane >> 4));
        bf16x8_t vf = lds_frag(Vs + (kb * 16 + (lane & 15)) * RS + ks * 32 + 8 * (lane >> 4));
        sacc = mfma16(qf[ks], kf, sacc);  // rows q, cols key
        pacc = mfma16(gf[ks], vf, pacc);  // dP[q][key]
      }
      float p[4], ds[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int qq = qb * 16 + 4 * (lane >> 4) + r;
        const bool masked = key >= Lp || qq >= Lp || (causal && key > qq);
        p[r] = masked ? 0.f : __expf(sacc[r] * scale - lse_r[r]);
        ds[r] = p[r] * (pacc[r] - d_r[r]) * scale;
      }
      // transposed fragment stores: [key][q..q+3], 8 bytes
      uint2 w;
      w.x = pack2bf_hw(p[0], p[1]); w.y = pack2bf_hw(p[2], p[3]);
      *reinterpret_cast<uint2*>(Pt + key * TS + qb * 16 + 4 * (lane >> 4)) = w;
      w.x = pack2bf_hw(ds[0], ds[1]); w.y = pack2bf_hw(ds[2], ds[3]);
      *reinterpret_cast<uint2*>(dSt + key * TS + qb * 16 + 4 * (lane >> 4)) = w;
    }
  }
  __syncthreads();

  // ---- phase 2: wave owns row block rb (keys for dV/dK, queries for dQ); contraction length L16
  if (wave * 16 < Lp) {                    // (a row block beyond the sequence stores nothing)
    const int rb = wave;
    bf16_t* dq_g = dqkv + row0 * gs + h * HD;
    const int nks_live = (Lp + 31) >> 5;   // contraction steps of 32 rows that hold anything: P^T / dS^T are zero beyond the sequence
    // LINE: each of the three 16 x 64 output blocks leaves as whole 128-byte rows, staged through this wave's OWN rows of P^T (read by
    // nobody else, and by this wave only for dV, which goes first) -- see the forward kernel.  Needs TS * 2 >= 128 bytes per row (NKB >= 4).
    // Dense sequences only: on the packed captions (causal, 43 rows on average) the three separate passes cost more than the stores save
    // (text backward 74.7 -> 77.4 us, image backward 85 -> 78 us; profiles/r06_attention_line_stores.txt).
    constexpr bool LINE = ATTN_LINE_OUT && !VL && TS * 2 >= 128;
    bf16_t* St = Pt + rb * 16 * TS;
    auto flush = [&](long col0) {          // the staged block -> rows rb*16 .. of dqkv, columns col0 .. col0 + 63 of this head
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");   // LDS written by some lanes is read by others of the same wave
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int pass = 0; pass < 2; ++pass) {
        const int r = pass * 8 + (lane >> 3), c = lane & 7;
        const uint4 v = *reinterpret_cast<const uint4*>(St + r * TS + c * 8);
        if (rb * 16 + r < Lp) *reinterpret_cast<uint4*>(dq_g + (long)(rb * 16 + r) * gs + col0 + c * 8) = v;
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");   // the next block's pieces overwrite what other lanes have just read
      __builtin_amdgcn_wave_barrier();
    };
    auto emit = [&](const f32x4_t& a, int db, long col0) {
      // transposed fragments: lane&15 = row (query / key), registers = 4 consecutive head-dim columns -> 8-byte pieces
      uint2 w;
      w.x = pack2bf_hw(a[0], a[1]); w.y = pack2bf_hw(a[2], a[3]);
      if (LINE) {
        *reinterpret_cast<uint2*>(St + (lane & 15) * TS + db * 16 + 4 * (lane >> 4)) = w;
      } else {
        const int row = rb * 16 + (lane & 15);
        if (row < Lp) *reinterpret_cast<uint2*>(dq_g + (long)row * gs + col0 + db * 16 + 4 * (lane >> 4)) = w;
      }
    };
    if (LINE) {
      // one output at a time (same MFMAs and operand reads as the fused loop below; the P^T rows are dead once dV is done)
      f32x4_t acc[4];
      // dV[key][d] = sum_q P^T[key][q] dO[q][d]
#pragma unroll
      for (int db = 0; db < 4; ++db) {
        acc[db] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
          if (ks >= nks_live) continue;
          const bool dead = KTAIL && ks == NKS - 1 && (lane >> 4) >= 2;
          const int ln = dead ? (lane & 31) : lane;
          const int ro = (rb * 16 + (lane & 15)) * TS + ks * 32 + 8 * (ln >> 4);
          bf16x8_t gB = kmask(frag_tr(Gs, RS, ks * 32, db * 16, ln), dead);
          acc[db] = mfma16(gB, kmask(lds_frag(Pt + ro), dead), acc[db]);
        }
      }
#pragma unroll
      for (int db = 0; db < 4; ++db) emit(acc[db], db, 2 * d_model);
      flush(2 * d_model);
      // dK[key][d] = sum_q dS^T[key][q] Q[q][d]
#pragma unroll
      for (int db = 0; db < 4; ++db) {
        f32x4_t ak = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
          if (ks >= nks_live) continue;
          const bool dead = KTAIL && ks == NKS - 1 && (lane >> 4) >= 2;
          const int ln = dead ? (lane & 31) : lane;
          const int ro = (rb * 16 + (lane & 15)) * TS + ks * 32 + 8 * (ln >> 4);
          bf16x8_t qB = kmask(frag_tr(Qs, RS, ks * 32, db * 16, ln), dead);
          ak = mfma16(qB, kmask(lds_frag(dSt + ro), dead), ak);
        }
        emit(ak, db, d_model);
      }
      flush(d_model);
      // dQ[q][d] = sum_key dS[q][key] K[key][d]: A = dS via transpose read of dS^T, B = K via transpose read
#pragma unroll
      for (int db = 0; db < 4; ++db) {
        f32x4_t aq = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
          if (ks >= nks_live) continue;
          const bool dead = KTAIL && ks == NKS - 1 && (lane >> 4) >= 2;
          const int ln = dead ? (lane & 31) : lane;
          bf16x8_t dsA = kmask(frag_tr(dSt, TS, ks * 32, rb * 16, ln), dead);
          bf16x8_t kB = kmask(frag_tr(Ks, RS, ks * 32, db * 16, ln), dead);
          aq = mfma16(kB, dsA, aq);
        }
        emit(aq, db, 0);
      }
      flush(0);
    } else {
#pragma unroll
    for (int db = 0; db < 4; ++db) {
      f32x4_t av = {0.f, 0.f, 0.f, 0.f}, ak = {0.f, 0.f, 0.f, 0.f}, aq = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < NKS; ++ks) {
        if (ks >= nks_live) continue;
        const bool dead = KTAIL && ks == NKS - 1 && (lane >> 4) >= 2;
        const int ln = dead ? (lane & 31) : lane;           // dead lanes read in-bounds addresses, then get zeroed
        const int ro = (rb * 16 + (lane & 15)) * TS + ks * 32 + 8 * (ln >> 4);
        // dV[key][d] = sum_q P^T[key][q] dO[q][d];  dK[key][d] = sum_q dS^T[key][q] Q[q][d]
        bf16x8_t gB = kmask(frag_tr(Gs, RS, ks * 32, db * 16, ln), dead);
        bf16x8_t qB = kmask(frag_tr(Qs, RS, ks * 32, db * 16, ln), dead);
        av = mfma16(gB, kmask(lds_frag(Pt + ro), dead), av);      // operands swapped: transposed fragments (see the stores below)
        ak = mfma16(qB, kmask(lds_frag(dSt + ro), dead), ak);
        // dQ[q][d] = sum_key dS[q][key] K[key][d]: A = dS via transpose read of dS^T, B = K via transpose read
        bf16x8_t dsA = kmask(frag_tr(dSt, TS, ks * 32, rb * 16, ln), dead);
        bf16x8_t kB = kmask(frag_tr(Ks, RS, ks * 32, db * 16, ln), dead);
        aq = mfma16(kB, dsA, aq);
      }
      emit(aq, db, 0);
      emit(ak, db, d_model);
      emit(av, db, 2 * d_model);
    }
    }
  }
  __syncthreads();                         // every wave is done with the tiles before the next pair overwrites them
  }
  }
  if (VL) tail0 = tail0 < 0 ? cu[nb_total] : tail0;        // (see the forward kernel)
#undef SEQ
  if (VL && tail1 > tail0) {
    // packed layout: the rows between the last caption and the whole-tile row count are ZERO (they meet the weight-gradient GEMMs as
    // contraction rows); written here instead of by a separate fill launch per attention call (24 launches per CLIP step)
    uint4* tp_ = reinterpret_cast<uint4*>(dqkv + (long)tail0 * (3 * d_model));
    const long n16 = (long)(tail1 - tail0) * (3 * d_model) / 8;
    for (long i = (long)blockIdx.x * nthr + tid; i < n16; i += (long)gridDim.x * nthr) tp_[i] = make_uint4(0u, 0u, 0u, 0u);
  }
}

// ------------------------------------------------------------------------------------------
// fp32 validation path: one workgroup (256 threads) per (batch, head); hd <= 64, L <= 128
// ------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void attn_fwd_generic_kernel(const T* __restrict__ qkv, T* __restrict__ out,
                                                               float* __restrict__ lse, int Lmax, int heads, int hd,
                                                               int causal, float scale, const int* __restrict__ cu) {
  DH_DYN_LDS(float, smf);
  const int hs = hd + 1;
  const int bh = blockIdx.x, bi = bh / heads, h = bh % heads;
  const long row0 = cu ? (long)cu[bi] : (long)bi * Lmax;     // packed sequences: rows cu[bi] .. cu[bi+1]; lse stays [b][heads][Lmax]
  const int L = cu ? cu[bi + 1] - cu[bi] : Lmax;
  float* Q = smf; float* K = Q + L * hs; float* V = K + L * hs; float* S = V + L * hs;  // S [L][L+1]
  const int dm = heads * hd;
  const long gs = 3L * dm;
  const T* base = qkv + row0 * gs + h * hd;
  for (int i = threadIdx.x; i < L * hd; i += 256) {
    int r = i / hd, c = i % hd;
    Q[r * hs + c] = ld<T>(base + (long)r * gs + c);
    K[r * hs + c] = ld<T>(base + (long)r * gs + dm + c);
    V[r * hs + c] = ld<T>(base + (long)r * gs + 2 * dm + c);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < L * L; i += 256) {
    int q = i / L, k = i % L;
    float a = 0.f;
    for (int c = 0; c < hd; ++c) a = fmaf(Q[q * hs + c], K[k * hs + c], a);
    a *= scale;
    if (causal && k > q) a = -INFINITY;
    S[q * (L + 1) + k] = a;
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int q = wave; q < L; q += 4) {
    float mx = -INFINITY;
    for (int k = lane; k < L; k += 64) mx = fmaxf(mx, S[q * (L + 1) + k]);
    mx = wave_max(mx);
    float sum = 0.f;
    for (int k = lane; k < L; k += 64) { float p = __expf(S[q * (L + 1) + k] - mx); S[q * (L + 1) + k] = p; sum += p; }
    sum = wave_sum(sum);
    const float inv = 1.f / sum;
    for (int k = lane; k < L; k += 64) S[q * (L + 1) + k] *= inv;
    if (lane == 0) lse[((long)bi * heads + h) * Lmax + q] = mx + __logf(sum);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < L * hd; i += 256) {
    int q = i / hd, c = i % hd;
    float a = 0.f;
    for (int k = 0; k < L; ++k) a = fmaf(S[q * (L + 1) + k], V[k * hs + c], a);
    st<T>(out + (row0 + q) * dm + h * hd + c, a);
  }
}

template <typename T>
__global__ __launch_bounds__(256) void attn_bwd_generic_kernel(const T* __restrict__ qkv, const T* __restrict__ out,
                                                               const T* __restrict__ dout, const float* __restrict__ lse,
                                                               T* __restrict__ dqkv, int Lmax, int heads, int hd,
                                                               int causal, float scale, const int* __restrict__ cu) {
  DH_DYN_LDS(float, smf);
  const int bh = blockIdx.x, bi = bh / heads, h = bh % heads;
  const long row0 = cu ? (long)cu[bi] : (long)bi * Lmax;
  const int L = cu ? cu[bi + 1] - cu[bi] : Lmax;
  const int hs = hd + 1, ls = L + 1;
  float* Q = smf; float* K = Q + L * hs; float* V = K + L * hs; float* G = V + L * hs;
  float* P = G + L * hs; float* dS = P + L * ls; float* Dq = dS + L * ls;
  const int dm = heads * hd;
  const long gs = 3L * dm;
  const T* base = qkv + row0 * gs + h * hd;
  const T* ob = out + row0 * dm + h * hd;
  const T* gb = dout + row0 * dm + h * hd;
  for (int i = threadIdx.x; i < L * hd; i += 256) {
    int r = i / hd, c = i % hd;
    Q[r * hs + c] = ld<T>(base + (long)r * gs + c);
    K[r * hs + c] = ld<T>(base + (long)r * gs + dm + c);
    V[r * hs + c] = ld<T>(base + (long)r * gs + 2 * dm + c);
    G[r * hs + c] = ld<T>(gb + (long)r * dm + c);
  }
  __syncthreads();
  for (int q = threadIdx.x; q < L; q += 256) {
    float a = 0.f;
    for (int c = 0; c < hd; ++c) a = fmaf(G[q * hs + c], ld<T>(ob + (long)q * dm + c), a);
    Dq[q] = a;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < L * L; i += 256) {
    int q = i / L, k = i % L;
    float s = 0.f, dp = 0.f;
    for (int c = 0; c < hd; ++c) { s = fmaf(Q[q * hs + c], K[k * hs + c], s); dp = fmaf(G[q * hs + c], V[k * hs + c], dp); }
    float p = (causal && k > q) ? 0.f : __expf(s * scale - lse[((long)bi * heads + h) * Lmax + q]);
    P[q * ls + k] = p;
    dS[q * ls + k] = p * (dp - Dq[q]) * scale;
  }
  __syncthreads();
  T* dbase = dqkv + row0 * gs + h * hd;
  for (int i = threadIdx.x; i < L * hd; i += 256) {
    int r = i / hd, c = i % hd;
    float dq = 0.f, dk = 0.f, dv = 0.f;
    for (int j = 0; j < L; ++j) {
      dq = fmaf(dS[r * ls + j], K[j * hs + c], dq);
      dk = fmaf(dS[j * ls + r], Q[j * hs + c], dk);
      dv = fmaf(P[j * ls + r], G[j * hs + c], dv);
    }
    st<T>(dbase + (long)r * gs + c, dq);
    st<T>(dbase + (long)r * gs + dm + c, dk);
    st<T>(dbase + (long)r * gs + 2 * dm + c, dv);
  }
}

static int attn_cus() {
  static int n = 0;
  if (!n) {
    int dev = 0;
    hipGetDevice(&dev);
    hipDeviceProp_t p;
    n = (hipGetDeviceProperties(&p, dev) == hipSuccess && p.multiProcessorCount > 0) ? p.multiProcessorCount : 256;
  }
  return n;
}

template <int NKB>
int launch_fwd_mfma(const bf16_t* qkv, bf16_t* out, float* lse, int b, int L, int heads, int causal, float scale,
                    hipStream_t st, const int* cu = nullptr, int tail0 = 0, int tail1 = 0, const int* seq_list = nullptr,
                    const int* seq_range = nullptr) {
  constexpr int L16 = NKB * 16, TS = L16 + 8;
  size_t lds = (size_t)(3 * L16 * RS + L16 * TS + 64) * sizeof(bf16_t);
  static int pf_env = -1;         // DH_ATTN_PF (read once): prefetch depth of the packed (text) forward kernel: 1 (default), 2, 3 = 2 for the short bucket only (measured: profiles/r06_attention_variants.txt -- no gain, the kernel is not waiting for its loads)
  if (pf_env < 0) { const char* ev = getenv("DH_ATTN_PF"); pf_env = ev ? atoi(ev) : 1; }
  const bool pf2 = cu && (pf_env == 2 || (pf_env == 3 && NKB <= 4));      // (3: only the short bucket -- the 5-block instantiation spills two registers at depth 2)
  if (pf2) hipFuncSetAttribute((const void*)attn_fwd_mfma_kernel<NKB, true, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  else if (cu) hipFuncSetAttribute((const void*)attn_fwd_mfma_kernel<NKB, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  else hipFuncSetAttribute((const void*)attn_fwd_mfma_kernel<NKB, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  static int cap_env = -1;        // DH_ATTN_WG_CAP (read once): most workgroups per CU (LDS permitting)
  if (cap_env < 0) { const char* ev = getenv("DH_ATTN_WG_CAP"); cap_env = ev ? atoi(ev) : 4; }
  const int per_cu = (int)((160 * 1024) / lds) > cap_env ? cap_env : (int)((160 * 1024) / lds);
  int grid = attn_cus() * (per_cu < 1 ? 1 : per_cu);
  if (grid > b * heads) grid = b * heads;
  if (pf2) hipLaunchKernelGGL((attn_fwd_mfma_kernel<NKB, true, 2>), dim3(grid), dim3(64 * NKB), lds, st, qkv, out, lse, L, heads, causal, scale, b * heads, cu, tail0, tail1, seq_list, seq_range);
  else if (cu) hipLaunchKernelGGL((attn_fwd_mfma_kernel<NKB, true>), dim3(grid), dim3(64 * NKB), lds, st, qkv, out, lse, L, heads, causal, scale, b * heads, cu, tail0, tail1, seq_list, seq_range);
  else hipLaunchKernelGGL((attn_fwd_mfma_kernel<NKB, false>), dim3(grid), dim3(64 * NKB), lds, st, qkv, out, lse, L, heads, causal, scale, b * heads, cu, 0, 0, (const int*)nullptr, (const int*)nullptr);
  return 0;
}
template <int NKB>
int launch_bwd_mfma(const bf16_t* qkv, const bf16_t* out, const bf16_t* dout, const float* lse, bf16_t* dqkv, int b,
                    int L, int heads, int causal, float scale, hipStream_t st, const int* cu = nullptr, int tail0 = 0, int tail1 = 0,
                    const int* seq_list = nullptr, const int* seq_range = nullptr) {
  constexpr int L16 = NKB * 16, TS = L16 + 8;
  size_t lds = (size_t)(4 * L16 * RS + 2 * L16 * TS + 64) * sizeof(bf16_t) + L16 * sizeof(float);
  if (cu) hipFuncSetAttribute((const void*)attn_bwd_mfma_kernel<NKB, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  else hipFuncSetAttribute((const void*)attn_bwd_mfma_kernel<NKB, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  { static int extra = -1; if (extra < 0) { const char* ev = getenv("DH_ATTN_LDS_EXTRA"); extra = ev ? atoi(ev) : 0; } lds += extra; }   // occupancy probe (tools/bench_small.py)
  if (cu) hipFuncSetAttribute((const void*)attn_bwd_mfma_kernel<NKB, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  else hipFuncSetAttribute((const void*)attn_bwd_mfma_kernel<NKB, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  const int per_cu = (int)((160 * 1024) / lds) > 4 ? 4 : (int)((160 * 1024) / lds);
  int grid = attn_cus() * (per_cu < 1 ? 1 : per_cu);
  if (grid > b * heads) grid = b * heads;
  static int pf_env = -1;         // DH_ATTN_PF_BWD (read once): prefetch depth of the packed (text) backward kernel: 1, 2 (default), 3 = 2 for the short bucket only
  if (pf_env < 0) { const char* ev = getenv("DH_ATTN_PF_BWD"); pf_env = ev ? atoi(ev) : 1; }
  const bool pf2 = cu && (pf_env == 2 || (pf_env == 3 && NKB <= 4));
  if (pf2) {
    hipFuncSetAttribute((const void*)attn_bwd_mfma_kernel<NKB, true, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL((attn_bwd_mfma_kernel<NKB, true, 2>), dim3(grid), dim3(64 * NKB), lds, st, qkv, out, dout, lse, dqkv, L, heads, causal, scale, b * heads, cu, tail0, tail1, seq_list, seq_range);
  } else
  if (cu) hipLaunchKernelGGL((attn_bwd_mfma_kernel<NKB, true>), dim3(grid), dim3(64 * NKB), lds, st, qkv, out, dout, lse, dqkv, L, heads, causal, scale, b * heads, cu, tail0, tail1, seq_list, seq_range);
  else hipLaunchKernelGGL((attn_bwd_mfma_kernel<NKB, false>), dim3(grid), dim3(64 * NKB), lds, st, qkv, out, dout, lse, dqkv, L, heads, causal, scale, b * heads, cu, 0, 0, (const int*)nullptr, (const int*)nullptr);
  return 0;
}

}  // namespace



#define DISPATCH_NKB(nkb, CALL)                 \
  switch (nkb) {                                \
    case 1: CALL(1); break; case 2: CALL(2); break; case 3: CALL(3); break; case 4: CALL(4); break; \
    case 5: CALL(5); break; case 6: CALL(6); break; case 7: CALL(7); break; case 8: CALL(8); break; \
    default: DH_FAIL(DH_ERR_UNSUPPORTED, "attention: L=%d unsupported", L); \
  }

static int attn_fwd_impl(int dtype, const void* qkv, void* out, float* lse, int b, int L, int heads, int hd, int causal,
                         const int* cu, dh_stream_t stream, int tail0 = 0, int tail1 = 0) {
  hipStream_t st = (hipStream_t)stream;
  DH_REQUIRE(qkv && out && lse && b > 0 && L > 0 && heads > 0, "dh_attn_fwd: bad args");
  DH_REQUIRE(L <= 128 && hd <= 64, "dh_attn_fwd: L<=128 and hd<=64 required (got %d, %d)", L, hd);
  const float scale = 1.0f / sqrtf((float)hd);
  if (dtype == DH_BF16 && hd == 64) {
    const int nkb = (L + 15) / 16;
#define CALL(N) launch_fwd_mfma<N>((const bf16_t*)qkv, (bf16_t*)out, lse, b, L, heads, causal, scale, st, cu, tail0, tail1)
    DISPATCH_NKB(nkb, CALL)
#undef CALL
  } else {
    size_t lds = (size_t)(3 * L * (hd + 1) + L * (L + 1)) * sizeof(float);
    DH_REQUIRE(lds <= 160 * 1024, "dh_attn_fwd: sequence too long for the fp32 / generic path (L <= 126 at hd = 64; the bf16 MFMA path takes L <= 128)");
    if (dtype == DH_BF16) {
      hipFuncSetAttribute((const void*)attn_fwd_generic_kernel<bf16_t>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      hipLaunchKernelGGL(attn_fwd_generic_kernel<bf16_t>, dim3(b * heads), dim3(256), lds, st, (const bf16_t*)qkv, (bf16_t*)out, lse, L, heads, hd, causal, scale, cu);
    } else {
      hipFuncSetAttribute((const void*)attn_fwd_generic_kernel<float>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      hipLaunchKernelGGL(attn_fwd_generic_kernel<float>, dim3(b * heads), dim3(256), lds, st, (const float*)qkv, (float*)out, lse, L, heads, hd, causal, scale, cu);
    }
  }
  DH_CHECK_LAUNCH();
  return DH_OK;
}

extern "C" int dh_attn_fwd(int dtype, const void* qkv, void* out, float* lse, int b, int L, int heads, int hd,
                           int causal, dh_stream_t stream) {
  return attn_fwd_impl(dtype, qkv, out, lse, b, L, heads, hd, causal, nullptr, stream);
}
// rows / rows_pad: valid rows (= cu_seqlens[b]) and allocated rows of the packed layout; rows [rows, rows_pad) of `out` are
// written as zeros (inside the attention kernel on the bf16 path, by a memset otherwise)
extern "C" int dh_attn_varlen_fwd(int dtype, const void* qkv, void* out, float* lse, const int* cu_seqlens, int b, int Lmax, int heads,
                                  int hd, int causal, int rows, int rows_pad, dh_stream_t stream) {
  DH_REQUIRE(cu_seqlens, "dh_attn_varlen_fwd: cu_seqlens is NULL");
  const bool in_kernel = dtype == DH_BF16 && hd == 64;
  DH_REQUIRE((rows >= 0 && rows_pad >= rows) || (rows == -1 && in_kernel && rows_pad > 0),
             "dh_attn_varlen_fwd: rows %d, rows_pad %d (rows = -1, 'read cu_seqlens[b] on the device', needs the bf16 / hd = 64 kernels)", rows, rows_pad);
  if (!in_kernel && rows_pad > rows) {
    const size_t esz = dtype == DH_BF16 ? 2 : 4, w = (size_t)heads * hd;
    if (hipMemsetAsync((char*)out + (size_t)rows * w * esz, 0, (size_t)(rows_pad - rows) * w * esz, (hipStream_t)stream) != hipSuccess)
      DH_FAIL(DH_ERR_LAUNCH, "dh_attn_varlen_fwd: memset failed");
  }
  return attn_fwd_impl(dtype, qkv, out, lse, b, Lmax, heads, hd, causal, cu_seqlens, stream, rows, in_kernel ? rows_pad : rows);
}

static int attn_bwd_impl(int dtype, const void* qkv, const void* out, const void* dout, const float* lse, void* dqkv, int b, int L,
                         int heads, int hd, int causal, const int* cu, dh_stream_t stream, int tail0 = 0, int tail1 = 0) {
  hipStream_t st = (hipStream_t)stream;
  DH_REQUIRE(qkv && out && dout && lse && dqkv && b > 0 && L > 0 && heads > 0, "dh_attn_bwd: bad args");
  DH_REQUIRE(L <= 128 && hd <= 64, "dh_attn_bwd: L<=128 and hd<=64 required (got %d, %d)", L, hd);
  const float scale = 1.0f / sqrtf((float)hd);
  if (dtype == DH_BF16 && hd == 64) {
    const int nkb = (L + 15) / 16;
#define CALL(N) launch_bwd_mfma<N>((const bf16_t*)qkv, (const bf16_t*)out, (const bf16_t*)dout, lse, (bf16_t*)dqkv, b, L, heads, causal, scale, st, cu, tail0, tail1)
    DISPATCH_NKB(nkb, CALL)
#undef CALL
  } else {
    size_t lds = (size_t)(4 * L * (hd + 1) + 2 * L * (L + 1) + L) * sizeof(float);
    DH_REQUIRE(lds <= 160 * 1024, "dh_attn_bwd: sequence too long for the fp32 / generic path (L <= 91 at hd = 64; the bf16 MFMA path takes L <= 128)");
    if (dtype == DH_BF16) {
      hipFuncSetAttribute((const void*)attn_bwd_generic_kernel<bf16_t>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      hipLaunchKernelGGL(attn_bwd_generic_kernel<bf16_t>, dim3(b * heads), dim3(256), lds, st, (const bf16_t*)qkv, (const bf16_t*)out, (const bf16_t*)dout, lse, (bf16_t*)dqkv, L, heads, hd, causal, scale, cu);
    } else {
      hipFuncSetAttribute((const void*)attn_bwd_generic_kernel<float>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      hipLaunchKernelGGL(attn_bwd_generic_kernel<float>, dim3(b * heads), dim3(256), lds, st, (const float*)qkv, (const float*)out, (const float*)dout, lse, (float*)dqkv, L, heads, hd, causal, scale, cu);
    }
  }
  DH_CHECK_LAUNCH();
  return DH_OK;
}

extern "C" int dh_attn_bwd(int dtype, const void* qkv, const void* out, const void* dout, const float* lse, void* dqkv,
                           int b, int L, int heads, int hd, int causal, dh_stream_t stream) {
  return attn_bwd_impl(dtype, qkv, out, dout, lse, dqkv, b, L, heads, hd, causal, nullptr, stream);
}
extern "C" int dh_attn_varlen_bwd(int dtype, const void* qkv, const void* out, const void* dout, const float* lse, void* dqkv,
                                  const int* cu_seqlens, int b, int Lmax, int heads, int hd, int causal, int rows, int rows_pad,
                                  dh_stream_t stream) {
  DH_REQUIRE(cu_seqlens, "dh_attn_varlen_bwd: cu_seqlens is NULL");
  const bool in_kernel = dtype == DH_BF16 && hd == 64;
  DH_REQUIRE((rows >= 0 && rows_pad >= rows) || (rows == -1 && in_kernel && rows_pad > 0),
             "dh_attn_varlen_bwd: rows %d, rows_pad %d (rows = -1, 'read cu_seqlens[b] on the device', needs the bf16 / hd = 64 kernels)", rows, rows_pad);
  if (!in_kernel && rows_pad > rows) {
    const size_t esz = dtype == DH_BF16 ? 2 : 4, w = (size_t)3 * heads * hd;
    if (hipMemsetAsync((char*)dqkv + (size_t)rows * w * esz, 0, (size_t)(rows_pad - rows) * w * esz, (hipStream_t)stream) != hipSuccess)
      DH_FAIL(DH_ERR_LAUNCH, "dh_attn_varlen_bwd: memset failed");
  }
  return attn_bwd_impl(dtype, qkv, out, dout, lse, dqkv, b, Lmax, heads, hd, causal, cu_seqlens, stream, rows, in_kernel ? rows_pad : rows);
}

// Length-bucketed packed attention (bf16, hd = 64): the sequences of `order[0 .. n_short)` are at most L_short rows long and run on
// the instantiation with ceil(L_short / 16) key blocks (fewer waves per workgroup, less LDS, more workgroups per CU); the others on
// the Lmax one.  order: int32 [b] (a permutation of the sequences, short ones first), ranges: int32 [4] = {0, n_short, n_short,
// b - n_short}, both in DEVICE memory (the bookkeeping of a packed batch: no host-side count enters a launch).  Measured on 512
// captions of 9..48 tokens (profiles/r03_small_kernels.txt): forward 38 -> 17 us, backward 57 -> 39 us through the 48-row kernels.
extern "C" int dh_attn_bucketed_fwd(int dtype, const void* qkv, void* out, float* lse, const int* cu_seqlens, const int* order, const int* ranges,
                                    int b, int Lmax, int L_short, int heads, int hd, int causal, int rows, int rows_pad, dh_stream_t stream) {
  hipStream_t st = (hipStream_t)stream;
  DH_REQUIRE(qkv && out && lse && cu_seqlens && order && ranges && b > 0 && heads > 0, "dh_attn_bucketed_fwd: bad args");
  DH_REQUIRE(dtype == DH_BF16 && hd == 64 && Lmax <= 128 && L_short > 0 && L_short <= Lmax, "dh_attn_bucketed_fwd: bf16, hd 64, 0 < L_short <= Lmax <= 128");
  DH_REQUIRE(rows == -1 || (rows >= 0 && rows_pad >= rows), "dh_attn_bucketed_fwd: rows %d, rows_pad %d", rows, rows_pad);
  const float scale = 1.0f / sqrtf((float)hd);
  const int L = Lmax;
  {
#define CALL(N) launch_fwd_mfma<N>((const bf16_t*)qkv, (bf16_t*)out, lse, b, Lmax, heads, causal, scale, st, cu_seqlens, 0, 0, order, ranges)
    DISPATCH_NKB((L_short + 15) / 16, CALL)
#undef CALL
  }
  {
#define CALL(N) launch_fwd_mfma<N>((const bf16_t*)qkv, (bf16_t*)out, lse, b, Lmax, heads, causal, scale, st, cu_seqlens, rows, rows_pad, order, ranges + 2)
    DISPATCH_NKB((Lmax + 15) / 16, CALL)
#undef CALL
  }
  DH_CHECK_LAUNCH();
  return DH_OK;
}
extern "C" int dh_attn_bucketed_bwd(int dtype, const void* qkv, const void* out, const void* dout, const float* lse, void* dqkv,
                                    const int* cu_seqlens, const int* order, const int* ranges, int b, int Lmax, int L_short, int heads, int hd,
                                    int causal, int rows, int rows_pad, dh_stream_t stream) {
  hipStream_t st = (hipStream_t)stream;
  DH_REQUIRE(qkv && out && dout && lse && dqkv && cu_seqlens && order && ranges && b > 0 && heads > 0, "dh_attn_bucketed_bwd: bad args");
  DH_REQUIRE(dtype == DH_BF16 && hd == 64 && Lmax <= 128 && L_short > 0 && L_short <= Lmax, "dh_attn_bucketed_bwd: bf16, hd 64, 0 < L_short <= Lmax <= 128");
  DH_REQUIRE(rows == -1 || (rows >= 0 && rows_pad >= rows), "dh_attn_bucketed_bwd: rows %d, rows_pad %d", rows, rows_pad);
  const float scale = 1.0f / sqrtf((float)hd);
  const int L = Lmax;
  {
#define CALL(N) launch_bwd_mfma<N>((const bf16_t*)qkv, (const bf16_t*)out, (const bf16_t*)dout, lse, (bf16_t*)dqkv, b, Lmax, heads, causal, scale, st, cu_seqlens, 0, 0, order, ranges)
    DISPATCH_NKB((L_short + 15) / 16, CALL)
#undef CALL
  }
  {
#define CALL(N) launch_bwd_mfma<N>((const bf16_t*)qkv, (const bf16_t*)out, (const bf16_t*)dout, lse, (bf16_t*)dqkv, b, Lmax, heads, causal, scale, st, cu_seqlens, rows, rows_pad, order, ranges + 2)
    DISPATCH_NKB((Lmax + 15) / 16, CALL)
#undef CALL
  }
  DH_CHECK_LAUNCH();
  return DH_OK;
}

// ------------------------------------------------------------------------------------------
// Pooled-query attention: ONE query per sequence (the row that is pooled: CLS of the vision tower, <|endoftext|> of the text
// tower).  In the LAST transformer block only that row's output is ever used (visual_transformer.py:70-72, text_transformer.py:203),
// so its query projection, attention, out_proj and MLP are needed for b rows instead of b*L -- K and V still come from every row.
// One wave per (sequence, head), hd == 64 == the wave: lane l owns key l (and l + 64) for the scores and output dim l for the
// weighted sums.  Keys of sequence i are kv rows row0[i] .. row0[i] + nkeys[i] - 1 (nkeys = position + 1 encodes the causal mask).
// ------------------------------------------------------------------------------------------
namespace {
// Round 5: 16-byte accesses throughout.  The first version read every key row element by element (64 scalar loads per lane and key)
// and walked the keys one at a time in the weighted sums (n iterations of 2-byte loads / stores per lane): 29 / 50 us per call at
// b = 512 on the critical path of a step (end of forward, start of backward, both towers at once).  Now: scores with 8 x ld8 per key row;
// weighted sums with lane = (key group lane >> 3, 8-column chunk lane & 7), 8 keys per iteration, partial sums combined by shuffles.
__device__ __forceinline__ float dot8(const float* a, const float* b) {
  return ((a[0] * b[0] + a[1] * b[1]) + (a[2] * b[2] + a[3] * b[3])) + ((a[4] * b[4] + a[5] * b[5]) + (a[6] * b[6] + a[7] * b[7]));
}
// sum over the 8 key groups (lanes that differ in bits 3..5)
__device__ __forceinline__ float kg_sum(float v) {
  v += __shfl_xor(v, 8, 64);
  v += __shfl_xor(v, 16, 64);
  v += __shfl_xor(v, 32, 64);
  return v;
}

template <typename T>
__global__ __launch_bounds__(256) void attn_pooled_fwd_kernel(const T* __restrict__ q, const T* __restrict__ kv, T* __restrict__ out,
                                                              float* __restrict__ lse, const int* __restrict__ row0,
                                                              const int* __restrict__ nkeys, int npairs, int heads, float scale) {
  __shared__ float ps[4][128];
  __shared__ __attribute__((aligned(16))) float qs[4][64];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int kg = lane >> 3, ch = lane & 7;
  const int d = heads * 64;
  for (int pair = blockIdx.x * 4 + wv; pair < npairs; pair += gridDim.x * 4) {
    const int bi = pair / heads, h = pair % heads;
    const long r0 = row0[bi];
    const int n = nkeys[bi];
    qs[wv][lane] = ld<T>(q + (long)bi * d + h * 64 + lane);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");   // LDS written by some lanes is read by others of the same wave
    __builtin_amdgcn_wave_barrier();
    float s[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int key = lane + 64 * j;
      float acc = -INFINITY;
      if (key < n) {
        acc = 0.f;
        const T* kr = kv + (r0 + key) * (2L * d) + h * 64;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
          float kf[8];
          ld8(kr + 8 * c, kf);
          acc += dot8(&qs[wv][8 * c], kf);
        }
        acc *= scale;
      }
      s[j] = acc;
    }
    const float mx = wave_max(fmaxf(s[0], s[1]));
    const float p0 = s[0] == -INFINITY ? 0.f : __expf(s[0] - mx), p1 = s[1] == -INFINITY ? 0.f : __expf(s[1] - mx);
    const float sum = wave_sum(p0 + p1);
    ps[wv][lane] = p0 / sum;
    ps[wv][lane + 64] = p1 / sum;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");   // LDS written by some lanes is read by others of the same wave
    __builtin_amdgcn_wave_barrier();
    float o[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int key = kg; key < n; key += 8) {
      float vf[8];
      ld8(kv + (r0 + key) * (2L * d) + d + h * 64 + 8 * ch, vf);
      const float pk = ps[wv][key];
#pragma unroll
      for (int x = 0; x < 8; ++x) o[x] += pk * vf[x];
    }
#pragma unroll
    for (int x = 0; x < 8; ++x) o[x] = kg_sum(o[x]);
    if (kg == 0) st8_fast(out + (long)bi * d + h * 64 + 8 * ch, o);
    if (lane == 0) lse[pair] = mx + __logf(sum);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");   // LDS written by some lanes is read by others of the same wave
    __builtin_amdgcn_wave_barrier();
  }
}

// dq [b][d]; dkv rows of the sequence's keys are WRITTEN (every (row, head) pair has exactly one owner).  total_rows > 0: the sequences
// are in row order (row0 ascending) and the rows between one sequence's keys and the next sequence (the last one: up to total_rows)
// are ZEROED here too, so the caller hands over an uninitialised dkv (no 45 / 78 MB fill launch per tower and step); total_rows = 0:
// rows outside the sequences stay as the caller initialised them
template <typename T>
__global__ __launch_bounds__(256) void attn_pooled_bwd_kernel(const T* __restrict__ q, const T* __restrict__ kv, const T* __restrict__ dout,
                                                              const float* __restrict__ lse, T* __restrict__ dq, T* __restrict__ dkv,
                                                              const int* __restrict__ row0, const int* __restrict__ nkeys, int npairs,
                                                              int heads, float scale, int total_rows) {
  __shared__ float ps[4][128], ds[4][128];
  __shared__ __attribute__((aligned(16))) float qs[4][64], gs[4][64];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int kg = lane >> 3, ch = lane & 7;
  const int d = heads * 64;
  for (int pair = blockIdx.x * 4 + wv; pair < npairs; pair += gridDim.x * 4) {
    const int bi = pair / heads, h = pair % heads;
    const long r0 = row0[bi];
    const int n = nkeys[bi];
    qs[wv][lane] = ld<T>(q + (long)bi * d + h * 64 + lane);
    gs[wv][lane] = ld<T>(dout + (long)bi * d + h * 64 + lane);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");   // LDS written by some lanes is read by others of the same wave
    __builtin_amdgcn_wave_barrier();
    const float l = lse[pair];
    float p[2], dp[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int key = lane + 64 * j;
      p[j] = 0.f; dp[j] = 0.f;
      if (key < n) {
        const T* kr = kv + (r0 + key) * (2L * d) + h * 64;
        float sc = 0.f, g = 0.f;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
          float kf[8], vf[8];
          ld8(kr + 8 * c, kf);
          ld8(kr + d + 8 * c, vf);
          sc += dot8(&qs[wv][8 * c], kf);
          g += dot8(&gs[wv][8 * c], vf);
        }
        p[j] = __expf(sc * scale - l);
        dp[j] = g;
      }
    }
    const float D = wave_sum(p[0] * dp[0] + p[1] * dp[1]);
#pragma unroll
    for (int j = 0; j < 2; ++j) { ps[wv][lane + 64 * j] = p[j]; ds[wv][lane + 64 * j] = p[j] * (dp[j] - D) * scale; }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");   // LDS written by some lanes is read by others of the same wave
    __builtin_amdgcn_wave_barrier();
    float aq[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    float qd[8], gd[8];
#pragma unroll
    for (int x = 0; x < 8; ++x) { qd[x] = qs[wv][8 * ch + x]; gd[x] = gs[wv][8 * ch + x]; }
    for (int key = kg; key < n; key += 8) {
      const long ro = (r0 + key) * (2L * d) + h * 64 + 8 * ch;
      float kf[8], ok[8], ov[8];
      ld8(kv + ro, kf);
      const float dsk = ds[wv][key], psk = ps[wv][key];
#pragma unroll
      for (int x = 0; x < 8; ++x) { aq[x] += dsk * kf[x]; ok[x] = dsk * qd[x]; ov[x] = psk * gd[x]; }
      st8_fast(dkv + ro, ok);                        // dK[key][8 ch ..]
      st8_fast(dkv + ro + d, ov);                    // dV[key][8 ch ..]
    }
    if (total_rows > 0) {
      const int span = (bi + 1 < npairs / heads ? row0[bi + 1] : total_rows) - (int)r0;       // rows up to the next sequence
      const float zero[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      for (int key = (bi == 0 ? -(int)r0 : n) + kg; key < span; key += 8) {      // (the first sequence also takes the rows before it)
        if (key >= 0 && key < n) continue;
        const long ro = (r0 + key) * (2L * d) + h * 64 + 8 * ch;
        st8_fast(dkv + ro, zero);
        st8_fast(dkv + ro + d, zero);
      }
    }
#pragma unroll
    for (int x = 0; x < 8; ++x) aq[x] = kg_sum(aq[x]);
    if (kg == 0) st8_fast(dq + (long)bi * d + h * 64 + 8 * ch, aq);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");   // LDS written by some lanes is read by others of the same wave
    __builtin_amdgcn_wave_barrier();
  }
}
}  // namespace

extern "C" int dh_attn_pooled_fwd(int dtype, const void* q, const void* kv, void* out, float* lse, const int* row0, const int* nkeys, int b,
                                  int heads, int hd, int Lmax, dh_stream_t stream) {
  hipStream_t st = (hipStream_t)stream;
  DH_REQUIRE(q && kv && out && lse && row0 && nkeys && b > 0 && heads > 0, "dh_attn_pooled_fwd: bad args");
  DH_REQUIRE(hd == 64 && Lmax >= 1 && Lmax <= 128, "dh_attn_pooled_fwd: head dim 64 and at most 128 keys per sequence (got %d, %d)", hd, Lmax);
  const int npairs = b * heads;
  int grid = dh_cdiv(npairs, 4);
  if (grid > 4096) grid = 4096;
  const float scale = 1.0f / sqrtf((float)hd);
  if (dtype == DH_BF16) hipLaunchKernelGGL(attn_pooled_fwd_kernel<bf16_t>, dim3(grid), dim3(256), 0, st, (const bf16_t*)q, (const bf16_t*)kv, (bf16_t*)out, lse, row0, nkeys, npairs, heads, scale);
  else if (dtype == DH_F32) hipLaunchKernelGGL(attn_pooled_fwd_kernel<float>, dim3(grid), dim3(256), 0, st, (const float*)q, (const float*)kv, (float*)out, lse, row0, nkeys, npairs, heads, scale);
  else DH_FAIL(DH_ERR_ARG, "dh_attn_pooled_fwd: bad dtype");
  DH_CHECK_LAUNCH();
  return DH_OK;
}

extern "C" int dh_attn_pooled_bwd(int dtype, const void* q, const void* kv, const void* dout, const float* lse, void* dq, void* dkv,
                                  const int* row0, const int* nkeys, int b, int heads, int hd, int Lmax, int total_rows, dh_stream_t stream) {
  hipStream_t st = (hipStream_t)stream;
  DH_REQUIRE(q && kv && dout && lse && dq && dkv && row0 && nkeys && b > 0 && heads > 0, "dh_attn_pooled_bwd: bad args");
  DH_REQUIRE(hd == 64 && Lmax >= 1 && Lmax <= 128, "dh_attn_pooled_bwd: head dim 64 and at most 128 keys per sequence (got %d, %d)", hd, Lmax);
  DH_REQUIRE(total_rows >= 0, "dh_attn_pooled_bwd: total_rows %d", total_rows);
  const int npairs = b * heads;
  int grid = dh_cdiv(npairs, 4);
  if (grid > 4096) grid = 4096;
  const float scale = 1.0f / sqrtf((float)hd);
  if (dtype == DH_BF16) hipLaunchKernelGGL(attn_pooled_bwd_kernel<bf16_t>, dim3(grid), dim3(256), 0, st, (const bf16_t*)q, (const bf16_t*)kv, (const bf16_t*)dout, lse, (bf16_t*)dq, (bf16_t*)dkv, row0, nkeys, npairs, heads, scale, total_rows);
  else if (dtype == DH_F32) hipLaunchKernelGGL(attn_pooled_bwd_kernel<float>, dim3(grid), dim3(256), 0, st, (const float*)q, (const float*)kv, (const float*)dout, lse, (float*)dq, (float*)dkv, row0, nkeys, npairs, heads, scale, total_rows);
  else DH_FAIL(DH_ERR_ARG, "dh_attn_pooled_bwd: bad dtype");
  DH_CHECK_LAUNCH();
  return DH_OK;
}
