// GEMM v5 for gfx950: 256 x 128 tiles, two accumulator sets, the epilogue of tile i inside the K loop of tile i+1.
//
// Why: the v4 kernel (gemm_v4.hip) stops the matrix pipe for every epilogue -- 9.5 k cycles for a bias-only bf16 tile against a main
// loop of 23-35 k cycles at K = 512 / 768 (the text tower, out_proj), 17-22 k for the GELU flavours.  A 256 x 256 fp32 tile is half
// of the CU's register file, so there is no room for a second one; a 256 x 128 tile (64 accumulator registers per wave) leaves room
// for TWO: while tile i+1 accumulates, tile i's values wait in the other set and are converted, staged through LDS and stored in
// the LOAD segments of tile i+1's first two K-tiles -- the segments in which a wave only issues LDS reads and waits, while its
// partner wave on the same SIMD owns the matrix pipe.
//
// Data path and schedule are v4's (same LDS images, same swizzles, same fragment reads, same MFMA):
//   * a K-tile (64 deep) = half-tiles A0, A1 (128 rows each) and B (128 columns), 16 KiB each; 2 stages = 96 KiB of ring;
//   * two phases per K-tile, 8 MFMAs (v_mfma_f32_32x32x16_bf16) per wave each: phase A = A0 x B, phase B = A1 x B; waves 4-7 run one
//     barrier behind waves 0-3;
//   * ROLES: waves 4-7 issue every LDS-DMA (and never store), waves 0-3 do every global store (and never wait on vmcnt in the
//     loop): a wave's vmcnt retires in order, so a wave with stores in flight could not wait for a younger load without waiting
//     for the stores; with the roles split the loaders' counted waits see loads only;
//   * the K-tiles of all items a workgroup processes form ONE stream: the half-tiles of the next item's first K-tiles are requested
//     while the current item's last K-tiles are multiplied, so the pipeline never drains between items;
//   * loader waits (4 DMA ops per half-tile and loader wave): phase A needs A1(kt) -> all but the 3 youngest half-tiles
//     (A0(kt+1), B(kt+1), A1(kt+1)): vmcnt(12); phase B needs A0(kt+1), B(kt+1) -> all but A1(kt+1), A0(kt+2), B(kt+2): vmcnt(12).
//   * epilogue of the previous item, spread over the load segments of K-tiles 0 and 1 of the current one: [kt 0, phase A] all waves
//     convert + bias + stage rows 0..127 (32 KiB staging tile behind the ring); [kt 0, phase B] waves 0-3 read the staged rows back
//     as 16-byte chunks and store them (+ residual); [kt 1, phase A / B] the same for rows 128..255.  The phase barriers of the K
//     loop order staging writes against the read-back (the wave groups are one barrier apart, see the timeline in the code).
//
// MEASURED (profiles/r02_gemm_v5_probe.txt, MI355X): results identical to v4 (bit-exact on most shapes, 1 bf16 ulp elsewhere),
// but 0.44-0.70 x v4's speed on the tower shapes (370-480 vs 530-900 TFLOP/s): a 256 x 128 tile needs 1.5 x the LDS-DMA
// instructions per flop of a 256 x 256 tile (48 vs 64 one-KiB pieces for half the MFMAs), and with the loader / storer roles the
// four loader waves issue 12 pieces per K-tile against 16 MFMAs -- at the ~100-185 cycles an LDS-DMA issue costs inside a loaded
// phase (MI355X_MICROARCH.md) the loaders need ~2300 cycles per K-tile against a matrix-pipe floor of 1030.  Hiding the epilogue
// cannot pay for a main loop at 45 % of the pipe.  What would: pieces issued by all eight waves (6 each; needs counted waits that
// see the epilogue's stores) or fewer M0 writes per piece; the model says parity with v4 at K = 768 and ~+13 % at K = 512 only.
// Kept opt-in (DH_GEMM_V5=1; =2 for K <= 1024 only) as the starting point of that work; v4 stays the production kernel.
//
// Scope: bf16 outputs with whole tiles (M % 256 == 0, N % 128 == 0), A [M][K] K-contiguous, B = weight [N][K] (forward) or
// contraction-major [K][N] (dX), optional bias, optional residual; no split-K, no tail slicing (everything else stays on v4).
#include "dh_common.h"
#include <stdlib.h>
#include <string.h>

namespace v5 {

typedef __attribute__((ext_vector_type(4))) uint32_t u32x4_t;
typedef __attribute__((ext_vector_type(4))) short s16x4;

constexpr int BM = 256, BN = 128, BK = 64;
constexpr int HALF_BYTES = 128 * BK * 2;          // 16 KiB
constexpr int STAGE_BYTES = 3 * HALF_BYTES;       // A0 A1 B
constexpr int BIAS_OFF = 2 * STAGE_BYTES;         // 96 KiB
constexpr int MAX_BIAS_N = 4096;
constexpr int CS_OFF = BIAS_OFF + MAX_BIAS_N * 4; // staging tile [128 rows][256 B]
constexpr int LDS_BYTES = CS_OFF + 128 * 256;     // 144 KiB

struct KA {
  const bf16_t* A; long lda;
  const bf16_t* B; long ldb;
  int M, N, K, ntx, nty, nitems;
  void* C; long ldc;
  const float* bias;
  const void* residual; long ldr;
};

__device__ __forceinline__ void dma16(const bf16_t* src, unsigned char* lds_dst_uniform) {
  const uint32_t l = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)lds_dst_uniform;
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(l), "v"(src) : "memory", "m0");
}
// same with a wave-uniform base (SGPR pair) + a 32-bit per-lane byte offset: half the address registers of the 64-bit form
__device__ __forceinline__ void dma16_so(const bf16_t* base_uniform, uint32_t off_bytes, unsigned char* lds_dst_uniform) {
  const uint32_t l = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)lds_dst_uniform;
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(l), "v"(off_bytes), "s"(base_uniform) : "memory", "m0");
}
template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ void wait_lgkm0() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

__device__ __forceinline__ bf16x8_t frag_kcontig(const unsigned char* tile, int r0, int s, int lane) {
  const int row = r0 + (lane & 31);
  const int chunk = 2 * s + (lane >> 5);
  return *reinterpret_cast<const bf16x8_t*>(tile + row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4));
}
__device__ __forceinline__ uint32_t kmajor_lane_off(int o0, int lane) {
  const int t = lane & 15;
  const int n = o0 + ((lane >> 4) & 1) * 16 + 4 * (t & 3);
  const int k = 8 * (lane >> 5) + (t >> 2);
  const int sw = (t >> 2) << 2;
  return k * 256 + ((((n >> 3) ^ sw)) << 4) + ((n & 7) << 1);
}
template <int OFF> __device__ __forceinline__ bf16x8_t frag_km(uint32_t addr) {
  s16x4 lo, hi;
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(lo) : "v"(addr), "n"(OFF));
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(hi) : "v"(addr), "n"(OFF + 1024));
  union { struct { s16x4 a, b; } s; bf16x8_t v; } u;
  u.s.a = lo; u.s.b = hi;
  return u.v;
}
// the 4 k16-step fragments of one 32-row / 32-column group of half-tile REGION (0 A0, 1 A1, 2 B)
template <bool KM, int REGION>
__device__ __forceinline__ void frag4(bf16x8_t* dst, uint32_t km_addr, const unsigned char* tile, int r0, int lane) {
  if (KM) {
    dst[0] = frag_km<REGION * 16384>(km_addr);
    dst[1] = frag_km<REGION * 16384 + 4096>(km_addr);
    dst[2] = frag_km<REGION * 16384 + 8192>(km_addr);
    dst[3] = frag_km<REGION * 16384 + 12288>(km_addr);
  } else {
#pragma unroll
    for (int s = 0; s < 4; ++s) dst[s] = frag_kcontig(tile, r0, s, lane);
  }
}
// DMA source of 1-KiB piece q (0..15) of a half-tile whose first out-row/column is o0
template <bool KM>
__device__ __forceinline__ const bf16_t* piece_src(const bf16_t* P, long ld, int q, int lane, int o0, int outs, long kbeg) {
  if (KM) {
    const int kl = q * 4 + (lane >> 4);
    const int c = (lane & 15) ^ ((kl & 3) << 2);
    int o = o0 + c * 8;
    o = o < outs ? o : 0;
    return P + (kbeg + kl) * ld + o;
  } else {
    const int rl = q * 8 + (lane >> 3);
    const int c = (lane & 7) ^ ((rl >> 1) & 7);
    int r = o0 + rl;
    r = r < outs ? r : outs - 1;
    return P + (long)r * ld + kbeg + c * 8;
  }
}

#define V5_BARRIER()                      \
  do {                                    \
    __builtin_amdgcn_sched_barrier(0);    \
    __builtin_amdgcn_s_barrier();         \
    __builtin_amdgcn_sched_barrier(0);    \
  } while (0)

__device__ __forceinline__ void chunk_of(int n, int x, int& start, int& len) {
  const int q = n >> 3, r = n & 7;
  len = q + (x < r ? 1 : 0);
  start = x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q;
}
__device__ __forceinline__ int wgs_on_xcd(int x, int grid) { return x < grid ? ((grid - 1 - x) >> 3) + 1 : 0; }

// position l of XCD x's contiguous chunk of the row-major tile list -> (tile_x, tile_y); false past the end
__device__ __forceinline__ bool decode(int x, int l, int nitems, int ntx, int& tx, int& ty) {
  int s0, len;
  chunk_of(nitems, x, s0, len);
  if (l >= len) return false;
  const int t = s0 + l;
  ty = t / ntx;
  tx = t - ty * ntx;
  return true;
}

template <bool TB, bool RES>
__global__ __launch_bounds__(512, 2) void gemm_v5_kernel(const KA ka) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int t = threadIdx.x;
  const int lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  const int ar = wm * 64, br = wn * 32;             // this wave's rows inside an A half / columns inside the B half
  const bool loader = wm == 1;                      // waves 4-7: all LDS-DMA; waves 0-3: all global stores
  unsigned char* const wdst = smem + (wave & 3) * 4096;
  const uint32_t lds0 = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) unsigned char*)smem;
  const bf16_t* A = ka.A;
  const bf16_t* B = ka.B;
  const long lda = ka.lda, ldb = ka.ldb;
  const int M = ka.M, N = ka.N, ntx = ka.ntx, nitems = ka.nitems;
  const int nk = ka.K / BK;                         // >= 2 (host)
  const long astep = BK;
  const long bstep = TB ? (long)BK * ldb : BK;
  const long a_dh = 128 * lda;

  // ---- this workgroup's items: positions my_l, my_l + step, ... of its XCD's chunk
  const int xcd = blockIdx.x & 7, grid = gridDim.x;
  const int my_l = blockIdx.x >> 3, lstep = wgs_on_xcd(xcd, grid);
  int chunk_s, chunk_len;
  chunk_of(nitems, xcd, chunk_s, chunk_len);
  const int n_my = my_l < chunk_len ? (chunk_len - 1 - my_l) / lstep + 1 : 0;
  if (n_my == 0) return;
  const long total = (long)n_my * nk;               // K-tiles of the whole stream

  // bias -> LDS once
  for (int q = t; q < N / 4; q += 512)
    *reinterpret_cast<float4*>(smem + BIAS_OFF + 16 * q) = ka.bias ? *reinterpret_cast<const float4*>(ka.bias + 4 * q) : make_float4(0.f, 0.f, 0.f, 0.f);

  // ---- issue side of the stream (loader waves): pointers of K-tile `js`
  // per-lane BYTE offsets (32-bit; operands are < 4 GiB) of the 4 pieces this wave stages, relative to the operand's base pointer
  uint32_t ap[4], bp[4];
  int issue_item = 0, issue_kt = 0;                 // item / K-tile of the half-tiles issued next
#define SETUP_SRC(item_)                                                                                   \
  do {                                                                                                     \
    int tx_, ty_;                                                                                          \
    decode(xcd, my_l + (item_) * lstep, nitems, ntx, tx_, ty_);                                            \
    _Pragma("unroll") for (int q = 0; q < 4; ++q) {                                                        \
      ap[q] = (uint32_t)((piece_src<false>(A, lda, (wave & 3) * 4 + q, lane, ty_ * BM, M, 0) - A) * 2);    \
      bp[q] = (uint32_t)((piece_src<TB>(B, ldb, (wave & 3) * 4 + q, lane, tx_ * BN, N, 0) - B) * 2);       \
    }                                                                                                      \
  } while (0)
#define ISSUE_H(BASE, P, OFFB, REGION, buf)                                                                \
  do {                                                                                                     \
    _Pragma("unroll") for (int q = 0; q < 4; ++q)                                                          \
      dma16_so((BASE), (P)[q] + (uint32_t)(OFFB), wdst + (buf) * STAGE_BYTES + (REGION) * HALF_BYTES + q * 1024); \
  } while (0)
  // advance the issue offsets to the next K-tile of the stream (re-deriving them at an item boundary)
#define ADVANCE_ISSUE()                                                                                    \
  do {                                                                                                     \
    if (++issue_kt == nk) {                                                                                \
      issue_kt = 0;                                                                                        \
      if (++issue_item < n_my) SETUP_SRC(issue_item);                                                      \
    } else {                                                                                               \
      _Pragma("unroll") for (int q = 0; q < 4; ++q) { ap[q] += (uint32_t)(astep * 2); bp[q] += (uint32_t)(bstep * 2); } \
    }                                                                                                      \
  } while (0)

  if (loader) {
    SETUP_SRC(0);
    ISSUE_H(A, ap, 0, 0, 0); ISSUE_H(B, bp, 0, 2, 0); ISSUE_H(A, ap, a_dh * 2, 1, 0);          // K-tile 0: A0, B, A1 -> stage 0
    ADVANCE_ISSUE();
    ISSUE_H(A, ap, 0, 0, 1); ISSUE_H(B, bp, 0, 2, 1);                                   // K-tile 1: A0, B -> stage 1 (its A1 follows in phase A)
    wait_vmcnt<12>();                                                             // A0, B of K-tile 0 have landed
  }
  __syncthreads();                                  // (also publishes the bias)
  if (wm == 1) V5_BARRIER();                        // waves 4-7 run one barrier behind waves 0-3 from here on

  const uint32_t akm0 = 0, akm1 = 0;                // (A is K-contiguous in every flavour of this kernel)
  (void)akm0; (void)akm1;
  const uint32_t bkm = lds0 + kmajor_lane_off(br, lane);
  f32x16_t acc[4], prev[4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc[i][r] = 0.f; prev[i][r] = 0.f; }
  bool have_prev = false;
  int pm0 = 0, pn0 = 0;                             // origin of the tile held in `prev`
  int cur_item = 0, kt = 0;
  int cm0, cn0;
  {
    int tx_, ty_;
    decode(xcd, my_l, nitems, ntx, tx_, ty_);
    cm0 = ty_ * BM; cn0 = tx_ * BN;
  }
  unsigned char* const Cs = smem + CS_OFF;

  // stage rows [half*128, half*128 + 128) of `prev` (+ bias) as bf16: staging row = 256 B, 8-byte unit u of row r at u ^ (r & 15).
  // Per-lane indexing restarts from an OPAQUE copy of the lane id inside each epilogue piece: otherwise the compiler hoists the
  // ~30 loop-invariant address registers out of the K-tile stream and keeps them live across the MFMA loop (spills).
  auto stage_half = [&](int half) {
    int lo = lane;
    asm volatile("" : "+v"(lo));
    const int h2 = lo >> 5;
#pragma unroll
    for (int ii = 0; ii < 2; ++ii) {
      const int ml = ar + ii * 32 + (lo & 31);
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) {
        const int nl = br + 8 * rg + 4 * h2;
        const float4 bq = *reinterpret_cast<const float4*>(smem + BIAS_OFF + 4 * (pn0 + nl));
        const f32x16_t& a = half == 0 ? prev[ii] : prev[2 + ii];
        uint2 pk;
        pk.x = pack2bf_hw(a[rg * 4 + 0] + bq.x, a[rg * 4 + 1] + bq.y);
        pk.y = pack2bf_hw(a[rg * 4 + 2] + bq.z, a[rg * 4 + 3] + bq.w);
        *reinterpret_cast<uint2*>(Cs + ml * 256 + (((nl >> 2) ^ (ml & 15)) << 3)) = pk;
      }
    }
  };
  // storer waves: staged rows -> global (16-byte chunks; odd rows have their 8-byte halves swapped by the unit swizzle).  Every
  // global access = wave-uniform row base (SGPRs) + one 32-bit per-lane byte offset: no per-row 64-bit address registers.
  auto store_half = [&](int half) {
    int to = t;
    asm volatile("" : "+v"(to));
    const int cc = to & 15, r0 = (to >> 4) & 15;      // 16-byte chunk cc of staged rows r0 + 16 * it
    unsigned char* Cb = reinterpret_cast<unsigned char*>(ka.C) + ((long)(pm0 + half * 128) * ka.ldc + pn0) * 2;
    const unsigned char* Rb = reinterpret_cast<const unsigned char*>(ka.residual) + ((long)(pm0 + half * 128) * ka.ldr + pn0) * 2;
    const uint32_t c_off = ((uint32_t)r0 * (uint32_t)ka.ldc + cc * 8) * 2;
    const uint32_t r_off = ((uint32_t)r0 * (uint32_t)ka.ldr + cc * 8) * 2;
#pragma unroll
    for (int it0 = 0; it0 < 8; it0 += 4) {
      uint4 res[4];
      if (RES) {
#pragma unroll
        for (int it = 0; it < 4; ++it) res[it] = *reinterpret_cast<const uint4*>(Rb + (long)(16 * (it0 + it)) * ka.ldr * 2 + r_off);
      }
#pragma unroll
      for (int itl = 0; itl < 4; ++itl) {
        const int it = it0 + itl;
        const int row = r0 + 16 * it;
        const int pc = cc ^ ((row & 15) >> 1);
        uint4 raw = *reinterpret_cast<const uint4*>(Cs + row * 256 + pc * 16);
        if (row & 1) { uint32_t tx = raw.x, ty = raw.y; raw.x = raw.z; raw.y = raw.w; raw.z = tx; raw.w = ty; }
        if (RES) {
          const uint32_t wv[4] = {raw.x, raw.y, raw.z, raw.w}, rv[4] = {res[itl].x, res[itl].y, res[itl].z, res[itl].w};
          uint32_t o[4];
#pragma unroll
          for (int x = 0; x < 4; ++x)
            o[x] = pack2bf_hw(__uint_as_float(wv[x] << 16) + __uint_as_float(rv[x] << 16),
                              __uint_as_float(wv[x] & 0xffff0000u) + __uint_as_float(rv[x] & 0xffff0000u));
          raw = make_uint4(o[0], o[1], o[2], o[3]);
        }
        u32x4_t w = {raw.x, raw.y, raw.z, raw.w};
        __builtin_nontemporal_store(w, reinterpret_cast<u32x4_t*>(Cb + (long)(16 * it) * ka.ldc * 2 + c_off));
      }
    }
  };

  // ---- the stream of K-tiles.  Timeline of one K-tile (global barrier numbers; waves 4-7 are one barrier behind):
  //   waves 0-3: [load A] b1 [mfma A] b2 [load B] b3 [mfma B] b4        waves 4-7: b1 [load A] b2 [mfma A] b3 [load B] b4 [mfma B] b5
  //   staging written in [load A] of both groups is complete at b2; waves 0-3 read it back in their [load B] (after b2); the next
  //   staging writes happen in the [load A] segments of the next K-tile (after b4): the read-back is finished by then.
  for (long s = 0; s < total; ++s) {
    const int buf = (int)(s & 1), nbuf = buf ^ 1;
    const bool has1 = s + 1 < total, has2 = s + 2 < total;
    const unsigned char* sA0 = smem + buf * STAGE_BYTES;
    const unsigned char* sA1 = sA0 + HALF_BYTES;
    const unsigned char* sB = sA0 + 2 * HALF_BYTES;
    const uint32_t boff = buf * STAGE_BYTES;
    bf16x8_t fa[2][4], fb[4];

    // ---------------- phase A: A0 x B
    frag4<TB, 2>(fb, bkm + boff, sB, br, lane);
    frag4<false, 0>(fa[0], 0, sA0, ar, lane);
    frag4<false, 0>(fa[1], 0, sA0, ar + 32, lane);
    if (loader) {
      if (has1) { ISSUE_H(A, ap, a_dh * 2, 1, nbuf); ADVANCE_ISSUE(); wait_vmcnt<12>(); }       // A1(s+1) requested; A1(s) has landed
      else wait_vmcnt<0>();
    }
    if (have_prev && kt < 2) { if (kt == 0) stage_half(0); else stage_half(1); }
    wait_lgkm0();
    V5_BARRIER();
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
      for (int ii = 0; ii < 2; ++ii) acc[ii] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[ks], fa[ii][ks], acc[ii], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
    V5_BARRIER();

    // ---------------- phase B: A1 x B
    frag4<false, 1>(fa[0], 0, sA1, ar, lane);
    frag4<false, 1>(fa[1], 0, sA1, ar + 32, lane);
    if (loader) {
      if (has2) { ISSUE_H(A, ap, 0, 0, buf); ISSUE_H(B, bp, 0, 2, buf); wait_vmcnt<12>(); }    // A0, B of s+2 requested; A0, B of s+1 have landed
      else if (has1) wait_vmcnt<4>();                                                     // only A1(s+1) may still fly
    }
    if (have_prev && kt < 2 && !loader) { if (kt == 0) store_half(0); else store_half(1); }
    wait_lgkm0();
    V5_BARRIER();
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
      for (int ii = 0; ii < 2; ++ii) acc[2 + ii] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[ks], fa[ii][ks], acc[2 + ii], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
    V5_BARRIER();

    // ---------------- end of an item: its accumulators move to `prev`, the epilogue rides on the next item's first K-tiles
    if (++kt == nk) {
      kt = 0;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        prev[i] = acc[i];
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
      }
      pm0 = cm0; pn0 = cn0;
      have_prev = true;
      if (++cur_item < n_my) {
        int tx_, ty_;
        decode(xcd, my_l + cur_item * lstep, nitems, ntx, tx_, ty_);
        cm0 = ty_ * BM; cn0 = tx_ * BN;
      }
    }
  }
  // ---- the last item's epilogue (nothing to hide it behind).  Re-align the wave groups first.
  if (wm == 0) V5_BARRIER();
#pragma unroll
  for (int half = 0; half < 2; ++half) {
    stage_half(half);
    wait_lgkm0();
    V5_BARRIER();
    if (!loader) store_half(half);
    wait_lgkm0();
    V5_BARRIER();
  }
#undef SETUP_SRC
#undef ISSUE_H
#undef ADVANCE_ISSUE
}

static int num_cus() {
  static int n = 0;
  if (!n) {
    int dev = 0;
    hipGetDevice(&dev);
    hipDeviceProp_t p;
    n = (hipGetDeviceProperties(&p, dev) == hipSuccess && p.multiProcessorCount > 0) ? p.multiProcessorCount : 256;
  }
  return n;
}

template <bool TB, bool RES>
void launch(const KA& ka, hipStream_t st) {
  static bool attr_set = false;
  if (!attr_set) {
    hipFuncSetAttribute((const void*)gemm_v5_kernel<TB, RES>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    attr_set = true;
  }
  int grid = num_cus();
  if (grid > ka.nitems) grid = ka.nitems;
  hipLaunchKernelGGL((gemm_v5_kernel<TB, RES>), dim3(grid), dim3(512), LDS_BYTES, st, ka);
}

}  // namespace v5

// Returns true if the v5 kernel took the problem (called from dh_gemm before v4).  DH_GEMM_V5=1 enables it.
bool dh_gemm_try_v5(const dh_gemm_args* a, hipStream_t st) {
  using namespace v5;
  static int on = -1;
  if (on < 0) { const char* ev = getenv("DH_GEMM_V5"); on = ev ? atoi(ev) : 0; }
  if (!on && a->force_generic != 5) return false;   // opt-in (measured slower than v4 on the tower shapes, see the header); tests force it
  if (a->dtype != DH_BF16 || a->c_dtype != DH_BF16 || a->a_kmajor || a->accumulate || a->a_colsum) return false;
  if (a->epilogue != DH_EPI_NONE || a->alpha != 1.f) return false;
  if ((a->M % BM) || (a->N % BN) || (a->K % BK) || a->K < 2 * BK || a->N > MAX_BIAS_N) return false;
  if ((a->lda % 8) || (a->ldb % 8) || ((uintptr_t)a->A & 15) || ((uintptr_t)a->B & 15)) return false;
  if (((uintptr_t)a->C & 15) || ((a->ldc * 2) & 15)) return false;
  if (a->residual && (((uintptr_t)a->residual & 15) || ((a->ldr * 2) & 15))) return false;
  if (a->bias && ((uintptr_t)a->bias & 15)) return false;
  if (on == 2 && a->K > 1024) return false;        // 2: only the short-K calls (where the epilogue weighs most)
  KA ka;
  ka.A = (const bf16_t*)a->A; ka.lda = a->lda; ka.B = (const bf16_t*)a->B; ka.ldb = a->ldb;
  ka.M = a->M; ka.N = a->N; ka.K = a->K; ka.ntx = a->N / BN; ka.nty = a->M / BM; ka.nitems = ka.ntx * ka.nty;
  ka.C = a->C; ka.ldc = a->ldc; ka.bias = a->bias; ka.residual = a->residual; ka.ldr = a->ldr;
  if (a->b_kmajor) { if (a->residual) launch<true, true>(ka, st); else launch<true, false>(ka, st); }
  else { if (a->residual) launch<false, true>(ka, st); else launch<false, false>(ka, st); }
  return true;
}
