// GEMM v4 for gfx950: 256 x 256 x 64 block tile, 8 waves, "ping-pong" phase schedule.
//
// Why another GEMM: the v2 kernel (128 x 128, one drain-everything barrier per K-step) tops out at
// ~650-900 TFLOP/s on the tower shapes -- the structure's ceiling (cdna_hip_programming.md s5).  v4 keeps the
// v2 DATA PATH (HBM -> LDS by LDS-DMA with the swizzle on the source address, fragments by ds_read_b128 or
// ds_read_b64_tr_b16, no transposed operand copies anywhere) and changes the SCHEDULE:
//
//   * the 256 x 256 C tile is four 128 x 128 quadrants (A-half i) x (B-half j); the 8 waves form a 2 x 4 grid
//     and every wave owns a 64 x 32 piece of EVERY quadrant (acc[4][2] 32x32 fragments = 128 VGPRs);
//   * a K-tile (64 deep) is 4 phases, one quadrant each: (A0,B0) (A0,B1) (A1,B1) (A1,B0); a phase is a
//     LOAD segment (fragment ds_reads for the quadrant + one 16 KiB half-tile of LDS-DMA for a later K-tile)
//     and an MFMA segment (8 x v_mfma_f32_32x32x16_bf16), separated by raw s_barriers;
//   * waves 4-7 run ONE barrier behind waves 0-3, so on every SIMD one wave is in its MFMA segment while its
//     partner is in its load segment: the matrix pipe and the LDS/DMA path stay busy at the same time;
//   * each half-tile (A0/A1/B0/B1 of a K-tile) is consumed in exactly one phase, so it is re-filled two phases
//     later for K-tile t+2: 4 half-tiles (64 KiB) are always in flight and the waits are COUNTED
//     (s_waitcnt vmcnt(8): "everything but the 4 youngest half-tiles"), never a drain;
//   * bf16 outputs: the MFMAs are issued with swapped operands (acc holds C^T fragments: a lane owns 4
//     consecutive columns of one row), so the accumulators are packed to bf16 and staged in LDS with
//     ds_write_b64, then written out as 16-byte vectors on full 512-byte row segments; alpha/bias are applied
//     in fp32 before the packing, GELU / dGELU / residual on the staged value;
//   * fp32 accumulate (dW, split-K): plain orientation, atomics straight from the accumulators (a lane group
//     covers 32 consecutive floats of a row).
//
// LDS: 2 stages x {A0, A1, B0, B1} x 16 KiB = 128 KiB (1 block / CU); the epilogue reuses it as the 256 x 256
// bf16 staging tile.  Half-tile images are exactly the v2 operand tiles:
//   K-contiguous  [128 rows][64 k]   128-B rows, 16-B chunk c of row r at slot c ^ ((r>>1)&7)
//   contraction-major [64 k][128 out] 256-B rows, 16-B chunk c of row k at slot c ^ ((k&3)<<2)
#include "dh_common.h"
#include <stdlib.h>

#ifndef V4_ABL
#define V4_ABL 0
#endif

namespace v4 {

struct EpiParams {
  int M, N;
  void* C; long ldc;
  const float* bias;
  int epilogue;
  const void* residual; long ldr;
  void* aux; long ldaux;
  float alpha;
  float* a_colsum;
  int delay;   // experiment: DH_V4_DELAY shader cycles of start-up delay for every other first-round CU
};

constexpr int BM = 256, BN = 256, BK = 64;
constexpr int HALF_BYTES = 128 * BK * 2;          // 16 KiB
constexpr int STAGE_BYTES = 4 * HALF_BYTES;       // A0 A1 B0 B1
constexpr int LDS_BYTES = 2 * STAGE_BYTES;        // 128 KiB; epilogue staging = 256 rows x 512 B = the same 128 KiB

typedef __attribute__((ext_vector_type(4))) short s16x4;

__device__ __forceinline__ void dma16(const bf16_t* src, unsigned char* lds_dst_uniform) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                   (__attribute__((address_space(3))) void*)lds_dst_uniform, 16, 0, 0);
}
template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// fragment of a K-contiguous half-tile: rows r0 + (lane&31), k = 16*s + 8*(lane>>5) .. +7
__device__ __forceinline__ bf16x8_t frag_kcontig(const unsigned char* tile, int r0, int s, int lane) {
  const int row = r0 + (lane & 31);
  const int chunk = 2 * s + (lane >> 5);
  return *reinterpret_cast<const bf16x8_t*>(tile + row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4));
}
// fragment of a contraction-major half-tile via the transpose read: out columns o0 + (lane&31)
__device__ __forceinline__ bf16x8_t frag_kmajor(const unsigned char* tile, int o0, int s, int lane) {
  constexpr int ROWB = 256;
  const int t = lane & 15;
  const int n = o0 + ((lane >> 4) & 1) * 16 + 4 * (t & 3);
  const int k = 16 * s + 8 * (lane >> 5) + (t >> 2);
  const int sw = (t >> 2) << 2;
  const unsigned char* p = tile + k * ROWB + ((((n >> 3) ^ sw)) << 4) + ((n & 7) << 1);
  s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p));
  s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p + 4 * ROWB));
  union { struct { s16x4 a, b; } s; bf16x8_t v; } u;
  u.s.a = lo; u.s.b = hi;
  return u.v;
}
template <bool KM> __device__ __forceinline__ bf16x8_t frag(const unsigned char* tile, int o0, int s, int lane) {
  return KM ? frag_kmajor(tile, o0, s, lane) : frag_kcontig(tile, o0, s, lane);
}

// DMA source of 1-KiB piece q (0..15) of a half-tile whose first out-row/column is o0
template <bool KM>
__device__ __forceinline__ const bf16_t* piece_src(const bf16_t* P, long ld, int q, int lane, int o0, int outs, int kbeg) {
  if (KM) {
    const int kl = q * 4 + (lane >> 4);
    const int c = (lane & 15) ^ ((kl & 3) << 2);
    int o = o0 + c * 8;
    o = o < outs ? o : 0;
    return P + (long)(kbeg + kl) * ld + o;
  } else {
    const int rl = q * 8 + (lane >> 3);
    const int c = (lane & 7) ^ ((rl >> 1) & 7);
    int r = o0 + rl;
    r = r < outs ? r : outs - 1;
    return P + (long)r * ld + kbeg + c * 8;
  }
}

#define V4_BARRIER()                      \
  do {                                    \
    __builtin_amdgcn_sched_barrier(0);    \
    __builtin_amdgcn_s_barrier();         \
    __builtin_amdgcn_sched_barrier(0);    \
  } while (0)

// ACC = false: bf16 C with fused epilogue (swapped MFMA operands); ACC = true: fp32 atomic accumulate
template <bool TA, bool TB, bool ACC>
__global__ __launch_bounds__(512, 2) void gemm_v4_kernel(const bf16_t* __restrict__ A, long lda, const bf16_t* __restrict__ B,
                                                         long ldb, int M, int N, int K, int k_per_split, EpiParams e) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int t = threadIdx.x;
  const int lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  int tile_x, tile_y;
  {   // XCD-aware, L2-grouped tile order (block b runs on XCD b % 8; see gemm_glds.hip)
    const int ntx = gridDim.x, nb = gridDim.x * gridDim.y;
    const int b = blockIdx.y * ntx + blockIdx.x;
    const int q = nb >> 3, r = nb & 7, xcd = b & 7, idx = b >> 3;
    const int logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    constexpr int GROUP_M = 8;
    const int nty = gridDim.y;
    const int in_group = GROUP_M * ntx;
    const int gid = logical / in_group;
    const int first_m = gid * GROUP_M;
    const int gsz = min(nty - first_m, GROUP_M);
    const int rem = logical - gid * in_group;
    tile_y = first_m + rem % gsz;
    tile_x = rem / gsz;
  }
  if (e.delay > 0) {
    const int b = (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
    if (b < 256 && ((b >> 3) & 1)) {
      const long t0 = __builtin_readcyclecounter();
      while (__builtin_readcyclecounter() - t0 < e.delay) __builtin_amdgcn_s_sleep(8);
    }
  }
  const int m0 = tile_y * BM, n0 = tile_x * BN;
  const int kbeg = blockIdx.z * k_per_split;
  const int kend = min(K, kbeg + k_per_split);
  const int nk = (kend - kbeg) / BK;                 // host guarantees divisibility and nk >= 1

  // DMA sources: [half][piece]; wave w stages pieces 2w, 2w+1 of every half-tile
  const bf16_t* a00 = piece_src<TA>(A, lda, wave * 2 + 0, lane, m0, M, kbeg);
  const bf16_t* a01 = piece_src<TA>(A, lda, wave * 2 + 1, lane, m0, M, kbeg);
  const bf16_t* a10 = piece_src<TA>(A, lda, wave * 2 + 0, lane, m0 + 128, M, kbeg);
  const bf16_t* a11 = piece_src<TA>(A, lda, wave * 2 + 1, lane, m0 + 128, M, kbeg);
  const bf16_t* b00 = piece_src<TB>(B, ldb, wave * 2 + 0, lane, n0, N, kbeg);
  const bf16_t* b01 = piece_src<TB>(B, ldb, wave * 2 + 1, lane, n0, N, kbeg);
  const bf16_t* b10 = piece_src<TB>(B, ldb, wave * 2 + 0, lane, n0 + 128, N, kbeg);
  const bf16_t* b11 = piece_src<TB>(B, ldb, wave * 2 + 1, lane, n0 + 128, N, kbeg);
  const long astep = TA ? (long)BK * lda : BK;
  const long bstep = TB ? (long)BK * ldb : BK;
  unsigned char* const wdst = smem + wave * 2048;    // this wave's 2 KiB slice of a half-tile
  // compile-time ablation (tuning aid: -DV4_ABL=mask; 1 no DMA, 2 no MFMA, 4 no fragment reads, 8 no epilogue)
  constexpr bool do_dma = !(V4_ABL & 1), do_mfma = !(V4_ABL & 2), do_frag = !(V4_ABL & 4), do_epi = !(V4_ABL & 8);

#define ISSUE_A0(buf) do { if (!do_dma) break; dma16(a00, wdst + (buf) * STAGE_BYTES);                  dma16(a01, wdst + (buf) * STAGE_BYTES + 1024);                  a00 += astep; a01 += astep; } while (0)
#define ISSUE_A1(buf) do { if (!do_dma) break; dma16(a10, wdst + (buf) * STAGE_BYTES + HALF_BYTES);     dma16(a11, wdst + (buf) * STAGE_BYTES + HALF_BYTES + 1024);     a10 += astep; a11 += astep; } while (0)
#define ISSUE_B0(buf) do { if (!do_dma) break; dma16(b00, wdst + (buf) * STAGE_BYTES + 2 * HALF_BYTES); dma16(b01, wdst + (buf) * STAGE_BYTES + 2 * HALF_BYTES + 1024); b00 += bstep; b01 += bstep; } while (0)
#define ISSUE_B1(buf) do { if (!do_dma) break; dma16(b10, wdst + (buf) * STAGE_BYTES + 3 * HALF_BYTES); dma16(b11, wdst + (buf) * STAGE_BYTES + 3 * HALF_BYTES + 1024); b10 += bstep; b11 += bstep; } while (0)

  f32x16_t acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // ---- prologue: FIFO order matches the steady state (A0 B0 B1 A1 of tile 0, then A0 B0 of tile 1)
  ISSUE_A0(0); ISSUE_B0(0); ISSUE_B1(0); ISSUE_A1(0);
  if (nk > 1) { ISSUE_A0(1); ISSUE_B0(1); wait_vmcnt<8>(); } else { wait_vmcnt<4>(); }
  V4_BARRIER();
  if (wm == 1) V4_BARRIER();             // waves 4-7 run one barrier behind waves 0-3 from here on

  const int ar = wm * 64, br = wn * 32;  // this wave's rows inside an A half / columns inside a B half

#define MFMA(ACCV, AF, BF) \
  if (!do_mfma) { asm volatile("" ::"v"(AF), "v"(BF)); } else ACCV = ACC ? __builtin_amdgcn_mfma_f32_32x32x16_bf16(AF, BF, ACCV, 0, 0, 0) : __builtin_amdgcn_mfma_f32_32x32x16_bf16(BF, AF, ACCV, 0, 0, 0)

  for (int kt = 0; kt < nk; ++kt) {
    const int buf = kt & 1, nbuf = buf ^ 1;
    const bool has1 = kt + 1 < nk, has2 = kt + 2 < nk;
    const unsigned char* sA0 = smem + buf * STAGE_BYTES;
    const unsigned char* sA1 = sA0 + HALF_BYTES;
    const unsigned char* sB0 = sA0 + 2 * HALF_BYTES;
    const unsigned char* sB1 = sA0 + 3 * HALF_BYTES;
    bf16x8_t fa0[2][4], fa1[2][4], fb0[4], fb1[4];
    if (!do_frag) {
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        fb0[s] = bf16x8_t{}; fb1[s] = bf16x8_t{}; fa0[0][s] = bf16x8_t{}; fa0[1][s] = bf16x8_t{}; fa1[0][s] = bf16x8_t{}; fa1[1][s] = bf16x8_t{};
        asm volatile("" : "+v"(fb0[s]), "+v"(fb1[s]), "+v"(fa0[0][s]), "+v"(fa0[1][s]), "+v"(fa1[0][s]), "+v"(fa1[1][s]));
      }
    }

    // ---- phase 1: quadrant (A0, B0)
#pragma unroll
    for (int s = 0; s < 4; ++s) if (do_frag) fb0[s] = frag<TB>(sB0, br, s, lane);
#pragma unroll
    for (int ii = 0; ii < 2; ++ii)
#pragma unroll
      for (int s = 0; s < 4; ++s) if (do_frag) fa0[ii][s] = frag<TA>(sA0, ar + ii * 32, s, lane);
    if (has1) { ISSUE_B1(nbuf); wait_vmcnt<8>(); } else { wait_vmcnt<2>(); }      // B1(kt) has landed
    V4_BARRIER();
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int ii = 0; ii < 2; ++ii) MFMA(acc[ii][0], fa0[ii][s], fb0[s]);
    __builtin_amdgcn_s_setprio(0);
    V4_BARRIER();

    // ---- phase 2: quadrant (A0, B1)
#pragma unroll
    for (int s = 0; s < 4; ++s) if (do_frag) fb1[s] = frag<TB>(sB1, br, s, lane);
    if (has1) { ISSUE_A1(nbuf); wait_vmcnt<8>(); } else { wait_vmcnt<0>(); }      // A1(kt) has landed
    V4_BARRIER();
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int ii = 0; ii < 2; ++ii) MFMA(acc[ii][1], fa0[ii][s], fb1[s]);
    __builtin_amdgcn_s_setprio(0);
    V4_BARRIER();

    // ---- phase 3: quadrant (A1, B1)
#pragma unroll
    for (int ii = 0; ii < 2; ++ii)
#pragma unroll
      for (int s = 0; s < 4; ++s) if (do_frag) fa1[ii][s] = frag<TA>(sA1, ar + ii * 32, s, lane);
    if (has2) ISSUE_A0(buf);                                                      // A0(kt+2): A0(kt) was read in phase 1
    V4_BARRIER();
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int ii = 0; ii < 2; ++ii) MFMA(acc[2 + ii][1], fa1[ii][s], fb1[s]);
    __builtin_amdgcn_s_setprio(0);
    V4_BARRIER();

    // ---- phase 4: quadrant (A1, B0); no fragment reads
    if (has2) { ISSUE_B0(buf); wait_vmcnt<8>(); } else if (has1) { wait_vmcnt<4>(); }   // A0, B0 of tile kt+1 have landed
    V4_BARRIER();
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int ii = 0; ii < 2; ++ii) MFMA(acc[2 + ii][0], fa1[ii][s], fb0[s]);
    __builtin_amdgcn_s_setprio(0);
    V4_BARRIER();
  }
  if (wm == 0) V4_BARRIER();             // re-align the two wave groups
#undef MFMA
#undef ISSUE_A0
#undef ISSUE_A1
#undef ISSUE_B0
#undef ISSUE_B1

  if (!do_epi) {
    if (acc[0][0][0] == 123.456f) reinterpret_cast<float*>(e.C)[0] = acc[1][1][3] + acc[2][0][5] + acc[3][1][7];
    return;
  }
  if (ACC) {
    // split-K / dW: atomics straight from the accumulators (lanes 0..31 = 32 consecutive floats of a row)
    float* Cf = reinterpret_cast<float*>(e.C);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int ii = 0; ii < 2; ++ii)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int m = m0 + i * 128 + ar + ii * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            const int n = n0 + j * 128 + br + (lane & 31);
            if (m < M && n < N) atomicAdd(Cf + (long)m * e.ldc + n, acc[i * 2 + ii][j][r] * e.alpha);
          }
    return;
  }

  // ---- bf16 epilogue.  acc[i*2+ii][j] holds a C^T fragment: lane -> row m = (lane&31), registers 4*rg .. 4*rg+3
  // -> 4 consecutive columns n = 8*rg + 4*(lane>>5) + {0..3}.  Staging tile: [256 rows][512 B], 8-byte unit u of
  // row m stored at unit u ^ (m & 15)  (conflict-free ds_write_b64; ds_read_b128 sees whole 16-B chunks).
  unsigned char* Cs = smem;
  float4 bv[2][4];
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int rg = 0; rg < 4; ++rg) {
      bv[j][rg] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (e.bias) {
        int nb = n0 + j * 128 + br + 8 * rg + 4 * (lane >> 5);
        nb = nb + 3 < N ? nb : 0;
        bv[j][rg] = *reinterpret_cast<const float4*>(e.bias + nb);
      }
    }
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int ii = 0; ii < 2; ++ii) {
      const int ml = i * 128 + ar + ii * 32 + (lane & 31);
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
          const int nl = j * 128 + br + 8 * rg + 4 * (lane >> 5);
          uint2 pk;
          pk.x = pack2bf_hw(acc[i * 2 + ii][j][rg * 4 + 0] * e.alpha + bv[j][rg].x, acc[i * 2 + ii][j][rg * 4 + 1] * e.alpha + bv[j][rg].y);
          pk.y = pack2bf_hw(acc[i * 2 + ii][j][rg * 4 + 2] * e.alpha + bv[j][rg].z, acc[i * 2 + ii][j][rg * 4 + 3] * e.alpha + bv[j][rg].w);
          const int u = (nl >> 2) ^ (ml & 15);
          *reinterpret_cast<uint2*>(Cs + ml * 512 + u * 8) = pk;
        }
    }
  __syncthreads();
  {
    const int cc = t & 31;                 // 16-byte chunk of the row (8 columns)
    const int r0 = t >> 5;                 // 0..15
    const int n = n0 + cc * 8;
    bf16_t* Cb = reinterpret_cast<bf16_t*>(e.C);
    if (n < N) {
#pragma unroll 4
      for (int it = 0; it < 16; ++it) {
        const int row = r0 + 16 * it;
        const int m = m0 + row;
        if (m < M) {
          const int pc = cc ^ ((row & 15) >> 1);
          uint4 raw = *reinterpret_cast<const uint4*>(Cs + row * 512 + pc * 16);
          if (row & 1) { uint32_t tx = raw.x, ty = raw.y; raw.x = raw.z; raw.y = raw.w; raw.z = tx; raw.w = ty; }
          if (e.epilogue == DH_EPI_NONE && !e.residual) {
            *reinterpret_cast<uint4*>(Cb + (long)m * e.ldc + n) = raw;
          } else {
            const uint32_t w[4] = {raw.x, raw.y, raw.z, raw.w};
            float v[8];
#pragma unroll
            for (int x = 0; x < 4; ++x) { v[2 * x] = __uint_as_float(w[x] << 16); v[2 * x + 1] = __uint_as_float(w[x] & 0xffff0000u); }
            if (e.epilogue == DH_EPI_GELU) {
              if (e.aux) *reinterpret_cast<uint4*>(reinterpret_cast<bf16_t*>(e.aux) + (long)m * e.ldaux + n) = raw;
#pragma unroll
              for (int x = 0; x < 8; ++x) v[x] = quick_gelu_f(v[x]);
            } else if (e.epilogue == DH_EPI_DGELU) {
              float uu[8];
              ld8(reinterpret_cast<const bf16_t*>(e.aux) + (long)m * e.ldaux + n, uu);
#pragma unroll
              for (int x = 0; x < 8; ++x) v[x] *= quick_gelu_grad_f(uu[x]);
            }
            if (e.residual) {
              float rr[8];
              ld8(reinterpret_cast<const bf16_t*>(e.residual) + (long)m * e.ldr + n, rr);
#pragma unroll
              for (int x = 0; x < 8; ++x) v[x] += rr[x];
            }
            st8_hw(Cb + (long)m * e.ldc + n, v);
          }
        }
      }
    }
  }
}

template <bool TA, bool TB, bool ACC>
void launch(const dh_gemm_args* a, const EpiParams& e, int split, int kps, hipStream_t st) {
  static bool attr_set = false;
  if (!attr_set) {
    hipFuncSetAttribute((const void*)gemm_v4_kernel<TA, TB, ACC>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    attr_set = true;
  }
  dim3 grid(dh_cdiv(a->N, BN), dh_cdiv(a->M, BM), split);
  hipLaunchKernelGGL((gemm_v4_kernel<TA, TB, ACC>), grid, dim3(512), LDS_BYTES, st, (const bf16_t*)a->A, (long)a->lda,
                     (const bf16_t*)a->B, (long)a->ldb, a->M, a->N, a->K, kps, e);
}

}  // namespace v4

// Returns true if the v4 kernel took the problem (called from dh_gemm first).  DH_GEMM_V4=0 disables it.
bool dh_gemm_try_v4(const dh_gemm_args* a, int split, hipStream_t st) {
  using namespace v4;
  static int mode = -2;
  if (mode == -2) { const char* ev = getenv("DH_GEMM_V4"); mode = ev ? atoi(ev) : -1; }
  if (mode == 0 && a->force_generic != 4) return false;
  if (a->dtype != DH_BF16) return false;
  if (a->a_colsum) return false;
  if ((a->lda % 8) || (a->ldb % 8) || ((uintptr_t)a->A & 15) || ((uintptr_t)a->B & 15)) return false;
  if (a->K % BK) return false;
  // operand extents: the DMA reads whole 16-byte chunks; out-of-range rows/columns are clamped (their products
  // land in rows/columns that are never stored), but a contraction-major operand needs whole 8-element chunks
  if (a->a_kmajor && (a->M % 8)) return false;
  if (a->b_kmajor && (a->N % 8)) return false;
  if (a->force_generic != 4 && (a->M < 256 || a->N < 256)) return false;
  int kps = ((a->K + split - 1) / split + BK - 1) / BK * BK;
  split = (a->K + kps - 1) / kps;
  if (!a->accumulate) {
    if (a->c_dtype != DH_BF16) return false;
    if (a->N % 8) return false;
    if (((uintptr_t)a->C & 15) || ((a->ldc * 2) & 15)) return false;
    if (a->residual && (((uintptr_t)a->residual & 15) || ((a->ldr * 2) & 15))) return false;
    if (a->aux && (((uintptr_t)a->aux & 15) || ((a->ldaux * 2) & 15))) return false;
    if (a->bias && ((uintptr_t)a->bias & 15)) return false;
  }
  EpiParams e;
  e.M = a->M; e.N = a->N; e.C = a->C; e.ldc = a->ldc; e.bias = a->bias; e.epilogue = a->epilogue;
  e.residual = a->residual; e.ldr = a->ldr; e.aux = a->aux; e.ldaux = a->ldaux; e.alpha = a->alpha;
  e.a_colsum = a->a_colsum;
  static int delay = -1;
  if (delay < 0) { const char* ev = getenv("DH_V4_DELAY"); delay = ev ? atoi(ev) : 0; }
  e.delay = delay;
#define V4_LAUNCH(TA, TB)                                                   \
  do {                                                                      \
    if (a->accumulate) launch<TA, TB, true>(a, e, split, kps, st);          \
    else launch<TA, TB, false>(a, e, split, kps, st);                       \
  } while (0)
  if (a->a_kmajor && a->b_kmajor) V4_LAUNCH(true, true);
  else if (a->a_kmajor) V4_LAUNCH(true, false);
  else if (a->b_kmajor) V4_LAUNCH(false, true);
  else V4_LAUNCH(false, false);
#undef V4_LAUNCH
  return true;
}
