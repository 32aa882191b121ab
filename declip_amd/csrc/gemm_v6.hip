// GEMM v6 for gfx950 (round 5, EXPERIMENT behind DH_GEMM_V6=1): TWO 4-wave workgroups per CU on 128 x 256 x 32 tiles.
//
// Why: gemm_v4 (one 8-wave workgroup per CU, 256 x 256 tiles) pays its epilogue on the critical path of every tile -- ~3.2 ms of the
// 20.4 ms of GEMM time in a CLIP step (5.6 k cycles of a 36 k-cycle K = 768 tile, 14 k for the GELU flavour): the accumulators fill
// half the register file, so nothing of the next tile can start before they are stored.  VERDICT r4 #1b asks for the epilogue off the
// critical path.  This kernel gets there with what the hardware schedules by itself: two INDEPENDENT workgroups share a CU (one wave
// of each per SIMD, 256 VGPRs each, 74 KB of LDS each); while one converts / stages / stores its tile, the other one's MFMAs have the
// matrix pipe to themselves, and when both are in their main loops they alternate on it exactly like v4's two wave groups do -- but
// without a barrier coupling them.  Price: a 128 x 256 tile moves 1.5 x the operand bytes per MFMA through the LDS-DMA (24 KB per
// 128 x 256 x 32 step against 64 KB per 256 x 256 x 64), the resource v4's in-kernel trace shows close to saturation.
//
// Data path = v4's: operands HBM -> LDS by LDS-DMA (global_load_lds_dwordx4, 1 KiB per wave-instruction, swizzle on the SOURCE
// address), fragments for v_mfma_f32_32x32x16_bf16 by ds_read_b128, accumulators as C^T fragments (swapped operands: a lane owns 4
// consecutive columns of a row), bf16 epilogue staged through LDS into 16-byte stores on full 512-byte row segments.
//   LDS per workgroup: 3 stages x (A 128 x 32 + B 256 x 32) bf16 = 72 KiB + 1 KiB bias of the tile.  K-step kt of the workgroup's
//   continuous K-step stream (across its tiles) lives in stage kt % 3 and is requested two steps ahead; one barrier per step.
//   K-contiguous tile image: [rows][64 B], 16-byte chunk c of row r at slot c ^ ((r >> 2) & 3)  (conflict-free ds_read_b128).
// Stage 1 (this file): forward layout only (A [M][K], B [N][K], both K-contiguous), epilogues bias / bias + residual.
#include "dh_common.h"
#include <stdlib.h>
#include <string.h>

namespace v6 {

constexpr int BM = 128, BN = 256, BK = 32, NST = 3;
constexpr int A_BYTES = BM * BK * 2;                  // 8 KiB
constexpr int B_BYTES = BN * BK * 2;                  // 16 KiB
constexpr int STAGE_BYTES = A_BYTES + B_BYTES;        // 24 KiB
constexpr int BIAS_OFF = NST * STAGE_BYTES;           // 72 KiB
constexpr int LDS_BYTES = BIAS_OFF + BN * 4;          // + 1 KiB

struct Args {
  const bf16_t* A; long lda;
  const bf16_t* B; long ldb;
  bf16_t* C; long ldc;
  const float* bias;
  const bf16_t* residual; long ldr;
  bf16_t* aux; long ldaux;                 // EPI 1: QuickGELU' output; EPI 2: the factor read
  int M, N, K, ntx, nty;
  float alpha;
};

__device__ __forceinline__ void dma16(const bf16_t* base, uint32_t voff_bytes, uint32_t lds_dst_uniform) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(lds_dst_uniform), "v"(voff_bytes), "s"(base) : "memory", "m0");
}
template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ void wait_lgkm0() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
// (a raw s_barrier does not wait for this wave's outstanding LDS writes: without the lgkmcnt(0) a staging write could still be in
// flight when another wave reads it back -- seen as single wrong 16-byte chunks once two workgroups shared a CU)
#define V6_BARRIER() do { __builtin_amdgcn_sched_barrier(0); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); __builtin_amdgcn_sched_barrier(0); } while (0)

// work item -> tile: XCD x (= workgroup index mod 8) owns a contiguous chunk of the row-major tile list, its workgroups stride through it
__device__ __forceinline__ void chunk_of(int n, int x, int& start, int& len) {
  const int q = n >> 3, r = n & 7;
  len = q + (x < r ? 1 : 0);
  start = x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q;
}

// EPI: 0 bias (+ residual if RES); 1 QuickGELU with QuickGELU' as the aux output (DH_EPI_GELU); 2 value * aux (DH_EPI_DGELU)
template <bool RES, int EPI>
__global__ __launch_bounds__(256, 2) void gemm_v6_kernel(const Args a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;

  // ---- this workgroup's tiles
  const int xcd = blockIdx.x & 7, nwg_x = (gridDim.x - 1 - xcd) / 8 + 1;
  int cs, cl;
  chunk_of(a.ntx * a.nty, xcd, cs, cl);
  int pos = blockIdx.x >> 3;                              // position inside the XCD's chunk
  if (pos >= cl) return;
  const int nk = a.K / BK;

  // ---- LDS-DMA sources.  Wave w requests A pieces 2w, 2w+1 (rows 32w ..) and B pieces 4w .. 4w+3 (rows 64w ..) of every K-step: per
  // lane a 32-bit byte offset from a wave-uniform 64-bit base (row part of the tile origin + k0), so a step's six requests need six
  // VGPRs in all.  Lane l of a piece: row 16p + (l >> 2), LDS slot l & 3 <- source chunk (l & 3) ^ ((row >> 2) & 3).
  uint32_t aoff[2], boff[4];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int row = 16 * (2 * wave + i) + (lane >> 2);
    aoff[i] = (uint32_t)(((long)row * a.lda + (((lane & 3) ^ ((row >> 2) & 3)) << 3)) * 2);
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = 16 * (4 * wave + i) + (lane >> 2);
    boff[i] = (uint32_t)(((long)row * a.ldb + (((lane & 3) ^ ((row >> 2) & 3)) << 3)) * 2);
  }
  // wave-uniform origins of the current and the next tile (one division per TILE, not per K-step)
  const bf16_t *Ac, *Bc, *An = nullptr, *Bn = nullptr;
  auto origin = [&](int tile, const bf16_t*& Ao, const bf16_t*& Bo) {
    const int ty = __builtin_amdgcn_readfirstlane(tile / a.ntx), tx = tile - ty * a.ntx;
    Ao = a.A + (long)ty * BM * a.lda;
    Bo = a.B + (long)tx * BN * a.ldb;
  };
  auto issue = [&](const bf16_t* Ao, const bf16_t* Bo, int kt, int stage) {
    const bf16_t* Ab = Ao + (long)kt * BK;
    const bf16_t* Bb = Bo + (long)kt * BK;
    const uint32_t sb = lds0 + stage * STAGE_BYTES;
#pragma unroll
    for (int i = 0; i < 2; ++i) dma16(Ab, aoff[i], sb + (2 * wave + i) * 1024);
#pragma unroll
    for (int i = 0; i < 4; ++i) dma16(Bb, boff[i], sb + A_BYTES + (4 * wave + i) * 1024);
  };

  // ---- fragment addresses (per lane, inside a stage): A rows 32i + (lane & 31), B rows 64w + 32j + (lane & 31); k-step s, half lane >> 5
  // -> chunk 2s + (lane >> 5) at slot chunk ^ ((row >> 2) & 3)
  const int fr = lane & 31, fh = lane >> 5;
  uint32_t fa[2], fb[2];                       // byte offsets of k-step 0 / 1 for row fr (+ 32 i rows = + 2048 i bytes: (32 i >> 2) & 3 == 0, same swizzle)
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    fa[s] = fr * 64 + (((2 * s + fh) ^ ((fr >> 2) & 3)) << 4);
    fb[s] = A_BYTES + (64 * wave + fr) * 64 + (((2 * s + fh) ^ ((fr >> 2) & 3)) << 4);
  }

  // ---- prologue: the first two K-steps of the first tile
  int tile = cs + pos;
  int g = 0;                                   // K-step counter of the stream: step g sits in stage g % 3
  origin(tile, Ac, Bc);
  issue(Ac, Bc, 0, 0);
  issue(Ac, Bc, 1, 1);
  bool prev_issued = true;                     // was a K-step requested in the previous iteration (decides what the wait may leave in flight)
  int pend = 0;                                // epilogue stores of the previous tile still counted by vmcnt (first two steps of a tile)

  while (true) {
    const int nxt = pos + nwg_x < cl ? cs + pos + nwg_x : -1;
    if (nxt >= 0) origin(nxt, An, Bn);
    f32x16_t acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    if (t < 64) {                              // the tile's 256 bias values -> LDS (free: every wave is past the previous epilogue's last convert)
      const int tx = tile % a.ntx;
      float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
      if (a.bias) bv = *reinterpret_cast<const float4*>(a.bias + tx * BN + 4 * t);
      *reinterpret_cast<float4*>(smem + BIAS_OFF + 16 * t) = bv;
    }
    for (int kt = 0; kt < nk; ++kt, ++g) {
      // K-step g has landed: everything this wave requested but the youngest step (and the previous tile's stores, for two steps)
      if (prev_issued) {
        if (pend && kt < 2) wait_vmcnt<6 + (EPI == 1 ? 32 : 16)>(); else wait_vmcnt<6>();
      } else {
        wait_vmcnt<0>();
      }
      V6_BARRIER();
      const int stage = g % 3;
      // what step g + 2 is (of this tile, or one of the first two steps of the next one): its six 1-KiB requests are issued BETWEEN
      // the MFMAs below (a request costs the wave ~60-100 cycles of issue; behind an MFMA most of that is the MFMA's own 32 cycles)
      const int k2 = kt + 2;
      const bool own = k2 < nk, more = own || nxt >= 0;
      const bf16_t* Ab = (own ? Ac : An) + (long)(own ? k2 : k2 - nk) * BK;
      const bf16_t* Bb = (own ? Bc : Bn) + (long)(own ? k2 : k2 - nk) * BK;
      const uint32_t sbn = lds0 + ((g + 2) % 3) * STAGE_BYTES;
      prev_issued = more;
      const unsigned char* sb = smem + stage * STAGE_BYTES;
      bf16x8_t af[4][2], bfv[2][2];
#pragma unroll
      for (int s = 0; s < 2; ++s) {
#pragma unroll
        for (int i = 0; i < 4; ++i) af[i][s] = *reinterpret_cast<const bf16x8_t*>(sb + fa[s] + i * 2048);
#pragma unroll
        for (int j = 0; j < 2; ++j) bfv[j][s] = *reinterpret_cast<const bf16x8_t*>(sb + fb[s] + j * 2048);
      }
#define V6_DMA_A(i_) do { __builtin_amdgcn_sched_barrier(0); if (more) dma16(Ab, aoff[i_], sbn + (2 * wave + (i_)) * 1024); __builtin_amdgcn_sched_barrier(0); } while (0)
#define V6_DMA_B(i_) do { __builtin_amdgcn_sched_barrier(0); if (more) dma16(Bb, boff[i_], sbn + A_BYTES + (4 * wave + (i_)) * 1024); __builtin_amdgcn_sched_barrier(0); } while (0)
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
#pragma unroll
          for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bfv[j][s], af[i][s], acc[i][j], 0, 0, 0);
          // one request behind MFMAs 2, 4, 6 (A0, A1, B0) of k-step 0 and 2, 4, 6 (B1, B2, B3) of k-step 1
          if (s == 0 && i == 0) V6_DMA_A(0);
          if (s == 0 && i == 1) V6_DMA_A(1);
          if (s == 0 && i == 2) V6_DMA_B(0);
          if (s == 1 && i == 0) V6_DMA_B(1);
          if (s == 1 && i == 1) V6_DMA_B(2);
          if (s == 1 && i == 2) V6_DMA_B(3);
        }
      __builtin_amdgcn_s_setprio(0);
#undef V6_DMA_A
#undef V6_DMA_B
    }
    // ---- epilogue.  acc[i][j]: lane -> row 32 i + (lane & 31), registers 4 rg .. 4 rg + 3 -> columns 64 w + 32 j + 8 rg + 4 (lane >> 5) + {0..3}.
    // Four passes (row block i): staging tile [32 rows][512 B] in the stage the LAST K-step sat in (the other two hold / receive the next
    // tile's first steps); 8-byte unit u of row r at unit u ^ (r & 15); read back as whole 16-byte chunks (halves swapped on odd rows).
    {
      unsigned char* Cs = smem + ((g - 1) % 3) * STAGE_BYTES;
      const int ty = tile / a.ntx, tx = tile - ty * a.ntx;
      const long m0 = (long)ty * BM, n0 = (long)tx * BN;
      float4 bq[2][4];
      V6_BARRIER();                            // every wave has read its last fragments (and the bias is in LDS)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) bq[j][rg] = *reinterpret_cast<const float4*>(smem + BIAS_OFF + 4 * (64 * wave + 32 * j + 8 * rg + 4 * fh));
      const int rrow = t >> 5, cc = t & 31;    // read-back: thread -> rows rrow + 8 it (it = 0..3), 16-byte chunk cc
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        uint4 pre[4];
        if (RES || EPI == 2) {
          const unsigned char* pb = reinterpret_cast<const unsigned char*>(RES ? (const void*)a.residual : (const void*)a.aux);
          const long pld = RES ? a.ldr : a.ldaux;
#pragma unroll
          for (int it = 0; it < 4; ++it) pre[it] = *reinterpret_cast<const uint4*>(pb + ((m0 + 32 * i + rrow + 8 * it) * pld + n0 + cc * 8) * 2);
        }
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int rg = 0; rg < 4; ++rg) {
            const int nl = 64 * wave + 32 * j + 8 * rg + 4 * fh;
            uint2 pk;
            pk.x = pack2bf_hw(acc[i][j][rg * 4 + 0] * a.alpha + bq[j][rg].x, acc[i][j][rg * 4 + 1] * a.alpha + bq[j][rg].y);
            pk.y = pack2bf_hw(acc[i][j][rg * 4 + 2] * a.alpha + bq[j][rg].z, acc[i][j][rg * 4 + 3] * a.alpha + bq[j][rg].w);
            *reinterpret_cast<uint2*>(Cs + fr * 512 + (((nl >> 2) ^ (fr & 15)) << 3)) = pk;
          }
        V6_BARRIER();
#pragma unroll
        for (int it = 0; it < 4; ++it) {
          const int row = rrow + 8 * it;
          const int pc = cc ^ ((row & 15) >> 1);
          uint4 raw = *reinterpret_cast<const uint4*>(Cs + row * 512 + pc * 16);
          if (row & 1) { uint32_t tx_ = raw.x, ty_ = raw.y; raw.x = raw.z; raw.y = raw.w; raw.z = tx_; raw.w = ty_; }
          if (RES || EPI == 2) {
            const uint32_t wv[4] = {raw.x, raw.y, raw.z, raw.w}, pw[4] = {pre[it].x, pre[it].y, pre[it].z, pre[it].w};
            uint32_t o[4];
#pragma unroll
            for (int x = 0; x < 4; ++x) {
              const float v0 = __uint_as_float(wv[x] << 16), v1 = __uint_as_float(wv[x] & 0xffff0000u);
              const float p0 = __uint_as_float(pw[x] << 16), p1 = __uint_as_float(pw[x] & 0xffff0000u);
              o[x] = EPI == 2 ? pack2bf_hw(v0 * p0, v1 * p1) : pack2bf_hw(v0 + p0, v1 + p1);
            }
            raw = make_uint4(o[0], o[1], o[2], o[3]);
          }
          if (EPI == 1) {       // both from the staged (bf16) pre-activation, like gemm_v4's GELU flavour
            const uint32_t wv[4] = {raw.x, raw.y, raw.z, raw.w};
            uint32_t og[4], od[4];
#pragma unroll
            for (int x = 0; x < 4; ++x) {
              float g0, g1, d0, d1;
              quick_gelu_both_f(__uint_as_float(wv[x] << 16), g0, d0);
              quick_gelu_both_f(__uint_as_float(wv[x] & 0xffff0000u), g1, d1);
              og[x] = pack2bf_hw(g0, g1); od[x] = pack2bf_hw(d0, d1);
            }
            *reinterpret_cast<uint4*>(reinterpret_cast<unsigned char*>(a.aux) + ((m0 + 32 * i + row) * a.ldaux + n0 + cc * 8) * 2) = make_uint4(od[0], od[1], od[2], od[3]);
            raw = make_uint4(og[0], og[1], og[2], og[3]);
          }
          *reinterpret_cast<uint4*>(reinterpret_cast<unsigned char*>(a.C) + ((m0 + 32 * i + row) * a.ldc + n0 + cc * 8) * 2) = raw;
        }
        V6_BARRIER();                          // staging free again
      }
    }
    pend = 1;
    if (nxt < 0) break;
    tile = nxt;
    Ac = An; Bc = Bn;
    pos += nwg_x;
  }
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

}  // namespace v6

// Returns true if the v6 kernel took the problem.  Opt-in: DH_GEMM_V6=1 (read per call: A/B runs switch it).
bool dh_gemm_try_v6(const dh_gemm_args* a, hipStream_t st) {
  using namespace v6;
  // DH_GEMM_V6: 0 / unset off; 1 every problem the kernel can take (tools/bench_v6.py); 2 only the shape classes where it measured faster than
  // gemm_v4 alone on the chip (profiles/r05_gemm_v6.txt): narrow outputs (N <= 1536: the weight panel stays in the XCD's L2 although
  // every 128-row tile re-reads it), no GELU flavour, and either more than one round of tiles or a short K (epilogue-dominated tiles)
  const char* ev = getenv("DH_GEMM_V6");
  const int mode = ev ? atoi(ev) : 0;
  if (mode != 1 && mode != 2) return false;
  if (mode == 2) {
    const int tiles_ = (a->M / BM) * (a->N / BN);
    if (a->N > 1536 || a->epilogue != DH_EPI_NONE || !(tiles_ > 2 * num_cus() || a->K <= 1024)) return false;
  }
  if (a->dtype != DH_BF16 || a->c_dtype != DH_BF16 || a->accumulate || a->a_kmajor || a->b_kmajor || a->a_colsum) return false;
  int epi = 0;
  if (a->epilogue == DH_EPI_GELU) { if (a->residual || !a->aux) return false; epi = 1; }
  else if (a->epilogue == DH_EPI_DGELU) { if (a->residual || !a->aux || a->bias) return false; epi = 2; }
  else if (a->epilogue != DH_EPI_NONE) return false;
  if ((a->M % BM) || (a->N % BN) || (a->K % BK) || a->K < 2 * BK) return false;
  if ((a->lda % 8) || (a->ldb % 8) || ((uintptr_t)a->A & 15) || ((uintptr_t)a->B & 15) || ((uintptr_t)a->C & 15) || ((a->ldc * 2) & 15)) return false;
  if (a->residual && (((uintptr_t)a->residual & 15) || ((a->ldr * 2) & 15))) return false;
  if (a->aux && (((uintptr_t)a->aux & 15) || ((a->ldaux * 2) & 15))) return false;
  if (a->bias && ((uintptr_t)a->bias & 15)) return false;
  if ((long)BM * a->lda * 2 >= (1L << 31) || (long)BN * a->ldb * 2 >= (1L << 31)) return false;     // 32-bit lane offsets inside a tile
  Args g;
  g.A = (const bf16_t*)a->A; g.lda = a->lda; g.B = (const bf16_t*)a->B; g.ldb = a->ldb; g.C = (bf16_t*)a->C; g.ldc = a->ldc;
  g.bias = a->bias; g.residual = (const bf16_t*)a->residual; g.ldr = a->ldr; g.aux = (bf16_t*)a->aux; g.ldaux = a->ldaux;
  g.M = a->M; g.N = a->N; g.K = a->K; g.ntx = a->N / BN; g.nty = a->M / BM; g.alpha = a->alpha;
  int grid = 2 * num_cus();
  const int tiles = g.ntx * g.nty;
  if (grid > tiles) grid = tiles;
#define V6_LAUNCH(RES_, EPI_)                                                                                                          \
  do {                                                                                                                                 \
    static bool attr_ = false;                                                                                                         \
    if (!attr_) { hipFuncSetAttribute((const void*)gemm_v6_kernel<RES_, EPI_>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES); attr_ = true; } \
    hipLaunchKernelGGL((gemm_v6_kernel<RES_, EPI_>), dim3(grid), dim3(256), LDS_BYTES, st, g);                                         \
  } while (0)
  if (epi == 1) V6_LAUNCH(false, 1);
  else if (epi == 2) V6_LAUNCH(false, 2);
  else if (a->residual) V6_LAUNCH(true, 0);
  else V6_LAUNCH(false, 0);
#undef V6_LAUNCH
  return true;
}
