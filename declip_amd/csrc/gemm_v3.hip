// GEMM v3 for gfx950: the v2 data path (LDS-DMA in, ds_read_b128 / ds_read_b64_tr_b16 fragments,
// LDS-staged vector epilogue) with a DEEP software pipeline:
//   * K is cut in 32-deep tiles; NST (3 or 4) LDS stages are in flight;
//   * the DMA of tile t+NST-1 is issued while tile t is multiplied;
//   * waits are COUNTED (s_waitcnt vmcnt(N) in inline asm): a wave only waits for its own pieces of the
//     OLDEST tile, the younger tiles' DMAs stay in flight across the raw s_barrier
//     (cdna_hip_programming.md, "Pipelining across barriers": __syncthreads() would drain vmcnt to 0);
//   * one barrier per K-tile: it publishes tile t to all waves and frees stage (t-1) for the next DMA.
// Configurations: 256 x 256 (8 waves as 2 x 4, 128 x 64 per wave, 4 stages, 128 KiB LDS, 1 block / CU)
//                 256 x 128 (8 waves as 4 x 2,  64 x 64 per wave, 3 stages,  72 KiB LDS, 2 blocks / CU)
#include "dh_common.h"
#include <stdlib.h>

namespace v3 {

struct EpiParams {
  int M, N;
  void* C; long ldc;
  const float* bias;
  int epilogue;
  const void* residual; long ldr;
  void* aux; long ldaux;
  int accumulate;
  float alpha;
  float* a_colsum;
};

constexpr int BK = 32;
constexpr int CSP = 4;   // epilogue staging pad (floats)

typedef __attribute__((ext_vector_type(4))) short s16x4;

template <int WM, int WN, int FM, int FN, int NST> struct Cfg {
  static constexpr int NW = WM * WN;
  static constexpr int BM = WM * FM * 32, BN = WN * FN * 32;
  static constexpr int A_BYTES = BM * BK * 2, B_BYTES = BN * BK * 2;
  static constexpr int STAGE_BYTES = A_BYTES + B_BYTES;
  static constexpr int CS = BN + CSP;
  static constexpr int EROWS = NW == 4 ? 64 : 128;                   // epilogue staged EROWS rows at a time
  static constexpr int EPI_BYTES = EROWS * CS * 4;
  static constexpr int LDS_BYTES = EPI_BYTES > NST * STAGE_BYTES ? EPI_BYTES : NST * STAGE_BYTES;
  static constexpr int APW = (A_BYTES / 1024) / NW, BPW = (B_BYTES / 1024) / NW;   // 1-KiB DMA pieces per wave per tile
};

__device__ __forceinline__ void dma16(const bf16_t* src, unsigned char* lds_dst_uniform) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                   (__attribute__((address_space(3))) void*)lds_dst_uniform, 16, 0, 0);
}

// K-contiguous image [R rows][32 k]: 64-byte rows, 16-B chunk c (0..3) of row r at slot c ^ ((r>>2)&3).
// piece q (1 KiB) = rows 16q .. 16q+15; lane -> row 16q + (lane>>2), slot lane&3.
// contraction-major image [32 k][R out]: 2R-byte rows, chunk c of row k at slot c ^ ((k&3)<<2).
// piece q = (512/R) k-rows x (R/8) slots.
template <bool KM, int R>
__device__ __forceinline__ const bf16_t* piece_src(const bf16_t* P, long ld, int q, int lane, int o0, int outs, int kbeg) {
  if (KM) {
    constexpr int SLOTS = R / 8, RPP = 64 / SLOTS;
    const int kl = q * RPP + lane / SLOTS;
    const int c = (lane % SLOTS) ^ ((kl & 3) << 2);
    int o = o0 + c * 8;
    o = o < outs ? o : 0;
    return P + (long)(kbeg + kl) * ld + o;
  } else {
    const int rl = q * 16 + (lane >> 2);
    const int c = (lane & 3) ^ ((rl >> 2) & 3);
    int r = o0 + rl;
    r = r < outs ? r : outs - 1;
    return P + (long)r * ld + kbeg + c * 8;
  }
}
// 32x32x16 fragment, k16-substep s (0/1), rows r0 + (lane&31)
__device__ __forceinline__ bf16x8_t frag_kcontig(const unsigned char* tile, int r0, int s, int lane) {
  const int row = r0 + (lane & 31);
  const int chunk = 2 * s + (lane >> 5);
  return *reinterpret_cast<const bf16x8_t*>(tile + row * 64 + ((chunk ^ ((row >> 2) & 3)) << 4));
}
template <int ROWB>
__device__ __forceinline__ bf16x8_t frag_kmajor(const unsigned char* tile, int o0, int s, int lane) {
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

template <typename TI, typename TO>
__device__ __forceinline__ void epilogue8(const EpiParams& e, int m, int n, float* v) {
#pragma unroll
  for (int i = 0; i < 8; ++i) v[i] *= e.alpha;
  if (e.bias) {
    float bv[8];
    ld8(e.bias + n, bv);
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] += bv[i];
  }
  if (e.epilogue == DH_EPI_GELU) {           // aux = QuickGELU'(pre): what DH_EPI_DGELU multiplies by
    float d[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) quick_gelu_both_f(v[i], v[i], d[i]);
    if (e.aux) st8(reinterpret_cast<TO*>(e.aux) + (long)m * e.ldaux + n, d);
  } else if (e.epilogue == DH_EPI_DGELU) {
    float u[8];
    ld8(reinterpret_cast<const TI*>(e.aux) + (long)m * e.ldaux + n, u);
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] *= u[i];
  }
  if (e.residual) {
    float r[8];
    ld8(reinterpret_cast<const TO*>(e.residual) + (long)m * e.ldr + n, r);
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] += r[i];
  }
  st8(reinterpret_cast<TO*>(e.C) + (long)m * e.ldc + n, v);
}

template <int N> __device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

template <int WM, int WN, int FM, int FN, int NST, bool TA, bool TB, typename TO>
__global__ __launch_bounds__(64 * WM * WN, (WM * WN == 4) ? 3 : (WM * WN * (NST == 4 ? 1 : 2)) / 4)
void gemm_v3_kernel(const bf16_t* __restrict__ A, long lda, const bf16_t* __restrict__ B, long ldb, int M, int N, int K,
                    int k_per_split, EpiParams e) {
  using C = Cfg<WM, WN, FM, FN, NST>;
  constexpr int BM = C::BM, BN = C::BN, NW = C::NW, APW = C::APW, BPW = C::BPW, PPT = APW + BPW;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int t = threadIdx.x;
  const int lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wm = wave / WN, wn = wave % WN;
  int tile_x = blockIdx.x, tile_y = blockIdx.y;
  {   // XCD-aware tile order (see gemm_glds.hip)
    const int ntx = gridDim.x, nb = gridDim.x * gridDim.y;
    const int b = blockIdx.y * ntx + blockIdx.x;
    const int q = nb >> 3, r = nb & 7, xcd = b & 7, idx = b >> 3;
    const int logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    // L2-aware rasterisation: the ~64 tiles an XCD runs concurrently form a GROUP_M x (64/GROUP_M) patch of the
    // tile grid, so they share GROUP_M A-panels and ~8 B-panels that fit the 4 MiB L2 (without this every
    // m-tile re-streams the whole B operand from the Infinity Cache: measured 3-4x the necessary traffic).
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
  const int m0 = tile_y * BM, n0 = tile_x * BN;
  const int kbeg = blockIdx.z * k_per_split;
  const int kend = min(K, kbeg + k_per_split);
  const int nk = (kend - kbeg) / BK;

  const bf16_t* asrc[APW];
  const bf16_t* bsrc[BPW];
#pragma unroll
  for (int i = 0; i < APW; ++i) asrc[i] = piece_src<TA, BM>(A, lda, wave * APW + i, lane, m0, M, kbeg);
#pragma unroll
  for (int i = 0; i < BPW; ++i) bsrc[i] = piece_src<TB, BN>(B, ldb, wave * BPW + i, lane, n0, N, kbeg);
  const long astep = TA ? (long)BK * lda : BK;
  const long bstep = TB ? (long)BK * ldb : BK;

  auto issue = [&](int stage) {
    unsigned char* ta = smem + stage * C::STAGE_BYTES + wave * (APW * 1024);
    unsigned char* tb = smem + stage * C::STAGE_BYTES + C::A_BYTES + wave * (BPW * 1024);
#pragma unroll
    for (int i = 0; i < APW; ++i) { dma16(asrc[i], ta + i * 1024); asrc[i] += astep; }
#pragma unroll
    for (int i = 0; i < BPW; ++i) { dma16(bsrc[i], tb + i * 1024); bsrc[i] += bstep; }
  };

  f32x16_t acc[FM][FN];
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const bool do_colsum = TA && e.a_colsum != nullptr && tile_x == 0;
  float csum = 0.f;

  // prologue: NST-1 tiles in flight
#pragma unroll
  for (int s = 0; s < NST - 1; ++s)
    if (s < nk) issue(s);

  int stage = 0;
  for (int kt = 0; kt < nk; ++kt) {
    // wait for this wave's pieces of tile kt: tiles kt+1 .. kt+NST-2 (if they exist) may stay in flight
    const int younger = min(NST - 2, nk - 1 - kt);
    if (NST == 4) {
      if (younger >= 2) wait_vmcnt<2 * PPT>(); else if (younger == 1) wait_vmcnt<PPT>(); else wait_vmcnt<0>();
    } else {
      if (younger >= 1) wait_vmcnt<PPT>(); else wait_vmcnt<0>();
    }
    __builtin_amdgcn_s_barrier();          // tile kt visible to every wave; everyone has finished tile kt-1
    if (kt + NST - 1 < nk) {
      int st2 = stage + NST - 1;
      st2 = st2 >= NST ? st2 - NST : st2;
      issue(st2);                          // refills the stage tile kt-1 was read from
    }
    const unsigned char* ta = smem + stage * C::STAGE_BYTES;
    const unsigned char* tb = ta + C::A_BYTES;
    if (TA && do_colsum) {
      constexpr int TPB = 64 * NW, KPARTS = TPB / BM, KR = BK / KPARTS;
      const int mcol = t % BM, kh = t / BM;
#pragma unroll
      for (int kk = 0; kk < KR; ++kk) {
        const int k = kh * KR + kk;
        const unsigned char* p = ta + k * (2 * BM) + ((((mcol >> 3) ^ ((k & 3) << 2))) << 4) + ((mcol & 7) << 1);
        csum += bf2f(*reinterpret_cast<const bf16_t*>(p));
      }
    }
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      bf16x8_t a[FM], b[FN];
#pragma unroll
      for (int i = 0; i < FM; ++i)
        a[i] = TA ? frag_kmajor<2 * BM>(ta, (wm * FM + i) * 32, s, lane) : frag_kcontig(ta, (wm * FM + i) * 32, s, lane);
#pragma unroll
      for (int j = 0; j < FN; ++j)
        b[j] = TB ? frag_kmajor<2 * BN>(tb, (wn * FN + j) * 32, s, lane) : frag_kcontig(tb, (wn * FN + j) * 32, s, lane);
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
      __builtin_amdgcn_s_setprio(0);
    }
    stage = stage + 1 == NST ? 0 : stage + 1;
  }

  if (TA && do_colsum) {
    const int mcol = m0 + (t % BM);
    if (mcol < M) atomicAdd(e.a_colsum + mcol, csum);
  }
  if (e.accumulate) {
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
      for (int j = 0; j < FN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int m = m0 + (wm * FM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
          const int n = n0 + (wn * FN + j) * 32 + (lane & 31);
          if (m < M && n < N) atomicAdd(reinterpret_cast<float*>(e.C) + (long)m * e.ldc + n, acc[i][j][r] * e.alpha);
        }
    return;
  }

  // ---- epilogue through LDS, 128 rows per pass: Cs[128][CS] fp32
  constexpr int CS = C::CS, EROWS = C::EROWS;
  float* Cs = reinterpret_cast<float*>(smem);
#pragma unroll 1
  for (int pass = 0; pass < BM / EROWS; ++pass) {
    __syncthreads();
    if ((wm * FM * 32) / EROWS == pass) {
#pragma unroll
      for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int row = (wm * FM + i) * 32 - pass * EROWS + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            const int col = (wn * FN + j) * 32 + (lane & 31);
            Cs[row * CS + col] = acc[i][j][r];
          }
    }
    __syncthreads();
    constexpr int CPR = BN / 8;                 // 16-B chunks per row
    constexpr int RSTEP = (64 * NW) / CPR;
    const int cc = t % CPR;
    const int n = n0 + cc * 8;
    if (n < N) {
#pragma unroll 4
      for (int i = 0; i < EROWS / RSTEP; ++i) {
        const int row = (t / CPR) + i * RSTEP;
        const int m = m0 + pass * EROWS + row;
        if (m < M) {
          float v[8];
          const float4 x = *reinterpret_cast<const float4*>(Cs + row * CS + cc * 8);
          const float4 y = *reinterpret_cast<const float4*>(Cs + row * CS + cc * 8 + 4);
          v[0] = x.x; v[1] = x.y; v[2] = x.z; v[3] = x.w; v[4] = y.x; v[5] = y.y; v[6] = y.z; v[7] = y.w;
          epilogue8<bf16_t, TO>(e, m, n, v);
        }
      }
    }
  }
}

template <int WM, int WN, int FM, int FN, int NST, bool TA, bool TB, typename TO>
void launch_cfg(const dh_gemm_args* a, const EpiParams& e, int split, int kps, hipStream_t st) {
  using C = Cfg<WM, WN, FM, FN, NST>;
  static bool attr_set = false;
  if (!attr_set) {
    hipFuncSetAttribute((const void*)gemm_v3_kernel<WM, WN, FM, FN, NST, TA, TB, TO>, hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES);
    attr_set = true;
  }
  dim3 grid(dh_cdiv(a->N, C::BN), dh_cdiv(a->M, C::BM), split);
  hipLaunchKernelGGL((gemm_v3_kernel<WM, WN, FM, FN, NST, TA, TB, TO>), grid, dim3(64 * C::NW), C::LDS_BYTES, st,
                     (const bf16_t*)a->A, (long)a->lda, (const bf16_t*)a->B, (long)a->ldb, a->M, a->N, a->K, kps, e);
}

template <bool TA, bool TB, typename TO>
void launch_t(const dh_gemm_args* a, const EpiParams& e, int split, int kps, int big, hipStream_t st) {
  if (big == 1) launch_cfg<2, 4, 4, 2, 4, TA, TB, TO>(a, e, split, kps, st);        // 256 x 256, 4 stages, 1 block/CU
  else if (big == 2) launch_cfg<2, 2, 2, 2, 3, TA, TB, TO>(a, e, split, kps, st);   // 128 x 128, 3 stages, 3 blocks/CU
  else launch_cfg<4, 2, 2, 2, 3, TA, TB, TO>(a, e, split, kps, st);                 // 256 x 128, 3 stages, 2 blocks/CU
}

}  // namespace v3

// Returns true if the v3 kernel took the problem (called from dh_gemm before the v2 path).
// DH_GEMM_V3 env: 0 = off, 1 = 256x128 only, 2 = 256x256 only, unset = heuristic.
bool dh_gemm_try_v3(const dh_gemm_args* a, int split, hipStream_t st) {
  using namespace v3;
  static int mode = -2;
  if (mode == -2) { const char* ev = getenv("DH_GEMM_V3"); mode = ev ? atoi(ev) : -1; }
  int m = mode;
  const bool forced = a->force_generic >= 31 && a->force_generic <= 33;     // dh_gemm_args.force_generic 31 / 32 / 33: this kernel with tile mode 1 / 2 / 3 (tests)
  if (forced) m = a->force_generic - 30;
  if (m == 0) return false;
  if (a->dtype != DH_BF16 || (a->force_generic && !forced)) return false;
  if ((a->lda % 8) || (a->ldb % 8) || ((uintptr_t)a->A & 15) || ((uintptr_t)a->B & 15)) return false;
  if (a->M < 256 && m != 3) return false;
  if (!a->pad_ok) {
    if (a->N % 8) return false;
    if (a->a_kmajor && (a->M % 8)) return false;
  } else if (!a->accumulate && (a->N % 8)) {
    return false;
  }
  if (a->K % BK) return false;
  int kps = ((a->K + split - 1) / split + BK - 1) / BK * BK;
  split = (a->K + kps - 1) / kps;
  if (!a->accumulate) {
    const int esz = a->c_dtype == DH_BF16 ? 2 : 4;
    if (((uintptr_t)a->C & 15) || ((a->ldc * esz) & 15)) return false;
    if (a->residual && (((uintptr_t)a->residual & 15) || ((a->ldr * esz) & 15))) return false;
    if (a->aux) {
      const int asz = a->epilogue == DH_EPI_DGELU ? 2 : esz;
      if (((uintptr_t)a->aux & 15) || ((a->ldaux * asz) & 15)) return false;
    }
    if (a->bias && ((uintptr_t)a->bias & 15)) return false;
  }
  EpiParams e;
  e.M = a->M; e.N = a->N; e.C = a->C; e.ldc = a->ldc; e.bias = a->bias; e.epilogue = a->epilogue;
  e.residual = a->residual; e.ldr = a->ldr; e.aux = a->aux; e.ldaux = a->ldaux; e.accumulate = a->accumulate;
  e.alpha = a->alpha; e.a_colsum = a->a_colsum;
  int big;
  if (m == 1) big = 0;
  else if (m == 2) big = 1;
  else if (m == 3) big = 2;
  else {
    // measured on the tower shapes (profiles/r01_gemm_microbench_v2_v3.txt): the deep pipeline only beats the
    // v2 kernel for the dX products (B contraction-major); everything else stays on v2.
    if (!(a->b_kmajor && !a->a_kmajor)) return false;
    big = a->N >= 2048 ? 0 : 2;
  }
  if (big && a->N < 256) big = 0;
#define V3_LAUNCH(TA, TB)                                                                  \
  do {                                                                                     \
    if (a->c_dtype == DH_BF16) launch_t<TA, TB, bf16_t>(a, e, split, kps, big, st);        \
    else launch_t<TA, TB, float>(a, e, split, kps, big, st);                               \
  } while (0)
  if (a->a_kmajor && a->b_kmajor) V3_LAUNCH(true, true);
  else if (a->a_kmajor) V3_LAUNCH(true, false);
  else if (a->b_kmajor) V3_LAUNCH(false, true);
  else V3_LAUNCH(false, false);
#undef V3_LAUNCH
  return true;
}
