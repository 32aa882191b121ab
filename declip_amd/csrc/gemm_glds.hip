// GEMM v2 for gfx950: operands go HBM -> LDS by direct LDS-DMA (global_load_lds_dwordx4), never
// through VGPRs; fragments for v_mfma_f32_32x32x16_bf16 come out of LDS with one ds_read_b128
// (K-contiguous operands) or two ds_read_b64_tr_b16 (contraction-major operands: the hardware
// transpose read replaces v1's in-register 8x8 transposes); the epilogue is staged through LDS so
// every global access of bias / residual / aux / C is a 16-byte vector on a contiguous row segment.
//
//   block tile 128 x 128 x 64, 4 waves (2 x 2, 64 x 64 each), 2 LDS stages, one barrier per K-step;
//   the DMA of K-tile t+1 is in flight while tile t is multiplied.
//
// LDS images (both are "lane-linear" for the DMA; the swizzles live on the SOURCE address and on
// the fragment read, never on the destination -- cdna_hip_programming.md rule 21):
//   K-contiguous operand : [128 rows][64 k]   128 B rows, 16-B chunk c of row r at slot c ^ ((r>>1)&7)
//   contraction-major op.: [64 k][128 out]    256 B rows, 16-B chunk c of row k at slot c ^ ((k&3)<<2)
// ds_read_b64_tr_b16 semantics (measured on MI355X, profiles/r01_hw_probe_trread_glds.txt): inside
// each 16-lane group, lane i receives element (i&3) of the 8-byte chunks addressed by lanes
// 4j + (i>>2), j = 0..3.
#include "dh_common.h"
#include <stdlib.h>

namespace glds {

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

constexpr int BN = 128, BK = 64;
constexpr int CS = 132;                       // epilogue staging row stride (floats)
// block tile (64*WMR) x 128 x 64 with 2*WMR waves (each 64 x 64): WMR = 2 -> 128 x 128 (2 blocks/CU),
// WMR = 4 -> 256 x 128 (1 block/CU, 85 flop per staged byte: the L2 -> LDS stream stops being the limiter)
template <int WMR> struct Cfg {
  static constexpr int BM = 64 * WMR;
  static constexpr int NW = 2 * WMR;
  static constexpr int A_BYTES = BM * BK * 2;
  static constexpr int B_BYTES = BN * BK * 2;
  static constexpr int STAGE_BYTES = A_BYTES + B_BYTES;
  static constexpr int EPI_BYTES = BM * CS * 4;
  static constexpr int LDS_BYTES = EPI_BYTES > 2 * STAGE_BYTES ? EPI_BYTES : 2 * STAGE_BYTES;
};

typedef __attribute__((ext_vector_type(4))) short s16x4;

__device__ __forceinline__ void dma16(const bf16_t* src, unsigned char* lds_dst_uniform, int /*unused*/) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                   (__attribute__((address_space(3))) void*)lds_dst_uniform, 16, 0, 0);
}

// fragment of a K-contiguous tile: rows r0 + (lane&31), k = 16*s + 8*(lane>>5) .. +7
__device__ __forceinline__ bf16x8_t frag_kcontig(const unsigned char* tile, int r0, int s, int lane) {
  const int row = r0 + (lane & 31);
  const int chunk = 2 * s + (lane >> 5);
  return *reinterpret_cast<const bf16x8_t*>(tile + row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4));
}
// fragment of a contraction-major tile via the transpose read: out columns o0 + (lane&31)
template <int ROWB>  // bytes per k-row of the image (2 * number of out columns)
__device__ __forceinline__ bf16x8_t frag_kmajor(const unsigned char* tile, int o0, int s, int lane) {
  const int t = lane & 15;
  const int n = o0 + ((lane >> 4) & 1) * 16 + 4 * (t & 3);
  const int k = 16 * s + 8 * (lane >> 5) + (t >> 2);
  const int sw = (t >> 2) << 2;                                    // (k & 3) << 2, k & 3 == t >> 2
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

// DMA source pointer of piece q (1 KiB = 64 lanes x 16 B) of an operand tile, for this lane.
//   K-contiguous image [R rows][64 k]: piece q = rows 8q .. 8q+7
//   contraction-major image [64 k][R out]: piece q = (512/R) k-rows x (R/8) slots
template <bool KM, int R>
__device__ __forceinline__ const bf16_t* piece_src(const bf16_t* P, long ld, int q, int lane, int o0, int outs, int kbeg) {
  if (KM) {
    constexpr int SLOTS = R / 8, RPP = 64 / SLOTS;       // slots per k-row, k-rows per piece
    const int kl = q * RPP + lane / SLOTS;
    const int c = (lane % SLOTS) ^ ((kl & 3) << 2);
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

template <int WMR, bool TA, bool TB, typename TO>
__global__ __launch_bounds__(128 * WMR, 2) void gemm_glds_kernel(const bf16_t* __restrict__ A, long lda,
                                                                 const bf16_t* __restrict__ B, long ldb, int M, int N, int K,
                                                                 int k_per_split, EpiParams e) {
  using C = Cfg<WMR>;
  constexpr int BM = C::BM, NW = C::NW;
  constexpr int APW = (BM / 8) / NW;     // 1-KiB pieces of the A tile per wave (= 4)
  constexpr int BPW = (BN / 8) / NW;     // pieces of the B tile per wave (4 or 2)
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int t = threadIdx.x;
  const int lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  // XCD-aware tile order (MI355X: block b runs on XCD b % 8, each XCD has its own L2): hand every
  // XCD a contiguous run of logical tiles so that the n-tiles sharing an A row-panel hit one L2.
  int tile_x = blockIdx.x, tile_y = blockIdx.y;
  {
    const int ntx = gridDim.x, nb = gridDim.x * gridDim.y;
    const int b = blockIdx.y * ntx + blockIdx.x;
    const int q = nb >> 3, r = nb & 7, xcd = b & 7, idx = b >> 3;
    const int logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    // L2-aware rasterisation: the ~64 tiles an XCD runs concurrently form a GROUP_M x (64/GROUP_M) patch of the
    // tile grid, so they share GROUP_M A-panels and ~8 B-panels that fit the 4 MiB L2.
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
  const int nk = (kend - kbeg) / BK;                 // host guarantees divisibility

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
    for (int i = 0; i < APW; ++i) {
      dma16(asrc[i], ta + i * 1024, 0);
      asrc[i] += astep;
    }
#pragma unroll
    for (int i = 0; i < BPW; ++i) {
      dma16(bsrc[i], tb + i * 1024, 0);
      bsrc[i] += bstep;
    }
  };

  f32x16_t acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // fused bias gradient: the n-tile-0 blocks also column-sum the dY tiles they stage (TA only)
  const bool do_colsum = TA && e.a_colsum != nullptr && tile_x == 0;
  float csum = 0.f;

  if (nk > 0) issue(0);
  for (int kt = 0; kt < nk; ++kt) {
    const int stage = kt & 1;
    __syncthreads();                       // drains this wave's DMA (vmcnt) + block barrier: tile kt is in LDS,
                                           // and every wave is done reading the other stage
    if (kt + 1 < nk) issue(stage ^ 1);
    const unsigned char* ta = smem + stage * C::STAGE_BYTES;
    const unsigned char* tb = ta + C::A_BYTES;
    if (TA && do_colsum) {
      // thread -> column m = t % BM, k rows [kh*KR, +KR) with KR = 64 / (threads / BM)
      constexpr int TPB = 64 * NW, KPARTS = TPB / BM, KR = 64 / KPARTS;
      const int mcol = t % BM, kh = t / BM;
#pragma unroll 8
      for (int kk = 0; kk < KR; ++kk) {
        const int k = kh * KR + kk;
        const unsigned char* p = ta + k * (2 * BM) + ((((mcol >> 3) ^ ((k & 3) << 2))) << 4) + ((mcol & 7) << 1);
        csum += bf2f(*reinterpret_cast<const bf16_t*>(p));
      }
    }
    // fragment reads of k16-step s+1 are issued before the MFMAs of step s (register double buffer)
    bf16x8_t a[2][2], b[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      a[0][i] = TA ? frag_kmajor<2 * BM>(ta, wm * 64 + i * 32, 0, lane) : frag_kcontig(ta, wm * 64 + i * 32, 0, lane);
      b[0][i] = TB ? frag_kmajor<2 * BN>(tb, wn * 64 + i * 32, 0, lane) : frag_kcontig(tb, wn * 64 + i * 32, 0, lane);
    }
#pragma unroll
    for (int s = 0; s < BK / 16; ++s) {
      const int cur = s & 1, nxt = cur ^ 1;
      if (s + 1 < BK / 16) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          a[nxt][i] = TA ? frag_kmajor<2 * BM>(ta, wm * 64 + i * 32, s + 1, lane) : frag_kcontig(ta, wm * 64 + i * 32, s + 1, lane);
          b[nxt][i] = TB ? frag_kmajor<2 * BN>(tb, wn * 64 + i * 32, s + 1, lane) : frag_kcontig(tb, wn * 64 + i * 32, s + 1, lane);
        }
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[cur][i], b[cur][j], acc[i][j], 0, 0, 0);
    }
  }

  if (TA && do_colsum) {
    const int mcol = m0 + (t % BM);
    if (mcol < M) atomicAdd(e.a_colsum + mcol, csum);
  }
  if (e.accumulate) {
    // split-K partials: atomics straight from the accumulator layout (32 consecutive floats per row)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int m = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
          const int n = n0 + wn * 64 + j * 32 + (lane & 31);
          if (m < M && n < N) atomicAdd(reinterpret_cast<float*>(e.C) + (long)m * e.ldc + n, acc[i][j][r] * e.alpha);
        }
    return;
  }

  // ---- epilogue through LDS: Cs[128][CS] fp32
  __syncthreads();
  float* Cs = reinterpret_cast<float*>(smem);
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        const int col = wn * 64 + j * 32 + (lane & 31);
        Cs[row * CS + col] = acc[i][j][r];
      }
  __syncthreads();
  // 16 consecutive lanes cover one 128-column row (16 B each): every wave store instruction writes
  // 4 full 256/512-byte row segments -- fully coalesced (the first version had lanes 128 B apart).
  {
    const int cc = t & 15;
    constexpr int RSTEP = (64 * NW) / 16;
    const int n = n0 + cc * 8;
    if (n < N) {
#pragma unroll
      for (int i = 0; i < BM / RSTEP; ++i) {
        const int row = (t >> 4) + i * RSTEP;
        const int m = m0 + row;
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

template <int WMR, bool TA, bool TB, typename TO>
void launch_cfg(const dh_gemm_args* a, const EpiParams& e, int split, int kps, hipStream_t st) {
  using C = Cfg<WMR>;
  static bool attr_set = false;
  if (!attr_set) {
    hipFuncSetAttribute((const void*)gemm_glds_kernel<WMR, TA, TB, TO>, hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES);
    attr_set = true;
  }
  dim3 grid(dh_cdiv(a->N, BN), dh_cdiv(a->M, C::BM), split);
  hipLaunchKernelGGL((gemm_glds_kernel<WMR, TA, TB, TO>), grid, dim3(64 * C::NW), C::LDS_BYTES, st, (const bf16_t*)a->A,
                     (long)a->lda, (const bf16_t*)a->B, (long)a->ldb, a->M, a->N, a->K, kps, e);
}

template <bool TA, bool TB>
void launch(const dh_gemm_args* a, const EpiParams& e, int split, int kps, int wmr, hipStream_t st) {
  if (a->c_dtype == DH_BF16) {
    if (wmr == 4) launch_cfg<4, TA, TB, bf16_t>(a, e, split, kps, st); else launch_cfg<2, TA, TB, bf16_t>(a, e, split, kps, st);
  } else {
    if (wmr == 4) launch_cfg<4, TA, TB, float>(a, e, split, kps, st); else launch_cfg<2, TA, TB, float>(a, e, split, kps, st);
  }
}

}  // namespace glds

// Returns true if the v2 kernel took the problem (called from dh_gemm in gemm.hip).
bool dh_gemm_try_glds(const dh_gemm_args* a, int split, hipStream_t st) {
  using namespace glds;
  if (a->dtype != DH_BF16 || a->force_generic) return false;
  if ((a->lda % 8) || (a->ldb % 8) || ((uintptr_t)a->A & 15) || ((uintptr_t)a->B & 15)) return false;
  if (a->M < 1) return false;
  if (!a->pad_ok) {
    if (a->N % 8) return false;
    if (a->a_kmajor && (a->M % 8)) return false;
  } else if (!a->accumulate && (a->N % 8)) {
    return false;                                   // the vector epilogue needs whole 8-column chunks
  }
  if (!a->a_kmajor && a->M < 1) return false;
  int kps = ((a->K + split - 1) / split + BK - 1) / BK * BK;
  if (a->K % BK) return false;
  split = (a->K + kps - 1) / kps;
  if (!a->accumulate) {
    // vector epilogue alignment: 16-byte rows for C / residual / aux, 32-byte bias
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
  e.alpha = a->alpha;
  e.a_colsum = a->a_colsum;
  // tile choice: 256 x 128 when there is enough work to fill the chip with one block per CU
  int wmr = 2;
  const long tiles256 = (long)dh_cdiv(a->M, 256) * dh_cdiv(a->N, BN) * split;
  (void)tiles256;   // measured: 128 x 128 (2 blocks/CU) beats 256 x 128 (1 block/CU) on every tower shape
  static int tile_env = -1;     // DH_GEMM_TILE (A/B switch), read once
  if (tile_env < 0) { const char* ev = getenv("DH_GEMM_TILE"); tile_env = ev ? atoi(ev) : 0; }
  if (tile_env == 256 && a->M >= 256) wmr = 4;
  if (a->a_kmajor && a->b_kmajor) launch<true, true>(a, e, split, kps, wmr, st);
  else if (a->a_kmajor) launch<true, false>(a, e, split, kps, wmr, st);
  else if (a->b_kmajor) launch<false, true>(a, e, split, kps, wmr, st);
  else launch<false, false>(a, e, split, kps, wmr, st);
  return true;
}
