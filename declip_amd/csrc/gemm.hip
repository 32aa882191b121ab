// GEMM for the transformer towers: C[M,N] (+)= epi(alpha * A . B^T + bias).
//   * gemm_mfma_kernel  -- bf16 operands, v_mfma_f32_32x32x16_bf16, 128x128x64 block tile,
//                          4 waves (2x2, 64x64 each), double-buffered XOR-swizzled LDS,
//                          register-staged global loads issued one K-tile ahead.
//                          Operands whose contraction index is the SLOW dimension in memory
//                          (dX: B = W[N][K]; dW: both) are transposed in registers (8x8 bf16
//                          blocks) on the way into LDS, so the MFMA fragment reads are always
//                          one ds_read_b128 per fragment.
//   * gemm_generic_kernel -- any dtype / shape, VALU fp32 FMA; the validation-precision path.
// Replaces: nn.Linear / nn.MultiheadAttention projections / conv1 (see include/declip_hip.h).
#include "dh_common.h"

namespace {

struct EpiParams {
  int M, N;
  void* C; long ldc;
  const float* bias;
  int epilogue;
  const void* residual; long ldr;
  void* aux; long ldaux;
  int accumulate;
  float alpha;
};

template <typename TI, typename TO>
__device__ __forceinline__ void epilogue_store(const EpiParams& e, int m, int n, float acc) {
  if (m >= e.M || n >= e.N) return;
  float v = acc * e.alpha;
  if (e.accumulate) {  // split-K partial: bias/act not allowed here (checked on host)
    atomicAdd(reinterpret_cast<float*>(e.C) + (long)m * e.ldc + n, v);
    return;
  }
  if (e.bias) v += e.bias[n];
  if (e.epilogue == DH_EPI_GELU) {           // aux = QuickGELU'(pre): what DH_EPI_DGELU multiplies by
    float g, d;
    quick_gelu_both_f(v, g, d);
    if (e.aux) st<TO>(reinterpret_cast<TO*>(e.aux) + (long)m * e.ldaux + n, d);
    v = g;
  } else if (e.epilogue == DH_EPI_DGELU) {
    v *= ld<TI>(reinterpret_cast<const TI*>(e.aux) + (long)m * e.ldaux + n);
  }
  if (e.residual) v += ld<TO>(reinterpret_cast<const TO*>(e.residual) + (long)m * e.ldr + n);
  st<TO>(reinterpret_cast<TO*>(e.C) + (long)m * e.ldc + n, v);
}

// ------------------------------------------------------------------------------------------
// bf16 MFMA kernel
// ------------------------------------------------------------------------------------------
constexpr int BM = 128, BN = 128, BK = 64;
constexpr int TILE_ELEMS = 128 * BK;  // per operand per stage

// LDS tile: [128 rows][64 k] bf16, 128 B per row; the 16-B chunk index c (0..7) of row r is
// stored at position c ^ ((r >> 1) & 7): conflict-free for the 32x32x16 fragment read
// (ds_read_b128 lane groups, MI355X LDS banking) and for the 8-lane ds_write_b128 groups.
__device__ __forceinline__ int lds_off(int row, int chunk) { return row * BK + ((chunk ^ ((row >> 1) & 7)) << 3); }

struct Staged {
  uint4 v[8];
};

// K-contiguous operand: P[row][k].  thread t: chunk = t&7, rows (t>>3) + 32*i.
__device__ __forceinline__ void load_kcontig(Staged& s, const bf16_t* __restrict__ P, long ld, int row0, int rows,
                                             int k0, int kend, int t) {
  const int c = t & 7;
  const int k = k0 + c * 8;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    int r = row0 + (t >> 3) + 32 * i;
    r = r < rows ? r : rows - 1;
    if (k < kend)
      s.v[i] = *reinterpret_cast<const uint4*>(P + (long)r * ld + k);
    else
      s.v[i] = make_uint4(0, 0, 0, 0);
  }
}
__device__ __forceinline__ void store_kcontig(const Staged& s, bf16_t* tile, int t) {
  const int c = t & 7;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    int r = (t >> 3) + 32 * i;
    *reinterpret_cast<uint4*>(tile + lds_off(r, c)) = s.v[i];
  }
}
// Contraction-major operand: P[k][out].  task (kb = t&7, ob = (t>>3)&15): an 8(k) x 8(out) block.
// The host guarantees outs % 8 == 0, so a block is either fully inside or fully outside;
// outside blocks read column 0 (their products are dropped by the epilogue bounds check).
__device__ __forceinline__ void load_kmajor(Staged& s, const bf16_t* __restrict__ P, long ld, int out0, int outs,
                                            int k0, int kend, int t) {
  const int kb = t & 7, ob = (t >> 3) & 15;
  int o = out0 + ob * 8;
  o = (o < outs) ? o : 0;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    int k = k0 + kb * 8 + j;
    if (k < kend)
      s.v[j] = *reinterpret_cast<const uint4*>(P + (long)k * ld + o);
    else
      s.v[j] = make_uint4(0, 0, 0, 0);
  }
}
__device__ __forceinline__ uint32_t sel16(const uint4& v, int c) {  // element c (0..7) of a uint4 of bf16
  uint32_t w = (c >> 1) == 0 ? v.x : (c >> 1) == 1 ? v.y : (c >> 1) == 2 ? v.z : v.w;
  return (c & 1) ? (w >> 16) : (w & 0xffffu);
}
// 8x8 transpose in registers: LDS row (ob*8 + c) gets the 8 k-values of out-column c.
__device__ __forceinline__ void store_kmajor(const Staged& s, bf16_t* tile, int t) {
  const int kb = t & 7, ob = (t >> 3) & 15;
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    uint4 w;
    w.x = sel16(s.v[0], c) | (sel16(s.v[1], c) << 16);
    w.y = sel16(s.v[2], c) | (sel16(s.v[3], c) << 16);
    w.z = sel16(s.v[4], c) | (sel16(s.v[5], c) << 16);
    w.w = sel16(s.v[6], c) | (sel16(s.v[7], c) << 16);
    *reinterpret_cast<uint4*>(tile + lds_off(ob * 8 + c, kb)) = w;
  }
}

template <bool TA, bool TB, typename TO>
__global__ __launch_bounds__(256) void gemm_mfma_kernel(const bf16_t* __restrict__ A, long lda,
                                                        const bf16_t* __restrict__ B, long ldb, int M, int N, int K,
                                                        int k_per_split, EpiParams e) {
  DH_DYN_LDS_A16(unsigned char, smem_raw);
  bf16_t* smem = reinterpret_cast<bf16_t*>(smem_raw);  // [2 stages][A tile | B tile]

  const int t = threadIdx.x;
  const int lane = t & 63, wave = t >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
  const int kbeg = blockIdx.z * k_per_split;
  const int kend = min(K, kbeg + k_per_split);
  const int nk = (kend - kbeg + BK - 1) / BK;

  f32x16_t acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  Staged sa, sb;
  // which threads stage which operand (see file header): K-contiguous operands use all 256
  // threads (4 x 16 B each); contraction-major operands use 128 tasks of 8 x 16 B.
  const bool a_active = TA ? (TB ? t < 128 : t < 128) : true;
  const bool b_active = TB ? (TA ? t >= 128 : t < 128) : true;

  auto load_tiles = [&](int kt) {
    const int k0 = kbeg + kt * BK;
    if (TA) {
      if (a_active) load_kmajor(sa, A, lda, m0, M, k0, kend, t);
    } else {
      load_kcontig(sa, A, lda, m0, M, k0, kend, t);
    }
    if (TB) {
      if (b_active) load_kmajor(sb, B, ldb, n0, N, k0, kend, t);
    } else {
      load_kcontig(sb, B, ldb, n0, N, k0, kend, t);
    }
  };
  auto store_tiles = [&](int stage) {
    bf16_t* ta = smem + stage * 2 * TILE_ELEMS;
    bf16_t* tb = ta + TILE_ELEMS;
    if (TA) {
      if (a_active) store_kmajor(sa, ta, t);
    } else {
      store_kcontig(sa, ta, t);
    }
    if (TB) {
      if (b_active) store_kmajor(sb, tb, t);
    } else {
      store_kcontig(sb, tb, t);
    }
  };

  if (nk > 0) load_tiles(0);
  for (int kt = 0; kt < nk; ++kt) {
    const int stage = kt & 1;
    store_tiles(stage);
    __syncthreads();
    if (kt + 1 < nk) load_tiles(kt + 1);
    const bf16_t* ta = smem + stage * 2 * TILE_ELEMS;
    const bf16_t* tb = ta + TILE_ELEMS;
#pragma unroll
    for (int s = 0; s < BK / 16; ++s) {
      bf16x8_t a[2], b[2];
      const int chunk = 2 * s + (lane >> 5);
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        a[i] = *reinterpret_cast<const bf16x8_t*>(ta + lds_off(wm * 64 + i * 32 + (lane & 31), chunk));
        b[i] = *reinterpret_cast<const bf16x8_t*>(tb + lds_off(wn * 64 + i * 32 + (lane & 31), chunk));
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
    }
  }

  // C/D layout of 32x32 MFMA: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        int m = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        int n = n0 + wn * 64 + j * 32 + (lane & 31);
        epilogue_store<bf16_t, TO>(e, m, n, acc[i][j][r]);
      }
}

// ------------------------------------------------------------------------------------------
// generic fp32-FMA kernel (validation precision, odd shapes)
// ------------------------------------------------------------------------------------------
template <typename TI, typename TO>
__global__ __launch_bounds__(256) void gemm_generic_kernel(const TI* __restrict__ A, long lda, int a_kmajor,
                                                           const TI* __restrict__ B, long ldb, int b_kmajor, int M,
                                                           int N, int K, int k_per_split, EpiParams e) {
  constexpr int T = 64, KT = 16, PAD = 4;
  __shared__ float As[KT][T + PAD];
  __shared__ float Bs[KT][T + PAD];
  const int t = threadIdx.x;
  const int tx = t & 15, ty = t >> 4;
  const int m0 = blockIdx.y * T, n0 = blockIdx.x * T;
  const int kbeg = blockIdx.z * k_per_split;
  const int kend = min(K, kbeg + k_per_split);
  float acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;

  for (int k0 = kbeg; k0 < kend; k0 += KT) {
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      int idx = t + p * 256;  // 0..1023
      int kk, mm;
      if (a_kmajor) { mm = idx & 63; kk = idx >> 6; } else { kk = idx & 15; mm = idx >> 4; }
      int m = m0 + mm, k = k0 + kk;
      float v = 0.f;
      if (m < M && k < kend) v = a_kmajor ? ld<TI>(A + (long)k * lda + m) : ld<TI>(A + (long)m * lda + k);
      As[kk][mm] = v;
      int nn;
      if (b_kmajor) { nn = idx & 63; kk = idx >> 6; } else { kk = idx & 15; nn = idx >> 4; }
      int n = n0 + nn;
      k = k0 + kk;
      v = 0.f;
      if (n < N && k < kend) v = b_kmajor ? ld<TI>(B + (long)k * ldb + n) : ld<TI>(B + (long)n * ldb + k);
      Bs[kk][nn] = v;
    }
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < KT; ++kk) {
      float a[4], b[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) a[i] = As[kk][ty * 4 + i];
#pragma unroll
      for (int j = 0; j < 4; ++j) b[j] = Bs[kk][tx * 4 + j];
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(a[i], b[j], acc[i][j]);
    }
    __syncthreads();
  }
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) epilogue_store<TI, TO>(e, m0 + ty * 4 + i, n0 + tx * 4 + j, acc[i][j]);
}

template <typename T>
__global__ void colsum_kernel(const T* __restrict__ X, long ldx, int M, int N, float* __restrict__ out, int accumulate,
                              int rows_per_block) {
  // block: 256 threads = 64 columns x 4 row-lanes; grid (ceil(N/64), row chunks)
  __shared__ float red[4][64];
  const int c = blockIdx.x * 64 + (threadIdx.x & 63);
  const int rl = threadIdx.x >> 6;
  const int r0 = blockIdx.y * rows_per_block;
  const int r1 = min(M, r0 + rows_per_block);
  float s = 0.f;
  if (c < N)
    for (int r = r0 + rl; r < r1; r += 4) s += ld<T>(X + (long)r * ldx + c);
  red[rl][threadIdx.x & 63] = s;
  __syncthreads();
  if (rl == 0 && c < N) {
    s = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
    if (accumulate || gridDim.y > 1) atomicAdd(out + c, s); else out[c] = s;
  }
}

template <typename TI, typename TO>
int launch_generic(const dh_gemm_args* a, const EpiParams& e, int split, int kps, hipStream_t st) {
  dim3 grid(dh_cdiv(a->N, 64), dh_cdiv(a->M, 64), split);
  hipLaunchKernelGGL((gemm_generic_kernel<TI, TO>), grid, dim3(256), 0, st, (const TI*)a->A, (long)a->lda, a->a_kmajor,
                     (const TI*)a->B, (long)a->ldb, a->b_kmajor, a->M, a->N, a->K, kps, e);
  return 0;
}

template <bool TA, bool TB>
int launch_mfma(const dh_gemm_args* a, const EpiParams& e, int split, int kps, hipStream_t st) {
  dim3 grid(dh_cdiv(a->N, BN), dh_cdiv(a->M, BM), split);
  size_t lds = 2 * 2 * TILE_ELEMS * sizeof(bf16_t);
  if (a->c_dtype == DH_BF16)
    hipLaunchKernelGGL((gemm_mfma_kernel<TA, TB, bf16_t>), grid, dim3(256), lds, st, (const bf16_t*)a->A, (long)a->lda,
                       (const bf16_t*)a->B, (long)a->ldb, a->M, a->N, a->K, kps, e);
  else
    hipLaunchKernelGGL((gemm_mfma_kernel<TA, TB, float>), grid, dim3(256), lds, st, (const bf16_t*)a->A, (long)a->lda,
                       (const bf16_t*)a->B, (long)a->ldb, a->M, a->N, a->K, kps, e);
  return 0;
}

}  // namespace

bool dh_gemm_try_glds(const dh_gemm_args* a, int split, hipStream_t st);  // gemm_glds.hip
bool dh_gemm_try_v3(const dh_gemm_args* a, int split, hipStream_t st);    // gemm_v3.hip
bool dh_gemm_try_v4(const dh_gemm_args* a, int split, hipStream_t st);

// launches per kernel family since the last reset: [0] v4 persistent 256x256, [1] v3, [2] v2 LDS-DMA 128x128, [3] v1 MFMA tiles,
// [4] generic (VALU).  The parity tests assert with it that a fixture really ran on the benchmarked kernel.
static long long g_gemm_family_calls[5] = {0, 0, 0, 0, 0};
extern "C" int dh_gemm_stats(long long* out5, int reset) {
  if (out5) for (int i = 0; i < 5; ++i) out5[i] = g_gemm_family_calls[i];
  if (reset) for (int i = 0; i < 5; ++i) g_gemm_family_calls[i] = 0;
  return DH_OK;
}

bool dh_gemm_try_v4_group(const dh_gemm_args* a, int n, hipStream_t st);   // gemm_v4.hip

// Several weight-gradient GEMMs with the same contraction length (the dW problems of one transformer block) as one launch of the
// persistent kernel + one reduce pass; any group the kernel cannot take is issued as n dh_gemm calls (same results).
extern "C" int dh_gemm_group(const dh_gemm_args* a, int n, dh_stream_t stream) {
  hipStream_t st = (hipStream_t)stream;
  DH_REQUIRE(a && n >= 1, "dh_gemm_group: bad args");
  for (int i = 0; i < n; ++i) {
    DH_REQUIRE(a[i].A && a[i].B && a[i].C && a[i].M > 0 && a[i].N > 0 && a[i].K > 0, "dh_gemm_group: problem %d: null pointer / bad shape", i);
    DH_REQUIRE(a[i].accumulate && a[i].c_dtype == DH_F32 && a[i].a_kmajor && a[i].b_kmajor,
               "dh_gemm_group: problem %d is not a weight gradient (a_kmajor, b_kmajor, fp32 accumulate)", i);
  }
  if (a[0].force_generic == 0 && n >= 2 && dh_gemm_try_v4_group(a, n, st)) {
    g_gemm_family_calls[0] += n;
    DH_CHECK_LAUNCH();
    return DH_OK;
  }
  for (int i = 0; i < n; ++i) {
    const int rc = dh_gemm(&a[i], stream);
    if (rc != DH_OK) return rc;
  }
  return DH_OK;
}

extern "C" int dh_gemm(const dh_gemm_args* a, dh_stream_t stream) {
  hipStream_t st = (hipStream_t)stream;
  DH_REQUIRE(a && a->A && a->B && a->C, "dh_gemm: null pointer");
  DH_REQUIRE(a->M > 0 && a->N > 0 && a->K > 0, "dh_gemm: bad shape %d %d %d", a->M, a->N, a->K);
  DH_REQUIRE(a->dtype == DH_F32 || a->dtype == DH_BF16, "dh_gemm: bad dtype");
  DH_REQUIRE(a->c_dtype == DH_F32 || a->c_dtype == DH_BF16, "dh_gemm: bad c_dtype");
  int split = a->split_k < 1 ? 1 : a->split_k;
  if (a->accumulate) {
    DH_REQUIRE(a->c_dtype == DH_F32, "dh_gemm: accumulate needs fp32 C");
    DH_REQUIRE(!a->bias && a->epilogue == DH_EPI_NONE && !a->residual, "dh_gemm: accumulate excludes bias/epilogue/residual");
  } else {
    DH_REQUIRE(split == 1, "dh_gemm: split_k > 1 requires accumulate");
  }
  if (a->epilogue == DH_EPI_DGELU) DH_REQUIRE(a->aux, "dh_gemm: DGELU needs aux");
  EpiParams e;
  e.M = a->M; e.N = a->N; e.C = a->C; e.ldc = a->ldc; e.bias = a->bias; e.epilogue = a->epilogue;
  e.residual = a->residual; e.ldr = a->ldr; e.aux = a->aux; e.ldaux = a->ldaux; e.accumulate = a->accumulate;
  e.alpha = a->alpha;

  if (a->a_colsum) DH_REQUIRE(a->a_kmajor, "dh_gemm: a_colsum needs a_kmajor");
  // v2 (LDS-DMA + transpose-read) kernel when shapes/alignments allow it (fuses a_colsum)
  if ((a->force_generic == 0 || a->force_generic == 4) && dh_gemm_try_v4(a, split, st)) {
    DH_HELPER_FAILED();
    ++g_gemm_family_calls[0];
    DH_CHECK_LAUNCH();
    return DH_OK;
  }
  DH_REQUIRE(a->force_generic != 4, "dh_gemm: the v4 kernel does not support this problem");
  dh_gemm_args first_touch;
  if (a->accumulate == 2) {
    // first touch through the other kernel families (fp32 atomics into C): clear the slot here, then accumulate as usual
    if (a->ldc == a->N) DH_RT(hipMemsetAsync(a->C, 0, sizeof(float) * (size_t)a->M * a->N, st), "dh_gemm: clearing a first-touch gradient");
    else DH_RT(hipMemset2DAsync(a->C, sizeof(float) * a->ldc, 0, sizeof(float) * a->N, a->M, st), "dh_gemm: clearing a first-touch gradient");
    if (a->a_colsum) DH_RT(hipMemsetAsync(a->a_colsum, 0, sizeof(float) * (size_t)a->M, st), "dh_gemm: clearing a first-touch bias gradient");
    first_touch = *a;
    first_touch.accumulate = 1;
    a = &first_touch;
    e.accumulate = 1;
  }
  if ((a->force_generic == 0 || (a->force_generic >= 31 && a->force_generic <= 33)) && dh_gemm_try_v3(a, split, st)) {
    ++g_gemm_family_calls[1];
    DH_CHECK_LAUNCH();
    return DH_OK;
  }
  const bool v3_pref = a->force_generic >= 31 && a->force_generic <= 33;     // (a preference, not a demand: what v3 declines goes on like auto)
  if ((a->force_generic == 0 || a->force_generic == 3 || v3_pref) && dh_gemm_try_glds(a, split, st)) {
    ++g_gemm_family_calls[2];
    DH_CHECK_LAUNCH();
    return DH_OK;
  }
  if (a->a_colsum) {  // other kernels: separate column-sum pass over A = [K][M]
    int rc = dh_colsum(a->dtype, a->A, a->lda, a->K, a->M, a->a_colsum, 1, stream);
    if (rc != DH_OK) return rc;
  }
  // v1 MFMA path: bf16 operands, 16-B aligned rows, dims multiple of 8
  bool mfma = a->dtype == DH_BF16 && a->force_generic != 1 && (a->lda % 8 == 0) && (a->ldb % 8 == 0) &&
              (((uintptr_t)a->A & 15) == 0) && (((uintptr_t)a->B & 15) == 0);
  if (mfma) {
    // contiguous-dimension extents must be multiples of 8 elements (16-B vector loads)
    if (!a->a_kmajor && a->K % 8) mfma = false;
    if (a->a_kmajor && (a->M % 8 || a->M < 8)) mfma = false;
    if (!a->b_kmajor && a->K % 8) mfma = false;
    if (a->b_kmajor && (a->N % 8 || a->N < 8)) mfma = false;
  }
  ++g_gemm_family_calls[mfma ? 3 : 4];
  if (mfma) {
    int kps = ((a->K + split - 1) / split + BK - 1) / BK * BK;
    split = (a->K + kps - 1) / kps;
    if (a->a_kmajor && a->b_kmajor) launch_mfma<true, true>(a, e, split, kps, st);
    else if (a->a_kmajor) launch_mfma<true, false>(a, e, split, kps, st);
    else if (a->b_kmajor) launch_mfma<false, true>(a, e, split, kps, st);
    else launch_mfma<false, false>(a, e, split, kps, st);
  } else {
    int kps = ((a->K + split - 1) / split + 15) / 16 * 16;
    split = (a->K + kps - 1) / kps;
    if (a->dtype == DH_F32 && a->c_dtype == DH_F32) launch_generic<float, float>(a, e, split, kps, st);
    else if (a->dtype == DH_BF16 && a->c_dtype == DH_BF16) launch_generic<bf16_t, bf16_t>(a, e, split, kps, st);
    else if (a->dtype == DH_BF16 && a->c_dtype == DH_F32) launch_generic<bf16_t, float>(a, e, split, kps, st);
    else launch_generic<float, bf16_t>(a, e, split, kps, st);
  }
  DH_CHECK_LAUNCH();
  return DH_OK;
}

extern "C" int dh_colsum(int dtype, const void* X, int64_t ldx, int M, int N, float* out, int accumulate,
                         dh_stream_t stream) {
  hipStream_t st = (hipStream_t)stream;
  DH_REQUIRE(X && out && M > 0 && N > 0, "dh_colsum: bad args");
  int chunks = dh_cdiv(M, 512);
  if (chunks > 128) chunks = 128;
  int rpb = dh_cdiv(M, chunks);
  chunks = dh_cdiv(M, rpb);
  if (!accumulate && chunks > 1) DH_RT(hipMemsetAsync(out, 0, sizeof(float) * N, st), "dh_colsum: clearing the output");
  dim3 grid(dh_cdiv(N, 64), chunks);
  if (dtype == DH_BF16)
    hipLaunchKernelGGL(colsum_kernel<bf16_t>, grid, dim3(256), 0, st, (const bf16_t*)X, (long)ldx, M, N, out, accumulate, rpb);
  else
    hipLaunchKernelGGL(colsum_kernel<float>, grid, dim3(256), 0, st, (const float*)X, (long)ldx, M, N, out, accumulate, rpb);
  DH_CHECK_LAUNCH();
  return DH_OK;
}
