// GEMM v4 for gfx950: persistent 256 x 256 x 64 tiles, 8 waves, "ping-pong" phase schedule.
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
//     partner is in its load segment: the matrix pipe and the LDS/DMA path stay busy at the same time
//     (measured: the main loop alone runs at ~1.7 PFLOP/s on the tower shapes);
//   * each half-tile (A0/A1/B0/B1 of a K-tile) is consumed in exactly one phase, so it is re-filled two phases
//     later for K-tile t+2: 4 half-tiles (64 KiB) are always in flight and the waits are COUNTED
//     (s_waitcnt vmcnt(8): "everything but the 4 youngest half-tiles"), never a drain;
//   * PERSISTENT: one workgroup per CU walks a list of tiles as ONE continuous stream of K-tiles (round 3).  The four
//     half-tiles of the NEXT tile's K-tile 0 are requested from inside the last two K-tiles of the current one -- in the slots and
//     with the counted waits a longer tile would use for its own K-tiles nk, nk+1 -- and the next work item is decoded (tile
//     coordinates, DMA source addresses) in the load segments of the second-to-last K-tile, which request nothing of their own;
//     so the operands of the next tile land under the last MFMAs and the epilogue, and nothing is decoded, requested or waited
//     for between the main loops.  The epilogue stages through the ring buffer of the last K-tile (the buffer parity of a tile
//     flips when nk is odd); its global stores are never waited for: they drain while the next main loop runs (the counted
//     waits of that tile's first K-tile are widened by the number of stores in flight);
//   * bf16 outputs (MODE_STORE): MFMAs are issued with swapped operands (acc = C^T fragments: a lane owns 4
//     consecutive columns of a row), packed with v_cvt_pk_bf16_f32 and staged in LDS with conflict-free
//     ds_write_b64, then written as 16-byte vectors on full 512-byte row segments; alpha/bias in fp32 before the
//     packing, GELU / dGELU / residual on the staged value; residual / dGELU operands are PREFETCHED before the
//     staging so no load ever sits between two stores (a load behind a store waits for the store: in-order vmcnt);
//   * split-K weight gradients (MODE_PARTIAL): fp32 partial tiles go to a workspace [split][M][N] as fully
//     coalesced 16-byte stores (LDS-staged, 4 passes) and a small reduce kernel adds them into the gradient --
//     measured 2x cheaper than fp32 atomics from every split; MODE_ATOMIC (no workspace) keeps the atomics;
//   * the bias gradient (column sums of the dY operand) rides on the matrix pipe: one extra MFMA per A fragment
//     against an indicator fragment, spread over the tile columns so every workgroup pays ~1/ntx of it.
//
// LDS: 2 stages x {A0, A1, B0, B1} x 16 KiB = 128 KiB (1 block / CU).  Half-tile images are the v2 operand tiles:
//   K-contiguous  [128 rows][64 k]   128-B rows, 16-B chunk c of row r at slot c ^ ((r>>1)&7)
//   contraction-major [64 k][128 out] 256-B rows, 16-B chunk c of row k at slot c ^ ((k&3)<<2)
#include "dh_common.h"
#include <stddef.h>
#include <stdlib.h>
#include <string.h>

#ifndef V4_ABL
#define V4_ABL 0
#endif

#ifndef V4_TRACE
#define V4_TRACE 0
#endif
// Where a wave requests its LDS-DMA pieces inside a K-tile.  0 (rounds 1-2): two half-tiles in each LOAD segment (B1, A1 of the next
// K-tile in phase A, A0, B0 of the one after in phase B), none in the MFMA segments.  1 (round 3): ONE half-tile per segment, the
// MFMA segments included -- phase A load B1, phase A MFMA A1, phase B load A0, phase B MFMA B0.  The in-kernel trace shows why: a
// load segment of phase A (16 fragment reads + 4 LDS-DMA issues at 100+ cycles each inside a segment that also reads fragments)
// takes ~750 cycles against the 512 of the 16 MFMAs it is paired with, phase B's ~560 -- a K-tile costs 2 x 750 + 2 x 560, the
// matrix pipe idles a fifth of the main loop.  An LDS-DMA issued BETWEEN two MFMAs costs the wave ~60 cycles of which 32 are the
// MFMA it follows.
// 2: the same idea at PIECE granularity (a wave requests two 1-KiB pieces per half-tile), balanced against what the ablation
// builds measured (profiles/r03_gemm_trace_stream_sched.txt: fragment reads are free, a piece costs a load segment ~100 cycles and
// an MFMA segment ~60): load A (16 fragment reads) 2 pieces, MFMA A 1, load B (8 reads) 4, MFMA B 1 --
//   load A: B1.1, A1.0 of K-tile kt+1 | MFMA A: A1.1 of kt+1 | load B: A0.0, A0.1, B0.0, B0.1 of kt+2 | MFMA B: B1.0 of kt+2.
#ifndef V4_SCHED
#define V4_SCHED 2
#endif
// Cache policy of the epilogue traffic (measured in-step, CLIP b=512): 0 none; 1 output tiles stored non-temporal (+1.2 %: a
// 128 KB tile per CU per epilogue otherwise displaces the A/B panels the XCD's 4 MB L2 is holding for the next tiles; the 4d-wide
// GELU / dGELU calls gain 7-12 %); 2 only the GELU / dGELU outputs (+1.1 %); 3 = 1 + read-once epilogue operands (residual, dGELU
// pre-activation) loaded non-temporal (+0.3 % more); 4 = 3 + the split-K partial tiles (-1.6 %: the reduce pass wants them cached).
#ifndef V4_NT_STORE
#define V4_NT_STORE 3
#endif
// Round 6: the epilogue operand of the residual / dGELU flavours ("pre": the residual tile, or the stored QuickGELU' factor) -- 1: its first
// half-tile (128 rows x 512 B = 64 KiB) is requested by LDS-DMA from inside the last two K-tiles of the item, in exactly the eight slots per
// wave (and with the counted waits) that used to fetch the NEXT item's K-tile 0, into the ring buffer the epilogue does not stage through;
// the next item's K-tile 0 is requested from the middle of the epilogue instead (behind the last read of that half), and the second half is
// loaded to registers at the START of the epilogue.  0: rounds 2-5 (both halves loaded to registers inside the epilogue: with every CU
// of the chip at the same point of its tile the 32 MB burst took 16-20 k cycles per tile that nothing overlapped --
// profiles/r06_epilogue_traces.txt).
#ifndef V4_PRE_RING
#define V4_PRE_RING 1
#endif
#if V4_TRACE
// tuning aid (-DV4_TRACE=1): s_memtime stamps of workgroups 0 / 100 / 200, waves 0 and 4, read back with dh_v4_trace_read
__device__ long v4_trace_buf[6 * 256];
extern "C" int dh_v4_trace_read(long* host, int n) { return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(v4_trace_buf), sizeof(long) * n); }
extern "C" int dh_v4_trace_clear() { static long z[6 * 256]; return (int)hipMemcpyToSymbol(HIP_SYMBOL(v4_trace_buf), z, sizeof(z)); }
#endif

namespace v4 {

// The pieces of this file that are inline ISA or address-space tricks, behind macros -- so that the SAME kernel body also
// compiles as plain C++ for the host emulation of tests/hipemu (-DDH_HOST_EMU, test infrastructure: HIP threads as fibers, an
// LDS-DMA is a synchronous 16-byte copy per lane, waits are no-ops, the transpose read is the emulation's wave collective).  The
// emulation checks indexing, ring-buffer parity, barrier counts and the epilogues; it cannot see what the counted waits protect.
#ifdef DH_HOST_EMU
#define V4_EMU 1
#define V4_KARGP(ka) ((kargp_t)(&(ka)))
#define V4_OPAQUE_S(x) do { } while (0)
#define V4_OPAQUE_V(x) do { } while (0)
#define V4_KEEP_V2(a, b) do { (void)(a); (void)(b); } while (0)
#define V4_RFL(x) (x)
#define V4_LDS_ADDR(p) 0u
#define V4_RCP(x) (1.f / (x))
#else
#define V4_EMU 0
#define V4_KARGP(ka) ((kargp_t)__builtin_amdgcn_kernarg_segment_ptr())
#define V4_OPAQUE_S(x) asm volatile("" : "+s"(x))
#define V4_OPAQUE_V(x) asm volatile("" : "+v"(x))
#define V4_KEEP_V2(a, b) asm volatile("" ::"v"(a), "v"(b))
#define V4_RFL(x) __builtin_amdgcn_readfirstlane(x)
#define V4_LDS_ADDR(p) ((uint32_t)(uintptr_t)(const __attribute__((address_space(3))) unsigned char*)(p))
#define V4_RCP(x) __builtin_amdgcn_rcpf(x)
#endif

typedef __attribute__((ext_vector_type(4))) uint32_t u32x4_t;
// 16-byte store of an output tile row piece; V4_NT_STORE: non-temporal (the tile is not read again by this kernel: keep it
// from displacing the operand panels in the XCD's L2)
// 16-byte load of an epilogue operand that is read exactly once (residual / dGELU pre-activation)
template <bool NT>
__device__ __forceinline__ uint4 load_once16(const unsigned char* p) {
  if (NT) {
    const u32x4_t w = __builtin_nontemporal_load(reinterpret_cast<const u32x4_t*>(p));
    return make_uint4(w.x, w.y, w.z, w.w);
  }
  return *reinterpret_cast<const uint4*>(p);
}

template <bool NT>
__device__ __forceinline__ void store_c16(unsigned char* p, const uint4& v) {
  if (NT) {
    u32x4_t w = {v.x, v.y, v.z, v.w};
    __builtin_nontemporal_store(w, reinterpret_cast<u32x4_t*>(p));
  } else {
    *reinterpret_cast<uint4*>(p) = v;
  }
}


struct EpiParams {
  int M, N;
  void* C; long ldc;
  const float* bias;
  int epilogue;
  const void* residual; long ldr;
  void* aux; long ldaux;
  float alpha;
  float* a_colsum;
  float* ws;          // MODE_PARTIAL: [nsplit][M][N] fp32
  float* ws_cs;       // MODE_PARTIAL + a_colsum: [nsplit][ntx][M] fp32 bias-gradient partials
  // MODE_MAXSIM (FILIP token-wise max-sim, filip.py:96-105): C = raw [ms_b][ms_B] fp32 (+= mean_j max_m), aux = arg-max uint8 [M][ms_B]
  int ms_J, ms_b, ms_B;
  // MODE_CE_FWD / MODE_CE_BWD (vocabulary cross-entropy of the masked-LM head, declip.py:326-334, without fp32 logits in HBM):
  // bias = the [V] bias vector (read per tile, V may exceed MAX_BIAS_N), N = V (ragged last tile: columns >= V are masked)
  const long long* ce_labels;   // [ce_n] target ids
  const float* ce_lse;          // BWD: [ce_n] log-sum-exp of every row (from the forward)
  const float* ce_g;            // BWD: [ce_n] upstream gradient of every row loss
  float* ce_part;               // FWD: [M][ntx][2] per-tile (max, sum exp) partials
  float* ce_lab;                // FWD: [M] logit at the label column
  int ce_n, ce_V, ce_ldd;       // valid rows, vocabulary size, row stride of the dl output (BWD; columns >= ce_ldd are not stored)
};

// The kernel's ONLY parameter: the kernarg segment is exactly this struct, and the kernel reads most of it LATE, through an
// opaque pointer to the kernarg segment, right where a value is needed (prologue of the launch, start of each epilogue).
// Passed as ordinary by-value arguments, these ~45 scalars were preloaded and kept (then spilled: 150-280 SGPR spills, with
// v_readlane reloads inside the K loop of the dW kernel) across the main loop, which itself needs only four strides.
// MODE_GROUP: several weight-gradient problems with the same contraction length (the four dW GEMMs of a transformer block:
// dY^T X for in_proj, out_proj, c_fc, c_proj) as ONE persistent launch.  A problem on its own has 9-36 tiles for 256 CUs, so it
// was cut into 7-64 K-slices (28 slices of 14 K-tiles for a 768 x 768 weight: prologue / epilogue bound, 120-275 TFLOP/s) and
// every slice wrote a 256 KB fp32 partial tile; together the four have 108 (vision) / 48 (text) tiles, fill the chip with 5-7
// slices of >= 57 K-tiles, write 4x fewer partial tiles and need one reduce pass instead of four.
constexpr int MAX_GROUP = 4;
struct GProb {
  const bf16_t* A; const bf16_t* B;      // dY [K][M] and X [K][N] (both contraction-major)
  long lda, ldb;
  long ws_off;                            // float offset of this problem's [M][N] block inside one K-slice slab of the workspace
  long cs_off;                            // float offset of its [ntx][M] bias-gradient block inside one slab of the colsum area
  float* a_colsum;                        // bias gradient (or null)
  int M, N, ntx, nty, tile0, pad_;        // tile0: index of its first tile in the concatenated tile list (INT_MAX: unused entry)
};
struct KArgs {
  const bf16_t* A; long lda;
  const bf16_t* B; long ldb;
  int M, N, K, k_per_split, ntx, nty, nitems, n_full, S, dyn;   // dyn: 0 static partition, 1 dynamic after the first item, 2 fully dynamic + stealing
  int* sched;
  EpiParams e;
  int group_m;                            // tile order (tile_from_logical)
  // MODE_GROUP only
  int T, ngrp;                            // tiles of all problems together; problems
  long zs, cs_zs;                         // floats per K-slice slab: partial tiles (sum M*N), bias-gradient partials (sum ntx*M)
  GProb gp[MAX_GROUP];
};
#if V4_EMU
typedef const unsigned char* kargp_t;
template <typename T> __device__ __forceinline__ T karg_load(kargp_t kp, int off) { T v; memcpy(&v, kp + off, sizeof(T)); return v; }
#else
typedef const __attribute__((address_space(4))) unsigned char* kargp_t;
template <typename T> __device__ __forceinline__ T karg_load(kargp_t kp, int off) {
  typedef const __attribute__((address_space(4))) T* P;
  return *(P)(kp + off);
}
#endif
#define KARG(kp, T, field) karg_load<T>((kp), (int)offsetof(KArgs, field))
#define GPARG(kp, p, T, field) karg_load<T>((kp), (int)(offsetof(KArgs, gp) + offsetof(GProb, field)) + (p) * (int)sizeof(GProb))

constexpr int BM = 256, BN = 256, BK = 64;
constexpr int HALF_BYTES = 128 * BK * 2;          // 16 KiB
constexpr int STAGE_BYTES = 4 * HALF_BYTES;       // A0 A1 B0 B1
constexpr int BIAS_OFF = 2 * STAGE_BYTES;         // behind the ring: the whole bias vector (fp32, N <= MAX_BIAS_N), loaded once per workgroup
constexpr int MAX_BIAS_N = 4096;
constexpr int SCHED_OFF = BIAS_OFF + MAX_BIAS_N * 4;   // one word: the work item after next (dynamic scheduling)
constexpr int LDS_BYTES = SCHED_OFF + 16;               // 144 KiB + [0] next item
// MODE: what happens to the finished tile.  The bf16 epilogue flavours are separate instantiations (straight-line code:
// the kernel lives at the 256-VGPR cap, runtime epilogue switches cost spills).
constexpr int MODE_STORE = 0, MODE_STORE_GELU = 1, MODE_STORE_DGELU = 2, MODE_STORE_RES = 3, MODE_ATOMIC = 4, MODE_PARTIAL = 5, MODE_GROUP = 6, MODE_MAXSIM = 7, MODE_CE_FWD = 8, MODE_CE_BWD = 9;

typedef __attribute__((ext_vector_type(4))) short s16x4;

// LDS-DMA of 64 lanes x 16 B (1 KiB) to a wave-uniform LDS address.  INLINE ASM on purpose: when hipcc (ROCm 7.2) can see an
// LDS-DMA in flight it puts s_waitcnt vmcnt(0) in front of the next LDS access it cannot prove disjoint (every ds_read /
// ds_write of the epilogue, every transpose read), i.e. it drains the pipeline.  With the DMA (and the transpose reads)
// issued from asm the compiler sees no VMEM traffic in the main loop at all and every wait there is one of ours.
#if V4_EMU
__device__ __forceinline__ void dma16(const bf16_t* src, unsigned char* lds_dst_uniform) { memcpy(lds_dst_uniform + 16 * emu_lane(), src, 16); }
template <int N> __device__ __forceinline__ void wait_vmcnt() { }
#else
__device__ __forceinline__ void dma16(const bf16_t* src, unsigned char* lds_dst_uniform) {
  const uint32_t l = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)lds_dst_uniform;
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(l), "v"(src) : "memory", "m0");
}
template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
#endif
// "all but the 4 youngest half-tiles and the `s` epilogue stores issued between them" (s is wave-uniform)
template <int N> __device__ __forceinline__ void wait_vmcnt_plus(int s) {   // "all but the N youngest loads and the s stores between them"
  if (s == 0) wait_vmcnt<N>();
  else if (s == 16) wait_vmcnt<N + 16>();
  else if (s == 32) wait_vmcnt<N + 32>();
  else wait_vmcnt<N>();                            // unknown count: conservative
}
__device__ __forceinline__ void wait_vmcnt_7_plus(int s) {   // V4_SCHED 2: "all but the 7 youngest pieces and the s stores between them"
  if (s == 0) wait_vmcnt<7>();
  else if (s == 8) wait_vmcnt<15>();
  else if (s == 16) wait_vmcnt<23>();
  else if (s == 32) wait_vmcnt<39>();
  else wait_vmcnt<7>();
}
// end of an epilogue: "everything but the s youngest operations" (= the epilogue's own stores)
__device__ __forceinline__ void wait_vmcnt_tail(int s) {
  if (s == 16) wait_vmcnt<16>();
  else if (s == 8) wait_vmcnt<8>();
  else if (s == 32) wait_vmcnt<32>();
  else wait_vmcnt<0>();                            // no stores / unknown count: drain
}
#if V4_EMU
__device__ __forceinline__ void wait_lgkm0() { }
#else
__device__ __forceinline__ void wait_lgkm0() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
#endif

// fragment of a K-contiguous half-tile: rows r0 + (lane&31), k = 16*s + 8*(lane>>5) .. +7
__device__ __forceinline__ bf16x8_t frag_kcontig(const unsigned char* tile, int r0, int s, int lane) {
  const int row = r0 + (lane & 31);
  const int chunk = 2 * s + (lane >> 5);
  return *reinterpret_cast<const bf16x8_t*>(tile + row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4));
}
// Fragments of a contraction-major half-tile come out of LDS with the transpose read ds_read_b64_tr_b16, as INLINE ASM on
// purpose: for the builtin hipcc (ROCm 7.2) inserts s_waitcnt vmcnt(0) in front of every read while an LDS-DMA is in flight
// (it cannot prove the read does not alias the DMA target), which drains the whole pipeline once per phase.  The asm read is
// invisible to that pass; the consumer waits lgkmcnt(0) explicitly after the phase barrier (wait_lgkm0 + sched_barrier in
// the MFMA segments).  Address = per-lane VGPR (ring buffer + kmajor_lane_off) + immediate (region, k16-step).
// per-lane byte offset (inside a half-tile region, k16-step 0) of the lane's first transpose-read for out columns o0 + ..
__device__ __forceinline__ uint32_t kmajor_lane_off(int o0, int lane) {
  const int t = lane & 15;
  const int n = o0 + ((lane >> 4) & 1) * 16 + 4 * (t & 3);
  const int k = 8 * (lane >> 5) + (t >> 2);
  const int sw = (t >> 2) << 2;
  return k * 256 + ((((n >> 3) ^ sw)) << 4) + ((n & 7) << 1);
}
template <int OFF> __device__ __forceinline__ bf16x8_t frag_km(uint32_t addr) {
  s16x4 lo, hi;
#if V4_EMU
  const unsigned char* base = (const unsigned char*)emu_dyn_lds() + addr;
  lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(base + OFF);
  hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(base + OFF + 1024);
#else
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(lo) : "v"(addr), "n"(OFF));
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(hi) : "v"(addr), "n"(OFF + 1024));
#endif
  union { struct { s16x4 a, b; } s; bf16x8_t v; } u;
  u.s.a = lo; u.s.b = hi;
  return u.v;
}
// the 4 k16-step fragments of one 32-row / 32-column group of half-tile REGION (0 A0, 1 A1, 2 B0, 3 B1)
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

// work item w -> (tile_x, tile_y, split z).  Items with the same z are consecutive (they share A/B panels); inside
// a split the tile order is XCD-aware (workgroup p and all its items w = p + i*grid sit on XCD p % 8 when the grid
// is a multiple of 8) and grouped so that the tiles an XCD runs concurrently share A and B panels in its 4 MiB L2.
// GROUP_M (KArgs.group_m): rows of tiles per group.  An XCD owns a CONTIGUOUS chunk of the logical order, so the group shape
// decides which operand panels two XCDs both fetch from the fabric: with groups of 8 rows and ntx columns, a chunk boundary
// inside a group hands the same 8 A panels to two (ntx = 3) or three (ntx = 12: 8 x 4 tiles per XCD) XCDs -- the PMC counters
// showed 2.7x the algorithmic read traffic on the N = 3072 forward GEMM.  group_m = 1 is row-major: all ntx tiles of a row
// (one A panel) sit on one XCD; only the small weight operand is fetched by every XCD.
// a / b for 0 <= a < 2^21, b > 0 through the fp32 reciprocal (exact after one correction step).  The work-item decode runs once per
// tile on every wave between the main loop and the epilogue; the compiler's 32-bit integer division is a ~35-instruction dependent
// chain, and five of them were most of the ~900 cycles the in-kernel trace shows for "next tile requested".
__device__ __forceinline__ int fdiv(int a, int b) {
  // (the divisor's float is made opaque: a loop-invariant reciprocal would otherwise be hoisted out of the persistent tile loop and kept
  // live across the main loop -- in the residual flavour it was SPILLED, and its reload in the dynamic hand-over came with an
  // s_waitcnt vmcnt(0): the compiler cannot see the LDS-DMA in flight, so waiting for its scratch load drained the whole operand pipeline
  // once per tile)
  float fb = (float)b;
  V4_OPAQUE_V(fb);
  int q = (int)((float)a * V4_RCP(fb));
  const int r = a - q * b;
  q += r >= b ? 1 : 0;
  q -= r < 0 ? 1 : 0;
  return V4_RFL(q);      // every caller passes wave-uniform values: keep the quotient in a scalar register
}
__device__ __forceinline__ void tile_from_logical(int b, int ntx, int nty, int& tile_x, int& tile_y, int GROUP_M = 8) {
  const int in_group = GROUP_M * ntx;
  const int gid = fdiv(b, in_group);
  const int first_m = gid * GROUP_M;
  const int gsz = min(nty - first_m, GROUP_M);
  const int rem = b - gid * in_group;
  tile_x = fdiv(rem, gsz);
  tile_y = first_m + (rem - tile_x * gsz);
}
// Work items of a launch.  Plain launches: n_fullitems = tiles x splits whole items.  TAIL-SLICED launches (bf16 outputs
// whose tile count leaves the last round mostly empty, e.g. 300 tiles on 256 CUs): n_fullitems = n_full whole tiles (a
// multiple of the grid) and the remaining `rem` tiles are cut into S K-slices each -- slice j = lt*S + s computes K-tiles
// [s*nkt/S, (s+1)*nkt/S) of tile n_full + lt into a private fp32 tile of the workspace, and tail_fixup_kernel sums the slices
// and applies the epilogue.  The last round then takes ~1/S of a tile time on all CUs instead of a full one on a few.
//
// Distribution: XCD x (workgroups p = x mod 8) owns a CONTIGUOUS chunk of the whole-item list and a contiguous chunk of the
// slice list; its list = whole items first, slices last.  A workgroup is identified by (x, position l in that list).  The
// first position of a workgroup is static (l = p / 8); every further one comes from a per-XCD atomic counter (DYNAMIC
// scheduling: when other kernels -- RCCL during the overlapped gradient all-reduce -- hold some CUs, the workgroups that do
// run simply take more tiles; with a static partition a 4-CU hog slowed this GEMM by 39 %), with stealing from the other
// XCDs' lists once the own one is exhausted.  Without a scheduler buffer the partition is static (l += workgroups on x).
struct Item { int tile_x, tile_y, z, kbeg, nk, slice, p; };
__device__ __forceinline__ void chunk_of(int n, int x, int& start, int& len) {
  const int q = n >> 3, r = n & 7;
  len = q + (x < r ? 1 : 0);
  start = x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q;
}
__device__ __forceinline__ bool decode_pos(int x, int l, int ntx, int nty, int n_fullitems, int n_full, int S, int K, int k_per_split,
                                           Item& it, int gm = 8) {
  int sf, lf;
  chunk_of(n_fullitems, x, sf, lf);
  if (l < lf) {
    const int logical = sf + l, nb = ntx * nty;
    it.z = fdiv(logical, nb);
    tile_from_logical(logical - it.z * nb, ntx, nty, it.tile_x, it.tile_y, gm);
    it.kbeg = it.z * k_per_split;
    it.nk = (min(K, it.kbeg + k_per_split) - it.kbeg) / BK;
    it.slice = -1;
    it.p = 0;
    return true;
  }
  if (S == 0) return false;
  const int rem = ntx * nty - n_full;
  int ss, ls;
  chunk_of(rem * S, x, ss, ls);
  const int j0 = l - lf;
  if (j0 >= ls) return false;
  const int j = ss + j0, lt = fdiv(j, S), sl = j - lt * S, nkt = K / BK;
  tile_from_logical(n_full + lt, ntx, nty, it.tile_x, it.tile_y, gm);
  const int k0 = fdiv(sl * nkt, S), k1 = fdiv((sl + 1) * nkt, S);
  it.z = 0; it.kbeg = k0 * BK; it.nk = k1 - k0; it.slice = j; it.p = 0;
  return true;
}
// MODE_GROUP: position l of XCD x's chunk of the (K-slice, tile) list -> item.  Tiles of all problems are concatenated; inside a
// problem the usual grouped order (tile_from_logical), so the ~32 items an XCD runs together share the dY / X panels of one
// K-slice in its L2.
__device__ __forceinline__ bool decode_group(kargp_t kp, int x, int l, int nitems, int T, int K, int k_per_split, Item& it) {
  int sf, lf;
  chunk_of(nitems, x, sf, lf);
  if (l >= lf) return false;
  const int logical = sf + l;
  it.z = fdiv(logical, T);
  const int t = logical - it.z * T;
  int p = 0;
#pragma unroll
  for (int q = 1; q < MAX_GROUP; ++q) p = t >= GPARG(kp, q, int, tile0) ? q : p;
  it.p = p;
  tile_from_logical(t - GPARG(kp, p, int, tile0), GPARG(kp, p, int, ntx), GPARG(kp, p, int, nty), it.tile_x, it.tile_y);
  it.kbeg = it.z * k_per_split;
  it.nk = (min(K, it.kbeg + k_per_split) - it.kbeg) / BK;
  it.slice = -1;
  return true;
}
__device__ __forceinline__ int wgs_on_xcd(int x, int grid) { return x < grid ? ((grid - 1 - x) >> 3) + 1 : 0; }

template <bool TA, bool TB, int MODE>
__global__ __launch_bounds__(512, 2) void gemm_v4_kernel(const KArgs ka_unused) {
  // launch-time view of the arguments (prologue); the epilogues re-read what they need (see KArgs)
  kargp_t kp = V4_KARGP(ka_unused);
  V4_OPAQUE_S(kp);
  const bf16_t* A = KARG(kp, const bf16_t*, A);
  const bf16_t* B = KARG(kp, const bf16_t*, B);
  long lda = KARG(kp, long, lda), ldb = KARG(kp, long, ldb);
  EpiParams e;
  e.bias = KARG(kp, const float*, e.bias);
  e.a_colsum = KARG(kp, float*, e.a_colsum);
  int M = KARG(kp, int, M), N = KARG(kp, int, N), K = KARG(kp, int, K), k_per_split = KARG(kp, int, k_per_split);
  int ntx = KARG(kp, int, ntx), nty = KARG(kp, int, nty), nitems = KARG(kp, int, nitems), n_full = KARG(kp, int, n_full), S = KARG(kp, int, S);
  int* sched = KARG(kp, int*, sched);
  const int dyn = sched ? KARG(kp, int, dyn) : 0;
  DH_DYN_LDS_A16(unsigned char, smem);
  constexpr bool SWAP = MODE != MODE_ATOMIC;
  constexpr bool STORE = MODE <= MODE_STORE_RES;
  constexpr bool GROUP = MODE == MODE_GROUP;
  constexpr bool HAS_PRE = MODE == MODE_STORE_DGELU || MODE == MODE_STORE_RES;
  constexpr bool PRE_RING = HAS_PRE && V4_PRE_RING && V4_SCHED == 2;
  // (a launch is either all whole items or -- S > 0, the few-tile launches -- all K-slices, which leave fp32 tiles and need no operand)
  const bool pring = PRE_RING && S == 0;
  // item decoding and the per-problem operands (MODE_GROUP: from the problem table, read late like everything else)
#define DECODE(x_, l_, it_) (GROUP ? decode_group(kp, (x_), (l_), nitems, KARG(kp, int, T), K, k_per_split, (it_)) \
                                   : decode_pos((x_), (l_), ntx, nty, n_fullitems, n_full, S, K, k_per_split, (it_), KARG(kp, int, group_m)))
#define LOAD_PROBLEM(kq_, p_)                                                                   \
  do {                                                                                          \
    if (GROUP) {                                                                                \
      A = GPARG(kq_, p_, const bf16_t*, A); B = GPARG(kq_, p_, const bf16_t*, B);               \
      lda = GPARG(kq_, p_, long, lda); ldb = GPARG(kq_, p_, long, ldb);                         \
      M = GPARG(kq_, p_, int, M); N = GPARG(kq_, p_, int, N);                                   \
      astep = (long)BK * lda; bstep = (long)BK * ldb;                                           \
    }                                                                                           \
  } while (0)
  constexpr int NP = 2;                          // 1-KiB pieces of a half-tile per wave
  // compile-time ablation (tuning aid: -DV4_ABL=mask; 1 no DMA, 2 no MFMA, 4 no fragment reads, 8 no epilogue)
  constexpr bool do_dma = !(V4_ABL & 1), do_mfma = !(V4_ABL & 2), do_frag = !(V4_ABL & 4), do_epi = !(V4_ABL & 8);
  const int t = threadIdx.x;
  const int lane = t & 63;
  const int wave = V4_RFL(t >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  const int ar = wm * 64, br = wn * 32;             // this wave's rows inside an A half / columns inside a B half
  unsigned char* const wdst = smem + wave * 2048;   // this wave's slice of a half-tile
  const uint32_t lds0 = V4_LDS_ADDR(smem);
  long astep = TA ? (long)BK * lda : BK;             // (MODE_GROUP: per problem, LOAD_PROBLEM)
  long bstep = TB ? (long)BK * ldb : BK;

  // DMA sources.  Shapes are whole tiles (host-checked).  Each wave stages pieces 2w, 2w+1 of every half-tile: two per-lane
  // pointers per operand (the source swizzle of a K-contiguous piece depends on the piece), half +1 = a wave-uniform offset
  // (128 rows / 128 columns).  ap* / bp* point at K-tile kt+1 while K-tile kt is multiplied (A1, B1 of kt+1 and A0, B0 of kt+2
  // are issued there); from the second-to-last K-tile of an item on they point at K-tile 0 of the NEXT item (see the main loop).
  const long a_dh = TA ? 128 : 128 * lda;
  const long b_dh = TB ? 128 : 128 * ldb;
  const bf16_t *ap[NP], *bp[NP];
  // The cross-entropy modes take a vocabulary that is not a multiple of the tile (N = 49409 in the tests; the reference's 49408 is):
  // piece_src clamps the rows of half 0 to the last row, and the wave-uniform "+128 rows" of half 1 would then point up to 127 rows
  // PAST the matrix (an out-of-bounds LDS-DMA read: harmless inside the allocator's pool, a memory fault at the end of a mapping --
  // seen once as an abort of the full test suite).  There the offset to half 1 is per lane and clamped like half 0.
  constexpr bool RAGGED_B = (MODE == MODE_CE_FWD || MODE == MODE_CE_BWD) && !TB;
  long bdh[NP];
#define BDHQ (RAGGED_B ? bdh[q] : b_dh)
#define BDH_(qq) (RAGGED_B ? bdh[qq] : b_dh)
  // (ln_: the lane id -- the OPAQUE per-tile copy inside the tile loop, or the per-lane parts of the source addresses are hoisted out of
  // the persistent loop, kept live across it and spilled: their reloads come with an s_waitcnt vmcnt(0) that drains the LDS-DMA pipeline)
#define SETUP_SRC(m0_, n0_, kbeg_, ln_)                                                       \
  do {                                                                                        \
    _Pragma("unroll") for (int q = 0; q < NP; ++q) {                                          \
      ap[q] = piece_src<TA>(A, lda, wave * 2 + q, (ln_), (m0_), M, (kbeg_));                  \
      bp[q] = piece_src<TB>(B, ldb, wave * 2 + q, (ln_), (n0_), N, (kbeg_));                  \
      if (RAGGED_B) {                                                                         \
        const int rl_ = (wave * 2 + q) * 8 + ((ln_) >> 3);                                    \
        const int r0_ = min((n0_) + rl_, N - 1), r1_ = min((n0_) + rl_ + 128, N - 1);         \
        bdh[q] = (long)(r1_ - r0_) * ldb;                                                     \
      }                                                                                       \
    }                                                                                         \
  } while (0)
#define ISSUE_H(P, OFF, REGION, buf)                                                          \
  do {                                                                                        \
    if (!do_dma) break;                                                                       \
    _Pragma("unroll") for (int q = 0; q < NP; ++q)                                            \
      dma16((P)[q] + (OFF), wdst + (buf) * STAGE_BYTES + (REGION) * HALF_BYTES + q * 1024);   \
  } while (0)
#define ISSUE_PIECE(P, OFF, REGION, buf, q)                                                   \
  do {                                                                                        \
    if (!do_dma) break;                                                                       \
    dma16((P)[q] + (OFF), wdst + (buf) * STAGE_BYTES + (REGION) * HALF_BYTES + (q) * 1024);   \
  } while (0)
  // piece q of half-tile region r of ring buffer `buf` <- rows r*32 + wave*4 + q*2 + (lane >> 5), 16-byte chunk lane & 31 of the "pre"
  // tile: the buffer then holds rows 0..127 of the tile row-major, 512 B per row (pre_ptr: this lane's address in row wave*4 + (lane>>5))
#define ISSUE_PRE(r, q, buf)                                                                  \
  do {                                                                                        \
    if (!do_dma) break;                                                                       \
    dma16(pre_ptr + (long)((r) * 32 + (q) * 2) * pre_ldv, wdst + (buf) * STAGE_BYTES + (r) * HALF_BYTES + (q) * 1024); \
  } while (0)
#define ADVANCE_SRC()                                                                         \
  do {                                                                                        \
    _Pragma("unroll") for (int q = 0; q < NP; ++q) { ap[q] += astep; bp[q] += bstep; }        \
  } while (0)
#define WAITV(N) wait_vmcnt<(N)>()
#define MFMA(ACCV, AF, BF) \
  if (!do_mfma) { V4_KEEP_V2(AF, BF); } else ACCV = SWAP ? __builtin_amdgcn_mfma_f32_32x32x16_bf16(BF, AF, ACCV, 0, 0, 0) : __builtin_amdgcn_mfma_f32_32x32x16_bf16(AF, BF, ACCV, 0, 0, 0)

#if V4_TRACE
  const int trace_slot = (blockIdx.x == 0 ? 0 : blockIdx.x == 100 ? 1 : blockIdx.x == 200 ? 2 : -1);
  const bool tracing = trace_slot >= 0 && (t == 0 || t == 256);
  long* trace_p = v4_trace_buf + (trace_slot * 2 + (t >> 8)) * 256;
  int trace_n = 0;
#define TRACE() do { if (tracing && trace_n < 255) trace_p[1 + trace_n++] = __builtin_readcyclecounter(); } while (0)
#if V4_TRACE >= 2          // 2: + four stamps inside each of the last two K-tiles of an item; 3: + four stamps inside K-tile 2 of every item (a steady-state K-tile)
#define TRACE_FINE() TRACE()
#else
#define TRACE_FINE() do { } while (0)
#endif
#else
#define TRACE() do { } while (0)
#define TRACE_FINE() do { } while (0)
#endif
  // ---- work distribution state (see decode_pos)
  const int xcd = blockIdx.x & 7, grid = gridDim.x;
  int n_fullitems = S ? n_full : nitems;
  int* const sched_lds = reinterpret_cast<int*>(smem + SCHED_OFF);      // [0] = packed (x << 24 | l) of the item after next, or -1
  auto list_len = [&](int x) {
    int sf, lf, ss = 0, ls = 0;
    chunk_of(n_fullitems, x, sf, lf);
    if (S) chunk_of((ntx * nty - n_full) * S, x, ss, ls);
    return lf + ls;
  };
  // thread 0 only (dyn == 1: every workgroup's first position is static, the further ones come from its XCD's counter).
  // FETCH: one returning atomic on the own XCD's counter AND the wait for it in ONE asm statement.  Rounds 1-2 issued the atomic
  // at the start of the epilogue and read its result register at the end, relying on the compiler neither copying, spilling nor
  // re-using that register in between ("checked in the ISA") -- an invariant nothing enforces: after round 3 re-shaped the
  // hand-over the residual flavour re-used the register inside the epilogue, the late-arriving counter value overwrote a bias
  // operand of lane 0 and a handful of outputs per launch lost their bias under DH_V4_DYNAMIC=1
  // (test_v4_dynamic_tile_distribution_is_bit_identical_to_the_static_one found it).  No register reservation mechanism exists
  // in this toolchain (amdgpu_num_vgpr is ignored under __launch_bounds__), so the fetch is synchronous now: it sits at the very
  // end of the epilogue, behind the wait that already covers everything older than the epilogue's stores, and costs the store
  // drain + one L2 atomic round trip per item (measured: see DESIGN_HISTORY.md s4).
  // (launch constants of this workgroup's XCD, computed once: the fetch at the end of every epilogue needs nothing else alive)
  const int my_wgs = V4_RFL(wgs_on_xcd(xcd, grid)), my_len = V4_RFL(list_len(xcd));
#if V4_EMU
#define FETCH(RAW) do { RAW = 1 << 22; if (my_wgs < my_len) RAW = atomicAdd(sched + xcd, 1); } while (0)
#else
#define FETCH(RAW)                                                                                             \
  do {                                                                                                        \
    RAW = 1 << 22;                                                                                            \
    if (my_wgs < my_len) {                                                                                    \
      const int one_ = 1;                                                                                     \
      int* p_ = sched + xcd;                                                                                  \
      asm volatile("global_atomic_add %0, %1, %2, off sc0\n\ts_waitcnt vmcnt(0)" : "=&v"(RAW) : "v"(p_), "v"(one_) : "memory"); \
    }                                                                                                         \
  } while (0)
#endif
  // packed position of the next item, or -1
  auto fetch_finish = [&](int raw) -> int {
    const int l0 = my_wgs + raw;
    return l0 < my_len ? ((xcd << 24) | l0) : -1;
  };
  Item nxt;
  bool have;
  {
    const int my_l = blockIdx.x >> 3;                                    // position in XCD `xcd`'s list
    have = DECODE(xcd, my_l, nxt);                                       // grid <= items: always true
    if (t == 0) {
      int v;
      if (dyn) { int r0; FETCH(r0); v = fetch_finish(r0); }
      else { const int l = my_l + wgs_on_xcd(xcd, grid); v = (xcd << 24) | l; }     // static partition (validity checked at decode)
      sched_lds[0] = v;
    }
  }
  // FAST hand-over to the next item (the forward / dX launches: one K-slice, row-major tile order, static partition, no tail
  // slices, whole tiles).  A workgroup's items are positions l, l + s, l + 2s ... of its XCD's chunk, so the next tile is the
  // current one plus a constant (s mod ntx, s / ntx) with a carry, and its DMA sources are the current ones plus a wave-uniform
  // 64-bit offset (the per-lane part of a source address does not depend on the tile): ~15 scalar + 8 vector instructions.  The
  // general decode (kernel arguments re-read, five integer divisions, four per-lane 64-bit address computations) measured
  // ~1.8 k cycles per wave group in the in-kernel trace -- three MFMA segments, which nothing can hide.  It stays for the
  // launches that need it (grouped / split-K weight gradients, tail slices, dynamic scheduling, the ragged vocabulary of the
  // cross-entropy modes): their items are 50+ K-tiles long.
  // Round 5: the DYNAMIC distribution (every rank of a multi-GPU job) takes the fast hand-over too.  Its next position l comes from
  // the XCD's counter instead of l + s, but for these launches a position is still just a row-major tile index (chunk start + l):
  // one reciprocal division gives the tile, and the sources move by the same wave-uniform offsets.  The general decode was most of
  // what the dynamic mode cost when nothing else ran (~2 % of the step).
  const bool fast = !GROUP && !RAGGED_B && (dyn == 0 || dyn == 1) && S == 0 && nitems == ntx * nty && KARG(kp, int, group_m) == 1;
  int left = 0, sx_f = 0, sy_f = 0, d_m = 0, d_n = 0, sf_f = 0;
  const int ntx_f = ntx;
  if (fast) {
    int sf, lf;
    chunk_of(nitems, xcd, sf, lf);
    sf_f = sf;
    const int st = wgs_on_xcd(xcd, grid), my_l = blockIdx.x >> 3;
    left = fdiv(lf - my_l + st - 1, st);               // items of this workgroup, the current one included (static partition)
    sy_f = fdiv(st, ntx);
    sx_f = st - sy_f * ntx;
  }
  if (STORE) {   // bias -> LDS once (no global load may sit between the epilogue stores of the persistent loop)
    for (int q = t; q < N / 4; q += 512)
      *reinterpret_cast<float4*>(smem + BIAS_OFF + 16 * q) = e.bias ? *reinterpret_cast<const float4*>(e.bias + 4 * q) : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  // ---- ONE continuous stream of K-tiles across the items of this workgroup.  K-tile kt of an item sits in ring buffer
  // (kt ^ tp) & 1; the four half-tiles of the NEXT item's K-tile 0 are requested from inside the last two K-tiles of the current
  // one, exactly where a longer item would request its own K-tiles nk and nk+1 (same slots, same counted waits), so they land
  // under the last MFMAs and the whole epilogue instead of being requested -- and waited for -- after the main loop; the next
  // item is decoded (tile coordinates, source addresses) in the load segments of the second-to-last K-tile, which have no
  // requests of their own to make.  The epilogue stages through the buffer of the last K-tile; the buffer of the next item's
  // K-tile 0 is the other one, so the tile parity `tp` flips when nk is odd.
  int tp = 0;
  if (have) {   // first item: the four half-tiles of its K-tile 0, in steady-state FIFO order
    LOAD_PROBLEM(kp, nxt.p);
    SETUP_SRC(nxt.tile_y * BM, nxt.tile_x * BN, nxt.kbeg, lane);
    ISSUE_H(ap, 0, 0, 0); ISSUE_H(bp, 0, 2, 0); ISSUE_H(bp, BDHQ, 3, 0); ISSUE_H(ap, a_dh, 1, 0);
    wait_vmcnt<0>();                   // (once per launch; later items find their K-tile 0 landed at the end of the previous epilogue)
  }
  wait_lgkm0();
  V4_BARRIER();
  int pend = 0;                      // epilogue stores of the previous tile that may still be in flight (per wave-instruction stream)

  // everything the epilogue needs, re-read from the kernarg segment (see KArgs)
#define RELOAD_ARGS()                                                                                   \
  do {                                                                                                  \
    kargp_t kq = V4_KARGP(ka_unused);                                                                   \
    V4_OPAQUE_S(kq);                                                                                    \
    e.M = KARG(kq, int, M); e.N = KARG(kq, int, N); e.C = KARG(kq, void*, e.C); e.ldc = KARG(kq, long, e.ldc); \
    e.bias = KARG(kq, const float*, e.bias); e.epilogue = KARG(kq, int, e.epilogue);                    \
    e.residual = KARG(kq, const void*, e.residual); e.ldr = KARG(kq, long, e.ldr);                      \
    e.aux = KARG(kq, void*, e.aux); e.ldaux = KARG(kq, long, e.ldaux); e.alpha = KARG(kq, float, e.alpha); \
    e.a_colsum = KARG(kq, float*, e.a_colsum); e.ws = KARG(kq, float*, e.ws); e.ws_cs = KARG(kq, float*, e.ws_cs); \
    e.ms_J = KARG(kq, int, e.ms_J); e.ms_b = KARG(kq, int, e.ms_b); e.ms_B = KARG(kq, int, e.ms_B);               \
    e.ce_labels = KARG(kq, const long long*, e.ce_labels); e.ce_lse = KARG(kq, const float*, e.ce_lse);            \
    e.ce_g = KARG(kq, const float*, e.ce_g); e.ce_part = KARG(kq, float*, e.ce_part); e.ce_lab = KARG(kq, float*, e.ce_lab); \
    e.ce_n = KARG(kq, int, e.ce_n); e.ce_V = KARG(kq, int, e.ce_V); e.ce_ldd = KARG(kq, int, e.ce_ldd);            \
  } while (0)
  // what the choice and the source addresses of the next item need (read in the second-to-last K-tile of every item)
#define RELOAD_SCHED_ARGS()                                                                             \
  do {                                                                                                  \
    kargp_t kq = V4_KARGP(ka_unused);                                                                   \
    V4_OPAQUE_S(kq);                                                                                    \
    A = KARG(kq, const bf16_t*, A); B = KARG(kq, const bf16_t*, B);                                     \
    lda = KARG(kq, long, lda); ldb = KARG(kq, long, ldb);                                               \
    M = KARG(kq, int, M); N = KARG(kq, int, N); K = KARG(kq, int, K); k_per_split = KARG(kq, int, k_per_split); \
    ntx = KARG(kq, int, ntx); nty = KARG(kq, int, nty); nitems = KARG(kq, int, nitems);                 \
    n_full = KARG(kq, int, n_full); S = KARG(kq, int, S); sched = KARG(kq, int*, sched);                \
    n_fullitems = S ? n_full : nitems;                                                                  \
  } while (0)

  // One K-tile in ring buffer `buf_`: two phases.  Phase A: fragments of A0, B0, B1 (16 reads), 16 MFMAs (quadrants A0B0, A0B1);
  // phase B: fragments of A1 (8 reads), 16 MFMAs (A1B1, A1B0).  Every load segment ends with lgkmcnt(0) BEFORE its barrier, so a
  // half-tile can be refilled in the very next phase: phase B requests A0, B0 of the K-tile after next (read in phase A), phase A
  // of the next K-tile requests its B1, A1.  Waits: phase A for A1 of this K-tile = all but the 4 youngest half-tiles; phase B
  // for A0, B0, B1 of the next K-tile = all but the 3 youngest.
  // KIND 0: a K-tile with at least two more of the same item behind it.  KIND 1: the second-to-last K-tile -- its phase A decodes
  // the next item, its phase B points the sources at that item's K-tile 0 and requests A0, B0 of it.  KIND 2: the last K-tile --
  // phase A requests B1, A1 of the next item's K-tile 0, phase B requests nothing.
#define KTILE(KIND, bufx_, pk_, tr_)                                                                                                   \
  do {                                                                                                                           \
    const int buf_ = V4_RFL(bufx_);     /* wave-uniform by construction; said explicitly (it ends up in M0 of the LDS-DMA) */   \
    const int nbuf_ = buf_ ^ 1;                                                                                                  \
    const unsigned char* sA0 = smem + (buf_) * STAGE_BYTES;                                                                      \
    const unsigned char* sA1 = sA0 + HALF_BYTES;                                                                                 \
    const unsigned char* sB0 = sA0 + 2 * HALF_BYTES;                                                                             \
    const unsigned char* sB1 = sA0 + 3 * HALF_BYTES;                                                                             \
    bf16x8_t fa0[2][4], fa1[2][4], fb0[4], fb1[4];                                                                               \
    if (!do_frag) {                                                                                                              \
      _Pragma("unroll") for (int s = 0; s < 4; ++s) {                                                                            \
        fb0[s] = bf16x8_t{}; fb1[s] = bf16x8_t{}; fa0[0][s] = bf16x8_t{}; fa0[1][s] = bf16x8_t{}; fa1[0][s] = bf16x8_t{}; fa1[1][s] = bf16x8_t{}; \
        V4_OPAQUE_V(fb0[s]); V4_OPAQUE_V(fb1[s]); V4_OPAQUE_V(fa0[0][s]); V4_OPAQUE_V(fa0[1][s]); V4_OPAQUE_V(fa1[0][s]); V4_OPAQUE_V(fa1[1][s]); \
      }                                                                                                                          \
    }                                                                                                                            \
    const bool cs_now = want_cs && cs_ctr == txcur;                                                                              \
    cs_ctr = cs_ctr + 1 == ntx_cs ? 0 : cs_ctr + 1;                                                                              \
    const uint32_t boff = (buf_) * STAGE_BYTES;                                                                                  \
    if (do_frag) {                                                                                                               \
      frag4<TB, 2>(fb0, bkm + boff, sB0, br, lm);                                                                                \
      frag4<TB, 3>(fb1, bkm + boff, sB1, br, lm);                                                                                \
      frag4<TA, 0>(fa0[0], akm0 + boff, sA0, ar, lm);                                                                            \
      frag4<TA, 0>(fa0[1], akm1 + boff, sA0, ar + 32, lm);                                                                       \
    }                                                                                                                            \
    const bool more_ = (KIND) != 2 || have;   /* a K-tile follows (KIND 2: the next item's K-tile 0) */                           \
    if ((KIND) == 2 && pring) {         /* the slots of the next item's B1.1 / A1.0 carry pieces of this item's epilogue operand (V4_PRE_RING) */ \
      ISSUE_PRE(3, 1, nbuf_); ISSUE_PRE(1, 0, nbuf_); wait_vmcnt_7_plus(pk_);                                                    \
    } else if (more_) {                 /* B1 (V4_SCHED 0: and A1) of that K-tile -> the other buffer; then: A1 of THIS K-tile has landed */ \
      if (V4_SCHED == 2) { ISSUE_PIECE(bp, BDH_(1), 3, nbuf_, 1); ISSUE_PIECE(ap, a_dh, 1, nbuf_, 0); wait_vmcnt_7_plus(pk_); }     \
      else if (V4_SCHED) { ISSUE_H(bp, BDHQ, 3, nbuf_); wait_vmcnt_plus<6>(pk_); }                                                \
      else { ISSUE_H(bp, BDHQ, 3, nbuf_); ISSUE_H(ap, a_dh, 1, nbuf_); wait_vmcnt_plus<8>(pk_); }                                 \
    } else { WAITV(0); }                                                                                                         \
    if ((KIND) == 1) {                  /* the next item: which one, and its tile coordinates (scalar work in a load segment that requests little) */ \
      if (fast && dyn) {                /* position from the XCD's counter (published by thread 0 at the end of the previous epilogue) */ \
        const int packed_ = V4_RFL(sched_lds[0]);                                                                                \
        nxt_packed = packed_;                                                                                                    \
        have = packed_ >= 0;                                                                                                     \
        if (have) {                                                                                                              \
          const int lg_ = sf_f + (packed_ & 0xffffff);                                                                           \
          const int ty_ = fdiv(lg_, ntx_f), tx_ = lg_ - ty_ * ntx_f;                                                             \
          d_m = (ty_ - nxt.tile_y) * BM; d_n = (tx_ - nxt.tile_x) * BN;                                                          \
          nxt.tile_x = tx_; nxt.tile_y = ty_;                                                                                    \
        }                                                                                                                        \
      } else if (fast) {                                                                                                         \
        left -= 1;                                                                                                               \
        have = left > 0;                                                                                                         \
        if (have) {                                                                                                              \
          int tx_ = nxt.tile_x + sx_f;                                                                                           \
          const int cy_ = tx_ >= ntx_f ? 1 : 0;                                                                                  \
          tx_ -= cy_ ? ntx_f : 0;                                                                                                \
          const int ty_ = nxt.tile_y + sy_f + cy_;                                                                               \
          d_m = (ty_ - nxt.tile_y) * BM; d_n = (tx_ - nxt.tile_x) * BN;                                                          \
          nxt.tile_x = tx_; nxt.tile_y = ty_;                                                                                    \
        }                                                                                                                        \
      } else {                                                                                                                   \
        RELOAD_SCHED_ARGS();                                                                                                     \
        const int packed_ = V4_RFL(sched_lds[0]);   /* published before the previous epilogue's last barrier */                  \
        nxt_packed = packed_;                                                                                                    \
        have = packed_ >= 0 && DECODE(packed_ >> 24, packed_ & 0xffffff, nxt);                                                   \
      }                                                                                                                          \
    }                                                                                                                            \
    wait_lgkm0();                                                                                                                \
    if (tr_) TRACE_FINE();      /* load segment A done (before its barrier) */                                           \
    V4_BARRIER();                                                                                                                \
    if (tr_) TRACE_FINE();      /* barrier passed */                                                                     \
    __builtin_amdgcn_s_setprio(1);                                                                                               \
    _Pragma("unroll") for (int s = 0; s < 4; ++s) {                                                                              \
      _Pragma("unroll") for (int ii = 0; ii < 2; ++ii) { MFMA(acc[ii][0], fa0[ii][s], fb0[s]); MFMA(acc[ii][1], fa0[ii][s], fb1[s]); } \
      if (V4_SCHED == 1 && more_ && (s == 0 || s == 2)) {   /* A1 of the next K-tile: one piece behind the 4th, one behind the 12th MFMA */ \
        __builtin_amdgcn_sched_barrier(0);                                                                                       \
        if (s == 0) ISSUE_PIECE(ap, a_dh, 1, nbuf_, 0); else ISSUE_PIECE(ap, a_dh, 1, nbuf_, 1);                                 \
        __builtin_amdgcn_sched_barrier(0);                                                                                       \
      }                                                                                                                          \
      if (V4_SCHED == 2 && ((KIND) == 2 && pring) && s == 1) {                                                                   \
        __builtin_amdgcn_sched_barrier(0);                                                                                       \
        ISSUE_PRE(1, 1, nbuf_);                                                                                                  \
        __builtin_amdgcn_sched_barrier(0);                                                                                       \
      } else if (V4_SCHED == 2 && more_ && s == 1) {   /* A1.1 of the next K-tile behind the 8th MFMA */                         \
        __builtin_amdgcn_sched_barrier(0);                                                                                       \
        ISSUE_PIECE(ap, a_dh, 1, nbuf_, 1);                                                                                      \
        __builtin_amdgcn_sched_barrier(0);                                                                                       \
      }                                                                                                                          \
    }                                                                                                                            \
    if (TA && cs_now) {                                                                                                          \
      /* column sums of A rows (bias gradient): wave wn takes k16-step wn; indicator fragment = ones in column c */              \
      _Pragma("unroll") for (int ii = 0; ii < 2; ++ii) {                                                                         \
        const uint32_t one = ((lane & 31) == ii) ? 0x3f803f80u : 0u;                                                             \
        union { uint32_t u[4]; bf16x8_t v; } ind;                                                                                \
        ind.u[0] = ind.u[1] = ind.u[2] = ind.u[3] = one;                                                                         \
        const bf16x8_t af = wn == 0 ? fa0[ii][0] : wn == 1 ? fa0[ii][1] : wn == 2 ? fa0[ii][2] : fa0[ii][3];                     \
        cs = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, ind.v, cs, 0, 0, 0);                                                    \
      }                                                                                                                          \
    }                                                                                                                            \
    __builtin_amdgcn_s_setprio(0);                                                                                               \
    V4_BARRIER();                                                                                                                \
    if (tr_) TRACE_FINE();      /* MFMA segment A done, barrier passed */                                                \
                                                                                                                                 \
    if (do_frag) {                                                                                                               \
      frag4<TA, 1>(fa1[0], akm0 + boff, sA1, ar, lm);                                                                            \
      frag4<TA, 1>(fa1[1], akm1 + boff, sA1, ar + 32, lm);                                                                       \
    }                                                                                                                            \
    if ((KIND) == 0) {                  /* A0 (V4_SCHED 0: and B0) of the K-tile after next -> this buffer (read in phase A); then: A0, B0, B1 of the next K-tile have landed */ \
      if (V4_SCHED == 2) { ADVANCE_SRC(); ISSUE_H(ap, 0, 0, (buf_)); ISSUE_H(bp, 0, 2, (buf_)); WAITV(6); }                      \
      else if (V4_SCHED) { ADVANCE_SRC(); ISSUE_H(ap, 0, 0, (buf_)); WAITV(4); }                                                 \
      else { ISSUE_H(ap, astep, 0, (buf_)); ISSUE_H(bp, bstep, 2, (buf_)); WAITV(6); ADVANCE_SRC(); }                            \
    } else if ((KIND) == 1) {                                                                                                    \
      if (have) {                       /* sources -> K-tile 0 of the next item; its A0, B0 take the slots a K-tile nk would take */ \
        if (fast) {                     /* the sources point at K-tile nk - 1 of this item: step back to k = 0, over to the next tile */ \
          const long back_ = (long)(nk - 1) * BK;                                                                                \
          const long dA_ = TA ? ((long)d_m - back_ * lda) : ((long)d_m * lda - back_);                                           \
          const long dB_ = TB ? ((long)d_n - back_ * ldb) : ((long)d_n * ldb - back_);                                           \
          _Pragma("unroll") for (int q = 0; q < NP; ++q) { ap[q] += dA_; bp[q] += dB_; }                                         \
        } else {                                                                                                                 \
          LOAD_PROBLEM(kp, nxt.p);                                                                                               \
          SETUP_SRC(nxt.tile_y * BM, nxt.tile_x * BN, nxt.kbeg, lm);                                                               \
        }                                                                                                                        \
        if (pring) { }                  /* (issued from the middle of the epilogue: these slots carry the epilogue operand) */    \
        else if (V4_SCHED == 2) { ISSUE_H(ap, 0, 0, (buf_)); ISSUE_H(bp, 0, 2, (buf_)); WAITV(6); }                              \
        else if (V4_SCHED) { ISSUE_H(ap, 0, 0, (buf_)); WAITV(4); }                                                              \
        else { ISSUE_H(ap, 0, 0, (buf_)); ISSUE_H(bp, 0, 2, (buf_)); WAITV(6); }                                                 \
      } else if (!pring) { WAITV(2); }  /* A0, B0, B1 of the last K-tile have landed */                                          \
      if (pring) {                      /* rows 0..127 of this item's epilogue operand -> the buffer the epilogue does not stage through */ \
        kargp_t kr_ = V4_KARGP(ka_unused);                                                                                       \
        V4_OPAQUE_S(kr_);                                                                                                        \
        const bf16_t* pb_ = reinterpret_cast<const bf16_t*>(MODE == MODE_STORE_DGELU ? (const void*)KARG(kr_, void*, e.aux) : KARG(kr_, const void*, e.residual)); \
        pre_ldv = MODE == MODE_STORE_DGELU ? KARG(kr_, long, e.ldaux) : KARG(kr_, long, e.ldr);                                  \
        pre_ptr = pb_ + (long)(m0 + wave * 4 + (lm >> 5)) * pre_ldv + n0 + (lm & 31) * 8;                                        \
        ISSUE_PRE(0, 0, (buf_)); ISSUE_PRE(0, 1, (buf_)); ISSUE_PRE(2, 0, (buf_)); ISSUE_PRE(2, 1, (buf_)); WAITV(6);            \
      }                                                                                                                          \
    }                                                                                                                            \
    wait_lgkm0();                                                                                                                \
    if (tr_) TRACE_FINE();      /* load segment B done */                                                                \
    V4_BARRIER();                                                                                                                \
    __builtin_amdgcn_s_setprio(1);                                                                                               \
    _Pragma("unroll") for (int s = 0; s < 4; ++s) {                                                                              \
      _Pragma("unroll") for (int ii = 0; ii < 2; ++ii) { MFMA(acc[2 + ii][1], fa1[ii][s], fb1[s]); MFMA(acc[2 + ii][0], fa1[ii][s], fb0[s]); } \
      if (V4_SCHED == 2 && ((KIND) == 1 && pring) && s == 1) {                                                                   \
        __builtin_amdgcn_sched_barrier(0);                                                                                       \
        ISSUE_PRE(3, 0, (buf_));                                                                                                 \
        __builtin_amdgcn_sched_barrier(0);                                                                                       \
      } else if (V4_SCHED == 2 && ((KIND) == 0 || ((KIND) == 1 && have)) && s == 1) {   /* B1.0 of the K-tile after next (KIND 1: of the next item's K-tile 0) behind the 8th MFMA */ \
        __builtin_amdgcn_sched_barrier(0);                                                                                       \
        ISSUE_PIECE(bp, BDH_(0), 3, (buf_), 0);                                                                                  \
        __builtin_amdgcn_sched_barrier(0);                                                                                       \
      }                                                                                                                          \
      if (V4_SCHED == 1 && ((KIND) == 0 || ((KIND) == 1 && have)) && (s == 0 || s == 2)) {   /* B0 of the K-tile after next (KIND 1: of the next item's K-tile 0) */ \
        __builtin_amdgcn_sched_barrier(0);                                                                                       \
        if (s == 0) ISSUE_PIECE(bp, 0, 2, (buf_), 0); else ISSUE_PIECE(bp, 0, 2, (buf_), 1);                                     \
        __builtin_amdgcn_sched_barrier(0);                                                                                       \
      }                                                                                                                          \
    }                                                                                                                            \
    if (TA && cs_now) {                                                                                                          \
      _Pragma("unroll") for (int ii = 0; ii < 2; ++ii) {                                                                         \
        const uint32_t one = ((lane & 31) == 2 + ii) ? 0x3f803f80u : 0u;                                                         \
        union { uint32_t u[4]; bf16x8_t v; } ind;                                                                                \
        ind.u[0] = ind.u[1] = ind.u[2] = ind.u[3] = one;                                                                         \
        const bf16x8_t af = wn == 0 ? fa1[ii][0] : wn == 1 ? fa1[ii][1] : wn == 2 ? fa1[ii][2] : fa1[ii][3];                     \
        cs = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, ind.v, cs, 0, 0, 0);                                                    \
      }                                                                                                                          \
    }                                                                                                                            \
    __builtin_amdgcn_s_setprio(0);                                                                                               \
    V4_BARRIER();                                                                                                                \
  } while (0)

  while (have) {
    const int m0 = nxt.tile_y * BM, n0 = nxt.tile_x * BN;
    const int nk = nxt.nk;                                         // host guarantees nk >= 2
    const int zcur = nxt.z;
    const int txcur = nxt.tile_x;
    const int cur_slice = nxt.slice;
    const int pcur = nxt.p;
    const int ntx_cs = GROUP ? GPARG(kp, pcur, int, ntx) : ntx;
    int nxt_packed = -1;
    const bf16_t* pre_ptr = nullptr;           // V4_PRE_RING: this lane's source of the epilogue-operand pieces (set in K-tile nk - 2)
    long pre_ldv = 0;

    // fragment addressing restarts from an opaque lane id per tile, so its ~12 address registers are not kept live across
    // the epilogue of the previous tile (same trick as `te` below)
    int lm = lane;
    V4_OPAQUE_V(lm);
    const uint32_t akm0 = lds0 + kmajor_lane_off(ar, lm), akm1 = lds0 + kmajor_lane_off(ar + 32, lm), bkm = lds0 + kmajor_lane_off(br, lm);
    f32x16_t acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    f32x16_t cs;                     // bias-gradient accumulator: column c = row-tile c of this wave (TA only)
#pragma unroll
    for (int r = 0; r < 16; ++r) cs[r] = 0.f;
    const bool want_cs = TA && (GROUP ? GPARG(kp, pcur, float*, a_colsum) != nullptr : e.a_colsum != nullptr);
    int cs_ctr = 0;                  // K-tiles with cs_ctr == tile_x contribute (spreads the extra MFMAs over the tile columns)

    // ---- K-tile 0 of this item is in LDS (requested by the previous item, or before the loop); A0, B0 of K-tile 1 go to the
    // other buffer: it was the previous epilogue's staging area, and every wave is past that epilogue's last barrier
    ADVANCE_SRC();
    { const int b1_ = V4_RFL(tp ^ 1); ISSUE_H(ap, 0, 0, b1_); ISSUE_H(bp, 0, 2, b1_); if (V4_SCHED == 2) ISSUE_PIECE(bp, BDH_(0), 3, b1_, 0); }   // (V4_SCHED 2: + the piece B1.0 that MFMA segment B of a K-tile "-1" would have requested)
    TRACE();                                 // [0] tile started
    if (wm == 1) V4_BARRIER();               // waves 4-7 run one barrier behind waves 0-3 from here on

    for (int kt = 0; kt < nk - 2; ++kt) KTILE(0, (kt ^ tp) & 1, (kt == 0 ? pend : 0), (V4_TRACE >= 3 && kt == 2));
    TRACE();                                 // [1] K-tiles 0 .. nk-3 done
    KTILE(1, (nk ^ tp) & 1, (nk == 2 ? pend : 0), (V4_TRACE == 2));   // K-tile nk - 2
    TRACE();                                 // [2] K-tile nk-2 done (next item decoded, its A0 / B0 requested)
    KTILE(2, (nk ^ tp ^ 1) & 1, 0, (V4_TRACE == 2));                    // K-tile nk - 1
    if (wm == 0) V4_BARRIER();               // re-align the two wave groups; the buffer of the last K-tile is free from here
    TRACE();                                 // [3] main loop done
    const int sbuf = V4_RFL((nk ^ tp ^ 1) & 1);      // ring buffer of the last K-tile = the epilogue's staging area (the other one holds / receives the next item's K-tile 0)
    tp = sbuf ^ 1;

    // ---- the item after next (thread 0): requested now, published at the end of this epilogue
    RELOAD_ARGS();
    int fut = -1;
    if (have && t == 0 && !fast && !dyn) { const int x = nxt_packed >> 24, l = (nxt_packed & 0xffffff) + wgs_on_xcd(x, grid); fut = (x << 24) | l; }
    TRACE();                                 // [4] epilogue arguments loaded
    pend = 0;
    do {   // the epilogue flavours leave with `break`
    // the epilogue's per-lane indexing starts from an OPAQUE copy of the thread id: otherwise the compiler hoists ~30 loop-
    // invariant address registers out of the persistent tile loop and keeps them live across the main loop (spills)
    int te = t;
    V4_OPAQUE_V(te);
    const int le = te & 63;
    if (!do_epi) {
      if (acc[0][0][0] == 123.456f) reinterpret_cast<float*>(e.C)[0] = acc[1][1][3] + acc[2][0][5] + acc[3][1][7] + cs[2];
      V4_BARRIER();
      break;
    }

    if (TA && want_cs) {
      // cs: lane -> column c = lane & 31 (row-tile c = i*2+ii of this wave), register r -> row (r&3) + 8*(r>>2) + 4*(lane>>5).
      // The four wn-waves of a row hold partial sums over different k16-steps: reduce them through LDS (the bias area, unused
      // here), then one value per row leaves the workgroup -- to its workspace slot (summed by the reduce pass: no atomics),
      // or as ONE atomic per row (MODE_ATOMIC).
      float* csl = reinterpret_cast<float*>(smem + BIAS_OFF);       // [4 wn][256 rows]
      const int c = le & 31;
      if (c < 4) {
        const int rb = (c >> 1) * 128 + ar + (c & 1) * 32 + 4 * (le >> 5);
#pragma unroll
        for (int r = 0; r < 16; ++r) csl[wn * 256 + rb + (r & 3) + 8 * (r >> 2)] = cs[r];
      }
      wait_lgkm0();
      V4_BARRIER();
      if (te < 256) {
        const float v = (csl[te] + csl[256 + te]) + (csl[512 + te] + csl[768 + te]);
        if (GROUP) e.ws_cs[(long)zcur * KARG(kp, long, cs_zs) + GPARG(kp, pcur, long, cs_off) + (long)txcur * GPARG(kp, pcur, int, M) + m0 + te] = v;
        else if (MODE == MODE_PARTIAL) e.ws_cs[((long)zcur * ntx + txcur) * M + m0 + te] = v;
        else atomicAdd(e.a_colsum + m0 + te, v);
      }
      wait_lgkm0();
      V4_BARRIER();
    }

    if (MODE == MODE_ATOMIC) {
      // fp32 atomics straight from the accumulators (plain orientation: lanes 0..31 = 32 consecutive floats of a row)
      float* Cf = reinterpret_cast<float*>(e.C);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int ii = 0; ii < 2; ++ii)
#pragma unroll
          for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const int m = m0 + i * 128 + ar + ii * 32 + (r & 3) + 8 * (r >> 2) + 4 * (le >> 5);
              const int n = n0 + j * 128 + br + (le & 31);
              atomicAdd(Cf + (long)m * e.ldc + n, acc[i * 2 + ii][j][r] * e.alpha);
            }
      wait_vmcnt<0>();                       // 128 atomics per lane cannot ride along: drain (also lands the next tile's loads)
      V4_BARRIER();
      break;
    }

    if (MODE == MODE_MAXSIM) {
      // FILIP token-wise max-sim fused into the epilogue (filip.py:96-105): the tile holds S[(i,j), (l,m)] for 256 token rows and
      // 16 captions x 16 selected tokens; what leaves the chip is max_m per (row, l) reduced to mean_j per (sample, l) -- one fp32
      // atomic per (sample-in-tile, l) into raw[b][B] -- and the arg-max m as one byte per (row, l).  S itself (fp32 [b*J, B*16]:
      // 1.6 + 2.6 GB per step at the FILIP batch) never exists.
      // acc[i*2+ii][j]: lane -> row (lane&31), registers 4*rg .. +3 -> columns 8*rg + 4*(lane>>5) + {0..3} of a 32-column group,
      // i.e. l-group g = rg>>1 of the fragment, m = 8*(rg&1) + 4*(lane>>5) + x.
      float* mv = reinterpret_cast<float*>(smem + sbuf * STAGE_BYTES);             // [256 rows][16 l] max values
      unsigned char* ma = smem + sbuf * STAGE_BYTES + 256 * 16 * 4;                  // [256 rows][16 l] arg-max
      const int hh = le >> 5;
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int ii = 0; ii < 2; ++ii)
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            float bv[2]; int bm[2];
#pragma unroll
            for (int g = 0; g < 2; ++g) {
              float v = acc[i * 2 + ii][j][8 * g]; int m = 4 * hh;
#pragma unroll
              for (int q = 1; q < 8; ++q) {
                const float c = acc[i * 2 + ii][j][8 * g + q];
                const int mq = 8 * (q >> 2) + 4 * hh + (q & 3);
                if (c > v) { v = c; m = mq; }
              }
              const float v2 = __shfl_xor(v, 32, 64);
              const int m2 = __shfl_xor(m, 32, 64);
              if (v2 > v || (v2 == v && m2 < m)) { v = v2; m = m2; }           // first maximum wins (torch.max)
              bv[g] = v; bm[g] = m;
            }
            // half-wave h publishes l-group h of the fragment
            const int row = i * 128 + ar + ii * 32 + (le & 31);
            const int lt = j * 8 + wn * 2 + hh;
            mv[row * 16 + lt] = hh ? bv[1] : bv[0];
            ma[row * 16 + lt] = (unsigned char)(hh ? bm[1] : bm[0]);
          }
      wait_lgkm0();
      V4_BARRIER();
      if (te < 256) {
        const int J = e.ms_J, Bc = e.ms_B;
        // arg-max bytes: one 16-byte store per row
        const uint4 a16 = *reinterpret_cast<const uint4*>(ma + te * 16);
        *reinterpret_cast<uint4*>(reinterpret_cast<unsigned char*>(e.aux) + (long)(m0 + te) * Bc + (n0 >> 4)) = a16;
        // mean over the rows of each sample present in this tile: thread (l = te & 15, segment = te >> 4)
        const int l = te & 15, sg = te >> 4;
        const int smp = m0 / J + sg;                                             // sample of this segment
        int r0 = smp * J - m0, r1 = r0 + J;
        r0 = r0 < 0 ? 0 : r0;
        r1 = r1 > 256 ? 256 : r1;
        if (smp < e.ms_b && r0 < r1) {
          float sum = 0.f;
          for (int r = r0; r < r1; ++r) sum += mv[r * 16 + l];
          atomicAdd(reinterpret_cast<float*>(e.C) + (long)smp * Bc + (n0 >> 4) + l, sum / (float)J);
        }
      }
      wait_lgkm0();
      V4_BARRIER();                          // staging area free again (the next tile's K-tile 1 lands there)
      pend = 0;                              // few, unevenly spread stores: the next prologue waits for them all (conservative count)
      break;
    }

    if (MODE == MODE_CE_FWD || MODE == MODE_CE_BWD) {
      // Cross-entropy over the vocabulary on the logits tile while it is still in registers (masked-LM head: rows = masked
      // positions, columns = 49409 vocabulary entries).  Forward: per-row (max, sum exp) of the tile's 256 columns + the label
      // logit leave the chip (8 bytes per row and tile instead of 1 KB of fp32 logits); a finalize kernel merges the 194 tiles
      // of a row.  Backward: the tile is recomputed and dl = g (softmax - onehot) is stored as bf16 for the two gradient GEMMs.
      unsigned char* const ce_base = smem + sbuf * STAGE_BYTES;                 // the staging buffer (ring buffer of the last K-tile)
      float* cb = reinterpret_cast<float*>(ce_base);                            // [256] bias of the tile's columns
      int* lb = reinterpret_cast<int*>(ce_base + 1024);                         // [256] label of the tile's rows (-1: padding row)
      float* rl = reinterpret_cast<float*>(ce_base + 2048);                     // BWD: [256] lse, FWD: unused
      float* rg = reinterpret_cast<float*>(ce_base + 3072);                     // BWD: [256] upstream gradient
      float* red = reinterpret_cast<float*>(ce_base + 4096);                    // FWD: [256 rows][4 wn][2]
      const int V = e.ce_V, nrows = e.ce_n;
      if (te < 256) {
        const int c = n0 + te, r = m0 + te;
        cb[te] = (c < V && e.bias) ? e.bias[c] : 0.f;
        lb[te] = r < nrows ? (int)e.ce_labels[r] : -1;
        if (MODE == MODE_CE_BWD) { rl[te] = r < nrows ? e.ce_lse[r] : 0.f; rg[te] = r < nrows ? e.ce_g[r] : 0.f; }
      }
      __syncthreads();
      const int hh = le >> 5;
      if (MODE == MODE_CE_FWD) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int ii = 0; ii < 2; ++ii) {
            const int row = i * 128 + ar + ii * 32 + (le & 31);
            const int lab = lb[row] - n0;                   // label column inside this tile (or out of [0, 256))
            float mx = -INFINITY;
            float vals[32];
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
              for (int q = 0; q < 16; ++q) {
                const int cl = j * 128 + br + 8 * (q >> 2) + 4 * hh + (q & 3);
                float v = acc[i * 2 + ii][j][q] + cb[cl];
                if (cl == lab) e.ce_lab[m0 + row] = v;
                v = (n0 + cl < V) ? v : -INFINITY;
                vals[j * 16 + q] = v;
                mx = fmaxf(mx, v);
              }
            float sm = 0.f;
#pragma unroll
            for (int q = 0; q < 32; ++q) sm += (mx == -INFINITY) ? 0.f : __expf(vals[q] - mx);
            // the other half-wave holds the other 32 columns of this wave's 64
            const float m2 = __shfl_xor(mx, 32, 64), s2 = __shfl_xor(sm, 32, 64);
            const float mm = fmaxf(mx, m2);
            const float ss = (mm == -INFINITY) ? 0.f : sm * __expf(mx - mm) + s2 * __expf(m2 - mm);
            if (hh == 0) { red[(row * 4 + wn) * 2] = mm; red[(row * 4 + wn) * 2 + 1] = ss; }
          }
        __syncthreads();
        if (te < 256) {
          float mm = -INFINITY;
#pragma unroll
          for (int w = 0; w < 4; ++w) mm = fmaxf(mm, red[(te * 4 + w) * 2]);
          float ss = 0.f;
#pragma unroll
          for (int w = 0; w < 4; ++w) { const float mw = red[(te * 4 + w) * 2]; ss += (mw == -INFINITY) ? 0.f : red[(te * 4 + w) * 2 + 1] * __expf(mw - mm); }
          float2 o2; o2.x = mm; o2.y = ss;
          *reinterpret_cast<float2*>(e.ce_part + ((long)(m0 + te) * ntx_cs + txcur) * 2) = o2;
        }
        __syncthreads();
      } else {
        // dl tile, bf16, staged through LDS like the plain bf16 epilogue (two passes of 128 rows, 512-byte staging rows,
        // 8-byte unit u of row r at unit u ^ (r & 15)); columns >= ce_ldd (beyond the dl buffer's row) are not stored
        unsigned char* Cd = ce_base + 8192;
        unsigned char* Db = reinterpret_cast<unsigned char*>(e.C) + ((long)m0 * e.ce_ldd + n0) * 2;
        const int cc = te & 31, r0 = te >> 5;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
#pragma unroll
          for (int ii = 0; ii < 2; ++ii) {
            const int ml = ar + ii * 32 + (le & 31);
            const int row = i * 128 + ml;
            const float lse_r = rl[row], g_r = rg[row];
            const int lab = lb[row] - n0;
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
              for (int rgp = 0; rgp < 4; ++rgp) {
                const int nl = j * 128 + br + 8 * rgp + 4 * hh;
                float d[4];
#pragma unroll
                for (int x = 0; x < 4; ++x) {
                  const int cl = nl + x;
                  const float v = acc[i * 2 + ii][j][rgp * 4 + x] + cb[cl];
                  const float pr = (n0 + cl < V) ? __expf(v - lse_r) : 0.f;
                  d[x] = g_r * (pr - (cl == lab ? 1.f : 0.f));
                }
                uint2 pk;
                pk.x = pack2bf_hw(d[0], d[1]); pk.y = pack2bf_hw(d[2], d[3]);
                *reinterpret_cast<uint2*>(Cd + ml * 512 + (((nl >> 2) ^ (ml & 15)) << 3)) = pk;
              }
          }
          wait_lgkm0();
          V4_BARRIER();
#pragma unroll
          for (int it = 0; it < 8; ++it) {
            const int row = r0 + 16 * it;
            const int pc = cc ^ ((row & 15) >> 1);
            uint4 raw = *reinterpret_cast<const uint4*>(Cd + row * 512 + pc * 16);
            if (row & 1) { uint32_t tx = raw.x, ty = raw.y; raw.x = raw.z; raw.y = raw.w; raw.z = tx; raw.w = ty; }
            if (n0 + cc * 8 < e.ce_ldd)
              *reinterpret_cast<uint4*>(Db + ((long)(i * 128 + row) * e.ce_ldd + cc * 8) * 2) = raw;
          }
          wait_lgkm0();
          V4_BARRIER();
        }
      }
      pend = 0;
      break;
    }

    unsigned char* Cs = smem + sbuf * STAGE_BYTES;  // the ring buffer of the last K-tile (the other one holds the next item's K-tile 0)
    constexpr bool SLICEABLE = MODE == MODE_STORE || MODE == MODE_STORE_RES;   // the N = d GEMMs (few tiles) use these flavours
    if (MODE == MODE_PARTIAL || GROUP || (SLICEABLE && cur_slice >= 0)) {
      // fp32 partial tile: MODE_PARTIAL -> ws[z][m][n]; tail slice -> its private [256][256] tile of the workspace.
      // 4 passes of 64 rows: pass (i, ii) holds rows i*128 + wm*64 + ii*32 + 0..31 of both wave groups; staging row =
      // wm*32 + (lane&31), 1024 B per row, 16-byte unit u of row r stored at unit u ^ (r & 7).
      float* Wp;
      long wld;
      if (GROUP) {   // this item's problem (M, N above already belong to the NEXT item)
        const int Nc = GPARG(kp, pcur, int, N);
        Wp = e.ws + (long)zcur * KARG(kp, long, zs) + GPARG(kp, pcur, long, ws_off) + (long)m0 * Nc + n0; wld = Nc;
      }
      else if (MODE == MODE_PARTIAL) { Wp = e.ws + (long)zcur * M * N + (long)m0 * N + n0; wld = N; }
      else { Wp = e.ws + (long)cur_slice * (BM * BN); wld = BN; }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int ii = 0; ii < 2; ++ii) {
          const int rl = wm * 32 + (le & 31);
#pragma unroll
          for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) {
              const int nl = j * 128 + br + 8 * rg + 4 * (le >> 5);
              f32x4_t v;
#pragma unroll
              for (int x = 0; x < 4; ++x) v[x] = acc[i * 2 + ii][j][rg * 4 + x] * e.alpha;
              *reinterpret_cast<f32x4_t*>(Cs + rl * 1024 + (((nl >> 2) ^ (rl & 7)) << 4)) = v;
            }
          wait_lgkm0();
          V4_BARRIER();
          {
            const int c = te & 63;            // 16-byte unit of the row (4 columns)
#pragma unroll
            for (int half = 0; half < 2; ++half) {
            f32x4_t rb[4];                     // four LDS reads first: read -> wait -> store per row was a dependent round trip each
#pragma unroll
            for (int it4 = 0; it4 < 4; ++it4) {
              const int rl2 = wave + 8 * (half * 4 + it4);
              rb[it4] = *reinterpret_cast<const f32x4_t*>(Cs + rl2 * 1024 + ((c ^ (rl2 & 7)) << 4));
            }
            wait_lgkm0();
#pragma unroll
            for (int it4 = 0; it4 < 4; ++it4) V4_OPAQUE_V(rb[it4]);
#pragma unroll
            for (int it4 = 0; it4 < 4; ++it4) {
              const int it = half * 4 + it4;
              const int rl2 = wave + 8 * it;                                       // staging row 0..63 (wave-uniform)
              const long mr = i * 128 + (rl2 >> 5) * 64 + ii * 32 + (rl2 & 31);    // row inside the tile
              const f32x4_t v = rb[it4];
#if V4_NT_STORE >= 4
              __builtin_nontemporal_store(v, reinterpret_cast<f32x4_t*>(reinterpret_cast<unsigned char*>(Wp + mr * wld) + (uint32_t)(c * 16)));
#else
              *reinterpret_cast<f32x4_t*>(reinterpret_cast<unsigned char*>(Wp + mr * wld) + (uint32_t)(c * 16)) = v;
#endif
            }
            }
          }
          wait_lgkm0();
          V4_BARRIER();
          TRACE();                           // [3..6] partial pass stored
        }
      pend = 32;
      break;
    }

    // ---- MODE_STORE: bf16 epilogue.  acc[i*2+ii][j] holds a C^T fragment: lane -> row (lane&31), registers 4*rg .. 4*rg+3
    // -> 4 consecutive columns 8*rg + 4*(lane>>5) + {0..3}.  Two passes (A-half i = rows i*128 ..): staging tile
    // [128 rows][512 B] in ring buffer 1, 8-byte unit u of row r at unit u ^ (r & 15)  (conflict-free ds_write_b64;
    // ds_read_b128 sees whole 16-byte chunks, halves swapped on odd rows).
    {
      // Four quarter passes q = (i, ii): rows i*128 + wm*64 + ii*32 + 0..31 of both wave groups -> staging rows wm*32 + (lane&31)
      // of buffer q & 1 (two [64 rows][512 B] halves of ring buffer 1).  Phase q converts and stages quarter q (VALU + LDS
      // writes) right after ISSUING the read-back + global stores of quarter q-1 (LDS reads + the vector-memory path): the two
      // waves of a SIMD overlap one's store issue with the other's conversions instead of meeting at a barrier between a
      // VALU-only and a store-only phase.  In-kernel trace (profiles/r02_gemm_trace_epilogue.txt): the two-pass epilogue costs
      // 2 x (2.3 k stage + 1.7 k store) cycles per tile whether 30 or 256 CUs store at the same time -- issue-bound, not HBM-bound.
      const int cc = te & 31;                 // 16-byte chunk of the row (8 columns)
      const int r0 = te >> 5;                 // 0..15
      unsigned char* Cb = reinterpret_cast<unsigned char*>(e.C) + ((long)m0 * e.ldc + n0) * 2;
      const uint32_t c_off = ((uint32_t)r0 * (uint32_t)e.ldc + cc * 8) * 2;
      unsigned char* Xb = reinterpret_cast<unsigned char*>(e.aux) + ((long)m0 * e.ldaux + n0) * 2;
      const uint32_t x_off = ((uint32_t)r0 * (uint32_t)e.ldaux + cc * 8) * 2;
      const unsigned char* Rb = reinterpret_cast<const unsigned char*>(e.residual) + ((long)m0 * e.ldr + n0) * 2;
      const uint32_t r_off = ((uint32_t)r0 * (uint32_t)e.ldr + cc * 8) * 2;
      constexpr bool is_gelu = MODE == MODE_STORE_GELU, is_dgelu = MODE == MODE_STORE_DGELU;
      constexpr bool NT_OUT = V4_NT_STORE == 1 || V4_NT_STORE >= 3 || (V4_NT_STORE == 2 && (is_gelu || is_dgelu));
      constexpr bool NT_IN = V4_NT_STORE >= 3;
      constexpr bool has_pre = MODE == MODE_STORE_DGELU || MODE == MODE_STORE_RES;
      const unsigned char* pre_base = is_dgelu ? Xb : Rb;
      const uint32_t pre_off = is_dgelu ? x_off : r_off;
      const long pre_ld = is_dgelu ? e.ldaux : e.ldr;
      // fused-epilogue operands of rows i*128 + 16*x + r0 (x = 0..7): half 0 requested before any store of the tile, half 1 when
      // the accumulators of half 0 are dead (after quarter 1 is staged; those loads queue behind quarter 0's stores, which is
      // fine: they are consumed two phases later)
      // V4_PRE_RING (round 6): half 0 is in LDS already -- requested by LDS-DMA from inside the last two K-tiles into the ring buffer
      // this epilogue does not stage through (rows 0..127 row-major, 512 B each) -- and half 1 is requested HERE, before anything else
      // of the epilogue, so that it has three quarter phases to land.
      uint4 pre0[8], pre1[8];
      const unsigned char* const pre_lds = smem + (sbuf ^ 1) * STAGE_BYTES + (uint32_t)(r0 * 512 + cc * 16);
      if (has_pre && !PRE_RING) {
#pragma unroll
        for (int x = 0; x < 8; ++x) pre0[x] = load_once16<NT_IN>(pre_base + (long)(16 * x) * pre_ld * 2 + pre_off);
      }
      if (PRE_RING) {
#pragma unroll
        for (int x = 0; x < 8; ++x) pre1[x] = load_once16<NT_IN>(pre_base + (long)(128 + 16 * x) * pre_ld * 2 + pre_off);
      }
      const int mlq = wm * 32 + (le & 31);
      // the lane's 8 bias quads (columns j*128 + br + 8*rg + 4*(lane>>5) ..+3) are the same for all four quarters: ONE batch of
      // LDS reads.  (Read next to their use, every conversion group was a dependent LDS round trip -- the compiler cannot move a
      // read of the bias area across the staging writes -- and 16 of them per half were most of the epilogue's 9.5 k cycles.)
      float4 bq[2][4];
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int rg = 0; rg < 4; ++rg)
          bq[j][rg] = *reinterpret_cast<const float4*>(smem + BIAS_OFF + 4 * (n0 + j * 128 + br + 8 * rg + 4 * (le >> 5)));
      wait_lgkm0();
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) { V4_OPAQUE_V(bq[j][rg].x); V4_OPAQUE_V(bq[j][rg].y); V4_OPAQUE_V(bq[j][rg].z); V4_OPAQUE_V(bq[j][rg].w); }
#pragma unroll
      for (int q = 0; q <= 4; ++q) {
        // order inside a phase: (1) the four LDS reads of quarter q-1 are ISSUED, (2) quarter q is converted and staged while they
        // are in flight (the staging writes go to the other half of the buffer; LDS operations retire in order, so the wait the
        // compiler places before the first use of the read data leaves the eight younger writes outstanding), (3) the global stores
        // of quarter q-1.  With the reads waited for before the conversions, every phase exposed one LDS round trip.
        uint4 raws[4];
        if (q >= 1) {
          const unsigned char* Cq = Cs + ((q - 1) & 1) * (64 * 512);
#pragma unroll
          for (int it = 0; it < 4; ++it) {
            const int row = r0 + 16 * it;                                        // staging row 0..63
            const int pc = cc ^ ((row & 15) >> 1);
            raws[it] = *reinterpret_cast<const uint4*>(Cq + row * 512 + pc * 16);
          }
        }
        if (q <= 3) {                          // convert + stage quarter q
          unsigned char* Cq = Cs + (q & 1) * (64 * 512);
#pragma unroll
          for (int j = 0; j < 2; ++j) {
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) {
              const int nl = j * 128 + br + 8 * rg + 4 * (le >> 5);
              uint2 pk;
              pk.x = pack2bf_hw(acc[q][j][rg * 4 + 0] * e.alpha + bq[j][rg].x, acc[q][j][rg * 4 + 1] * e.alpha + bq[j][rg].y);
              pk.y = pack2bf_hw(acc[q][j][rg * 4 + 2] * e.alpha + bq[j][rg].z, acc[q][j][rg * 4 + 3] * e.alpha + bq[j][rg].w);
              *reinterpret_cast<uint2*>(Cq + mlq * 512 + (((nl >> 2) ^ (mlq & 15)) << 3)) = pk;
            }
          }
        }
        if (q >= 1) {                          // store quarter q - 1
          const int qs = q - 1, i = qs >> 1, ii = qs & 1;
#pragma unroll
          for (int it = 0; it < 4; ++it) {
            const int row = r0 + 16 * it;
            const long mu = i * 128 + (it >> 1) * 64 + ii * 32 + 16 * (it & 1);  // wave-uniform part of the tile row (+ r0 in c_off)
            const int xo = (it >> 1) * 4 + ii * 2 + (it & 1);                    // index of this row group in pre0 / pre1
            uint4 raw = raws[it];
            if (row & 1) { uint32_t tx = raw.x, ty = raw.y; raw.x = raw.z; raw.y = raw.w; raw.z = tx; raw.w = ty; }
            if (MODE == MODE_STORE) {
              store_c16<NT_OUT>(Cb + mu * e.ldc * 2 + c_off, raw);
            } else {
              const uint32_t wv[4] = {raw.x, raw.y, raw.z, raw.w};
              float v[8];
#pragma unroll
              for (int x = 0; x < 4; ++x) { v[2 * x] = __uint_as_float(wv[x] << 16); v[2 * x + 1] = __uint_as_float(wv[x] & 0xffff0000u); }
              if (is_gelu) {      // aux = QuickGELU' of the staged (bf16) pre-activation: the factor the dX GEMM of c_proj multiplies by
                float dv[8];
#pragma unroll
                for (int x = 0; x < 8; ++x) quick_gelu_both_f(v[x], v[x], dv[x]);
                uint4 dk;
                dk.x = pack2bf_hw(dv[0], dv[1]); dk.y = pack2bf_hw(dv[2], dv[3]); dk.z = pack2bf_hw(dv[4], dv[5]); dk.w = pack2bf_hw(dv[6], dv[7]);
                store_c16<NT_OUT>(Xb + mu * e.ldaux * 2 + x_off, dk);
              }
              if (has_pre) {
                const uint4 pr = i == 0 ? (PRE_RING ? *reinterpret_cast<const uint4*>(pre_lds + xo * (16 * 512)) : pre0[xo]) : pre1[xo];
                const uint32_t pw[4] = {pr.x, pr.y, pr.z, pr.w};
                float pf[8];
#pragma unroll
                for (int x = 0; x < 4; ++x) { pf[2 * x] = __uint_as_float(pw[x] << 16); pf[2 * x + 1] = __uint_as_float(pw[x] & 0xffff0000u); }
                if (is_dgelu) {
#pragma unroll
                  for (int x = 0; x < 8; ++x) v[x] *= pf[x];
                } else {
#pragma unroll
                  for (int x = 0; x < 8; ++x) v[x] += pf[x];
                }
              }
              uint4 pk;
              pk.x = pack2bf_hw(v[0], v[1]); pk.y = pack2bf_hw(v[2], v[3]); pk.z = pack2bf_hw(v[4], v[5]); pk.w = pack2bf_hw(v[6], v[7]);
              store_c16<NT_OUT>(Cb + mu * e.ldc * 2 + c_off, pk);
            }
            if (it & 1) __builtin_amdgcn_sched_barrier(0);      // keep the unrolled iterations from being interleaved (register pressure)
          }
        }
        if (q == 1 && has_pre && !PRE_RING) {
#pragma unroll
          for (int x = 0; x < 8; ++x) pre1[x] = load_once16<NT_IN>(pre_base + (long)(128 + 16 * x) * pre_ld * 2 + pre_off);
        }
        if (q == 0 && PRE_RING) wait_vmcnt<8>();   // this wave's pieces of half 0 have landed (everything but the 8 loads of half 1 just issued); the barrier below publishes them
        if (q <= 3) {                          // (q = 4: the end-of-item barrier below)
          wait_lgkm0();
          V4_BARRIER();                        // quarter q staged / buffer (q - 1) & 1 free again
          TRACE();
        }
        if (q == 2 && PRE_RING && have) {
          // every wave has read its last piece of half 0 (the reads of phase 2, waited for before the barrier): the buffer takes the next
          // item's K-tile 0 now -- the four half-tiles in steady-state FIFO order, as before the first item -- and has phases 3 and 4 to land
          const int nb_ = sbuf ^ 1;
          ISSUE_H(ap, 0, 0, nb_); ISSUE_H(bp, 0, 2, nb_); ISSUE_H(bp, BDHQ, 3, nb_); ISSUE_H(ap, a_dh, 1, nb_);
        }
      }
      // stores the next item's first counted wait has to look past / the end-of-item wait leaves in flight: all of this epilogue's, or
      // (V4_PRE_RING) the 8 of phases 3 and 4 -- the K-tile 0 requested behind phase 2 must have landed
      pend = is_gelu ? 32 : (PRE_RING && have) ? 8 : 16;
    }
      } while (0);
    // ---- end of the item.  The next item's K-tile 0 (requested during the last two K-tiles) has landed: everything this wave
    // requested before the epilogue's `pend` stores is older than they are (in-order vmcnt).  The scheduler word is published.
    // Both BEFORE the barrier that every wave passes on its way into the next item.
    wait_vmcnt_tail(pend);
    if (t == 0) {
      if (dyn && have) { int raw_; FETCH(raw_); fut = fetch_finish(raw_); }
      sched_lds[0] = fut;                        // read by every wave in the second-to-last K-tile of the NEXT item
    }
    wait_lgkm0();
    V4_BARRIER();
    TRACE();                                 // [n] epilogue done
  }
  if (dyn && t == 0) {
    // self-resetting scheduler state: the last workgroup to leave zeroes the counters for the next launch on this stream
    const int d = atomicAdd(sched + 8, 1);
    if (d == grid - 1) {
#pragma unroll
      for (int x = 0; x < 9; ++x) sched[x] = 0;
    }
  }
#if V4_TRACE
  if (tracing) trace_p[0] = trace_n;
#endif
#undef MFMA
#undef TRACE
#undef TRACE_FINE
#undef ISSUE_H
#undef SETUP_SRC
#undef RELOAD_ARGS
#undef FETCH
#undef ADVANCE_SRC
#undef ISSUE_PIECE
#undef WAITV
#undef DECODE
#undef LOAD_PROBLEM
#undef KTILE
#undef RELOAD_SCHED_ARGS
#undef BDHQ
#undef BDH_
}

// out[m][n] (+)= sum_z ws[z][m][n]   (the split-K partial tiles of MODE_PARTIAL)
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ ws, float* __restrict__ out, long ldo, int M,
                                                            int N, int nsplit, int accumulate, const float* __restrict__ ws_cs,
                                                            float* __restrict__ colsum, int ncs) {
  const long n4 = (long)M * N / 4;
  const long stride = (long)gridDim.x * blockDim.x;
  const long zs = (long)M * N;
  const int nq = N / 4;
  const long tid = (long)blockIdx.x * blockDim.x + threadIdx.x;
  for (long i = tid; i < n4; i += stride) {
    f32x4_t s0 = {0.f, 0.f, 0.f, 0.f}, s1 = s0, s2 = s0, s3 = s0;
    int zz = 0;
    for (; zz + 4 <= nsplit; zz += 4) {        // 4 independent loads in flight per thread
      s0 += *reinterpret_cast<const f32x4_t*>(ws + (zz + 0) * zs + i * 4);
      s1 += *reinterpret_cast<const f32x4_t*>(ws + (zz + 1) * zs + i * 4);
      s2 += *reinterpret_cast<const f32x4_t*>(ws + (zz + 2) * zs + i * 4);
      s3 += *reinterpret_cast<const f32x4_t*>(ws + (zz + 3) * zs + i * 4);
    }
    for (; zz < nsplit; ++zz) s0 += *reinterpret_cast<const f32x4_t*>(ws + zz * zs + i * 4);
    f32x4_t s = (s0 + s1) + (s2 + s3);
    const long m = i / nq, n = (i - m * nq) * 4;
    f32x4_t* o = reinterpret_cast<f32x4_t*>(out + m * ldo + n);
    if (accumulate) s += *o;
    *o = s;
  }
  if (ws_cs) {                                 // bias gradient: colsum[m] += sum over the ncs = nsplit * ntx partial slots
    // 8 threads per row, each over a strided subset of the slots with 4 loads in flight; combined by lane shuffles
    const long gt = (long)gridDim.x * blockDim.x - 1 - tid;      // the LAST threads of the grid take this (short) job
    const int part = (int)(gt & 7);
    for (long m = gt >> 3; m < M; m += stride >> 3) {
      float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
      int q = part;
      for (; q + 24 < ncs; q += 32) {
        a0 += ws_cs[(long)q * M + m];
        a1 += ws_cs[(long)(q + 8) * M + m];
        a2 += ws_cs[(long)(q + 16) * M + m];
        a3 += ws_cs[(long)(q + 24) * M + m];
      }
      for (; q < ncs; q += 8) a0 += ws_cs[(long)q * M + m];
      float a = (a0 + a1) + (a2 + a3);
      a += __shfl_xor(a, 1, 64);
      a += __shfl_xor(a, 2, 64);
      a += __shfl_xor(a, 4, 64);
      if (part == 0) colsum[m] = accumulate ? colsum[m] + a : a;
    }
  }
}

// Tail-sliced launches: C tile = epilogue(sum of the S fp32 slice tiles + bias).  One thread = 8 consecutive columns.
__global__ __launch_bounds__(256) void tail_fixup_kernel(const float* __restrict__ ws, int S, int n_full, int ntx, int nty, EpiParams e, int gm) {
  const int lt = blockIdx.x >> 5;
  const int id = (blockIdx.x & 31) * 256 + threadIdx.x;
  const int row = id >> 5, cc = id & 31;
  int tile_x, tile_y;
  tile_from_logical(n_full + lt, ntx, nty, tile_x, tile_y, gm);
  const long m = (long)tile_y * BM + row;
  const int n = tile_x * BN + cc * 8;
  float v[8];
  {
    const float* p = ws + (long)lt * S * (BM * BN) + row * BN + cc * 8;
    ld8(p, v);
    for (int sl = 1; sl < S; ++sl) {
      float u[8];
      ld8(p + (long)sl * (BM * BN), u);
#pragma unroll
      for (int x = 0; x < 8; ++x) v[x] += u[x];
    }
  }
  if (e.bias) {
    float b[8];
    ld8(e.bias + n, b);
#pragma unroll
    for (int x = 0; x < 8; ++x) v[x] += b[x];
  }
  bf16_t* C = reinterpret_cast<bf16_t*>(e.C);
  if (e.epilogue == DH_EPI_GELU) {
    float d[8];
#pragma unroll
    for (int x = 0; x < 8; ++x) quick_gelu_both_f(bf2f(f2bf(v[x])), v[x], d[x]);      // like the in-kernel path: of the bf16-rounded pre-activation
    if (e.aux) st8_hw(reinterpret_cast<bf16_t*>(e.aux) + m * e.ldaux + n, d);
  } else if (e.epilogue == DH_EPI_DGELU) {
    float u[8];
    ld8(reinterpret_cast<const bf16_t*>(e.aux) + m * e.ldaux + n, u);
#pragma unroll
    for (int x = 0; x < 8; ++x) v[x] *= u[x];
  }
  if (e.residual) {
    float r[8];
    ld8(reinterpret_cast<const bf16_t*>(e.residual) + m * e.ldr + n, r);
#pragma unroll
    for (int x = 0; x < 8; ++x) v[x] += r[x];
  }
  st8_hw(C + m * e.ldc + n, v);
}

static int num_cus() {
#if V4_EMU
  return 8;      // the work distribution deals items to 8 XCD lists: the emulated "chip" has one workgroup per list
#endif
  static int n = 0;
  if (!n) {
    int dev = 0;
    hipGetDevice(&dev);
    hipDeviceProp_t p;
    n = (hipGetDeviceProperties(&p, dev) == hipSuccess && p.multiProcessorCount > 0) ? p.multiProcessorCount : 256;
  }
  return n;
}

static int g_v4_dynamic = -2;     // -2 unread; see sched_slot / dh_gemm_v4_set_dynamic
// Scheduler state for the dynamic tile distribution: 16 launch slots x 16 ints (8 per-XCD counters, 1 exit counter), one
// slot per stream that launches v4 GEMMs (two launches on different streams may overlap in time; launches on one stream do
// not).  Library-internal, allocated once, zero at rest (the kernel restores the zeros on its way out).
static int* sched_slot(hipStream_t st, int* dyn) {
  // DH_V4_DYNAMIC: 0 (default) static partition; 1 dynamic after each workgroup's first item -- declip_amd.dist sets it for
  // multi-GPU jobs, where RCCL kernels hold CUs during the overlapped gradient all-reduce (measured with a 16-CU hog:
  // static 83 -> 119 us, dynamic 83 -> 101 us; costs ~1.5 % when nothing else runs); 2 fully dynamic + stealing (experimental)
  if (g_v4_dynamic == -2) { const char* ev = getenv("DH_V4_DYNAMIC"); g_v4_dynamic = ev ? atoi(ev) : 0; }   // environment: read ONCE; dh_gemm_v4_set_dynamic overrides
  int mode = g_v4_dynamic;
  if (mode != 0 && mode != 1) mode = 0;
  *dyn = mode;
  if (!mode) return nullptr;
  constexpr int NSLOT = 16;
  static int* buf = nullptr;
  static hipStream_t owner[NSLOT];
  static int nown = 0;
  if (!buf) {
    if (hipMalloc(&buf, NSLOT * 16 * sizeof(int)) != hipSuccess) { buf = nullptr; *dyn = 0; return nullptr; }
    hipMemset(buf, 0, NSLOT * 16 * sizeof(int));
  }
  for (int i = 0; i < nown; ++i)
    if (owner[i] == st) return buf + i * 16;
  if (nown == NSLOT) { *dyn = 0; return nullptr; }   // more streams than slots: static partition for the extra ones
  owner[nown] = st;
  return buf + (nown++) * 16;
}

// tile order of a launch (see tile_from_logical): DH_V4_GROUP_M overrides (A/B runs); the weight-gradient layout keeps groups
// of 8 (its items are ordered K-slice-major, both operands are streamed once per slice)
static int g_group_m_override = 0;   // set around one launch() by entry points whose big operand is B (the vocabulary matrix of the CE modes)
static int v4_group_m(bool ta) {
  if (g_group_m_override > 0) return g_group_m_override;
  static int env = -2;
  if (env == -2) { const char* ev = getenv("DH_V4_GROUP_M"); env = ev ? atoi(ev) : -1; }
  if (env > 0) return env;
  return ta ? 8 : 1;
}

template <bool TA, bool TB, int MODE>
void launch(const dh_gemm_args* a, const EpiParams& e, int split, int kps, int n_full, int S, hipStream_t st) {
  static bool attr_set = false;
  if (!attr_set) {
    hipFuncSetAttribute((const void*)gemm_v4_kernel<TA, TB, MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    attr_set = true;
  }
  const int ntx = dh_cdiv(a->N, BN), nty = dh_cdiv(a->M, BM);
  const int nitems = S ? n_full + (ntx * nty - n_full) * S : ntx * nty * split;
  int grid = num_cus();
  if (grid > nitems) grid = nitems;
  KArgs ka;
  ka.A = (const bf16_t*)a->A; ka.lda = a->lda; ka.B = (const bf16_t*)a->B; ka.ldb = a->ldb;
  ka.M = a->M; ka.N = a->N; ka.K = a->K; ka.k_per_split = kps; ka.ntx = ntx; ka.nty = nty; ka.nitems = nitems; ka.n_full = n_full; ka.S = S;
  ka.sched = sched_slot(st, &ka.dyn); ka.e = e;
  ka.group_m = v4_group_m(TA);
  hipLaunchKernelGGL((gemm_v4_kernel<TA, TB, MODE>), dim3(grid), dim3(512), LDS_BYTES, st, ka);
}

}  // namespace v4

// Returns true if the v4 kernel took the problem (called from dh_gemm first).  DH_GEMM_V4=0 disables it.
static int g_v4_mode = -2;   // -2 unread, -1 default (on), 0 off, 1 on
extern "C" int dh_gemm_v4_enable(int on) {
  if (g_v4_mode == -2) { const char* ev = getenv("DH_GEMM_V4"); g_v4_mode = ev ? atoi(ev) : -1; }
  const int prev = g_v4_mode;
  g_v4_mode = on;
  return prev;
}
// Tile distribution of the persistent kernel: 0 static partition, 1 dynamic after each workgroup's first item (see sched_slot).  The
// environment (DH_V4_DYNAMIC) is read once, at the first launch; this call overrides it (declip_amd.dist for multi-GPU jobs, the tests).
// Returns the previous setting.
extern "C" int dh_gemm_v4_set_dynamic(int mode) {
  if (v4::g_v4_dynamic == -2) { const char* ev = getenv("DH_V4_DYNAMIC"); v4::g_v4_dynamic = ev ? atoi(ev) : 0; }
  const int prev = v4::g_v4_dynamic;
  v4::g_v4_dynamic = (mode == 1) ? 1 : 0;
  return prev;
}
bool dh_gemm_try_v4(const dh_gemm_args* a, int split, hipStream_t st) {
  using namespace v4;
  if (g_v4_mode == -2) { const char* ev = getenv("DH_GEMM_V4"); g_v4_mode = ev ? atoi(ev) : -1; }
  const int mode = g_v4_mode;
  if (mode == 0 && a->force_generic != 4) return false;
  if (a->dtype != DH_BF16) return false;
  if ((a->lda % 8) || (a->ldb % 8) || ((uintptr_t)a->A & 15) || ((uintptr_t)a->B & 15)) return false;
  if (a->K % BK) return false;
  // operand extents: the DMA reads whole 16-byte chunks; out-of-range rows/columns are clamped (their products
  // land in rows/columns that are never stored), but a contraction-major operand needs whole 8-element chunks
  if (a->a_kmajor && (a->M % 8)) return false;
  if (a->b_kmajor && (a->N % 8)) return false;
  if ((a->M % BM) || (a->N % BN)) return false;   // whole tiles only (every tower shape at b = 256/512 is)
  if (a->accumulate) {
    // own split choice (the caller's split_k is sized for the 128 x 128 kernels): one work item per CU if K allows it,
    // at most 32 slices (reduction traffic), at least 4 K-tiles per slice
    const int tiles = dh_cdiv(a->N, BN) * dh_cdiv(a->M, BM);
    int sp = num_cus() / tiles;
    if (sp > 32 && tiles > 4) sp = 32;            // reduction traffic grows with the slice count ...
    if (sp > 64) sp = 64;                         // ... except for the smallest weights (<= 4 tiles: 1 MB of fp32 per slice)
    if (sp > a->K / (4 * BK)) sp = a->K / (4 * BK);
    if (sp < 1) sp = 1;
    if (a->ws) while (sp > 1 && a->ws_bytes < (int64_t)sp * a->M * a->N * 4 + (int64_t)sp * dh_cdiv(a->N, BN) * a->M * 4) --sp;
    split = sp;
  }
  int kps = ((a->K + split - 1) / split + BK - 1) / BK * BK;
  if (kps < 2 * BK) kps = 2 * BK;                  // the pipeline wants >= 2 K-tiles per work item
  if (a->K < 2 * BK) return false;
  split = (a->K + kps - 1) / kps;
  if (a->K - (split - 1) * kps < 2 * BK) return false;   // short tail split
  int md = MODE_STORE;
  if (a->accumulate) {
    if (!(a->a_kmajor && a->b_kmajor)) return false;           // only the dW layout is instantiated
    md = MODE_ATOMIC;
    // split-K through the workspace: partial tiles + one reduce pass
    if (a->ws && split > 1 && (a->ldc % 4 == 0) && (((uintptr_t)a->C & 15) == 0) && (((uintptr_t)a->ws & 15) == 0) &&
        a->ws_bytes >= (int64_t)split * a->M * a->N * 4 + (int64_t)split * dh_cdiv(a->N, BN) * a->M * 4)
      md = MODE_PARTIAL;
  } else {
    if (a->a_kmajor || a->a_colsum) return false;
    if (a->c_dtype != DH_BF16) return false;
    if (((uintptr_t)a->C & 15) || ((a->ldc * 2) & 15)) return false;
    if (a->residual && (((uintptr_t)a->residual & 15) || ((a->ldr * 2) & 15))) return false;
    if (a->aux && (((uintptr_t)a->aux & 15) || ((a->ldaux * 2) & 15))) return false;
    if (a->bias && ((uintptr_t)a->bias & 15)) return false;
    if (a->N > MAX_BIAS_N) return false;
    // instantiated flavours: forward (B = weight [N][K]): plain / GELU+aux / residual; dX (B contraction-major): plain / dGELU
    if (a->epilogue == DH_EPI_GELU) { if (a->b_kmajor || a->residual || !a->aux) return false; md = MODE_STORE_GELU; }
    else if (a->epilogue == DH_EPI_DGELU) { if (!a->b_kmajor || a->residual) return false; md = MODE_STORE_DGELU; }
    else if (a->residual) { if (a->b_kmajor) return false; md = MODE_STORE_RES; }
  }
  EpiParams e;
  e.M = a->M; e.N = a->N; e.C = a->C; e.ldc = a->ldc; e.bias = a->bias; e.epilogue = a->epilogue;
  e.residual = a->residual; e.ldr = a->ldr; e.aux = a->aux; e.ldaux = a->ldaux; e.alpha = a->alpha;
  e.a_colsum = a->a_colsum;
  e.ws = (float*)a->ws;
  e.ws_cs = (md == MODE_PARTIAL && a->a_colsum) ? (float*)a->ws + (int64_t)split * a->M * a->N : nullptr;
  // tail slicing (bf16 outputs), FEW-tile launches only (the M = b GEMMs of the pooled last block: 6-24 tiles of 12-48 K-tiles ran on
  // 6-24 CUs, 20-68 us each at 12-90 TFLOP/s): every tile is cut in K over the whole chip, tail_fixup_kernel sums the slices and applies
  // the epilogue.  DH_V4_TAIL=0 switches it off (read once).  The two LONG-list variants of rounds 3-5 (300 tiles on 256 CUs: the 44
  // tiles of the last round cut in K, summed by a fix-up launch or by the last slice to arrive inside the kernel) each won on the launch
  // alone and lost in the step (profiles/r03_ab_tail_slicing.txt, r05_tail_in_kernel.txt); round 6 moved them out of the shipped
  // library: tools/experiments/gemm_v4_r05_with_tail_modes.hip.txt.
  int n_full = 0, S = 0;
  if ((md == MODE_STORE || md == MODE_STORE_RES) && a->ws && (((uintptr_t)a->ws & 15) == 0)) {
    static int tail = -1;
    if (tail < 0) { const char* ev = getenv("DH_V4_TAIL"); tail = ev ? (atoi(ev) != 0) : 1; }
    const int T = dh_cdiv(a->N, BN) * dh_cdiv(a->M, BM), G = num_cus(), nkt = a->K / BK;
    if (tail && T <= G / 2 && nkt >= 8) {
      int s_ = G / T;
      if (s_ > 32) s_ = 32;
      if (s_ > nkt / 2) s_ = nkt / 2;
      if (s_ >= 2 && a->ws_bytes >= (int64_t)T * s_ * BM * BN * 4) { S = s_; n_full = 0; }
    }
  }
  if (md == MODE_ATOMIC && a->accumulate == 2) {           // first touch through the atomic path: clear, then add
    if (a->ldc == a->N) DH_RT_NOTE(hipMemsetAsync(a->C, 0, sizeof(float) * (size_t)a->M * a->N, st), "gemm_v4: clearing a first-touch gradient");
    else DH_RT_NOTE(hipMemset2DAsync(a->C, sizeof(float) * a->ldc, 0, sizeof(float) * a->N, a->M, st), "gemm_v4: clearing a first-touch gradient");
    if (a->a_colsum) DH_RT_NOTE(hipMemsetAsync(a->a_colsum, 0, sizeof(float) * (size_t)a->M, st), "gemm_v4: clearing a first-touch bias gradient");
    if (dh_helper_error()) return true;         // (the entry point reports it: DH_HELPER_FAILED)
  }
  switch (md) {
    case MODE_ATOMIC: launch<true, true, MODE_ATOMIC>(a, e, split, kps, n_full, S, st); break;
    case MODE_PARTIAL: launch<true, true, MODE_PARTIAL>(a, e, split, kps, n_full, S, st); break;
    case MODE_STORE_GELU: launch<false, false, MODE_STORE_GELU>(a, e, split, kps, n_full, S, st); break;
    case MODE_STORE_RES: launch<false, false, MODE_STORE_RES>(a, e, split, kps, n_full, S, st); break;
    case MODE_STORE_DGELU: launch<false, true, MODE_STORE_DGELU>(a, e, split, kps, n_full, S, st); break;
    default:
      if (a->b_kmajor) launch<false, true, MODE_STORE>(a, e, split, kps, n_full, S, st);
      else launch<false, false, MODE_STORE>(a, e, split, kps, n_full, S, st);
  }
  if (S) hipLaunchKernelGGL(tail_fixup_kernel, dim3((dh_cdiv(a->N, BN) * dh_cdiv(a->M, BM) - n_full) * 32), dim3(256), 0, st,
                            (const float*)a->ws, S, n_full, dh_cdiv(a->N, BN), dh_cdiv(a->M, BM), e, v4_group_m(false));
  if (md == MODE_PARTIAL) {
    const long n4 = (long)a->M * a->N / 4;
    int blocks = (int)((n4 + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    // accumulate == 2 (first touch of this gradient in the step): the reduce pass WRITES the sum -- C is neither zeroed beforehand nor read
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3(blocks), dim3(256), 0, st, (const float*)a->ws, (float*)a->C, (long)a->ldc, a->M,
                       a->N, split, a->accumulate == 2 ? 0 : 1, (const float*)e.ws_cs, a->a_colsum, split * dh_cdiv(a->N, BN));
  }
  return true;
}

// ---- grouped weight gradients (MODE_GROUP) -------------------------------------------------------------------------------------
namespace v4 {
struct GReduce {
  int overwrite[MAX_GROUP];       // problem is the FIRST gradient contribution of the step: the sum is written, the slot not read (dh_gemm_args.accumulate == 2)
  float* out[MAX_GROUP];          // gradient [M][N], contiguous (ldc == N)
  float* colsum[MAX_GROUP];       // bias gradient [M] or null
  long ws_off[MAX_GROUP + 1];     // float offsets inside a slab (ascending; [n] = zs)
  long cs_off[MAX_GROUP];
  int M[MAX_GROUP], ntx[MAX_GROUP];
  int n;
};
// out_p[i] += sum_z ws[z][ws_off_p + i]  for every problem of the group; bias gradients: colsum_p[m] += sum_{z, tx} cs[z][cs_off_p + tx*M_p + m]
__global__ __launch_bounds__(256) void splitk_reduce_group_kernel(const float* __restrict__ ws, const float* __restrict__ cs, long zs, long cs_zs,
                                                                  int nsplit, GReduce g) {
  const long n4 = zs / 4;
  const long stride = (long)gridDim.x * blockDim.x;
  const long tid = (long)blockIdx.x * blockDim.x + threadIdx.x;
  for (long i = tid; i < n4; i += stride) {
    const long e = i * 4;
    f32x4_t s0 = {0.f, 0.f, 0.f, 0.f}, s1 = s0;
    int zz = 0;
    for (; zz + 2 <= nsplit; zz += 2) {
      s0 += *reinterpret_cast<const f32x4_t*>(ws + (long)zz * zs + e);
      s1 += *reinterpret_cast<const f32x4_t*>(ws + (long)(zz + 1) * zs + e);
    }
    if (zz < nsplit) s0 += *reinterpret_cast<const f32x4_t*>(ws + (long)zz * zs + e);
    int p = 0;
#pragma unroll
    for (int q = 1; q < MAX_GROUP; ++q) p = (q < g.n && e >= g.ws_off[q]) ? q : p;
    float* op = nullptr;
    long off = 0;
#pragma unroll
    for (int q = 0; q < MAX_GROUP; ++q) if (q == p) { op = g.out[q]; off = g.ws_off[q]; }
    f32x4_t* o = reinterpret_cast<f32x4_t*>(op + (e - off));
    bool ow = false;
#pragma unroll
    for (int q = 0; q < MAX_GROUP; ++q) if (q == p) ow = g.overwrite[q] != 0;
    *o = ow ? (s0 + s1) : (*o + (s0 + s1));
  }
  if (cs) {
    // the LAST threads of the grid take the (short) bias-gradient job: one thread per (problem, row)
    long gt = (long)gridDim.x * blockDim.x - 1 - tid;
#pragma unroll
    for (int q = 0; q < MAX_GROUP; ++q) {
      if (q >= g.n) break;
      const int Mq = g.M[q];
      if (g.colsum[q] && gt >= 0 && gt < Mq) {
        float a = 0.f;
        for (int z = 0; z < nsplit; ++z)
          for (int tx = 0; tx < g.ntx[q]; ++tx) a += cs[(long)z * cs_zs + g.cs_off[q] + (long)tx * Mq + gt];
        g.colsum[q][gt] = g.overwrite[q] ? a : g.colsum[q][gt] + a;
      }
      gt -= Mq;
    }
  }
}
}  // namespace v4

// The n weight-gradient problems C_p[M_p][N_p] += A_p^T B_p (dh_gemm_args with a_kmajor = b_kmajor = accumulate = 1, fp32 C, same K)
// as one launch + one reduce pass.  Returns false (nothing launched) if the group does not fit the kernel: the caller then issues
// the problems one by one.
bool dh_gemm_try_v4_group(const dh_gemm_args* a, int n, hipStream_t st) {
  using namespace v4;
  if (g_v4_mode == -2) { const char* ev = getenv("DH_GEMM_V4"); g_v4_mode = ev ? atoi(ev) : -1; }
  if (g_v4_mode == 0) return false;
  static int grp_on = -1;
  if (grp_on < 0) { const char* ev = getenv("DH_V4_GROUP"); grp_on = ev ? atoi(ev) : 1; }
  if (!grp_on || n < 2 || n > MAX_GROUP) return false;
  const int K = a[0].K;
  if (K % BK || K < 8 * BK || !a[0].ws || ((uintptr_t)a[0].ws & 15)) return false;
  KArgs ka;
  memset(&ka, 0, sizeof(ka));
  GReduce gr;
  memset(&gr, 0, sizeof(gr));
  int T = 0;
  long zs = 0, cs_zs = 0;
  bool any_cs = false;
  for (int p = 0; p < MAX_GROUP; ++p) ka.gp[p].tile0 = 0x7fffffff;
  for (int p = 0; p < n; ++p) {
    const dh_gemm_args& q = a[p];
    if (q.dtype != DH_BF16 || q.c_dtype != DH_F32 || !q.a_kmajor || !q.b_kmajor || !q.accumulate || q.K != K) return false;
    if (q.bias || q.residual || q.epilogue != DH_EPI_NONE || q.alpha != 1.f) return false;
    if ((q.lda % 8) || (q.ldb % 8) || ((uintptr_t)q.A & 15) || ((uintptr_t)q.B & 15) || ((uintptr_t)q.C & 15)) return false;
    if ((q.M % BM) || (q.N % BN) || q.ldc != q.N) return false;
    GProb& g = ka.gp[p];
    g.A = (const bf16_t*)q.A; g.B = (const bf16_t*)q.B; g.lda = q.lda; g.ldb = q.ldb; g.M = q.M; g.N = q.N;
    g.ntx = q.N / BN; g.nty = q.M / BM; g.tile0 = T; g.ws_off = zs; g.cs_off = cs_zs; g.a_colsum = (float*)q.a_colsum;
    gr.overwrite[p] = q.accumulate == 2;
    gr.out[p] = (float*)q.C; gr.colsum[p] = (float*)q.a_colsum; gr.ws_off[p] = zs; gr.cs_off[p] = cs_zs; gr.M[p] = q.M; gr.ntx[p] = g.ntx;
    T += g.ntx * g.nty;
    zs += (long)q.M * q.N;
    cs_zs += (long)g.ntx * q.M;
    any_cs = any_cs || q.a_colsum;
  }
  gr.n = n;
  for (int p = n; p <= MAX_GROUP; ++p) gr.ws_off[p] = zs;
  // K-slices: minimise  rounds x (K-tiles per slice + epilogue) + reduce traffic   [unit: one K-tile of one CU, ~1.3 us]
  const int G = num_cus(), nkt = K / BK;
  const double epi = 8.0, red = (double)zs * 4.0 / 5.3e6;
  int best = 1;
  double best_cost = 1e30;
  for (int S = 1; S <= 16 && nkt / S >= 8; ++S) {
    const int rounds = (T * S + G - 1) / G;
    const int per = (nkt + S - 1) / S;
    if (nkt - (S - 1) * per < 2) continue;              // short tail slice
    const double cost = rounds * (per + epi) + (S > 1 ? S * red : 0.0);
    if (a[0].ws_bytes < (int64_t)S * (zs + cs_zs) * 4) break;
    if (cost < best_cost) { best_cost = cost; best = S; }
  }
  static int force_s = -1;
  if (force_s < 0) { const char* ev = getenv("DH_V4_GROUP_SPLIT"); force_s = ev ? atoi(ev) : 0; }
  int split = force_s > 0 ? force_s : best;
  int kps = ((nkt + split - 1) / split) * BK;
  split = (K + kps - 1) / kps;
  if (K - (split - 1) * kps < 2 * BK || kps < 2 * BK) return false;
  if (a[0].ws_bytes < (int64_t)split * (zs + cs_zs) * 4) return false;
  static bool attr_set = false;
  if (!attr_set) {
    hipFuncSetAttribute((const void*)gemm_v4_kernel<true, true, MODE_GROUP>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    attr_set = true;
  }
  ka.K = K; ka.k_per_split = kps; ka.nitems = T * split; ka.T = T; ka.ngrp = n; ka.zs = zs; ka.cs_zs = cs_zs;
  ka.A = ka.gp[0].A; ka.B = ka.gp[0].B; ka.lda = ka.gp[0].lda; ka.ldb = ka.gp[0].ldb; ka.M = ka.gp[0].M; ka.N = ka.gp[0].N;
  ka.ntx = ka.gp[0].ntx; ka.nty = ka.gp[0].nty;
  ka.sched = sched_slot(st, &ka.dyn);
  ka.group_m = 8;
  ka.e.alpha = 1.f;
  ka.e.ws = (float*)a[0].ws;
  ka.e.ws_cs = any_cs ? (float*)a[0].ws + (int64_t)split * zs : nullptr;
  ka.e.a_colsum = any_cs ? ka.gp[0].a_colsum : nullptr;
  int grid = G;
  if (grid > ka.nitems) grid = ka.nitems;
  hipLaunchKernelGGL((gemm_v4_kernel<true, true, MODE_GROUP>), dim3(grid), dim3(512), LDS_BYTES, st, ka);
  int blocks = (int)((zs / 4 + 255) / 256);
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(splitk_reduce_group_kernel, dim3(blocks), dim3(256), 0, st, (const float*)a[0].ws, (const float*)ka.e.ws_cs, zs, cs_zs, split, gr);
  return true;
}

// FILIP token-wise max-sim, forward (filip.py:96-105): raw[i][l] = mean_j max_m <Q[(i,j)], K[(l,m)]>, argmax[(i,j)][l] = that m.
// Q [rows_pad][D] bf16 (rows_pad = b*J rounded up to 256, the padding rows anything finite), K [B*16][D] bf16.  raw is zeroed here.
bool dh_maxsim_try_v4(const void* Q, const void* Ksel, int rows_pad, int b, int B, int J, int D, float* raw, uint8_t* arg, hipStream_t st) {
  using namespace v4;
  if (g_v4_mode == -2) { const char* ev = getenv("DH_GEMM_V4"); g_v4_mode = ev ? atoi(ev) : -1; }
  if (g_v4_mode == 0) return false;
  const int N = B * 16;
  if ((rows_pad % BM) || (N % BN) || (D % BK) || D < 2 * BK || ((uintptr_t)Q & 15) || ((uintptr_t)Ksel & 15) || ((uintptr_t)arg & 15)) return false;
  if (J < 1 || 256 / J + 2 > 16) return false;               // at most 16 sample segments per 256-row tile
  DH_RT_NOTE(hipMemsetAsync(raw, 0, sizeof(float) * (size_t)b * B, st), "maxsim: clearing the raw similarities");
  if (dh_helper_error()) return true;           // (the entry point reports it: DH_HELPER_FAILED)
  dh_gemm_args a;
  memset(&a, 0, sizeof(a));
  a.dtype = DH_BF16; a.c_dtype = DH_F32; a.M = rows_pad; a.N = N; a.K = D; a.A = Q; a.lda = D; a.B = Ksel; a.ldb = D; a.C = raw; a.ldc = B;
  a.alpha = 1.f;
  EpiParams e;
  memset(&e, 0, sizeof(e));
  e.M = rows_pad; e.N = N; e.C = raw; e.ldc = B; e.aux = arg; e.ldaux = B; e.alpha = 1.f; e.ms_J = J; e.ms_b = b; e.ms_B = B;
  launch<false, false, MODE_MAXSIM>(&a, e, 1, D, 0, 0, st);
  return true;
}

// ---- masked-LM cross-entropy without fp32 logits (MODE_CE_FWD / MODE_CE_BWD) -----------------------------------------------------
namespace v4 {
// row r: merge its ntx (max, sum exp) partials -> lse; loss = lse - label logit
__global__ __launch_bounds__(256) void ce_finalize_kernel(const float* __restrict__ part, const float* __restrict__ lab, int n, int ntx,
                                                          float* __restrict__ row_loss, float* __restrict__ row_lse) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int r = blockIdx.x * 4 + wave; r < n; r += gridDim.x * 4) {
    const float* p = part + (long)r * ntx * 2;
    float mm = -INFINITY, ss = 0.f;
    for (int t = lane; t < ntx; t += 64) {
      const float m = p[2 * t], s = p[2 * t + 1];
      if (m > mm) { ss = ss * __expf(mm - m) + s; mm = m; } else if (m != -INFINITY) ss += s * __expf(m - mm);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const float m2 = __shfl_xor(mm, o, 64), s2 = __shfl_xor(ss, o, 64);
      const float m = fmaxf(mm, m2);
      ss = (m == -INFINITY) ? 0.f : ss * __expf(mm - m) + s2 * __expf(m2 - m);
      mm = m;
    }
    if (lane == 0) { const float lse = mm + __logf(ss); row_lse[r] = lse; row_loss[r] = lse - lab[r]; }
  }
}
}  // namespace v4

static bool ce_v4_ok(const void* X, const void* W, int n_pad, int V, int K) {
  using namespace v4;
  if (g_v4_mode == -2) { const char* ev = getenv("DH_GEMM_V4"); g_v4_mode = ev ? atoi(ev) : -1; }
  return g_v4_mode != 0 && n_pad % BM == 0 && V >= BN && K % BK == 0 && K >= 2 * BK && !((uintptr_t)X & 15) && !((uintptr_t)W & 15);
}
// forward: X [n_pad][K] bf16 (rows >= n zero), W [V][K] bf16, bias [V] fp32 or null, labels [n] int64 -> row_loss / row_lse [n];
// ws: n_pad * (ceil(V/256) * 2 + 1) floats
bool dh_ce_try_v4_fwd(const void* X, const void* W, const float* bias, const long long* labels, int n, int n_pad, int V, int K,
                      float* row_loss, float* row_lse, float* ws, int64_t ws_bytes, hipStream_t st) {
  using namespace v4;
  if (!ce_v4_ok(X, W, n_pad, V, K)) return false;
  const int ntx = dh_cdiv(V, BN);
  if (ws_bytes < (int64_t)n_pad * (ntx * 2 + 1) * 4) return false;
  dh_gemm_args a;
  memset(&a, 0, sizeof(a));
  a.dtype = DH_BF16; a.c_dtype = DH_F32; a.M = n_pad; a.N = V; a.K = K; a.A = X; a.lda = K; a.B = W; a.ldb = K; a.C = ws; a.ldc = 1; a.alpha = 1.f;
  EpiParams e;
  memset(&e, 0, sizeof(e));
  e.M = n_pad; e.N = V; e.bias = bias; e.alpha = 1.f; e.ce_labels = labels; e.ce_part = ws; e.ce_lab = ws + (int64_t)n_pad * ntx * 2;
  e.ce_n = n; e.ce_V = V;
  g_group_m_override = dh_cdiv(n_pad, BM);      // column-major: the 50 MB vocabulary matrix is fetched once, the 6 MB of rows stay in L2
  launch<false, false, MODE_CE_FWD>(&a, e, 1, K, 0, 0, st);
  g_group_m_override = 0;
  int blocks = dh_cdiv(n, 4);
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(ce_finalize_kernel, dim3(blocks), dim3(256), 0, st, (const float*)e.ce_part, (const float*)e.ce_lab, n, ntx, row_loss, row_lse);
  return true;
}
// backward: dl [n_pad][ldd] bf16 = g[row] * (softmax(X W^T + bias)[row] - onehot(label[row])), zero for rows >= n and columns >= V
bool dh_ce_try_v4_bwd(const void* X, const void* W, const float* bias, const long long* labels, const float* row_lse, const float* g_row,
                      int n, int n_pad, int V, int K, void* dl, int64_t ldd, hipStream_t st) {
  using namespace v4;
  if (!ce_v4_ok(X, W, n_pad, V, K) || ((uintptr_t)dl & 15) || (ldd % 8) || ldd < V) return false;
  dh_gemm_args a;
  memset(&a, 0, sizeof(a));
  a.dtype = DH_BF16; a.c_dtype = DH_BF16; a.M = n_pad; a.N = V; a.K = K; a.A = X; a.lda = K; a.B = W; a.ldb = K; a.C = dl; a.ldc = ldd; a.alpha = 1.f;
  EpiParams e;
  memset(&e, 0, sizeof(e));
  e.M = n_pad; e.N = V; e.C = dl; e.ldc = ldd; e.bias = bias; e.alpha = 1.f; e.ce_labels = labels; e.ce_lse = row_lse; e.ce_g = g_row;
  e.ce_n = n; e.ce_V = V; e.ce_ldd = (int)ldd;
  g_group_m_override = dh_cdiv(n_pad, BM);
  launch<false, false, MODE_CE_BWD>(&a, e, 1, K, 0, 0, st);
  g_group_m_override = 0;
  return true;
}
