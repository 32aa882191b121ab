// One ResidualAttentionBlock per C-ABI call (dh_block_fwd / dh_block_bwd): the host enqueues the block's kernels from C instead of
// one Python -> ctypes round trip per kernel.
//
// Reference: prototype/model/image_encoder/base_transformer.py:29-53 (ResidualAttentionBlock.forward: x + attn(ln_1(x)), then
// x + mlp(ln_2(x)) with QuickGELU) and what autograd derives from it.  The reference steps through this with ~20 eager torch ops per
// block and direction; the engine of rounds 1-3 did the same with 9 + 13 ctypes calls from Python (42 us of host time per launch:
// ~20 ms per CLIP step, as long as the step itself -- hidden by the hipGraph only while the batch is resident, never in the solver
// or behind the input pipeline).  Here the sequence LN -> QKV GEMM -> attention -> out_proj (+residual) -> LN -> c_fc (+GELU) ->
// c_proj (+residual) and its backward (incl. the block's four weight gradients as ONE grouped launch) are enqueued by one call.
// Every kernel is the one the per-op entry points launch; this file adds no arithmetic.
//
// Activations of a block live in ONE caller-owned slab (dh_block_act_bytes), laid out here, kept for the backward:
//   h1 [R,d] | qkv [R,3d] | a [R,d] | x_mid [R,d] | h2 [R,d] | u [R,4d] | g [R,4d] | mean1 rstd1 mean2 rstd2 [R] f32 | lse [b,H,L] f32
// backward temporaries in a second slab (dh_block_bwd_scratch_bytes): du [R,4d] | dqkv [R,3d] | dh2 | dx_mid | da | dh1 [R,d].
#include "dh_common.h"
#include <string.h>

namespace {

inline int64_t al256(int64_t v) { return (v + 255) & ~(int64_t)255; }
inline int esz(int dtype) { return dtype == DH_BF16 ? 2 : 4; }

struct ActLayout {
  int64_t h1, qkv, a, x_mid, h2, u, g, mean1, rstd1, mean2, rstd2, lse, total;
};
ActLayout act_layout(int dtype, int rows, int d, int heads, int b, int L) {
  ActLayout o;
  const int64_t e = esz(dtype), rd = (int64_t)rows * d * e;
  int64_t p = 0;
  o.h1 = p; p += al256(rd);
  o.qkv = p; p += al256(3 * rd);
  o.a = p; p += al256(rd);
  o.x_mid = p; p += al256(rd);
  o.h2 = p; p += al256(rd);
  o.u = p; p += al256(4 * rd);
  o.g = p; p += al256(4 * rd);
  o.mean1 = p; p += al256((int64_t)rows * 4);
  o.rstd1 = p; p += al256((int64_t)rows * 4);
  o.mean2 = p; p += al256((int64_t)rows * 4);
  o.rstd2 = p; p += al256((int64_t)rows * 4);
  o.lse = p; p += al256((int64_t)b * heads * L * 4);
  o.total = p;
  return o;
}
struct BwdLayout {
  int64_t du, dqkv, dh2, dx_mid, da, dh1, total;
};
BwdLayout bwd_layout(int dtype, int rows, int d) {
  BwdLayout o;
  const int64_t rd = (int64_t)rows * d * esz(dtype);
  int64_t p = 0;
  o.du = p; p += al256(4 * rd);
  o.dqkv = p; p += al256(3 * rd);
  o.dh2 = p; p += al256(rd);
  o.dx_mid = p; p += al256(rd);
  o.da = p; p += al256(rd);
  o.dh1 = p; p += al256(rd);
  o.total = p;
  return o;
}

int check_common(const dh_block_args* a, const char* who) {
  DH_REQUIRE(a, "%s: null args", who);
  DH_REQUIRE(a->dtype == DH_BF16 || a->dtype == DH_F32, "%s: bad dtype", who);
  DH_REQUIRE(a->rows > 0 && a->d > 0 && a->heads > 0 && a->d % a->heads == 0 && a->b > 0 && a->L > 0, "%s: bad geometry", who);
  DH_REQUIRE(a->cu || (int64_t)a->b * a->L == a->rows, "%s: dense attention needs rows == b * L (%d vs %d x %d)", who, a->rows, a->b, a->L);
  DH_REQUIRE(!a->seq_order || (a->cu && a->seq_ranges && a->L_short > 0 && a->L_short <= a->L), "%s: length buckets need cu, ranges and 0 < L_short <= L", who);
  DH_REQUIRE(!a->cu || a->rows_valid == -1 || (a->rows_valid > 0 && a->rows_valid <= a->rows), "%s: packed rows %d of %d", who, a->rows_valid, a->rows);
  DH_REQUIRE(a->act && a->act_bytes >= act_layout(a->dtype, a->rows, a->d, a->heads, a->b, a->L).total, "%s: activation slab too small", who);
  const dh_block_params& p = a->p;
  DH_REQUIRE(p.w_in && p.w_out && p.w_fc && p.w_proj && p.b_in && p.b_out && p.b_fc && p.b_proj && p.ln1_w && p.ln1_b && p.ln2_w && p.ln2_b,
             "%s: null parameter", who);
  return DH_OK;
}

// C[M,N] = epi(A[M,K] W^T + bias) (forward, W stored [N][K]) or A[M,K] W (dX, W stored [K][N]: b_kmajor)
int linear(const dh_block_args* a, const void* A, int M, int K, const void* W, int N, bool b_kmajor, const float* bias, int epi,
           const void* residual, void* aux, void* C, bool use_ws, dh_stream_t st) {
  dh_gemm_args g;
  memset(&g, 0, sizeof(g));
  g.dtype = a->dtype; g.c_dtype = a->dtype; g.b_kmajor = b_kmajor ? 1 : 0;
  g.M = M; g.N = N; g.K = K;
  g.A = A; g.lda = K; g.B = W; g.ldb = b_kmajor ? N : K; g.C = C; g.ldc = N;
  g.bias = bias; g.epilogue = epi;
  g.residual = residual; g.ldr = N;
  g.aux = aux; g.ldaux = N;
  g.split_k = 1; g.alpha = 1.f;
  if (use_ws && a->dtype == DH_BF16) { g.ws = a->ws; g.ws_bytes = a->ws_bytes; }     // (few-tile launches are cut in K over the chip: gemm_v4.hip)
  return dh_gemm(&g, st);
}

}  // namespace

extern "C" int64_t dh_block_act_bytes(int dtype, int rows, int d, int heads, int b, int L) {
  if (rows <= 0 || d <= 0 || heads <= 0 || b <= 0 || L <= 0) return 0;
  return act_layout(dtype, rows, d, heads, b, L).total;
}
extern "C" int dh_block_act_offsets(int dtype, int rows, int d, int heads, int b, int L, int64_t* out12) {
  DH_REQUIRE(out12 && rows > 0 && d > 0 && heads > 0 && b > 0 && L > 0, "dh_block_act_offsets: bad args");
  const ActLayout o = act_layout(dtype, rows, d, heads, b, L);
  const int64_t v[12] = {o.h1, o.qkv, o.a, o.x_mid, o.h2, o.u, o.g, o.mean1, o.rstd1, o.mean2, o.rstd2, o.lse};
  for (int i = 0; i < 12; ++i) out12[i] = v[i];
  return DH_OK;
}
extern "C" int64_t dh_block_bwd_scratch_bytes(int dtype, int rows, int d) {
  if (rows <= 0 || d <= 0) return 0;
  return bwd_layout(dtype, rows, d).total;
}

#define RUN(expr)                    \
  do {                               \
    const int rc__ = (expr);         \
    if (rc__ != DH_OK) return rc__;  \
  } while (0)

extern "C" int dh_block_fwd(const dh_block_args* a, dh_stream_t st) {
  RUN(check_common(a, "dh_block_fwd"));
  DH_REQUIRE(a->x && a->x_out, "dh_block_fwd: null x / x_out");
  const dh_block_params& p = a->p;
  const int R = a->rows, d = a->d, hd = d / a->heads;
  const ActLayout o = act_layout(a->dtype, R, d, a->heads, a->b, a->L);
  char* s = (char*)a->act;
  void *h1 = s + o.h1, *qkv = s + o.qkv, *at = s + o.a, *x_mid = s + o.x_mid, *h2 = s + o.h2, *u = a->save ? s + o.u : nullptr, *g = s + o.g;
  float *mean1 = (float*)(s + o.mean1), *rstd1 = (float*)(s + o.rstd1), *mean2 = (float*)(s + o.mean2), *rstd2 = (float*)(s + o.rstd2);
  float* lse = (float*)(s + o.lse);
  RUN(dh_layernorm_fwd(a->dtype, a->x, p.ln1_w, p.ln1_b, h1, mean1, rstd1, R, d, p.eps1, st));
  RUN(linear(a, h1, R, d, p.w_in, 3 * d, false, p.b_in, DH_EPI_NONE, nullptr, nullptr, qkv, false, st));
  if (a->cu && a->seq_order) RUN(dh_attn_bucketed_fwd(a->dtype, qkv, at, lse, a->cu, a->seq_order, a->seq_ranges, a->b, a->L, a->L_short, a->heads, hd,
                                                      a->causal, a->rows_valid, R, st));
  else if (a->cu) RUN(dh_attn_varlen_fwd(a->dtype, qkv, at, lse, a->cu, a->b, a->L, a->heads, hd, a->causal, a->rows_valid, R, st));
  else RUN(dh_attn_fwd(a->dtype, qkv, at, lse, a->b, a->L, a->heads, hd, a->causal, st));
  RUN(linear(a, at, R, d, p.w_out, d, false, p.b_out, DH_EPI_NONE, a->x, nullptr, x_mid, true, st));
  RUN(dh_layernorm_fwd(a->dtype, x_mid, p.ln2_w, p.ln2_b, h2, mean2, rstd2, R, d, p.eps2, st));
  RUN(linear(a, h2, R, d, p.w_fc, 4 * d, false, p.b_fc, DH_EPI_GELU, nullptr, u, g, false, st));
  RUN(linear(a, g, R, 4 * d, p.w_proj, d, false, p.b_proj, DH_EPI_NONE, x_mid, nullptr, a->x_out, true, st));
  return DH_OK;
}

extern "C" int dh_block_bwd(dh_block_args* a, dh_stream_t st) {
  RUN(check_common(a, "dh_block_bwd"));
  const dh_block_params& p = a->p;
  DH_REQUIRE(a->x && a->dx_out && a->dx, "dh_block_bwd: null x / dx_out / dx");
  DH_REQUIRE(p.g_w_in && p.g_w_out && p.g_w_fc && p.g_w_proj && p.g_b_in && p.g_b_out && p.g_b_fc && p.g_b_proj && p.g_ln1_w && p.g_ln1_b &&
             p.g_ln2_w && p.g_ln2_b, "dh_block_bwd: null gradient slot");
  const int R = a->rows, d = a->d, hd = d / a->heads;
  DH_REQUIRE(a->scratch && a->scratch_bytes >= bwd_layout(a->dtype, R, d).total, "dh_block_bwd: scratch slab too small");
  DH_REQUIRE(a->ln_part1 && a->ln_part2 && a->ln_part_bytes >= dh_layernorm_bwd_ws_bytes(R, d), "dh_block_bwd: LayerNorm partial slices too small");
  const ActLayout o = act_layout(a->dtype, R, d, a->heads, a->b, a->L);
  const BwdLayout q = bwd_layout(a->dtype, R, d);
  char* s = (char*)a->act;
  char* t = (char*)a->scratch;
  void *h1 = s + o.h1, *qkv = s + o.qkv, *at = s + o.a, *x_mid = s + o.x_mid, *h2 = s + o.h2, *u = s + o.u, *g = s + o.g;
  float *mean1 = (float*)(s + o.mean1), *rstd1 = (float*)(s + o.rstd1), *mean2 = (float*)(s + o.mean2), *rstd2 = (float*)(s + o.rstd2);
  float* lse = (float*)(s + o.lse);
  void *du = t + q.du, *dqkv = t + q.dqkv, *dh2 = t + q.dh2, *dx_mid = t + q.dx_mid, *da = t + q.da, *dh1 = t + q.dh1;
  // MLP: x_out = x_mid + gelu(h2 Wfc^T + bfc) Wproj^T + bproj
  RUN(linear(a, a->dx_out, R, d, p.w_proj, 4 * d, true, nullptr, DH_EPI_DGELU, nullptr, u, du, false, st));
  RUN(linear(a, du, R, 4 * d, p.w_fc, d, true, nullptr, DH_EPI_NONE, nullptr, nullptr, dh2, true, st));
  RUN(dh_layernorm_bwd_part(a->dtype, dh2, x_mid, p.ln2_w, mean2, rstd2, a->dx_out, dx_mid, p.g_ln2_w, p.g_ln2_b, R, d, a->ln_part2,
                            a->ln_part_bytes, &a->ln_nb2, st));
  // attention: x_mid = x + attn(h1) Wout^T + bout
  RUN(linear(a, dx_mid, R, d, p.w_out, d, true, nullptr, DH_EPI_NONE, nullptr, nullptr, da, true, st));
  if (a->cu && a->seq_order) RUN(dh_attn_bucketed_bwd(a->dtype, qkv, at, da, lse, dqkv, a->cu, a->seq_order, a->seq_ranges, a->b, a->L, a->L_short,
                                                      a->heads, hd, a->causal, a->rows_valid, R, st));
  else if (a->cu) RUN(dh_attn_varlen_bwd(a->dtype, qkv, at, da, lse, dqkv, a->cu, a->b, a->L, a->heads, hd, a->causal, a->rows_valid, R, st));
  else RUN(dh_attn_bwd(a->dtype, qkv, at, da, lse, dqkv, a->b, a->L, a->heads, hd, a->causal, st));
  RUN(linear(a, dqkv, R, 3 * d, p.w_in, d, true, nullptr, DH_EPI_NONE, nullptr, nullptr, dh1, true, st));
  RUN(dh_layernorm_bwd_part(a->dtype, dh1, a->x, p.ln1_w, mean1, rstd1, dx_mid, a->dx, p.g_ln1_w, p.g_ln1_b, R, d, a->ln_part1,
                            a->ln_part_bytes, &a->ln_nb1, st));
  // the block's four weight gradients (+ bias gradients): one grouped launch, all inputs are final here
  dh_gemm_args w[4];
  memset(w, 0, sizeof(w));
  const void* dys[4] = {a->dx_out, du, dx_mid, dqkv};
  const void* xs[4] = {g, h2, at, h1};
  float* gws[4] = {p.g_w_proj, p.g_w_fc, p.g_w_out, p.g_w_in};
  float* gbs[4] = {p.g_b_proj, p.g_b_fc, p.g_b_out, p.g_b_in};
  const int outs[4] = {d, 4 * d, d, 3 * d}, ins[4] = {4 * d, d, d, d};
  for (int i = 0; i < 4; ++i) {
    dh_gemm_args& g4 = w[i];
    g4.dtype = a->dtype; g4.c_dtype = DH_F32; g4.a_kmajor = 1; g4.b_kmajor = 1;
    g4.M = outs[i]; g4.N = ins[i]; g4.K = R;
    g4.A = dys[i]; g4.lda = outs[i]; g4.B = xs[i]; g4.ldb = ins[i]; g4.C = gws[i]; g4.ldc = ins[i];
    g4.accumulate = a->dw_first_touch ? 2 : 1; g4.alpha = 1.f; g4.a_colsum = gbs[i];
    const int tiles = ((outs[i] + 127) / 128) * ((ins[i] + 127) / 128);
    int sk = 1024 / (tiles > 0 ? tiles : 1);
    if (sk > R / 512) sk = R / 512;
    g4.split_k = sk < 1 ? 1 : sk;                      // (used only when the group is issued one by one)
    if (a->dtype == DH_BF16) { g4.ws = a->ws; g4.ws_bytes = a->ws_bytes; }
  }
  RUN(dh_gemm_group(w, 4, st));
  return DH_OK;
}
