/* libdeclip_hip.so -- C-ABI of the MI355X (gfx950) contrastive-training hot path.
 *
 * The reference (Sense-GVT/DeCLIP) has NO native boundary: its hot path is eager
 * PyTorch behind a Python plug-in surface (prototype/model/__init__.py:15-21,
 * prototype/solver/clip_solver.py:413-566).  This header is the boundary a
 * maintainer would bind instead (ctypes stub in INTEGRATION.md); every entry
 * point cites the reference code whose arithmetic it replaces (paths relative
 * to /root/reference/prototype unless noted).
 *
 * Conventions
 *  - extern "C"; every function returns 0 (DH_OK) or a negative error code and
 *    never throws; dh_last_error() returns a thread-local message.
 *  - All pointers are BORROWED raw device pointers (row-major, contiguous unless a
 *    leading dimension is given); the caller keeps them alive until the stream op
 *    completes.  The library never allocates or frees tensor memory.
 *  - Every function only ENQUEUES work on `stream` (a hipStream_t passed as void*);
 *    nothing synchronises the device.
 *  - dtype: DH_F32 = validation precision (parity <= 1e-3 vs the fp32 reference),
 *    DH_BF16 = throughput precision (bf16 storage, MFMA bf16, fp32 accumulate,
 *    fp32 statistics).  Parameters' gradients are always fp32.
 */
#ifndef DECLIP_HIP_H
#define DECLIP_HIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define DH_OK 0
#define DH_ERR_ARG -1
#define DH_ERR_LAUNCH -2
#define DH_ERR_UNSUPPORTED -3

#define DH_F32 0
#define DH_BF16 1

typedef void* dh_stream_t; /* hipStream_t */

const char* dh_last_error(void);
int dh_version(void);
/* device properties the host needs for grid sizing: out[0]=CUs, out[1]=clock kHz, out[2]=LDS bytes/CU, out[3]=gfx arch number */
int dh_device_info(int device, int* out4);
/* Ends a stream capture that was left open on `stream` (an invalidated capture that its owner gave up on: declip_amd/graph.py, the
 * fallback of a data-parallel step whose capture failed); 1 = a capture was open and is ended now, 0 = the stream was not capturing. */
int dh_stream_abandon_capture(dh_stream_t stream);

/* ---------------------------------------------------------------- GEMM ------------------
 * C[M,N] (+)= epi( alpha * sum_k A(m,k) * B(n,k) + bias[n] )
 *   a_kmajor = 0: A stored [M][K] (lda)      1: A stored [K][M] (lda)
 *   b_kmajor = 0: B stored [N][K] (ldb; the nn.Linear weight layout)   1: B stored [K][N] (ldb)
 * Replaces every dense contraction of the towers: conv1 patch-embed
 * (model/image_encoder/visual_transformer.py:14-15,56-59), MHA in/out projections and
 * the MLP (base_transformer.py:33-41,45-53), `x @ proj` (visual_transformer.py:72-73),
 * text_projection (text_encoder/text_transformer.py:203) and their autograd backward
 * (dX: b_kmajor=1; dW: a_kmajor=b_kmajor=1 with accumulate).
 * Epilogues:
 *   DH_EPI_NONE
 *   DH_EPI_GELU   C = QuickGELU(pre), aux_out[m,n] = QuickGELU'(pre)  (base_transformer.py:24-26; the derivative is what autograd's
 *                 backward of x * sigmoid(1.702 x) multiplies by -- both come from one sigmoid, round 5)
 *   DH_EPI_DGELU  C = value * aux_in[m,n]                              (backward of the above: aux_in = the forward's aux_out)
 *   residual != NULL: C += residual[m,n] (type c_dtype)               (base_transformer.py:51-52)
 *   accumulate = 1: C is fp32 and is atomically accumulated into (split_k > 1 allowed).
 *   accumulate = 2: the same product as the FIRST contribution to C in this step: C (and a_colsum) are overwritten -- the caller
 *                   need not have zeroed them, and the split-K reduce pass does not read them (torch's "grad is None -> assign").
 *   a_colsum (with a_kmajor): the weight-gradient call dW = dY^T X also emits db = colsum(dY) from the dY
 *   tiles it already staged in LDS (nn.Linear bias gradient without re-reading dY from HBM).
 */
#define DH_EPI_NONE 0
#define DH_EPI_GELU 1
#define DH_EPI_DGELU 2

typedef struct dh_gemm_args {
  int dtype;   /* element type of A, B, aux (DGELU) */
  int c_dtype; /* element type of C, residual, aux (GELU) */
  int a_kmajor, b_kmajor;
  int M, N, K;
  const void* A; int64_t lda;
  const void* B; int64_t ldb;
  void* C; int64_t ldc;
  const float* bias;     /* [N] fp32 or NULL */
  int epilogue;
  const void* residual; int64_t ldr; /* or NULL */
  void* aux; int64_t ldaux;          /* GELU: out, DGELU: in; or NULL */
  int accumulate;
  int split_k;           /* >=1; >1 requires accumulate */
  float alpha;
  int force_generic;     /* tests: 1 = VALU fp32-FMA kernel, 2 = v1 register-staged MFMA kernel, 3 = v2 LDS-DMA kernel, 4 = v4 256x256 ping-pong kernel (error if unsupported), 31 / 32 / 33 = prefer gemm_v3 with tile mode 1 / 2 / 3, 0 = auto */
  float* a_colsum;       /* optional, a_kmajor only: a_colsum[m] += sum_k A(m,k) (bias gradient fused into dW) */
  int pad_ok;            /* caller guarantees operand rows are readable (finite) up to the next multiple of 8
                            elements / 128 rows beyond M,N: lifts the M%8 / N%8 conditions of the MFMA kernels */
  void* ws; int64_t ws_bytes; /* optional caller scratch (16-B aligned).  With accumulate and split_k > 1 the v4 kernel writes fp32
                            partial tiles [split][M][N] here and adds them into C with one reduce pass instead of fp32
                            atomics from every split (needs split*M*N*4 bytes; ignored when too small or NULL) */
} dh_gemm_args;
int dh_gemm(const dh_gemm_args* args, dh_stream_t stream);
/* n weight-gradient problems with the SAME contraction length K -- args[i] with a_kmajor = b_kmajor = accumulate = 1, fp32 C --
 * as one launch: the dW GEMMs of one ResidualAttentionBlock (in_proj, out_proj, c_fc, c_proj of base_transformer.py:29-48, i.e. what
 * autograd computes as four separate `grad_output.t() @ input` matmuls).  On their own these problems have 9-36 output tiles for
 * 256 CUs; together they fill the chip with 2-7 K-slices instead of 7-64.  Uses args[0].ws / ws_bytes as the split-K workspace.
 * Same results as n dh_gemm calls (which is also what happens whenever the group does not fit the persistent kernel). */
int dh_gemm_group(const dh_gemm_args* args, int n, dh_stream_t stream);
/* Auto-dispatch switch for the 256 x 256 persistent kernel (on by default; DH_GEMM_V4=0 in the environment turns it off).
 * Returns the previous setting (-1 = default).  Used by the parity tests to run one model through both GEMM families. */
int dh_gemm_v4_enable(int on);
/* Tile distribution of the persistent kernel: 0 = static XCD-contiguous partition (one process per node), 1 = every workgroup's
 * items after its first come from per-XCD counters (multi-GPU ranks: RCCL kernels hold CUs during the overlapped gradient
 * all-reduce of utils/dist.py:63-88).  DH_V4_DYNAMIC in the environment is the default, read once.  Returns the previous setting. */
int dh_gemm_v4_set_dynamic(int mode);
/* dh_gemm launches per kernel family since the last reset: out5[0] persistent 256x256 (gemm_v4.hip), [1] gemm_v3.hip, [2] LDS-DMA
 * 128x128 (gemm_glds.hip), [3] MFMA-builtin tiles, [4] generic VALU kernel (gemm.hip).  Test instrumentation: lets a parity test
 * assert that a fixture ran on the kernel that bench.py measures.  Host counters, not thread-safe. */
int dh_gemm_stats(long long* out5, int reset);

/* out[n] (+)= sum_m X[m,n]  (fp32 out; bias gradients).  X: dtype, [M][N] with ldx. */
int dh_colsum(int dtype, const void* X, int64_t ldx, int M, int N, float* out, int accumulate, dh_stream_t stream);

/* ---------------------------------------------------------------- LayerNorm -------------
 * base_transformer.py:10-18 (nn.LayerNorm, eps 1e-5): y = (x-mean)*rstd*w + b.
 * mean/rstd [rows] fp32 are saved for backward.  w,b fp32.
 * bwd: dx = LN'(dy) (+ dres if non-NULL: the residual-branch gradient, fused add);
 *      dw[d], db[d] fp32 are ACCUMULATED into; ws: caller scratch of dh_layernorm_bwd_ws_bytes(). */
int dh_layernorm_fwd(int dtype, const void* x, const float* w, const float* b, void* y, float* mean, float* rstd,
                     int rows, int d, float eps, dh_stream_t stream);
int64_t dh_layernorm_bwd_ws_bytes(int rows, int d);
int dh_layernorm_bwd(int dtype, const void* dy, const void* x, const float* w, const float* mean, const float* rstd,
                     const void* dres, void* dx, float* dw, float* db, int rows, int d, void* ws, int64_t ws_bytes,
                     dh_stream_t stream);
/* The same with the reduction of the weight / bias gradient DEFERRED: the per-block partials stay in `part` ([*nb_out][2 d] fp32,
 * at least dh_layernorm_bwd_ws_bytes(rows, d) bytes, owned by the caller until the reduce ran) and dh_ln_reduce_many adds the
 * partials of up to 32 LayerNorms per launch into their dw / db.  A tower's backward (24 LayerNorms + ln_pre / ln_post of
 * base_transformer.py:29-53, visual_transformer.py:55-82) then issues ONE reduce instead of one per LayerNorm.  *nb_out = 0: this
 * shape was accumulated into dw / db directly (nothing to reduce). */
typedef struct dh_ln_part {
  const float* part; /* [nb][2 d] partials written by dh_layernorm_bwd_part */
  float* dw;         /* [d] accumulated into */
  float* db;         /* [d] accumulated into */
  int32_t nb, d;
} dh_ln_part;
int dh_layernorm_bwd_part(int dtype, const void* dy, const void* x, const float* w, const float* mean, const float* rstd,
                          const void* dres, void* dx, float* dw, float* db, int rows, int d, void* part, int64_t part_bytes,
                          int* nb_out, dh_stream_t stream);
int dh_ln_reduce_many(const dh_ln_part* items, int n, dh_stream_t stream);

/* ---------------------------------------------------------------- attention -------------
 * nn.MultiheadAttention(x,x,x, attn_mask) core (base_transformer.py:33,45-48; causal mask
 * text_encoder/text_transformer.py:136-142): per (batch, head)
 *   P = softmax(q k^T / sqrt(hd) [+ causal -inf]);  o = P v.
 * qkv: [b, L, 3*heads*hd] (q | k | v blocks, each heads*hd wide, head-major), out: [b, L, heads*hd],
 * lse: [b, heads, L] fp32 (log-sum-exp of the scaled scores; saved for backward).
 * hd must be 64 for DH_BF16 (MFMA path); L <= 128.  The fp32 validation kernels keep the [L, L] matrices in LDS as well:
 * forward L <= 126, backward L <= 91 at hd = 64 (longer sequences return DH_ERR_ARG, nothing is launched).
 * bwd recomputes P from q,k,lse: dqkv [b, L, 3*heads*hd]. */
int dh_attn_fwd(int dtype, const void* qkv, void* out, float* lse, int b, int L, int heads, int hd, int causal,
                dh_stream_t stream);
int dh_attn_bwd(int dtype, const void* qkv, const void* out, const void* dout, const float* lse, void* dqkv, int b,
                int L, int heads, int hd, int causal, dh_stream_t stream);
/* The same on PACKED (variable-length) sequences: pair (bi, h) owns rows cu_seqlens[bi] .. cu_seqlens[bi+1] of qkv / out / dout /
 * dqkv ([rows][3*d] / [rows][d]; cu_seqlens int32 [b + 1] in device memory, every length <= Lmax); lse stays [b][heads][Lmax].
 * rows = cu_seqlens[b] (the caller knows it on the host), rows_pad >= rows = the allocated row count (whole GEMM tiles): rows
 * [rows, rows_pad) of out / dqkv are written as ZEROS -- they are contraction rows of the weight-gradient GEMMs -- by the
 * attention kernel itself on the bf16 path (no separate fill launch).  rows = -1 (bf16, hd = 64 only): the kernel reads
 * cu_seqlens[b] itself -- no host-side row count in the launch, so a captured step replays for ANY batch with this rows_pad.
 * Used by the packed text tower (captions computed up to <|endoftext|> only; DESIGN_HISTORY.md s11). */
int dh_attn_varlen_fwd(int dtype, const void* qkv, void* out, float* lse, const int* cu_seqlens, int b, int Lmax, int heads, int hd,
                       int causal, int rows, int rows_pad, dh_stream_t stream);
int dh_attn_varlen_bwd(int dtype, const void* qkv, const void* out, const void* dout, const float* lse, void* dqkv,
                       const int* cu_seqlens, int b, int Lmax, int heads, int hd, int causal, int rows, int rows_pad, dh_stream_t stream);
/* The same in two LENGTH BUCKETS (bf16, hd = 64): sequences order[0 .. n_short) are at most L_short rows long and run on the kernel
 * instantiation with ceil(L_short / 16) key blocks, the rest on the Lmax one -- a 20-token caption no longer occupies the workgroup
 * shape of a 77-token one (text_transformer.py:136-142 pads every caption to the context length).  order int32 [b] = a permutation of
 * the sequences, short ones first; ranges int32 [4] = {0, n_short, n_short, b - n_short}; both in device memory (the packed batch's
 * bookkeeping), like rows = -1: nothing of a launch depends on the batch beyond rows_pad. */
int dh_attn_bucketed_fwd(int dtype, const void* qkv, void* out, float* lse, const int* cu_seqlens, const int* order, const int* ranges,
                         int b, int Lmax, int L_short, int heads, int hd, int causal, int rows, int rows_pad, dh_stream_t stream);
int dh_attn_bucketed_bwd(int dtype, const void* qkv, const void* out, const void* dout, const float* lse, void* dqkv,
                         const int* cu_seqlens, const int* order, const int* ranges, int b, int Lmax, int L_short, int heads, int hd,
                         int causal, int rows, int rows_pad, dh_stream_t stream);
/* Pooled-query attention for the LAST block of a tower: only the pooled row's output is used downstream (CLS,
 * image_encoder/visual_transformer.py:70-72; <|endoftext|>, text_encoder/text_transformer.py:203), so that block's query projection,
 * attention, out_proj and MLP are needed for b rows, not b*L (K, V still come from every row).  q [b][d] (the pooled rows'
 * queries), kv [rows][2*d] (k | v, head-major, of every row); sequence i's keys are kv rows row0[i] .. row0[i] + nkeys[i] - 1
 * (row0, nkeys int32 [b] in device memory; nkeys = position + 1 is the causal mask of the text tower, nkeys <= Lmax <= 128);
 * hd == 64.  out [b][d], lse [b][heads].  bwd writes dq [b][d] and the dkv rows of every sequence's keys; total_rows > 0
 * (row0 ascending, kv / dkv have total_rows rows): the rows no sequence owns are zeroed by the same launch, dkv may be uninitialised;
 * total_rows = 0: those rows stay as the caller initialised them. */
int dh_attn_pooled_fwd(int dtype, const void* q, const void* kv, void* out, float* lse, const int* row0, const int* nkeys, int b,
                       int heads, int hd, int Lmax, dh_stream_t stream);
int dh_attn_pooled_bwd(int dtype, const void* q, const void* kv, const void* dout, const float* lse, void* dq, void* dkv,
                       const int* row0, const int* nkeys, int b, int heads, int hd, int Lmax, int total_rows, dh_stream_t stream);

/* ---------------------------------------------------------------- one transformer block per call
 * ResidualAttentionBlock (base_transformer.py:29-53): x_mid = x + out_proj(attn(in_proj(ln_1(x)))), x_out = x_mid +
 * c_proj(quick_gelu(c_fc(ln_2(x_mid)))), and its backward -- enqueued by ONE call: 9 launches forward, 13 backward (the four
 * weight gradients as one dh_gemm_group).  Same kernels and same results as the per-op entry points above; what goes away is one
 * host round trip per kernel (csrc/block.hip).
 *   rows x d activations of `dtype`; dense attention over b sequences of L rows (rows == b * L), or -- cu != NULL -- packed
 *   sequences rows cu[i] .. cu[i+1] (int32 [b + 1], device memory), rows_valid = cu[b] or -1 (read it on the device, see
 *   dh_attn_varlen_fwd), L = the longest allowed sequence.
 *   p: weights in `dtype` ([3d,d], [d,d], [4d,d], [d,4d], the nn.Linear layouts), biases / LayerNorm parameters fp32; g_*: fp32
 *   gradient slots, ACCUMULATED into (backward only).
 *   act: caller-owned slab of dh_block_act_bytes() bytes, written by forward, read by backward.  save = 0 (no backward will
 *   follow): the GELU pre-activation is not kept.
 *   backward: dx_out -> dx; scratch: dh_block_bwd_scratch_bytes() bytes of temporaries; ln_part1 / ln_part2: one slice of
 *   dh_layernorm_bwd_ws_bytes(rows, d) bytes each for the deferred LayerNorm weight / bias reductions of ln_1 / ln_2, alive
 *   until dh_ln_reduce_many ran on {ln_part, ln_nb, d, g_ln_w, g_ln_b}; ln_nb1 / ln_nb2 are written by the call.
 *   ws: the split-K / sliced-tile workspace of dh_gemm (bf16 only; may be NULL). */
typedef struct dh_block_params {
  const void *w_in, *w_out, *w_fc, *w_proj;
  const float *b_in, *b_out, *b_fc, *b_proj, *ln1_w, *ln1_b, *ln2_w, *ln2_b;
  float *g_w_in, *g_w_out, *g_w_fc, *g_w_proj, *g_b_in, *g_b_out, *g_b_fc, *g_b_proj, *g_ln1_w, *g_ln1_b, *g_ln2_w, *g_ln2_b;
  float eps1, eps2;
} dh_block_params;
typedef struct dh_block_args {
  int dtype, rows, d, heads, b, L, causal, save;
  const int* cu; int rows_valid;
  const int* seq_order; const int* seq_ranges; int L_short;   /* packed sequences in two length buckets (dh_attn_bucketed_fwd), or NULL */
  dh_block_params p;
  const void* x; void* x_out;
  void* act; int64_t act_bytes;
  void* ws; int64_t ws_bytes;
  const void* dx_out; void* dx;
  void* scratch; int64_t scratch_bytes;
  void* ln_part1; void* ln_part2; int64_t ln_part_bytes;
  int ln_nb1, ln_nb2;
  int dw_first_touch;      /* dh_block_bwd: the block's weight / bias gradients are the first contribution of this step to their slots: written
                              (dh_gemm_args.accumulate = 2), the caller need not have zeroed the slots */
} dh_block_args;
/* base[lo, hi) = 0 for the n ranges of table_dev ([n][2] int64 element offsets on the device, multiples of 4; max_len = the longest hi - lo):
 * the flat gradient buffer minus the slots whose first writer of the step is a weight-gradient GEMM with accumulate = 2 (csrc/fill.hip). */
int dh_zero_ranges(float* base, const int64_t* table_dev, int n, int64_t max_len, dh_stream_t stream);
int64_t dh_block_act_bytes(int dtype, int rows, int d, int heads, int b, int L);
/* byte offsets inside the slab of h1, qkv, a (attention output), x_mid, h2, u (GELU pre-activation), g, mean1, rstd1, mean2, rstd2, lse */
int dh_block_act_offsets(int dtype, int rows, int d, int heads, int b, int L, int64_t* out12);
int64_t dh_block_bwd_scratch_bytes(int dtype, int rows, int d);
int dh_block_fwd(const dh_block_args* args, dh_stream_t stream);
int dh_block_bwd(dh_block_args* args, dh_stream_t stream);

/* ---------------------------------------------------------------- embeddings ------------
 * Text: x[b,l,:] = table[ids[b,l],:] + pos[l,:]   (text_transformer.py:188-190); table/pos fp32.
 * bwd: dtable (fp32, atomic accumulate) and dpos (fp32, atomic accumulate).  hot_ids_host (<= 4, HOST array):
 * ids present in (almost) every caption -- pad 0, SOT, EOT -- which are block-reduced instead of serialising
 * thousands of atomics on one table row. */
int dh_text_embed_fwd(int dtype, const int64_t* ids, const float* table, const float* pos, void* x, int b, int L, int d,
                      dh_stream_t stream);
int dh_text_embed_bwd(int dtype, const int64_t* ids, const void* dx, float* dtable, float* dpos, int b, int L, int d,
                      const int64_t* hot_ids_host, int n_hot, dh_stream_t stream);
/* The token-table part of it as a sort-by-id segmented reduction: rows counting-sorted by id (one integer atomic per ROW), runs of
 * equal ids summed in registers by one wave per 16 sorted rows and added to dtable [vocab][d] with plain 16-byte stores; only
 * runs longer than a wave's chunk (frequent tokens) flush with float atomics, once per 16 rows.  ids outside [0, vocab) are
 * skipped.  ws: caller scratch of dh_embed_table_grad_ws_bytes(rows, vocab).  d % 8 == 0. */
int64_t dh_embed_table_grad_ws_bytes(int rows, int vocab);
int dh_embed_table_grad(int dtype, const int64_t* ids, const void* dx, float* dtable, int rows, int d, int vocab, void* ws,
                        int64_t ws_bytes, dh_stream_t stream);
/* Packed (variable-length) captions.  Under the causal mask a token never attends to a later one and only the <|endoftext|>
 * row is pooled (text_encoder/text_transformer.py:136-142,203), so the rows after EOT of the reference's [b][77] layout are dead
 * work; the packed text tower keeps only the rows up to and including EOT: [rows = sum len_i][d], zero rows up to rows_pad
 * (whole GEMM tiles).  x[r] = table[ids_p[r]] + pos[pos_idx[r]].  Token-table gradient: dh_text_embed_bwd on the packed ids
 * (b = rows, L = 1, dpos NULL); positional gradient: dh_packed_pos_grad, dpos[p] += sum_{i: len_i > p} dx[cu[i] + p]
 * (cu_seqlens int32 [b + 1], device). */
int dh_text_embed_packed_fwd(int dtype, const int64_t* ids_p, const int* pos_idx, const float* table, const float* pos, void* x,
                             int rows, int rows_pad, int d, dh_stream_t stream);
int dh_packed_pos_grad(int dtype, const void* dx, const int* cu_seqlens, int b, int Lmax, int d, float* dpos, dh_stream_t stream);
/* Vision: im2row of the stride-P patch conv (visual_transformer.py:14-15,56-59):
 * images [b,3,H,W] fp32 (channel offset c0 of C_total channels, for channel-stacked views,
 * data/transforms.py:38-41) -> rows [b*gh*gw, 3*P*P] (dtype), inner order (c,ph,pw). */
int dh_im2row(int dtype, const float* images, int c_total, int c0, void* rows, int b, int H, int W, int P,
              dh_stream_t stream);
/* uint8 HWC images [b][src_h][src_w][3] -> fp32 CHW channels c0..c0+2 of dst [b][c_total][H][W]:
 * dst = (src / 255 - mean[c]) / std[c] over the window at crop_xy_dev[b][2] = (x0, y0) (NULL: src is H x W), mirrored
 * horizontally where flip_dev[b] != 0 (NULL: never).  ToTensor + Normalize + crop + flip of the reference's input
 * pipelines (data/transforms.py, data/nvidia_dali_dataloader.py crop_mirror_normalize); resizing stays with the
 * decoder.  mean3 / std3 are HOST pointers; the crop window must lie inside the source (caller's contract). */
int dh_image_prep_u8(const uint8_t* src, int b, int src_h, int src_w, const int* crop_xy_dev, const uint8_t* flip_dev,
                     const float* mean3, const float* std3, float* dst, int c_total, int c0, int H, int W,
                     dh_stream_t stream);
/* RandomResizedCrop / Resize + CenterCrop on the GPU for decoded uint8 HWC images (what the reference's CPU workers / DALI do:
 * data/imagenet_dataloader.py:36-47 `STANDARD_SLIP` RandomResizedCrop(224, scale=(0.5, 1)), :105-111 `ONECROP` Resize(256) +
 * CenterCrop(224); torchvision on PIL images = Image.crop + Image.resize(BILINEAR): antialiased triangle filter with support
 * max(scale, 1), taps clamped to the crop box), fused with mirror, ToTensor and Normalize.  The images of a batch share one
 * [b][src_h][src_w][3] canvas (each in its top-left corner).  params_dev int32 [b][8] in DEVICE memory:
 *   x0, y0, w, h : the crop box in the source;   Wf, Hf : the size the box is resized to;
 *   ox, oy       : where the H x W output window sits inside that Wf x Hf image
 * (RandomResizedCrop: Wf = W, Hf = H, ox = oy = 0; Resize(s) + CenterCrop: box = whole image, Wf x Hf = the resized image,
 * ox, oy = the centre window).  round_u8 != 0 rounds the resized value to an integer grey level first, as the uint8 PIL image
 * between resize and ToTensor does.  dst as in dh_image_prep_u8. */
int dh_image_resized_crop_u8(const uint8_t* src, int b, int src_h, int src_w, const int* params_dev, const uint8_t* flip_dev,
                             const float* mean3, const float* std3, float* dst, int c_total, int c0, int H, int W, int round_u8,
                             dh_stream_t stream);
/* x[b,0,:] = cls + pos[0]; x[b,1+p,:] = patches[b,p,:] + pos[1+p]  (visual_transformer.py:60-62) */
int dh_vit_assemble_fwd(int dtype, const void* patches, const float* cls, const float* pos, void* x, int b, int np,
                        int d, dh_stream_t stream);
/* dcls[d] += sum_b dx[b,0,:]; dpos[1+np,d] += sum_b dx[b,:,:] (fp32 atomics) */
int dh_vit_assemble_bwd(int dtype, const void* dx, float* dcls, float* dpos, int b, int np, int d, dh_stream_t stream);
/* out[i,:] = x[i, idx[i], :] (idx NULL -> position 0): CLS / EOT pooling
 * (visual_transformer.py:66; text_transformer.py:203).  bwd: dx zero-filled then scattered. */
int dh_pool_rows_fwd(int dtype, const void* x, const int64_t* idx, void* out, int b, int L, int d, dh_stream_t stream);
int dh_pool_rows_bwd(int dtype, const void* dout, const int64_t* idx, void* dx, int b, int L, int d, dh_stream_t stream);

/* ---------------------------------------------------------------- features / loss -------
 * y = x / (||x|| + eps) rows of [rows,d], x of `dtype`, y fp32 (model/clip.py:129-130: eps 0 image, 1e-10 text).
 * bwd: dx (dtype) from dy (fp32). */
int dh_l2norm_fwd(int dtype, const void* x, float* y, float* norm, int rows, int d, float eps, dh_stream_t stream);
int dh_l2norm_bwd(int dtype, const void* x, const float* norm, const float* dy, void* dx, int rows, int d, float eps,
                  dh_stream_t stream);

/* Fused contrastive loss (model/clip.py:140-141 + loss_functions/loss.py:37-47 + utils/misc.py:415-428):
 * for each pair p: logits = scale * Q_p[b,D] . K_p[B,D]^T is streamed tile by tile (never written to HBM),
 * row-wise online log-sum-exp; label of local row i is label0 + i.
 *   row_loss[p,i] = lse_i - logit_{i,label},  row_lse[p,i], correct1/correct5 [p,i] (0/1: top-1/top-5 hit)
 * Q,K fp32 (normalised features); scale_dev points to ONE fp32 on the device (exp(logit_scale), clip.py:133-134)
 * so the step needs no host sync.  logits_out (optional, may be NULL) [p,b,B] fp32 for the API surface.
 * bwd: given g_row[p,i] = dLoss/d(row_loss[p,i]):
 *   dQ_p[i,:]  = scale * sum_j g_i (softmax_ij - 1[j=label]) K_p[j,:]
 *   dK_p[j,:] += scale * sum_i g_i (softmax_ij - 1[j=label]) Q_p[i,:]   (fp32 atomics; zero it first)
 *   dscale    += sum_ij g_i (softmax_ij - 1[j=label]) <Q_i,K_j>         (fp32 atomic) */
typedef struct dh_nce_pair {
  const float* Q; const float* K;   /* [b,D], [B,D] */
  float* dQ; float* dK;             /* bwd outputs (may be NULL in fwd) */
} dh_nce_pair;
int64_t dh_infonce_ws_bytes(int n_pairs, int b, int B);
/* label0s / excl0s (HOST int arrays of n_pairs, or NULL): per-pair label offset, and per-pair offset of a column
 * removed from row i's softmax (excl0 + i; -1 = none) -- the self-pair of SimCLR NT-Xent
 * (loss_functions/nt_xent.py:62-97: positives at label0+i, the row's own entry excluded). */
int dh_infonce_fwd(const dh_nce_pair* pairs_host, int n_pairs, int b, int B, int D, const float* scale_dev, int label0,
                   const int* label0s, const int* excl0s, float* row_loss, float* row_lse, float* correct1,
                   float* correct5, float* logits_out, void* ws, int64_t ws_bytes, dh_stream_t stream);
int dh_infonce_bwd(const dh_nce_pair* pairs_host, int n_pairs, int b, int B, int D, const float* scale_dev, int label0,
                   const int* label0s, const int* excl0s, const float* row_lse, const float* g_row, float* dscale,
                   dh_stream_t stream);

/* Linear(K -> V) + row-wise cross-entropy in ONE pass over the vocabulary, bf16 operands: the masked-LM head
 * `text_label_predictor(word_features)` + `F.cross_entropy(pred[mask], labels[mask])` of declip.py:326-334 without the fp32 logits
 * [n_masked, 49409] (1.2 GB per DeCLIP step at b = 512) ever reaching HBM.  X [n_pad][K] (rows >= n zero; n_pad % 256 == 0),
 * W [V][K], bias [V] or NULL, labels [n].  Forward: row_loss / row_lse [n] (per-tile max / sum-exp partials in ws, merged by a
 * finalize kernel).  Backward: dl [n_pad][ldd] bf16 = g_row * (softmax - onehot), recomputed tile by tile (zero for rows >= n
 * and columns >= V) -- the operand of the two gradient GEMMs (dh_gemm).  Both fail (no fallback) on shapes the persistent kernel
 * does not take. */
int64_t dh_ce_fused_ws_bytes(int n_pad, int V);
int dh_ce_fused_fwd(const void* X_bf16, const void* W_bf16, const float* bias, const int64_t* labels, int n, int n_pad, int V, int K,
                    float* row_loss, float* row_lse, void* ws, int64_t ws_bytes, dh_stream_t stream);
int dh_ce_fused_bwd(const void* X_bf16, const void* W_bf16, const float* bias, const int64_t* labels, const float* row_lse,
                    const float* g_row, int n, int n_pad, int V, int K, void* dl_bf16, int64_t ldd, dh_stream_t stream);
/* Row-wise softmax cross-entropy on MATERIALISED fp32 logits [rows,C] (leading dim ld): the form
 * loss_functions/loss.py:44-45 sees when handed tensors, and the MLM head CE (model/declip.py:326-334).
 * Rows whose label is outside [0,C) (e.g. -100, mask_tokens.py:17) give loss 0 / zero gradient. */
int dh_ce_rows_fwd(const float* logits, int64_t ld, const int64_t* labels, int rows, int C, float* row_loss,
                   float* row_lse, float* correct1, float* correct5, dh_stream_t stream);
int dh_ce_rows_bwd(const float* logits, int64_t ld, const int64_t* labels, int rows, int C, const float* row_lse,
                   const float* g_row, float* dlogits, int64_t ldd, dh_stream_t stream);

/* As dh_ce_rows_bwd, writing dlogits [rows_pad][ldd] as `out_dtype` with zeros outside [rows) x [C): the MLM
 * head's gradient in the padded layout its MFMA GEMMs consume (vocab 49409 padded to a multiple of 64). */
int dh_ce_rows_bwd_padded(const float* logits, int64_t ld, const int64_t* labels, int rows, int C, const float* row_lse,
                          const float* g_row, void* dlogits, int out_dtype, int64_t ldd, int rows_pad, int C_pad,
                          dh_stream_t stream);

/* ---------------------------------------------------------------- DeCLIP / SLIP heads ---
 * BatchNorm1d (+ optional ReLU) with batch statistics per GROUP of rows (the reference applies the SimSiam
 * projector / predictor once per view: model/declip.py:33-130,238-241; plain per-rank nn.BatchNorm1d).
 * x,y [groups*rows_per_group, C]; save_mean/save_invstd [groups, C]; running stats updated group after group
 * (momentum, unbiased variance) exactly like successive module calls.  training=0 uses the running stats. */
int dh_bn1d_fwd(int dtype, const void* x, const float* w, const float* b, void* y, float* save_mean, float* save_invstd,
                float* running_mean, float* running_var, int groups, int rows_per_group, int C, float eps, float momentum,
                int relu, int training, dh_stream_t stream);
int dh_bn1d_bwd(int dtype, const void* dy, const void* x, const void* y, const float* w, const float* save_mean,
                const float* save_invstd, void* dx, float* dw, float* db, int groups, int rows_per_group, int C, int relu,
                dh_stream_t stream);
/* cos[r] = <p_r,z_r>/(|p_r||z_r|)  (SimSiam D(p, stopgrad z), loss_functions/loss.py:49-55); bwd w.r.t. p only. */
int dh_cos_rows_fwd(int dtype, const void* p, const void* z, float* cosv, int rows, int d, dh_stream_t stream);
int dh_cos_rows_bwd(int dtype, const void* p, const void* z, const float* g_row, void* dp, int rows, int d,
                    dh_stream_t stream);
/* Nearest neighbour of every query row in the feature bank (model/utils/nnclr_modules/nn_memory_bank.py:42-65,
 * topk=1): bank [size][D] fp32 resident in HBM (the reference keeps it on the CPU and re-uploads 128 MiB per
 * call); exact fp32 dot products, first maximum wins.  idx_out [rows] int64, feat_out [rows][D] = bank[idx]. */
int64_t dh_nn_bank_ws_bytes(int rows, int size);
int dh_nn_bank_query(const float* q, const float* bank, int rows, int size, int D, int64_t* idx_out, float* feat_out,
                     void* ws, int64_t ws_bytes, dh_stream_t stream);
/* FIFO enqueue of the queue (nnclr_modules/memory_bank.py:71-87): store [size + spill][D] fp32 = the bank followed by `spill`
 * rows that are never searched; rows *ptr .. *ptr + b - 1 receive `batch` [b][D] (the reference drops the part of a batch that
 * would run over the end: here it lands in the spill rows, spill >= b), then *ptr = 0 if *ptr + b >= size else *ptr + b.  `ptr`
 * is int64 [1] in DEVICE memory: no host value enters a launch, a captured step advances the queue on every replay. */
int dh_nn_bank_enqueue(float* store, int64_t* ptr, const float* batch, int b, int size, int spill, int D, dh_stream_t stream);
/* out[r,:] = x[idx[r],:] for r < n, zero rows for n <= r < n_pad (masked-LM rows, model/declip.py:326-334);
 * scatter_rows_add: dx[idx[r],:] += dout[r,:] (unique indices). */
int dh_gather_rows(int dtype, const void* x, const int64_t* idx, void* out, int n, int n_pad, int d, dh_stream_t stream);
int dh_scatter_rows_add(int dtype, const void* dout, const int64_t* idx, void* dx, int n, int d, dh_stream_t stream);

/* ---------------------------------------------------------------- FILIP (model/filip.py:71-106) ---
 * dh_filip_select: per sample the 16 image tokens with the largest summed similarity to the caption's tokens and
 *   the 16 text tokens with the largest summed similarity to the image's tokens (filip.py:78-87);
 *   img_tok [b,J,D], txt_tok [b,T,D] fp32 L2-normalised; idx_* [b,16] int64 (token index inside the sample).
 * dh_maxsim_reduce: S [b*J, B*16] = token-similarity GEMM output (s_dtype) ->
 *   raw[i,l] = mean_j max_m S[(i,j),(l,m)], logits = scale * raw, argmax [b*J, B] uint8 (filip.py:96-105).
 * dh_maxsim_scatter: backward of the reduction: G[(i,j),(l,m)] = scale * dlogits[i,l] / J at m = argmax, else 0. */
int dh_filip_select(const float* img_tok, const float* txt_tok, int b, int J, int T, int D, int64_t* idx_img,
                    int64_t* idx_txt, dh_stream_t stream);
int dh_maxsim_reduce(int s_dtype, const void* S, int64_t lds, int b, int B, int J, const float* scale_dev, float* logits,
                     float* raw, uint8_t* argmax, dh_stream_t stream);
/* Fused forward for bf16 token features: the token-similarity GEMM on the persistent MFMA kernel with max_m / mean_j in its
 * epilogue -- S [b*J, B*16] is never written (filip.py:96-105 materialises it as [b, B, J, 16]).  Q [rows_pad][D] (rows_pad =
 * b*J rounded up to 256; padding rows anything finite), K [B*16][D], both bf16 row-major.  Writes raw [b][B] (unscaled means),
 * logits = raw * *scale_dev, argmax uint8 [rows_pad][B].  Fails (no fallback) unless rows_pad % 256 == 0, B % 16 == 0,
 * D % 64 == 0, D >= 128, J >= 19. */
int dh_maxsim_fused_fwd(const void* Q_bf16, const void* K_bf16, int rows_pad, int b, int B, int J, int D, const float* scale_dev,
                        float* logits, float* raw, uint8_t* argmax, dh_stream_t stream);
/* Rows [r0, r0 + nrows) of G (dh_maxsim_scatter) into a chunk buffer Gc [nrows][ldg]: the backward walks G in row chunks through
 * one small buffer (dQ_chunk = Gc K, dK += Gc^T Q_chunk) instead of allocating [b*J, B*16]; rows >= b*J are written as zeros. */
int dh_maxsim_scatter_rows(int g_dtype, const float* dlogits, const uint8_t* argmax, const float* scale_dev, void* Gc, int64_t ldg,
                           int b, int B, int J, int64_t r0, int64_t nrows, dh_stream_t stream);
int dh_maxsim_scatter(int g_dtype, const float* dlogits, const uint8_t* argmax, const float* scale_dev, void* G,
                      int64_t ldg, int b, int B, int J, dh_stream_t stream);

/* ---------------------------------------------------------------- optimizer / casts -----
 * Fused flat AdamW over a contiguous fp32 range (torch.optim.AdamW semantics, the optimizer of
 * experiments/clip_experiments/yfcc15m/yfcc15m_vit_clip/config.yaml:26-33), optionally refreshing the
 * bf16 mirror of the parameters.  step >= 1. */
int dh_adamw(float* p, const float* g, float* m, float* v, void* p_bf16_or_null, int64_t n, float lr, float beta1,
             float beta2, float eps, float weight_decay, int step, float grad_scale, dh_stream_t stream);
/* Same update with per-segment (lr, weight_decay): the parameter groups of utils/misc.py:267-412 laid out
 * over the flat buffer; seg_start[nseg] ascending element offsets (multiples of 4) in DEVICE memory.
 * Segments with seg_lr < 0 are left untouched (parameters the optimizer does not own, frozen or gradient-less ones: what
 * torch.optim.AdamW skips); lr == 0 is an ordinary value (the moments still move, as in torch). */
int dh_adamw_segmented(float* p, const float* g, float* m, float* v, void* p_bf16_or_null, int64_t n,
                       const int64_t* seg_start_dev, const float* seg_lr_dev, const float* seg_wd_dev, int nseg,
                       float beta1, float beta2, float eps, int step, float grad_scale, dh_stream_t stream);
int dh_cast(int src_dtype, const void* src, int dst_dtype, void* dst, int64_t n, dh_stream_t stream);

/* ---------------------------------------------------------------- ModifiedResNet tower ---
 * prototype/model/image_encoder/modified_resnet.py (BASELINE configs[0]: clip_res50).  Activations are NHWC, i.e.
 * row-major [N*H*W][C] "pixel rows" (the token-major layout of the transformer towers): a 1x1 convolution is dh_gemm
 * on the activation as stored, a 3x3 convolution is dh_conv_rows + dh_gemm against conv.weight viewed [Cout][Cin*9]
 * as stored (nn.Conv2d of modified_resnet.py:20-29,144-149; bias-free).  C must be a multiple of 8 everywhere.
 *
 * dh_conv_rows: rows[(n,oy,ox)][c*9 + ky*3 + kx] = src[n, oy*stride-pad+ky, ox*stride-pad+kx, c], 0 outside; k == 3.
 *   src_layout 0: NHWC activation of `dtype`, Kpad == 9*C.
 *   src_layout 1: fp32 NCHW image batch [N][c_total][H][W], the 3 channels from c0 (a channel-stacked view,
 *                 data/transforms.py:38-54), rows of `dtype` with K = 27 zero-padded to Kpad == 32 (stem conv1). */
int dh_conv_rows(int dtype, const void* src, int src_layout, int c_total, int c0, void* rows, int N, int H, int W, int C, int k,
                 int stride, int pad, int Kpad, dh_stream_t stream);
/* nn.BatchNorm2d (modified_resnet.py:138, use_sync_bn False: per-rank statistics) on [rows][C], fused with the ReLU and
 * the residual add that follow it in Bottleneck.forward (modified_resnet.py:44-56): y = relu?(bn(x) (+ residual)).
 * training: batch mean / biased variance (saved as save_mean / save_invstd [C]); running_mean / running_var (may both be
 * NULL) move by `momentum` towards the batch mean / unbiased variance.  training == 0: running statistics.
 * Deterministic two-level reductions through the caller's workspace (dh_bn2d_ws_bytes). */
int64_t dh_bn2d_ws_bytes(int rows, int C);
int dh_bn2d_fwd(int dtype, const void* x, const void* residual_or_null, const float* w, const float* b, void* y, float* save_mean,
                float* save_invstd, float* running_mean, float* running_var, int rows, int C, float eps, float momentum, int relu,
                int training, void* ws, int64_t ws_bytes, dh_stream_t stream);
/* Backward of the same: dyr = dy masked by (y > 0) when relu; dx = w*invstd*(dyr - mean(dyr) - xhat*mean(dyr*xhat));
 * dres_or_null receives dyr (the gradient of the residual branch); dw += sum dyr*xhat, db += sum dyr (fp32, accumulate). */
int dh_bn2d_bwd(int dtype, const void* dy, const void* x, const void* y_or_null, const float* w, const float* save_mean,
                const float* save_invstd, void* dx, void* dres_or_null, float* dw, float* db, int rows, int C, int relu, void* ws,
                int64_t ws_bytes, dh_stream_t stream);
/* The same in three stages for BatchNorm synchronised across ranks (`use_sync_bn: True`, modified_resnet.py:118-140;
 * torch.nn.SyncBatchNorm semantics): per-channel SUMS are what ranks exchange -- ONE SUM all-reduce of `sums`
 * ([2*C + 1] doubles: mode 0 (sum x, sum x^2), mode 1 (sum dyr, sum dyr*xhat); last element = row count, kept in device
 * memory so that nothing returns to the host between the collective and the apply pass).
 *   forward : dh_bn2d_sums(mode 0) -> all-reduce(sums) -> dh_bn2d_fwd_apply (mean / invstd / running stats from the global
 *             sums, then y = relu?(bn(x) (+ residual)));
 *   backward: dh_bn2d_sums(mode 1) -> copy, all-reduce -> dh_bn2d_bwd_apply (dw, db += LOCAL sums -- the gradient all-reduce
 *             sums them over ranks --, the two means of the dx formula from the GLOBAL sums). */
int dh_bn2d_sums(int dtype, int mode, const void* x, const void* dy_or_null, const void* y_or_null, const float* save_mean_or_null,
                 const float* save_invstd_or_null, int relu, int rows, int C, double* sums, void* ws, int64_t ws_bytes, dh_stream_t stream);
int dh_bn2d_fwd_apply(int dtype, const void* x, const void* residual_or_null, const float* w, const float* b, const double* sums, void* y,
                      float* save_mean, float* save_invstd, float* running_mean, float* running_var, int rows, int C, float eps,
                      float momentum, int relu, dh_stream_t stream);
int dh_bn2d_bwd_apply(int dtype, const void* dy, const void* x, const void* y_or_null, const float* w, const float* save_mean,
                      const float* save_invstd, const double* sums_local, const double* sums_global, void* dx, void* dres_or_null,
                      float* dw, float* db, int rows, int C, int relu, void* ws /* >= 2*C floats */, int64_t ws_bytes, dh_stream_t stream);
/* nn.AvgPool2d(k) on NHWC (modified_resnet.py:26,36,149); H and W multiples of k. */
int dh_avgpool_fwd(int dtype, const void* x, void* y, int N, int H, int W, int C, int k, dh_stream_t stream);
int dh_avgpool_bwd(int dtype, const void* dy, void* dx, int N, int H, int W, int C, int k, dh_stream_t stream);
/* AttentionPool2d token matrix (modified_resnet.py:70-72): tok [b][HW+1][C], tok[:,0] = mean over the HW pixel rows,
 * tok[:,1:] = the pixel rows, plus positional_embedding [HW+1][C] (fp32).  bwd: dx [b*HW][C], dpos += sum_b dtok. */
int dh_attnpool_tokens_fwd(int dtype, const void* x, const float* pos, void* tok, int b, int HW, int C, dh_stream_t stream);
int dh_attnpool_tokens_bwd(int dtype, const void* dtok, void* dx, float* dpos_or_null, int b, int HW, int C, dh_stream_t stream);

/* ---- host side: caption tokeniser (no device work) ----------------------------------------------
 * Byte-level BPE of model/utils/text_utils/simple_tokenizer.py:62-134 + the [SOT] ids [EOT] / zero-pad /
 * truncate-keeping-EOT layout of text_encoder/text_transformer.py:144-180, batch-parallel on host threads.
 * `merges` = decompressed text of the vocabulary file (header line + "left right" per line), the first n_merges
 * lines are used (49152 - 256 - 2 in the reference).  Captions must already be cleaned and lower-cased
 * (ftfy / html.unescape / whitespace collapse are the caller's); a caption containing a non-ASCII byte is not
 * tokenised (status 1, row zeroed): the caller runs its Unicode-aware tokeniser for that row.
 * dh_bpe_create returns NULL on error (dh_last_error()).  Thread-safe per handle. */
void* dh_bpe_create(const char* merges, int64_t nbytes, int n_merges);
void dh_bpe_destroy(void* handle);
int dh_bpe_vocab_size(void* handle);
int dh_bpe_encode(void* handle, const char* texts, const int64_t* offsets /* n + 1 */, int n, int ctx,
                  int64_t* out /* [n][ctx] */, int32_t* status /* [n] */, int n_threads);

/* ---- communicator context: the collectives of the data-parallel step on RCCL / xGMI -----------------
 * Replaces the `linklink` calls of the step (linklink/__init__.py:13-71): AllGather.forward/backward (model/clip.py:25-49:
 * one all_gather per feature tensor, backward = all-reduce of the whole gathered gradient + slice) and DistModule's
 * per-parameter gradient all-reduce hooks (utils/dist.py:49-88).  A context owns the RCCL communicator, a communication stream
 * and the events ordering it against the caller's streams.  Every call ENQUEUES only: the communication stream first waits for
 * what `stream` (the producer of the operands) holds at the time of the call; consumers are ordered with dh_comm_wait().
 * RCCL is bound at run time (dlopen) by dh_comm_unique_id / dh_init; any RCCL error fails the call (dh_last_error()).
 *   dh_comm_unique_id   rank 0: the 128-byte ncclUniqueId every rank passes to dh_init (exchanged by the launcher's own means)
 *   dh_init             binds `local_rank`'s device, creates the communicator (collective over all ranks); NULL on error
 *   dh_allgather_packed n tensors [rows][cols_k] (elem_bytes 2 | 4, rows of 16-byte multiples) -> gathered [world*rows][sum cols]
 *                       rank-major, ONE all-gather (the own rows are packed straight into the own slot: in-place collective)
 *   dh_reducescatter_packed  backward of it: grad_gathered [world*rows][sum cols] --reduce-scatter(SUM)--> scratch [rows][sum cols]
 *                       --split--> dst_k [rows][cols_k]; same sum as the reference's all-reduce + slice, world x less traffic
 *   dh_allreduce_bucket SUM all-reduce of grad[0..n) (fp32, in place).  bf16_stage != NULL ([n] bf16): the bucket crosses the
 *                       links as bf16 (cast, all-reduce, cast back -- all on the communication stream) */
#define DH_COMM_MAX_TENSORS 8
typedef struct dh_ctx dh_ctx;
int dh_comm_unique_id(void* id_out, int64_t bytes /* >= 128 */);
dh_ctx* dh_init(int rank, int world, int local_rank, const void* nccl_unique_id);
int dh_finalize(dh_ctx* ctx);
int dh_ctx_info(const dh_ctx* ctx, int* out3 /* rank, world, device */);
void* dh_comm_stream(const dh_ctx* ctx);                 /* the hipStream_t of the context (for profilers / external events) */
int dh_comm_wait(dh_ctx* ctx, dh_stream_t stream);       /* `stream` waits for every collective enqueued so far */
int dh_allgather_packed(dh_ctx* ctx, const void* const* src, const int* cols, int n, int rows, int elem_bytes, void* gathered,
                        dh_stream_t stream);
int dh_reducescatter_packed(dh_ctx* ctx, const void* grad_gathered, void* const* dst, const int* cols, int n, int rows, int elem_bytes,
                            void* scratch, dh_stream_t stream);
int dh_allreduce_bucket(dh_ctx* ctx, float* grad, int64_t n, void* bf16_stage, dh_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif
