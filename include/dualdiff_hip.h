/*
 * dualdiff_hip.h — C-ABI of the MI355X (gfx950) denoising hot path.
 *
 * Every entry point is a plain `extern "C"` launcher: device pointers + sizes + a
 * hipStream_t (passed as void*).  Launchers never allocate, never synchronise and
 * never throw: they validate arguments, enqueue kernels on `stream` and return 0
 * or a negative DD_ERR_* code.  All scratch memory is caller-owned (`ws`).
 * Launchers are graph-capture safe (hipGraph / torch.cuda.graph).
 *
 * The reference (yangzhaojason/DualDiff, MD_txt_con_fusion/) has no FFI: its hot
 * path reaches cuDNN/cuBLAS/xformers through torch.nn modules built by
 * diffusers-0.17.1.  Each launcher below names the reference call site whose
 * arithmetic it replaces (file:line relative to MD_txt_con_fusion/).
 *
 * Layout convention: activations are token-major / NHWC, i.e. a feature map
 * (M, C, H, W) is stored as rows = M*H*W, cols = C, row stride `ld*` in elements.
 * dtype: DD_F16 (reference eval dtype, misc/test_utils.py:98) or DD_BF16.
 * Accumulation, normalisation statistics and softmax are fp32.
 */
#ifndef DUALDIFF_HIP_H
#define DUALDIFF_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 2 (round 4): dd_gemm_desc gained splitk_inkernel / prefetch / prefetch_bytes and dd_attn_desc kv_batch_map2 in round 3
 * without a bump; the GroupNorm workspace's first 256 B are barrier state.
 * 3 (round 5): what was measured and never dispatched is gone — dd_gemm_desc lost ln_gamma / ln_beta / w_scale (row-panel
 * family), splitk_inkernel, prefetch / prefetch_bytes; dd_attn_desc.variant takes 0 only; the cooperative GroupNorm
 * entry points (dd_groupnorm_is_coop / _set_coop) are removed and its workspace carries no barrier state.
 * 4 (round 6): dd_attn_desc / dd_xattn_desc gained lk_dev — the key count of a cross-attention read from DEVICE memory,
 * so that one recorded HIP graph serves every context length (78 + N_box tokens: the reference's collate function pads
 * the boxes to the batch's maximum, dataset/utils.py:165-244, so the length changes from sample to sample) up to the
 * capacity it was recorded at.
 * A binding must check BOTH the version and the descriptor sizes (dd_desc_size) before the first launch: a stale
 * pair would read past the caller's struct. */
#define DD_ABI_VERSION 4

enum { DD_F16 = 0, DD_BF16 = 1 };

enum {
  DD_OK = 0,
  DD_ERR_BAD_ARG = -1,     /* null pointer / non-positive size / misaligned */
  DD_ERR_UNSUPPORTED = -2, /* shape or option not implemented by the kernels */
  DD_ERR_LAUNCH = -3,      /* hipGetLastError() != hipSuccess after launch */
  DD_ERR_WORKSPACE = -4    /* caller workspace too small */
};

typedef void* dd_stream_t; /* hipStream_t */

int dd_abi_version(void);
/* sizeof() of descriptor `which` as THIS library was compiled: 0 dd_gemm_desc, 1 dd_attn_desc, 2 dd_xattn_desc,
 * 3 dd_gemm8_desc, 4 dd_box_tokens_desc; -1 for an unknown index.  The binding compares with its own struct sizes at load time. */
int64_t dd_desc_size(int which);
const char* dd_error_string(int code);
/* Reports the compile-time gfx target string ("gfx950"). */
const char* dd_target_arch(void);

/* ------------------------------------------------------------------------- *
 * GEMM / implicit-GEMM convolution with fused epilogue.
 *
 *   out[r, n] (op)= alpha * ( sum_k A[r, k] * W[n, k] + bias[n]
 *                             + rowvec[r / rows_per_inst, n] ) + res[r, n]
 *
 * Replaces: every nn.Linear / 1x1 nn.Conv2d / 3x3 nn.Conv2d on the path —
 *   attention to_q/to_k/to_v/to_out        networks/box_adapter.py:102-110,163
 *   SFA projections                        networks/txt_con_fusion.py:110-116,169
 *   attn4 + connector                      networks/blocks.py:203-220
 *   ResnetBlock2D conv1/conv2/shortcut, Down/Upsample2D conv, proj_in/proj_out,
 *   FeedForward (GEGLU)                    diffusers-0.17.1 (instantiated at
 *                                          networks/unet_addon_rawbox.py:240-295,
 *                                          networks/unet_2d_condition_multiview.py:181-216)
 *   zero convs + conditioning_scale + branch sum
 *                                          networks/unet_addon_rawbox.py:1029-1055,
 *                                          pipeline/pipeline_bev_controlnet.py:421-429
 *   ControlNetConditioningEmbedding convs  networks/map_embedder.py:114-138
 *
 * W is [N][K] row-major (torch Linear layout; conv weights pre-packed to
 * [Cout][ky][kx][Cin]).  K % 8 == 0, N % 8 == 0, all pointers 16-byte aligned,
 * lda/ldc/ldres multiples of 8.
 * ------------------------------------------------------------------------- */
enum { DD_EPI_NONE = 0, DD_EPI_GEGLU = 1, DD_EPI_SILU = 2 };

typedef struct dd_gemm_desc {
  /* A operand (dense mode): rows x K, row stride lda.  Optional second source
   * concatenated along K (K = k1 + k2): columns [0,k1) from a, [k1,K) from a2. */
  const void* a;
  const void* a2;    /* may be NULL */
  int64_t lda, lda2;
  int32_t k1;        /* == K when a2 == NULL */
  /* Problem size */
  int32_t rows, n, k;
  /* Weights / epilogue operands */
  const void* w;       /* [n_w][k]; n_w = n (or 2n for DD_EPI_GEGLU) */
  const void* bias;    /* [n_w] or NULL */
  const void* rowvec;  /* [rows/rows_per_inst][ld_rowvec] or NULL (time-emb add) */
  int32_t rows_per_inst, ld_rowvec;
  const void* res;     /* [rows][ldres] or NULL */
  int64_t ldres;
  void* out;           /* [rows][ldc] */
  int64_t ldc;
  float alpha;         /* see formula */
  int32_t accumulate;  /* 1: out += value (read-modify-write), 0: out = value */
  int32_t epilogue;    /* DD_EPI_*; GEGLU: out[r,n] = h * gelu_erf(g) with h = col n,
                          g = col n + N of the 2N-wide product (FeedForward/GEGLU) */
  /* conv mode (conv != 0): A is an NHWC image batch and K = 9*cin (3x3, pad 1).
   * Optional nearest-neighbour upsample of the input to (hv, wv) before the conv
   * (Upsample2D with explicit output size,
   *  networks/unet_2d_condition_multiview.py:369-374,500-501). */
  int32_t conv;        /* 0 dense, 1 conv3x3 */
  int32_t hin, win, cin;   /* stored input image */
  int32_t hv, wv;          /* virtual (post-upsample) input size; == hin,win if none */
  int32_t hout, wout, stride;
  /* dtype / tuning */
  int32_t dtype;       /* DD_F16 / DD_BF16 */
  int32_t tile;        /* 0 = auto, else tile-config id (see dd_gemm_num_tiles) */
  int32_t split_k;     /* 0 = auto, 1 = off, >1 = number of K slices */
  void* ws;            /* split-K workspace (>= dd_gemm_workspace_bytes): 64 KiB reserved, then the fp32
                          partial slabs; one buffer serves all launches of a stream. */
  int64_t ws_bytes;
  /* LayerNorm fold (dense mode, K in {320, 640, 1280}, no a2, no split-K): `a` holds the
   * UN-normalised rows x; the kernel computes each row's mean / rstd over its K columns in its
   * prologue and evaluates  LN(x) W^T + b  as
   *     rstd_r * (x W'^T - mean_r * ln_colsum) + ln_bias,
   * with W' = W * gamma (passed as `w`), ln_colsum[n] = sum_k W'[n,k], ln_bias[n] = W beta + b
   * (both fp32, n_w entries).  Replaces LayerNorm + Linear of norm1->to_q/k/v, norm2->to_q,
   * norm4->attn4 q/k/v, norm3->GEGLU proj (blocks.py:150-236).  `bias` must be NULL.  */
  const void* ln_colsum;   /* NULL = no fold */
  const void* ln_bias;
  float ln_eps;
  int32_t out_f32;     /* 1: `out` is fp32 [rows][ldc] (attention logits of the VAE mid block, which must not be
                          rounded to the storage type before the softmax); no GEGLU / accumulate */
  /* LayerNorm statistics carried from the producer to the consumer (both optional, fp32):
   * ln_stats_out: this GEMM's epilogue also writes, per output row and per 32-column group, the sum and the
   *   sum of squares of the values it stores: [rows][n / 32][2] (n % 32 == 0, no split-K, no GEGLU).  Every
   *   transformer-block GEMM whose output feeds a LayerNorm (proj_in, to_out + residual; blocks.py:150-236)
   *   holds those values in registers anyway.
   * ln_stats_in: with the LayerNorm fold above, the row mean / rstd come from such a table of the `a`
   *   tensor ([rows][k / 32][2]) instead of a pass over the rows in every column tile's prologue. */
  void* ln_stats_out;
  const void* ln_stats_in;
  /* Head-major output (dense mode, plain epilogue, no split-K): with out_headmajor_d = D > 0 (D % 8 == 0,
   * n % D == 0) output column c of row r is stored at  out[((c / D) * rows + r) * D + c % D]  — one
   * contiguous [rows][D] plane per head of a fused Q|K|V projection, so that the attention kernel streams
   * a head's K/V rows as one linear range instead of D-element pieces of every fused row
   * (attn1 / attn4 of blocks.py:150-236).  The first `hm_scaled_planes` planes (the Q heads) are
   * multiplied by `hm_scale` in fp32 before the store rounding (softmax scale * log2 e, see
   * dd_attn_desc.q_prescaled).  ldc is ignored. */
  int32_t out_headmajor_d;
  int32_t hm_scaled_planes;
  float hm_scale;
  /* A split-K GEMM is two launches (partial slabs, then reduce + epilogue).  phase = 0 enqueues both (normal use);
   * 1 = the partial-slab launch only, 2 = the reduce launch only — so that a profiler-less caller (bench.py's
   * HIP-event brackets) can time the two kernels separately.  Ignored when split-K is off. */
  int32_t phase;
  /* LayerNorm EMITTED BY THE EPILOGUE (dense mode, n == 320, tile 40 = 80 whole rows x 320 columns per workgroup):
   * besides out = alpha * (A W^T + bias) + res, the kernel writes ln_out = LayerNorm(out) * lno_gamma + lno_beta
   * (two-pass fp32 statistics over the values as ROUNDED to the storage type, eps = ln_eps) — the producer of the
   * transformer's residual stream hands the next sub-layer its normalised input, so norm1 / norm2 / norm3 / norm4
   * of the 28x50 level (networks/blocks.py:150-236) need no launch of their own.  ln_out: T [rows][ld_ln_out]. */
  void* ln_out;            /* NULL = off */
  int64_t ld_ln_out;
  const void* lno_gamma;
  const void* lno_beta;
} dd_gemm_desc;

int dd_gemm(const dd_gemm_desc* d, dd_stream_t stream);
/* Workspace bytes dd_gemm needs for this descriptor (0 when split-K is off). */
int64_t dd_gemm_workspace_bytes(const dd_gemm_desc* d);
int dd_gemm_num_tiles(void);
/* id of the index-th tile configuration (for tuners); -1 when out of range. */
int dd_gemm_tile_id(int index);
/* Name of the kernel symbol dd_gemm would launch for `d` (for profile matching). */
const char* dd_gemm_kernel_name(const dd_gemm_desc* d);

/* ------------------------------------------------------------------------- *
 * GroupNorm (+ optional SiLU), NHWC, optional channel-concat of two sources.
 *   y[m, p, c] = act( (x[m,p,c] - mean[m,g]) * rstd[m,g] * gamma[c] + beta[c] )
 * Replaces torch.nn.GroupNorm + SiLU inside ResnetBlock2D (eps 1e-5), the
 * Transformer2DModel input norm (eps 1e-6, no SiLU) and conv_norm_out
 * (networks/unet_2d_condition_multiview.py:519-522).
 * ws: fp32 scratch, >= dd_groupnorm_workspace_bytes(M, G), private to the stream (first 256 bytes reserved).
 * C1 + C2 = C, C % G == 0, C1 % 8 == 0, C2 % 8 == 0.
 * ------------------------------------------------------------------------- */
int dd_groupnorm_nhwc(const void* x1, int32_t c1, const void* x2, int32_t c2,
                      const void* gamma, const void* beta, void* y,
                      int32_t m, int32_t hw, int32_t groups, float eps,
                      int32_t apply_silu, int32_t dtype, void* ws, int64_t ws_bytes,
                      dd_stream_t stream);
int64_t dd_groupnorm_workspace_bytes(int32_t m, int32_t groups);
/* Which kernels dd_groupnorm_nhwc launches for an image of hw pixels x c channels: 256 / 1024 = ONE launch of
 * dd_gn_fused_kernel with that many threads (slab in registers), 0 = dd_gn_stats_kernel + dd_gn_apply_kernel
 * (for profile matching, like dd_gemm_kernel_name). */
int dd_groupnorm_is_fused(int32_t hw, int32_t c, int32_t groups);
/* GroupNorm(+SiLU) that CONSUMES the partial slabs of a split-K dd_gemm launched with phase = 1 (in place of that GEMM's
 * reduce launch): x[r, c] = T( sum_z partial[z][r][c] + bias[c] + rowvec[r / hw][c] + res[r][c] ) — exactly what the
 * reduce launch would have stored (alpha = 1, no activation, no accumulate) — then y = GroupNorm(x) as above; x itself
 * is written only when x_out != NULL.  ResnetBlock2D: conv1 -> norm2 (x is not needed) and conv2 -> the Transformer2D
 * input norm (x is the residual stream).  partial = (char*)ws + 65536 of the dd_gemm call, nsplit = its slice count
 * (dd_gemm_kernel_name reports it).  Single-launch images only (dd_groupnorm_is_fused != 0), else DD_ERR_UNSUPPORTED. */
int dd_groupnorm_splitk(const float* partial, int32_t nsplit, const void* bias, const void* rowvec, int32_t ld_rowvec,
                        const void* res, int64_t ldres, void* x_out, const void* gamma, const void* beta, void* y,
                        int32_t m, int32_t hw, int32_t c, int32_t groups, float eps, int32_t apply_silu, int32_t dtype,
                        dd_stream_t stream);

/* LayerNorm over the last dim (eps 1e-5, affine) — BasicTransformerBlock
 * norm1/2/3 (diffusers) and norm4 (networks/blocks.py:67-71,191-194). */
int dd_layernorm(const void* x, const void* gamma, const void* beta, void* y,
                 int64_t rows, int32_t c, float eps, int32_t dtype, dd_stream_t stream);

/* ------------------------------------------------------------------------- *
 * Scaled-dot-product attention, flash-style (online softmax, fp32 state).
 *   O[b, i, h, :] (op)= softmax_j( scale * Q[b,i,h,:].K[kb,j,h,:] ) V[kb,j,h,:]
 * with kb = kv_batch_map ? kv_batch_map[b] : b.
 * Replaces xformers.ops.memory_efficient_attention at
 *   networks/box_adapter.py:150-156 (attn1/attn2 processor),
 *   networks/txt_con_fusion.py:156-162 (SFA), :313-318 (SFA+),
 *   networks/blocks.py:203-217 (attn4: one call per neighbour with
 *   accumulate=1 realises the sum over neighbours).
 * No mask (attention_mask is None on the inference path, blocks.py:166-187).
 * Q/K/V/O are (batch, len, heads*head_dim) views with row strides ld* (elements),
 * so they may alias slices of a fused QKV projection.  head_dim in {40,80,160}.
 * ------------------------------------------------------------------------- */
typedef struct dd_attn_desc {
  const void* q; const void* k; const void* v; void* o;
  int64_t ldq, ldk, ldv, ldo;         /* row strides in elements */
  int64_t q_batch_stride, k_batch_stride, v_batch_stride, o_batch_stride;
  int32_t batch, heads, head_dim, lq, lk;
  float scale;
  const int32_t* kv_batch_map;        /* device ptr [batch] or NULL */
  int32_t accumulate;                 /* 1: O += result */
  int32_t dtype;
  int32_t variant;                    /* 0 (the tuning variants of rounds 1-3 are gone: anything else is DD_ERR_UNSUPPORTED) */
  /* head h of q / k / v starts at element h * {q,k,v}_head_stride of its batch (0 = head_dim: the heads
   * are column blocks of one row, the reference's layout).  A head-major projection (dd_gemm_desc.
   * out_headmajor_d) passes ld = head_dim, batch stride = l * head_dim, head stride = rows * head_dim. */
  int64_t q_head_stride, k_head_stride, v_head_stride;
  int32_t q_prescaled;                /* 1: q already carries scale * log2(e); `scale` is ignored */
  /* Neighbour PAIR in one launch (networks/blocks.py:203-217, attn4 with neighboring_attn_type="add"):
   *   O[b] (op)= Attn(Q[b], K/V[kv_batch_map[b]]) + Attn(Q[b], K/V[kv_batch_map2[b]])
   * — two softmaxes with their own normalisation, summed in fp32 and rounded once.  Needs kv_batch_map and the default
   * variant (0); NULL = single attention. */
  const int32_t* kv_batch_map2;
  /* NULL, or a 4-byte-aligned DEVICE pointer to the number of keys every batch entry really has: the kernel reads it at
   * its start (clamped to [1, lk]) and `lk` is then only the CAPACITY — it sizes the strides the caller built and selects
   * the instantiation; keys lk_dev[0] .. lk-1 are never read.  Same arithmetic as a launch with lk = lk_dev[0] and the
   * same strides: bit-identical results. */
  const int32_t* lk_dev;
} dd_attn_desc;

int dd_attention(const dd_attn_desc* d, dd_stream_t stream);
/* Name of the kernel instantiation dd_attention would launch for `d` and its grid (query rows per wave, key tile and
 * waves per SIMD follow from the shape): for profile matching and for testing the dispatch without a GPU, like
 * dd_gemm_kernel_name.  "invalid" / "unsupported" mirror DD_ERR_BAD_ARG / DD_ERR_UNSUPPORTED; nothing is launched. */
const char* dd_attention_kernel_name(const dd_attn_desc* d);

/* ------------------------------------------------------------------------- *
 * Fused cross-attention of the 320-channel level (8 heads x 40, <= 128 context keys per view-instance), ONE launch:
 *   out[r, :] = ( softmax_j( scale * (x[r] Wq^T)_h . K[i, j]_h ) V[i, j]_h )_h Wo^T + bo + res[r, :]
 * for query row r of view-instance i = r / rows_per_inst; optionally also ln_out = LayerNorm(out) (two-pass fp32
 * statistics over the rounded values, eps ln_eps).  Replaces to_q -> memory_efficient_attention -> to_out (+ residual)
 * of Semantic Fusion Attention (networks/txt_con_fusion.py:108-181: x = res = the ORS condition map, K / V = to_k /
 * to_v of the 77 text tokens) and of the text / box cross-attention attn2 of the 28x50 transformer blocks
 * (networks/box_adapter.py:102-163 inside blocks.py:166-187: x = LayerNorm2(h), res = h).  q and the attention output
 * never reach HBM.  x / res / out: [instances * rows_per_inst][320] with row pitches ld* (elements).  K / V: element
 * (instance i, key j, head h, d) at  k + i * k_inst_stride + j * ldk + h * k_head_stride + d  (elements; all strides
 * multiples of 8) — row-major column slices of a wider projection (ldk = its width, head stride 40, instance stride
 * lk * ldk) and head-major planes written by dd_gemm's out_headmajor_d = 40 (ldk = 40, head stride = total rows * 40,
 * instance stride lk * 40: a head's keys are one contiguous block, which is what the kernel streams fastest).
 * wq / wo: the [320][320] torch Linear weights PACKED by dd_xattn_pack_weight (K-step-major, pre-swizzled, so that the
 * kernel's LDS-DMA copies them linearly); bo [320] (required: pass zeros for a bias-free projection).
 * ------------------------------------------------------------------------- */
typedef struct dd_xattn_desc {
  const void* x; int64_t ldx;
  const void* res; int64_t ldres;          /* NULL = no residual */
  const void* wq; const void* wo; const void* bo;
  const void* k; const void* v; int64_t ldk, ldv;
  int64_t k_inst_stride, k_head_stride, v_inst_stride, v_head_stride;
  void* out; int64_t ldo;
  int32_t instances, rows_per_inst, lk;
  int32_t channels, heads;                 /* 320, 8: anything else is DD_ERR_UNSUPPORTED */
  float scale;
  int32_t dtype;
  void* ln_out; int64_t ld_ln_out;         /* NULL = off */
  const void* ln_gamma; const void* ln_beta; float ln_eps;
  const int32_t* lk_dev;                   /* as dd_attn_desc.lk_dev: keys per instance from device memory, lk = capacity */
} dd_xattn_desc;

int dd_xattn320(const dd_xattn_desc* d, dd_stream_t stream);
/* w: [320][320] row-major (torch Linear: out x in) -> packed: 102,400 elements in the streaming order of dd_xattn320. */
int dd_xattn_pack_weight(const void* w, void* packed, int32_t dtype, dd_stream_t stream);

/* ------------------------------------------------------------------------- *
 * Small HBM-bound helpers.
 * ------------------------------------------------------------------------- */
/* y = a + b (+ c)   — ControlNet residual add into UNet skips
 * (networks/unet_2d_condition_multiview.py:464-473,487-488). c may be NULL. */
int dd_add(const void* a, const void* b, const void* c, void* y, int64_t n,
           int32_t dtype, dd_stream_t stream);
/* y = x * s (in place allowed) */
int dd_scale(const void* x, void* y, float s, int64_t n, int32_t dtype, dd_stream_t stream);
/* SiLU elementwise (time embedding act). */
int dd_silu(const void* x, void* y, int64_t n, int32_t dtype, dd_stream_t stream);

/* NCHW <-> NHWC (only at the 4-channel latent / drop-in boundary). */
int dd_nchw_to_nhwc(const void* x, void* y, int32_t m, int32_t c, int32_t hw,
                    int32_t c_pad, int32_t dtype, dd_stream_t stream);
int dd_nhwc_to_nchw(const void* x, void* y, int32_t m, int32_t c, int32_t hw,
                    int32_t ldx, int32_t dtype, dd_stream_t stream);

/* Sinusoidal timestep embedding, diffusers `Timesteps(dim, flip_sin_to_cos=True,
 * downscale_freq_shift=0)` (networks/unet_addon_rawbox.py:142-144,921-927;
 * unet_2d_condition_multiview.py:404-409): out[i, :] = [cos(t_i f), sin(t_i f)],
 * f_j = exp(-ln(10000) j / (dim/2)).  t: fp32 device array [n]. */
int dd_timestep_embedding(const float* t, void* out, int32_t n, int32_t dim,
                          int32_t flip_sin_to_cos, float freq_shift,
                          int32_t dtype, dd_stream_t stream);

/* NeRF-style Fourier features (networks/embedder.py:18-67, used by the camera and 3-D box token
 * embedders, unet_addon_rawbox.py:308-325, bbox_embedder.py:187): for every row of `dims` inputs
 *   out[r] = [x (if include_input), sin(f_0 x), cos(f_0 x), ..., sin(f_{F-1} x), cos(f_{F-1} x)],
 * each block `dims` wide.  x: fp32 / fp16 / bf16 per `in_dtype` (DD_F16, DD_BF16 or DD_F32); the
 * arithmetic is fp32; out in `out_dtype`.  freqs: HOST array of num_freqs (<= 16) frequencies. */
#define DD_F32 2
int dd_fourier_embed(const void* x, void* out, int64_t rows, int32_t dims, const float* freqs,
                     int32_t num_freqs, int32_t include_input, int32_t in_dtype, int32_t out_dtype,
                     dd_stream_t stream);

/* ------------------------------------------------------------------------- *
 * Token / condition preparation of a ControlNet branch (csrc/tokens.hip, round 4): what the reference does with ~40
 * tiny torch ops per step (unet_addon_rawbox.py:308-361,832-896,1007; bbox_embedder.py:164-203;
 * map_embedder.py:116-125) as four launches.
 * ------------------------------------------------------------------------- */
/* NCHW (m, c, h, views * w) -> NHWC rows of m * views instances (h, w, c_pad), channels zero-padded; views = 1 is the
 * plain layout change (LDS-tiled: both sides coalesced), views = 6 the panorama split of map_embedder.py:116-125.
 * c_pad % 8 == 0, y 16-byte aligned. */
int dd_nchw_to_nhwc_views(const void* x, void* y, int32_t m, int32_t c, int32_t h, int32_t w, int32_t views,
                          int32_t c_pad, int32_t dtype, dd_stream_t stream);
/* dd_fourier_embed with a strided source and a padded destination: row r = outer * inner + j reads its `dims` inputs at
 * x[outer * stride_outer + j * stride_inner + d * stride_dim] and writes its features at
 * out[outer * out_ld + j * width ...], width = dims * (include_input + 2 * num_freqs); the out_ld - inner * width
 * (<= dims) trailing columns of every outer row are zeroed (K padding of the Linear that follows).  Camera parameters
 * (b, n, 3, 7) -> (b n, 189 padded to 192): dims 3, inner 7, strides (21, 1, 7) (unet_addon_rawbox.py:308-325). */
int dd_fourier_embed_strided(const void* x, void* out, int64_t rows, int32_t dims, const float* freqs,
                             int32_t num_freqs, int32_t include_input, int32_t in_dtype, int32_t out_dtype,
                             int32_t inner, int64_t stride_outer, int64_t stride_inner, int64_t stride_dim,
                             int64_t out_ld, dd_stream_t stream);
/* Both operands of the box MLP (bbox_embedder.py:164-203).  For every box r: pos[r] = masks[r] ? Fourier features of its
 * points : null_pos  (T [rows][points_per_box * 3 * (include_input + 2 num_freqs)], the input of bbox_proj), and
 * cat[r][cls_offset ...] = masks[r] ? class_tokens[classes[r]] : null_class  (the right half of the concat
 * second_linear reads; bbox_proj writes the left half); cls_out (optional) gets the same class rows (box adapter).
 * points: [rows][points_per_box][3] in `points_dtype` (DD_F16 / DD_BF16 / DD_F32; values pass through that dtype as in
 * the reference); classes int64; masks uint8 (NULL = keep all); normalize: (p - xyz_min) / xyz_range first. */
typedef struct dd_box_tokens_desc {
  const void* points; const int64_t* classes; const uint8_t* masks;
  const void* class_tokens; const void* null_pos; const void* null_class;
  void* pos; void* cat; void* cls_out;
  int32_t rows, points_per_box, num_freqs, include_input, class_token_dim, cls_offset;
  int64_t ld_cat;
  int32_t normalize, points_dtype, dtype;
  int32_t n_classes;   /* rows of class_tokens: a kept box's class index is wrapped like torch indexing (-1 = last row); one still
                          outside [0, n_classes) gets NaN tokens instead of reading out of bounds.  0 = unchecked (ABI 2 callers) */
  float freqs[16]; float xyz_min[3]; float xyz_range[3];
} dd_box_tokens_desc;
int dd_box_tokens(const dd_box_tokens_desc* d, dd_stream_t stream);
/* Context assembly: full[i] = [cam_i | text | box tokens] and (optional) txt[i] = text for instance i = (scene, view)
 * (unet_addon_rawbox.py:337-361, :1007, :977).  cam [m][dim]; text [scenes][lt][dim] (text_per_view: [m][lt][dim]);
 * box [scenes * box_views][nbox][dim], box_views in {n_cam, 1} (NULL when nbox == 0); full [m][1 + lt + nbox][dim];
 * txt [m][lt][dim] or NULL.  dim % 8 == 0, all 16-byte aligned. */
int dd_ctx_assemble(const void* cam, const void* text, const void* box, void* full, void* txt, int32_t m,
                    int32_t n_cam, int32_t lt, int32_t nbox, int32_t dim, int32_t text_per_view, int32_t box_views,
                    int32_t dtype, dd_stream_t stream);

/* ORS projection (networks/occ3d_proj.py:49-113 + dataset/utils.py:412-420): every
 * latent pixel's ray is sampled at `samples` equidistant points (step metres apart) in a
 * 200 x 200 x 16 class volume (uint8, x-major, 0.4 m voxels: x, y in [-40, 40) m, z in [-1, 5.4) m);
 * the nearest voxel's class is taken, 17 outside the volume.
 *   origin [n_cam][3], dir [n_cam][hw][3]: fp32 ray origins / unit directions (ego frame);
 *   labels (may be NULL): uint8 [n_cam][hw][samples];
 *   cond   (may be NULL): [n_cam][samples][hw] in `dtype` = class / 17 after the optional
 *           foreground (class <= 10 -> 17, keep_fg == 0) / background (class >= 11 -> 17, keep_bg == 0)
 *           filtering — the ORS-3D ControlNet condition (unet_addon_rawbox.py:967-990).
 * Integer output: bit-exact against the reference's fp32 arithmetic (separate multiply / add, true
 * divisions, round-half-even). */
int dd_ors_project(const uint8_t* occ, const float* origin, const float* dir, uint8_t* labels, void* cond,
                   int32_t n_cam, int32_t hw, int32_t samples, float step, int32_t keep_fg, int32_t keep_bg,
                   int32_t dtype, dd_stream_t stream);

/* Row softmax of fp32 logits into the storage type: p[r][c] = softmax_c(s[r][c]) for c < cols
 * (diffusers AttnProcessor `get_attention_scores` with upcast_softmax — the single 512-wide head of
 * the VAE decoder's mid-block attention, decode_latents pipeline_bev_controlnet.py:101-113).
 * s: fp32 [rows][lds]; p: T [rows][ldp], columns cols..ldp-1 are zero-filled (K padding of the PV GEMM). */
int dd_softmax_rows(const float* s, void* p, int64_t rows, int32_t cols, int64_t lds, int64_t ldp,
                    int32_t dtype, dd_stream_t stream);

/* conv3x3 with tiny Cout (conv_out 320->4): y NCHW fp32/T. x NHWC (rows, cin),
 * w [cout][9*cin]; writes y as NCHW (m, cout, h, w) in dtype T.
 * (networks/unet_2d_condition_multiview.py:522) */
int dd_conv3x3_small_cout(const void* x, const void* w, const void* bias, void* y_nchw,
                          int32_t m, int32_t h, int32_t wd, int32_t cin, int32_t cout,
                          int32_t dtype, dd_stream_t stream);

/* conv3x3 / pad 1 / stride 1 or 2 for THIN channel counts on large NHWC images — the first layers of
 * ControlNetConditioningEmbedding (networks/map_embedder.py:79-113: 3 -> 16 -> 16 -> 32 -> 32 channels on 224x400 ..
 * 112x200).  x (m, hin, win, cin), w [cout][9*cin] (k = tap * cin + channel), bias [cout] or NULL,
 * y (m, hout, wout, cout) with hout = (hin - 1) / stride + 1; optional SiLU.  cin in {8, 16, 32} (3 input channels are
 * zero-padded to 8 by the caller), cout in {16, 32}; other combinations: DD_ERR_UNSUPPORTED. */
int dd_conv3x3_thin(const void* x, const void* w, const void* bias, void* y, int32_t m, int32_t hin, int32_t win,
                    int32_t cin, int32_t cout, int32_t stride, int32_t silu, int32_t dtype, dd_stream_t stream);

/* Classifier-free guidance + DDIM (eta = 0) update, fused:
 *   eps = eps_u + g (eps_c - eps_u);  x0 = (x - sqrt(1-a_t) eps)/sqrt(a_t);
 *   x' = sqrt(a_prev) x0 + sqrt(1-a_prev) eps
 * (pipeline/pipeline_bev_controlnet.py:487-499 with diffusers DDIMScheduler.step).
 * eps: [2][n] (uncond first, :489), x / x_out: [n] in dtype T; coef: device fp32[4] =
 * {sqrt(a_t), sqrt(1-a_t), sqrt(a_prev), sqrt(1-a_prev)} read at kernel time so the
 * launch can be replayed from a graph. x_dup: optional second copy of x' (the CFG
 * duplicate `torch.cat([latents]*2)`, :384-386) or NULL. */
int dd_cfg_ddim_step(const void* eps, const void* x, void* x_out, void* x_dup,
                     const float* coef, float guidance, int64_t n,
                     int32_t dtype, dd_stream_t stream);

/* Classifier-free guidance + UniPC (bh2, order <= 2, x0-prediction) step, fused — the scheduler the
 * reference's test pipeline installs (misc/test_utils.py:161-162) and steps at
 * pipeline/pipeline_bev_controlnet.py:487-499:
 *   eps = eps_u + g (eps_c - eps_u);  x0 = a_x x + a_e eps
 *   x_c = use_c ? c_l last + c_1 m1 + c_2 m2 + c_0 x0 : x        (corrector)
 *   x'  = p_x x_c + p_0 x0 + p_1 m1                              (predictor)
 *   last <- x_c;  m2 <- m1;  m1 <- x0
 * eps: [2][n] (uncond first), x / x_out / x_dup: [n] in dtype T; last, m1, m2: [n] fp32 history owned by
 * the caller (any values on the first step: use_c = 0, p_1 = 0 there); coef: device fp32[10] =
 * {a_x, a_e, use_c, c_l, c_1, c_2, c_0, p_x, p_0, p_1} read at kernel time (graph replay). */
int dd_cfg_unipc_step(const void* eps, const void* x, void* x_out, void* x_dup, float* last, float* m1,
                      float* m2, const float* coef, float guidance, int64_t n, int32_t dtype,
                      dd_stream_t stream);

/* ------------------------------------------------------------------------- *
 * EXTENSION (BASELINE configs[4]: "fp8 weights (CDNA4 fp8 MFMA)"; no reference semantics — README.md:47-49 is prose):
 * W8A8 projection on the fp8 matrix path.
 *   dd_rowquant_fp8: y = gamma ? LayerNorm(x) (dd_layernorm's arithmetic, rounded to the storage type) : x;
 *       scale[r] = max_c |y[r, c]| / 448 (1 for a zero row);  q[r, c] = e4m3fn(y[r, c] / scale[r]) (round to nearest even);
 *       q rows have pitch ldq bytes (multiple of 128, zero padded) — the LayerNorm launch of norm1 / norm2 / norm4
 *       (networks/blocks.py:150-222) doubles as the activation quantiser.
 *   dd_gemm8: out[r, n] = a_scale[r] * w_scale[n] * sum_k A8[r, k] W8[n, k] + bias[n] + res[r, n], fp32 accumulation in
 *       v_mfma_scale_f32_16x16x128_f8f6f4 (unit block scales), A8 [rows][lda] / W8 [n][ldw] OCP e4m3fn bytes whose rows are
 *       zero padded to k_padded (multiple of 128); output in `dtype`, row-major or head-major planes as dd_gemm's
 *       out_headmajor_d (the fused Q|K|V projection of attn1 / attn4, to_q of attn2), or with the GEGLU gate.
 * ------------------------------------------------------------------------- */
int dd_rowquant_fp8(const void* x, const void* gamma, const void* beta, void* q, float* scale, int64_t rows, int32_t c,
                    int64_t ldq, float eps, int32_t dtype, dd_stream_t stream);

typedef struct dd_gemm8_desc {
  const void* a; const float* a_scale; int64_t lda;
  const void* w; const float* w_scale; int64_t ldw;
  const void* bias; const void* res; int64_t ldres;
  void* out; int64_t ldc;
  int32_t rows, n, k_padded;
  int32_t dtype;                          /* of bias / res / out */
  int32_t out_headmajor_d, hm_scaled_planes; float hm_scale;
  int32_t geglu;                          /* 1: W8 / w_scale / bias have 2n rows (h | g); out[r, c] = h * gelu_erf(g) — the
                                             GEGLU projection of FeedForward behind norm3 (diffusers FeedForward / GEGLU) */
} dd_gemm8_desc;

int dd_gemm8(const dd_gemm8_desc* d, dd_stream_t stream);

/* Measurement aid (not on the data path): one wave busy-waits `ticks_100mhz` ticks of the constant-rate 100 MHz
 * s_memrealtime counter and stores its first and last counter reading in stamps[0], stamps[1] (device memory).
 * bench.py brackets it with HIP events to learn what an event pair adds to a kernel of KNOWN device-side duration. */
int dd_probe_spin(uint64_t* stamps, uint32_t ticks_100mhz, dd_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* DUALDIFF_HIP_H */
