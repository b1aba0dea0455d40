/* libinstantir_hip.so -- C ABI of the MI355X (gfx950) kernels behind the InstantIR denoising path.
 *
 * The reference (rebots-online/InstantIR) is pure Python on PyTorch ops; it has no FFI of its own
 * (SURVEY.md section 8b).  Each entry point below therefore replaces a *call site* of a PyTorch op in
 * the reference and cites it.  Conventions: plain C symbols, raw device pointers, explicit
 * dims/strides (element counts, fp16 unless noted), `stream` is a hipStream_t passed as void*,
 * caller owns every buffer, return 0 on success, IIR_EINVAL (-1) for rejected arguments,
 * IIR_ELAUNCH (-2) when the launch failed.  No global state besides one-time kernel attributes;
 * thread-safe per stream; nothing here allocates, synchronises or copies, so every entry point may
 * be captured into a hipGraph.
 *
 * Activation layout is NHWC / (rows, tokens, channels): a (R, C, H, W) reference tensor is the
 * (R, H, W, C) buffer here, which is also the (R, H*W, C) token matrix (reference permutes at
 * module/min_sdxl.py:581-583,590-594 -- free here).
 */
#ifndef INSTANTIR_HIP_H
#define INSTANTIR_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define IIR_ABI_VERSION 1

/* epilogue selectors */
#define IIR_EPI_PLAIN 0 /* C = act(acc + bias + rowbias) + res                                    */
#define IIR_EPI_GEGLU 1 /* C[:, j] = (acc_v + b_v) * gelu_erf(acc_g + b_g); W rows pair-permuted  */
#define IIR_EPI_SFT 2   /* C[:, j] = res[:, j] * (acc_gamma + b + 1) + (acc_beta + b)             */
#define IIR_EPI_XATTN 3 /* C = cross_attention(q = acc + bias, xattn_kv[0], xattn_kv[1]): see iir_gemm_desc.xattn_kv */
#define IIR_ACT_NONE 0
#define IIR_ACT_SILU 1
#define IIR_ACT_GELU 2 /* erf form */
#define IIR_ACT_QUICKGELU 3 /* x * sigmoid(1.702 x), CLIP-L text encoder */

/* Pair permutation expected by GEGLU / SFT epilogues: for an op with `n_out` outputs whose
 * "value" rows are V[0..n_out) and partner rows are G[0..n_out), the weight (and bias) handed to
 * the kernel has 2*n_out rows where rows [16*b, 16*b+8) = V[8*b, 8*b+8) and
 * rows [16*b+8, 16*b+16) = G[8*b, 8*b+8).  (Host helper: instantir_amd.packing.pair_rows.) */

#define IIR_DT_F16 0
#define IIR_DT_BF16 1

typedef struct iir_attn_kv {
    const void* K; int64_t ldk, k_batch_stride;     /* K[b][t][h*64+d]                               */
    const void* Vt; int64_t ldvt, vt_batch_stride;  /* Vt[h*64+d][b*vt_batch_stride + t]; rows finite */
    int32_t Tkv;                                    /*   and readable on [0, roundup8(Tkv))           */
} iir_attn_kv;

typedef struct iir_gemm_desc {
    const void* A; int64_t lda;    /* [M][K] activations, row stride lda                           */
    const void* W;                 /* [N][K] weights (torch nn.Linear layout), K contiguous        */
    void* C; int64_t ldc;          /* [M][N] (or [M][N/2] for paired epilogues)                    */
    int32_t M, N, K;               /* K % 64 == 0, N % 4 == 0                                      */
    const void* bias;              /* [N] or NULL                                                  */
    const void* rowbias; int64_t ldrb; int32_t rows_per_rb; /* + rowbias[m / rows_per_rb][n]       */
    const void* res; int64_t ldr;  /* residual / SFT `h` input, or NULL                            */
    int32_t epi, act;
    float out_scale;               /* 0 means 1                                                    */
    int32_t tile;                  /* 0 auto, 1 = 128x128, 2 = 128x64, 3 = 64x64, 4 = 128x160, 5 = 64x160 */
    const void* prefetch;          /* optional: range the workgroups touch (at most 1024 128-byte lines each) */
    int64_t prefetch_bytes;        /*   so it is in the Infinity Cache for a LATER launch (next layers' weights) */
    void* splitk_ws;               /* optional split-K workspace (iir_gemm_splitk_workspace_bytes), ZEROED once by the  */
    int64_t splitk_ws_bytes;       /*   caller and private to one stream; lets tile = 0 pick the 2-slice form for long K  */
    void* Ct; int64_t ldct;        /* optional (PLAIN epilogue, no residual on those columns): output columns n >= tr_from */
    int32_t tr_from;               /*   are stored TRANSPOSED, Ct[(n - tr_from) * ldct + m] -- the V third of a fused       */
                                   /*   q|k|v projection lands as the V^T image iir_attention_d64_f16 consumes              */
    int32_t dtype;                 /* IIR_DT_F16 (0) or IIR_DT_BF16 (1): element type of A, W, C, bias, rowbias, res.  The  */
                                   /*   bf16 build serves the VAE, which the reference runs in fp32 because the SDXL VAE   */
                                   /*   overflows fp16 (pipelines/sdxl_instantir.py:984-1001,1668-1674)                     */
    int32_t c_f32;                 /* != 0: C is float [M][N] (ldc in floats, plain epilogue, out_scale only): the VAE     */
                                   /*   mid-block attention scores stay fp32 through their softmax                          */
    const void* wscale;            /* != NULL: W holds fp8-E4M3 (OCP) BYTES [N][K] and wscale[n] (fp32, 16-byte aligned) is */
                                   /*   the scale of row n: C = epi((A8 . W8^T) * wscale[n] ...), A converted to fp8 in the */
                                   /*   kernel (BASELINE configs[4]: "fp8 MFMA weights" for the LCM single-step path)       */
    /* LayerNorm without a LayerNorm launch (nn.LayerNorm -> nn.Linear pairs of BasicTransformerBlock, module/min_sdxl.py:541-562):  */
    void* ln_stats_out;            /* producer side, optional: the launch that WRITES the residual stream also leaves, per row and   */
                                   /*   per column tile, (mean, M2) of the fp16 values it stored: float2 [iir_gemm_ln_parts()][M]    */
    const void* ln_stats_in;       /* consumer side, optional: A holds the RAW (un-normalised) rows; W must be W . diag(gamma),      */
    int32_t ln_parts, ln_part_cols;/*   bias must carry W . beta; the kernel merges the `ln_parts` partials of each row (each over   */
    float ln_eps;                  /*   `ln_part_cols` columns; parts x cols == K) to (mean, rstd) and forms                         */
    const void* ln_colsum;         /*   C = epi(rstd * (A . W^T) - rstd * mean * ln_colsum[n] + bias ...), ln_colsum[n] = sum_k W[n][k] (fp32) */
    /* GroupNorm statistics from the launch that produces the GroupNorm's input (nn.Conv2d / proj_out -> nn.GroupNorm pairs,          */
    /* module/min_sdxl.py:242-283,565-595): no separate statistics pass over the tensor                                               */
    void* gn_stats_out;            /* optional (PLAIN epilogue, whole tiles, M % 64 == 0): float2 [M / 64][N] = (mean, M2) of the 64   */
                                   /*   stored values of every channel per 64-row slab; consumed by iir_groupnorm_from_partials       */
    /* IIR_EPI_XATTN: the Linear is `attn2.to_q` of TA_IPAttnProcessor2_0 (module/ip_adapter/attention_processor.py:1140) with the     */
    /* softmax scale x log2(e) folded into W / bias by the caller; every workgroup finishes a 64-row x 2-head tile of q and runs the    */
    /* two SDPA calls + add of :1165,:1185,:1192 on it, so C receives `hidden_states + ip_hidden_states` (before to_out) and q never     */
    /* goes to memory.  Needs: fp16, N % 128 == 0 (head dimension 64), M % 64 == 0, xattn_tq % 64 == 0, 1 <= Tkv[0] <= 80,             */
    /* 1 <= Tkv[1] <= 64, ldk / ldvt / batch strides % 8 == 0, no res / rowbias / act / Ct / c_f32 / wscale / *_stats_out / split-K.      */
    const iir_attn_kv* xattn_kv;   /* [2]: text K / V^T, IP-token K / V^T (layouts as for iir_attention_d64_f16)                       */
    int32_t xattn_tq;              /* query rows per image (row m belongs to image m / xattn_tq)                                      */
    /* BASELINE configs[4], both operands in fp8: with `wscale` set and a_fp8 != 0, A holds fp8-E4M3 (OCP) BYTES [M][K] too (lda in    */
    /* bytes, % 16; K % 128 == 0): C = epi((A8 . W8^T) * wscale[n] * a_scale ...).  Half the 128-byte lines per FLOP of the fp16 form. */
    int32_t a_fp8;
    float a_scale;                 /* 0 means 1                                                                                       */
    int32_t c_fp8;                 /* != 0: C is a BYTE matrix of fp8-E4M3 (ldc in bytes, % 8; whole tiles; PLAIN / GEGLU epilogues, */
                                   /*   no *_stats_out / Ct / c_f32): the value is rounded to fp16 first, then to fp8 (scale 1)     */
} iir_gemm_desc;

/* Replaces nn.Linear / F.linear call sites: attention projections
 * module/ip_adapter/attention_processor.py:370,377-378,402,1140,1148-1149,1173-1174,1195; GEGLU
 * feed-forward module/min_sdxl.py:502-528; proj_in/out :572,575; time_emb_proj :266; embeddings
 * :226-240; AdaLayerNorm linear attention_processor.py:23; Resampler resampler.py:48-49,66-68,95-97. */
int iir_gemm_f16(const iir_gemm_desc* d, void* stream);
/* partials per row a tile = 0, plain-epilogue launch of (M, N, K) writes to `ln_stats_out`; 0 = that launch cannot (ragged tiles) */
int iir_gemm_ln_parts(int32_t M, int32_t N, int32_t K);
/* the tile `tile = 0` resolves to for an (M, N, K) problem (paired != 0 for GEGLU / SFT epilogues) */
int iir_gemm_pick_tile(int32_t M, int32_t N, int32_t K, int32_t paired);
/* The tile / kernel iir_gemm_f16(d) resolves to, without launching (91: the 8-wave 256x320 kernel of csrc/gemm8.hip, used for the
 * GEGLU projections of module/min_sdxl.py:502-528 when its tiles fill the chip); -1 for an invalid descriptor. */
int iir_gemm_resolve_tile(const iir_gemm_desc* d);
/* 1 when the tile = 0, plain-epilogue fp16 launch of (M, N, K) can leave GroupNorm partials in `gn_stats_out` */
int iir_gemm_gn_supported(int32_t M, int32_t N, int32_t K, int32_t is_conv);
/* 1 when the tile = 0 all-fp8 launch (a_fp8) of (M, N, K) can store its result as fp8 bytes (c_fp8): whole tiles */
int iir_gemm_fp8_out_supported(int32_t M, int32_t N, int32_t K, int32_t paired);
/* bytes of split-K workspace an (M, N) problem can use (0: the split form does not apply to it) */
int64_t iir_gemm_splitk_workspace_bytes(int32_t M, int32_t N);
/* 1 if a tile = 0 launch of this problem with a workspace of ws_bytes takes the two-slice split-K form (128x160 tile) */
int iir_gemm_uses_splitk(int32_t M, int32_t N, int32_t K, int64_t ws_bytes);
/* output-tile width BN of tile id `tile` (1..6) */
int iir_gemm_tile_bn(int32_t tile);

typedef struct iir_conv_desc {
    const void* X; int64_t ldx;    /* NHWC input (R, H, W, Cin), pixel stride ldx >= Cin           */
    int32_t R, H, Wd, Cin;         /* image count, height, width; Cin % 64 == 0 (host pads latents) */
    const void* Wt;                /* [Cout][k][k][Cin] weights                                    */
    void* Y; int64_t ldy;          /* NHWC output, pixel stride ldy                                */
    int32_t Cout, ksize, stride, upsample;   /* ksize 1|3, stride 1|2, nearest-2x folded if upsample */
    const void* bias;
    const void* rowbias; int64_t ldrb; int32_t rows_per_rb;
    const void* res; int64_t ldr;
    int32_t epi, act;
    float out_scale;
    int32_t tile;
    const void* zero_page;         /* >= 128 zero bytes: source of the padding pixels              */
    int64_t x_img_stride;          /* elements between input images; 0 = H*Wd*ldx (dense)          */
    int32_t y_img_rows;            /* rows between images in Y; 0 = Ho*Wo (dense)                  */
    int32_t res_img_rows;          /* rows between images in res; 0 = Ho*Wo                        */
    int32_t pad_mode;              /* 0: symmetric ksize/2; 1: pad 0 top/left, 1 bottom/right (VAE  */
                                   /*    Downsample2D(padding=0) + F.pad(0,1,0,1), vae.py:110)      */
    const void* prefetch;          /* as in iir_gemm_desc                                          */
    int64_t prefetch_bytes;
    void* splitk_ws;               /* as in iir_gemm_desc (M = R*Ho*Wo, N = Cout)                  */
    int64_t splitk_ws_bytes;
    int32_t dtype;                 /* IIR_DT_F16 / IIR_DT_BF16, as in iir_gemm_desc                */
    void* gn_stats_out;            /* as in iir_gemm_desc: float2 [R * Ho * Wo / 64][Cout] (rows of one image must fill whole 64-row slabs) */
} iir_conv_desc;

/* Replaces nn.Conv2d call sites: ResnetBlock2D module/min_sdxl.py:256-259,274 (+ the temb add :267 as
 * `rowbias`, the residual add :279 as `res`), Downsample2D :601-606, Upsample2D :612-618 (with
 * F.interpolate nearest folded in), conv_in/conv_out :827,912, SFT module/aggregator.py:62-67,76-86. */
int iir_conv2d_nhwc_f16(const iir_conv_desc* c, void* stream);

typedef struct iir_attn_desc {
    const void* Q; int64_t ldq, q_batch_stride;     /* Q[b][t][h*64+d]                               */
    void* O; int64_t ldo, o_batch_stride;           /* O[b][t][h*64+d]                               */
    int32_t batch, heads, Tq, nseg;                 /* nseg 1 or 2                                   */
    float scale;                                    /* 1/sqrt(64) for SDPA                           */
    iir_attn_kv kv[2];
    int32_t causal;                                 /* != 0: key j is visible to query i only if j <= i */
    int32_t q_prescaled;                            /* != 0: Q already holds q * scale * log2(e) (the caller folded the factor
                                                     *   into the projection weights); the kernel then uses Q as it stands   */
    int32_t o_fp8;                                  /* != 0: O is a BYTE matrix of fp8-E4M3 (ldo, o_batch_stride in bytes): the
                                                     *   A operand of an all-fp8 `to_out` GEMM (iir_gemm_desc.a_fp8)         */
} iir_attn_desc;

/* Replaces F.scaled_dot_product_attention at module/ip_adapter/attention_processor.py:394 (nseg=1)
 * and the two SDPA calls + add at :1165,:1185,:1192 (nseg=2: text KV, IP KV), head_dim 64. */
int iir_attention_d64_f16(const iir_attn_desc* a, void* stream);

/* nn.GroupNorm (+ fused nn.SiLU) on NHWC: module/min_sdxl.py:245,252,257,269-271,568,841.
 * workspace: iir_groupnorm_workspace_bytes(R, groups) bytes of fp32 partial sums. */
int iir_groupnorm_nhwc_f16(const void* X, int64_t ldx, void* Y, int64_t ldy, int32_t R, int32_t HW, int32_t C,
                           int32_t groups, const void* gamma, const void* beta, float eps, int32_t silu,
                           void* workspace, int64_t workspace_bytes, void* stream);
/* nn.GroupNorm (+ SiLU) whose statistics pass already happened in the producing launch (`gn_stats_out` of iir_gemm_f16 /
 * iir_conv2d_nhwc_f16: float2 [R * HW / 64][ldp] (mean, M2) per 64-row slab and channel): merges the slabs and channels of
 * every (image, group), then normalises X into Y.  HW % 64 == 0.  Same call sites as iir_groupnorm_nhwc_f16. */
int iir_groupnorm_from_partials(const void* partials, int64_t ldp, const void* X, int64_t ldx, void* Y, int64_t ldy, int32_t R,
                                int32_t HW, int32_t C, int32_t groups, const void* gamma, const void* beta, float eps, int32_t silu,
                                void* workspace, int64_t workspace_bytes, int32_t dtype, void* stream);
/* the same for fp16 or bf16 tensors (X, Y, gamma, beta of type `dtype`): the VAE's GroupNorms (eps 1e-6,
 * module/diffusers_vae/vae.py via unet_2d_ZeroSFT_blocks.py:2804-2874) run on bf16 activations */
int iir_groupnorm_nhwc(const void* X, int64_t ldx, void* Y, int64_t ldy, int32_t R, int32_t HW, int32_t C,
                       int32_t groups, const void* gamma, const void* beta, float eps, int32_t silu,
                       void* workspace, int64_t workspace_bytes, int32_t dtype, void* stream);
int64_t iir_groupnorm_workspace_bytes(int32_t R, int32_t groups);

/* nn.LayerNorm (module/min_sdxl.py:534-538; resampler.py:15,43-44,98) and AdaLayerNorm's
 * LN(x)*(1+scale)+shift (attention_processor.py:24-25).  gamma/beta/shift/scale may be NULL.
 * transposed == 2 stores y as fp8-E4M3 BYTES (Y a byte matrix, ldy in bytes: the A operand of an all-fp8 GEMM, iir_gemm_desc.a_fp8);
 * transposed == 1 stores y^T (used to emit the IP-adapter V^T image directly):
 *   Y[c][ (row / tr_rows) * tr_bstride + row % tr_rows ], row stride ldy. */
int iir_layernorm_f16(const void* X, int64_t ldx, void* Y, int64_t ldy, int32_t rows, int32_t C, const void* gamma,
                      const void* beta, float eps, const void* shift, const void* scale, int64_t ldmod,
                      int32_t rows_per_mod, int32_t transposed, int32_t tr_rows, int64_t tr_bstride, void* stream);

/* Every adaLN of one UNet forward in one launch.  The IP-adapter tokens are step invariant, so the raw to_k_ip /
 * to_v_ip projections are hoisted; what is left per step and per TA-IP block is AdaLayerNorm(raw, temb)
 * (module/ip_adapter/attention_processor.py:14-26 as called at :1173-1176): LN without affine, then * (1 + scale) + shift.
 * jobs_dev: DEVICE array of njobs records; all jobs share rows, eps, ldmod (row stride of the shift / scale matrix),
 * rows_per_mod, tr_rows and tr_bstride (meaning as in iir_layernorm_f16); max_C = largest C in the table. */
typedef struct iir_adaln_job {
    const void* X; void* Y; const void* shift; const void* scale;
    int64_t ldx, ldy;
    int32_t C, transposed;
} iir_adaln_job;
int iir_adaln_batch_f16(const iir_adaln_job* jobs_dev, int32_t njobs, int32_t rows, int32_t max_C, float eps, int64_t ldmod,
                        int32_t rows_per_mod, int32_t tr_rows, int64_t tr_bstride, void* stream);

/* In-place row softmax (fp32 math) of an fp16 matrix, cols <= 16384: the score matrix of the VAE mid-block
 * attention (1 head of dim 512, T = (H/8)*(W/8) tokens; torch.softmax inside F.scaled_dot_product_attention,
 * module/ip_adapter/attention_processor.py:394 as used by module/unet/unet_2d_ZeroSFT_blocks.py:776-790). */
int iir_softmax_rows_f16(void* X, int64_t ld, int32_t rows, int32_t cols, void* stream);
/* softmax over the columns of fp32 scores S[rows][cols] (cols % 4 == 0, <= 16384) written as fp16 / bf16 probabilities P:
 * the VAE mid-block attention (Attention(heads=1), unet_2d_ZeroSFT_blocks.py:776-790) keeps its scores in fp32 through the
 * softmax like the reference's upcast VAE (pipelines/sdxl_instantir.py:984-1001) */
int iir_softmax_rows_f32(const float* S, int64_t lds, void* P, int64_t ldp, int32_t rows, int32_t cols, int32_t dtype, void* stream);

/* Timesteps: module/min_sdxl.py:205-224.  out[r][col_off + v*dim + ...] = [cos | sin](vals[r][v] * w_k). */
int iir_sinusoid_f16(const float* vals, int32_t n_vals, int32_t rows, int32_t dim, void* out, int64_t ldo,
                     int32_t col_off, void* stream);
int iir_silu_f16(const void* x, void* y, int64_t n, void* stream);

/* dst[m][dst_off + c] = src[m][c] + add[m][c] * add_scale[m / rows_per_scale]: torch.cat of skips
 * (module/min_sdxl.py:712) with the aggregator residual scaling/add folded in
 * (pipelines/sdxl_instantir.py:1602-1603). add / add_scale may be NULL. */
int iir_copy_add_f16(const void* src, int64_t lds, void* dst, int64_t ldd, int64_t dst_off, int64_t M, int32_t C,
                     const void* add, int64_t lda, const float* add_scale, int32_t rows_per_scale, void* stream);

/* fp32 NCHW latents <-> fp16 NHWC rows (pipelines/sdxl_instantir.py:1503 cat([latents]*2) = rep 2). */
int iir_pack_latent(const float* x, int32_t B, int32_t C, int32_t HW, void* out, int64_t ldo, int32_t rep, float scale,
                    void* stream);
int iir_unpack_latent(const void* in, int64_t ldi, int32_t R, int32_t C, int32_t HW, float* out, void* stream);
/* the same with the 16-bit side as fp16 or bf16 (`dtype`): image / latent hand-over of the bf16 VAE */
int iir_pack_latent_t(const float* x, int32_t B, int32_t C, int32_t HW, void* out, int64_t ldo, int32_t rep, float scale,
                      int32_t dtype, void* stream);
int iir_unpack_latent_t(const void* in, int64_t ldi, int32_t R, int32_t C, int32_t HW, float* out, int32_t dtype, void* stream);

/* CFG + main scheduler step (pipelines/sdxl_instantir.py:1619-1633).  coef = device fp32[8]:
 * {guidance, sqrt(1-abar_t), sqrt(abar_t), k_x0, k_x, k_eps, k_noise, 0}; prev = k_x0*x0 + k_x*x + k_eps*eps
 * + k_noise*noise with x0 = (x - sqrt(1-abar_t)*eps)/sqrt(abar_t).  DDPM: k_eps = 0; DDIM: k_x = 0. */
int iir_sched_step(const void* eps_nhwc, int64_t lde, int32_t B, int32_t C, int32_t HW, int32_t cfg, const float* coef,
                   const float* x, const float* noise, float* prev, float* x0_out, float* eps_out,
                   const float* eps_factor, void* stream);
/* rescale_noise_cfg, pipelines/sdxl_instantir.py:181-192 (used at :1623-1626 when guidance_rescale > 0): per image
 * factor[b] = phi * std(eps_text) / std(eps_cfg) + (1 - phi) over (C,H,W); iir_sched_step multiplies the guided
 * eps of image b by eps_factor[b] (NULL = 1).  coef[0] = guidance scale, as for iir_sched_step. */
int iir_cfg_rescale_factor(const void* eps_nhwc, int64_t lde, int32_t B, int32_t C, int32_t HW, const float* coef,
                           float guidance_rescale, float* factor, void* stream);

/* LCMSingleStepScheduler.step, schedulers/lcm_single_step_scheduler.py:455-484.
 * coef = device fp32[4] {sqrt(1-abar_t), sqrt(abar_t), c_out, c_skip}. */
int iir_lcm_step(const void* eps_nhwc, int64_t lde, int32_t B, int32_t rep, int32_t C, int32_t HW, const float* coef,
                 const float* x, void* out_nhwc, int64_t ldo, float* out_nchw, void* stream);

/* Scheduler `.step()` on fp32 NCHW tensors (same linear form as iir_sched_step, no CFG), and
 * a*x + b*y for add_noise (schedulers/lcm_single_step_scheduler.py:492-513; pipelines/sdxl_instantir.py:931-939). */
int iir_sched_step_f32(const float* eps, const float* x, const float* noise, const float* coef, int64_t n, float* prev,
                       float* x0_out, void* stream);
int iir_axpby_f32(const float* x, const float* y, const float* coef, int64_t n, float* out, void* stream);

/* Weight prefetch: touches every 128-byte line of [p, p+bytes) with `blocks` workgroups so the range
 * sits in the 256 MiB Infinity Cache when the GEMM/conv that streams it starts (the reference has no
 * counterpart: PyTorch streams each layer's weights cold from HBM). */
int iir_prefetch(const void* p, int64_t bytes, int32_t blocks, void* stream);

/* Seam blending of the tiled VAE decode, in place on tile `b` (fp32 NCHW tiles, `planes` = batch*channels):
 * blend_v / blend_h of module/diffusers_vae/autoencoder_kl.py:311-321. */
int iir_blend_tiles_f32(const float* a, float* b, int32_t planes, int32_t Ha, int32_t Wa, int32_t Hb, int32_t Wb,
                        int32_t extent, int32_t vertical, void* stream);

int iir_transpose_f16(const void* in, int64_t ldi, int32_t rows, int32_t cols, void* out, int64_t ldo, int32_t rows_pad,
                      void* stream);
/* Measurement hook (bench.py roofline leg; no reference counterpart): arm a start/stop event pair and the NEXT
 * iir_gemm_f16 / iir_conv2d_nhwc_f16 / iir_attention_d64_f16 launch issued from this thread stamps them with the
 * kernel's own begin / end timestamps on its stream (hipExtLaunchKernelGGL).  After the stream has drained,
 * iir_timing_elapsed_us returns that kernel's duration.  Not usable while the stream is being captured. */
void* iir_timing_event_create(void);
void iir_timing_event_destroy(void* event);
int iir_timing_arm(void* start_event, void* stop_event);
int iir_timing_elapsed_us(void* start_event, void* stop_event, float* us);

int iir_abi_version(void);

#ifdef __cplusplus
}
#endif
#endif
