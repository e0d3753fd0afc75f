/*
 * st2.h -- C ABI of libst2_hip.so, the MI355X (gfx950) kernel library behind the
 * StyleTTS 2 text->waveform hot path (style-diffusion sampler, AdaIN acoustic decoder,
 * iSTFTNet / HiFi-GAN vocoder).
 *
 * The reference (yl4579/StyleTTS2) has no FFI of its own: the hot path is a chain of
 * torch ATen ops issued from nn.Module.forward.  Each entry point
 * below therefore replaces one *class* of ATen calls; the reference call sites it stands
 * in for are cited per function (paths relative to the reference tree).
 *
 * Conventions
 *   - every pointer is a DEVICE pointer to fp32 unless stated otherwise; the caller owns
 *     all memory; nothing is allocated or freed inside the library (one exception: the 64-byte
 *     host-mapped status word of st2_status(), allocated once per process);
 *   - tensors are row-major "NCL": element (b, c, l) of tensor t lives at
 *     t + b * t_bs + c * t_cs + l   (strides in ELEMENTS; bs = batch, cs = channel);
 *   - `stream` is a hipStream_t passed as void* (0 = the null stream); all work is
 *     stream-ordered and asynchronous; no entry point synchronises;
 *   - return value: 0 on success, non-zero on error; st2_last_error() gives the message
 *     of the last failing call on this thread.  No C++ exception crosses the ABI;
 *   - devices: every entry point works on the CURRENT HIP device of the calling thread; per-kernel settings
 *     (dynamic LDS limits, co-residency capacities, the autotuner's table) are kept per device ordinal, so one
 *     process may drive several GPUs.  The measurement hooks (st2_conv_timing*, st2_conv_tune*) are process-wide
 *     and meant for one driving thread.
 */
#ifndef ST2_H
#define ST2_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ST2_ABI_VERSION 22

/* ---- library ---------------------------------------------------------------------- */
int st2_abi_version(void);
const char* st2_last_error(void);
/* Fills name (<= cap bytes) with the gcnArchName of device `dev`, returns CU count or <0. */
int st2_device_info(int dev, char* name, int cap);

/* ---- sticky device-side status --------------------------------------------------------------------------------- *
 * Conditions a kernel can only detect on the device are OR-ed into ONE process-wide status word that lives in
 * host-mapped (pinned) memory -- the single allocation this library makes, 64 bytes, on the first call that needs
 * it -- so the host reads it without a device synchronisation; a bit is visible once the kernel that raised it has
 * completed.  Bits:
 *   ST2_STATUS_F16_RANGE     an operand of a split-f16 conv exceeded the f16 range after scaling (|x * x_scale| >
 *                            65504) and was clamped to +-65504 (st2_act_split, st2_conv1d_f16s): the result is finite
 *                            but not the fp32 conv's; re-run that layer with a smaller x_scale or on st2_conv1d.  A NaN
 *                            operand raises the same bit (ABI 18: the clamp maps it to -65504);
 *   ST2_STATUS_LSTM_TIMEOUT  a bounded spin of st2_lstm_bidir_coop expired (a group's workgroups were not
 *                            co-resident in time): the outputs of that call are invalid;
 *                            (st2_lstm_bidir_coop called directly; the launch plans use st2_lstm_bidir_coop_recovering, which
 *                            repairs the call in-stream and raises ST2_STATUS_LSTM_RECOVERED instead);
 *   ST2_STATUS_DURATION_SUM  a row of the durations handed to st2_expand_by_durations does not sum to T (caller-supplied
 *                            durations with a wrong `total_frames`): frames past the sum repeat the last phoneme.
 * st2_status(clear != 0) returns the word and atomically clears it.  Returns < 0 if no HIP device is usable. */
#define ST2_STATUS_F16_RANGE 1
#define ST2_STATUS_LSTM_TIMEOUT 2
#define ST2_STATUS_DURATION_SUM 4
#define ST2_STATUS_LSTM_RECOVERED 8  /* informational: a cooperative BiLSTM group timed out and st2_lstm_bidir_coop_recovering
                                        re-ran the call on the single-CU kernel -- the outputs are VALID, latency was lost */
int st2_status(int clear);

/* ---- fused Conv1d (implicit GEMM on v_mfma_f32_32x32x2_f32, exact fp32) ------------ *
 * y[b,co,l] = epi( bias[co] + sum_{ci,t} W[co,ci,t] * pro(x)[b,ci, l + t*dil - pad_left] )
 * with zero padding applied AFTER the prologue, stride 1.
 * Replaces: F.conv1d behind every weight-normed nn.Conv1d on the path
 *   Modules/istftnet.py:68-74 (AdaINResBlock1 convs1/convs2), :445-448 (AdainResBlk1d),
 *   :319-322,364 (ConvTranspose1d via its polyphase form, see st2_convt_interleave),
 *   :377 (conv_post), Modules/hifigan.py:65-74,292-294,344, models.py:255-263,
 *   and nn.Linear / 1x1 Conv1d of the denoiser, Modules/diffusion/modules.py:256-261,
 *   484-490,317-324 (tokens laid out channel-major so a Linear is a k=1 conv).
 */
enum st2_prologue {
  ST2_PRO_NONE = 0,
  ST2_PRO_LEAKY = 1,        /* x>=0 ? x : slope*x                       istftnet.py:360,376 */
  ST2_PRO_ADAIN_LEAKY = 2,  /* leaky((1+g)*(x-mean)*rstd + b)           istftnet.py:441-447 */
  ST2_PRO_ADAIN_SNAKE = 3,  /* u=(1+g)*(x-mean)*rstd+b; u+sin(a u)^2/a  istftnet.py:68-72   */
  ST2_PRO_SNAKE = 4,        /* x + sin(a x)^2 / a                       hifigan.py:329,343  */
  ST2_PRO_COLNORM = 5       /* (x-mean[b,l])*rstd[b,l]*G[b,c]+Bt[b,c]   modules.py:18-38,556 */
};
enum st2_epilogue_act {
  ST2_ACT_NONE = 0,
  ST2_ACT_GELU = 1,     /* exact erf GELU                                modules.py:484-490 */
  ST2_ACT_EXP_SIN = 2,  /* rows < act_split: exp, rows >= act_split: sin istftnet.py:378-379 */
  ST2_ACT_TANH = 3,     /*                                               hifigan.py:345     */
  ST2_ACT_LEAKY = 4,    /* leaky(act_slope)                                                  */
  ST2_ACT_GELU_TANH = 5 /* 0.5 x (1 + tanh(sqrt(2/pi)(x + 0.044715 x^3))): HF "gelu_new", the ALBERT FFN of PL-BERT
                           (Utils/PLBERT/util.py:19-20 -> transformers AlbertConfig.hidden_act)               */
};

typedef struct st2_conv_desc {
  /* geometry */
  int32_t B, C_in, C_out, L_in, L_out, ks, dil, pad_left;
  /* input */
  const float* x; int64_t x_bs; int32_t x_cs;
  /* weights packed K-major: wt[(ci*ks + t) * w_ld + co], w_ld >= C_out, w_ld % 4 == 0 */
  const float* wt; int32_t w_ld;
  const float* bias;                 /* [C_out] or NULL */
  /* output */
  float* y; int64_t y_bs; int32_t y_cs;
  /* prologue */
  int32_t pro;                       /* enum st2_prologue */
  float slope;                       /* LEAKY / ADAIN_LEAKY */
  const float* stats;                /* ADAIN_*: [B][C_in][2] = (mean, rstd); COLNORM: [B][L_in][2] */
  const float* gamma; const float* beta;  /* ADAIN_*: gamma[b*gb_bs + c]; the kernel applies (1+gamma)
                                             COLNORM: G = gamma[b*gb_bs+c] (+1 if gamma_plus_one) */
  int64_t gb_bs; int32_t gamma_plus_one;
  const float* alpha;                /* SNAKE modes: [C_in] */
  /* epilogue: v = acc + bias; v += res; v += res2; v /= div; v = act(v) */
  const float* res;  int64_t res_bs;  int32_t res_cs;  int32_t res_shift;  /* reads res[b,co,l>>res_shift] */
  const float* res2; int64_t res2_bs; int32_t res2_cs;
  float div;                         /* 1.0f = none */
  int32_t act; int32_t act_split; float act_slope;
  /* st2_conv1d_f16s only: split-f16 packed weights (see below); ignored by st2_conv1d */
  const void* wq; int32_t wq_co_pad; int32_t wq_cin_pad;
  float x_scale;                     /* power of two applied to pro(x) before the hi/lo split (by rule 8 after a normalising
                                        prologue, else 1; per layer after st2_calibrate) */
  float out_scale;                   /* applied to the accumulator first: 1 / (x_scale * weight scale), or 1 / x_scale
                                        when w_row_scale carries the per-row weight scales */
  const float* w_row_scale;          /* [wq_co_pad] or NULL: 1 / (per-output-row weight scale); the accumulator of row co
                                        is multiplied by out_scale * w_row_scale[co] (both powers of two: exact) */
  /* st2_conv1d_xs only: pre-activated, pre-split input planes written by st2_act_split (x/pro/stats/... unused) */
  const void* xs; int32_t xs_cg; int32_t xs_lp; int32_t xs_halo;
  /* st2_conv1d_xs only, optional: per-tile InstanceNorm partial sums of the STORED output,
     part[((b*C_out + co)*part_nt + l/128)*2 + {0,1}] = (sum, sum of squares) of (y - shift) over the 128 columns of that tile,
     shift = the tile's first stored value y[b][co][128*(l/128)], itself stored at part[B*C_out*part_nt*2 + (b*C_out + co)*part_nt +
     l/128]: the buffer holds B*C_out*part_nt*3 floats (see st2_stats_finalize) */
  float* part; int32_t part_nt;
  /* st2_conv1d_xs only (ABI v20): columns per partial-sum slot, 0 = 128.  A small-grid launch (fewer workgroups than CUs: one
     utterance) runs 128 x 64 or 128 x 32 tiles and emits its sums per 64 / 32 columns: the caller asks
     st2_conv1d_xs_part_cols(d) BEFORE sizing `part` (part_nt >= ceil(L_out / part_cols)) and passes the answer here; a caller
     that leaves it 0 keeps the 128-column tiles and the 128-column slots.  st2_stats_finalize sums whatever slots there are. */
  int32_t part_cols;
  /* st2_conv1d_f16s only, optional: workspace for split-K launches.  A layer whose grid leaves most of the chip idle and
     whose k loop is long (C_in >= 8 chunks, < 128 workgroups: the 1024 -> 2048 Linears of the denoiser over the ~100
     tokens of one utterance) runs as up to 8 K slices per tile + a fixed-order reduction that applies the epilogue; the
     split is a function of the geometry alone (every plan picks the same one: results are reproducible bit for bit) and
     needs st2_conv1d_f16s_splitk_bytes(d) bytes here (the slices are stored in the accumulator layout of whole tiles: ksplit * B *
     padded C_out * padded L_out * 4).  NULL / too small = no split. */
  void* splitk_ws; int64_t splitk_ws_bytes;
} st2_conv_desc;

int st2_conv1d(const st2_conv_desc* d, void* stream);

/* ---- fused Conv1d on the f16 matrix pipe, fp32-class accuracy ("f16 hi/lo split") ---- *
 * Same contract, prologues and epilogues as st2_conv1d.
 * Each operand v is carried as f16 hi + f16 lo of v*scale and the product is evaluated as
 * hi*hi + hi*lo + lo*hi by three v_mfma_f32_32x32x16_f16 into one fp32 accumulator (the dropped lo*lo
 * term is 2^-22 relative).  Through the whole decoder the waveform differs from an fp64 evaluation by
 * 2.6e-7 RMS, the fp32 ATen path by 2.4e-7; the rate ceiling is 5.3x that of the exact-fp32 MFMA.
 * Weights are pre-split once per load (the d.wt field is unused):
 *   wq[((ci/16 * ks + t) * 2 + (ci%16)/8) * wq_co_pad + co][16 halves] = hi[0..7] | lo[0..7]
 *   over the 8 channels ci..ci+7 of W[co, ., t] * w_scale[co], zero padded to wq_cin_pad input channels
 *   (a multiple of st2_conv1d_f16s_chunk(ks)) and wq_co_pad rows (a multiple of
 *   st2_conv1d_f16s_co_block(C_out)); w_scale[co] is the power of two that puts max_{ci,t}|W[co]| in [2^13, 2^14)
 *   (per OUTPUT ROW, so a checkpoint whose rows differ by many octaves keeps every row's lo halves in the normal
 *   f16 range; d.w_row_scale = 1 / w_scale[co]).  Operands are clamped to the f16 range (ST2_STATUS_F16_RANGE).
 * Replaces the same reference call sites as st2_conv1d (decoder / vocoder convolutions, denoiser Linears). */
int st2_conv1d_f16s(const st2_conv_desc* d, void* stream);
/* Bytes of d.splitk_ws the launch described by *d would use (0 = this geometry is not split): the caller allocates that
 * much (any alignment >= 16) and sets d.splitk_ws / d.splitk_ws_bytes before calling st2_conv1d_f16s. */
int64_t st2_conv1d_f16s_splitk_bytes(const st2_conv_desc* d);
int st2_conv1d_f16s_chunk(int ks);        /* input-channel padding granule of the packed weight */
int st2_conv1d_f16s_co_block(int C_out);  /* output-channel padding granule of the packed weight */
/* Measurement hook (process-wide): which build of the fused kernel a launch takes.  0 (default) = by rule: the
 * warp-specialised persistent build (512-thread workgroups, one per CU: four waves stage + activate the input, four issue
 * the MFMAs; bitwise the same results) for AdaIN + Snake convs with k = 3, C_out <= 64 and at least 512 tiles, the one-role
 * build otherwise; 1 = one-role build always; 2 = warp-specialised for every layer it can run (k = 3 / 7 / 11, C_out <= 64). */
void st2_conv1d_f16s_set_variant(int variant);
/* Measurement hook (process-wide): the depth of the split-K rule -- at most `max_slices` K slices per tile (1..32), each of at
 * least `min_chunks` input-channel chunks (1..16); out-of-range values restore the default (8, 4).  Every plan of a process sees
 * the same rule (results stay reproducible within the process); the default is what the tests and the bench run. */
void st2_conv1d_f16s_set_splitk(int max_slices, int min_chunks);
/* sizeof(st2_conv_desc) as the library was compiled: lets a binding verify its struct mirror. */
int st2_sizeof_conv_desc(void);

/* ---- activation pass + pure MFMA conv ("xs" path) -------------------------------------------------------------- *
 * The fused prologue of st2_conv1d_f16s costs ~40 VALU instructions per staged element inside the MFMA kernel.  The
 * xs path runs it ONCE per element in an HBM-bound pass instead and hands the conv pre-split f16 operands:
 *
 *   st2_act_split:  xs[b][plane][cg][pos][e] (f16; plane 0 = hi, 1 = lo; 16-byte slots of 8 channels) =
 *                   split( x_scale * pro(x)[b][cg*8 + e][pos - halo] ),  zero for pos - halo outside [0, L) and for
 *                   channels >= C.  Same prologue arithmetic (op for op) as st2_conv1d_f16s.  cg in [0, xs_cg),
 *                   pos in [0, Lp); xs_cg*8 >= C rounded up to st2_conv1d_f16s_chunk(ks) of the consuming conv.
 *                   gb_seg > 0 (ST2_PRO_COLNORM only): the gamma / beta row of position l is l / gb_seg instead of the
 *                   batch index b -- the token-merged view [1][C][B*N] of B utterances of N tokens whose LayerNorm
 *                   affine is per utterance (the multispeaker denoiser's AdaLayerNorm, Modules/diffusion/modules.py:
 *                   104-118), so that their q / kv projections run as ONE GEMM over B*N columns.
 *   st2_conv1d_xs:  same GEMM, weights (d.wq ...) and epilogue as st2_conv1d_f16s on those planes: chunks are staged
 *                   global -> LDS as plain 16-byte copies, no per-element arithmetic in the MFMA kernel.  Requires
 *                   pad_left <= xs_halo and Lp large enough for the last tile (checked).  If d.part != NULL the
 *                   epilogue also emits per-128-column partial (sum, sumsq) of the stored output so the next
 *                   layer's InstanceNorm statistics need no extra read of the tensor:
 *   st2_stats_finalize: stats[row] = (mean, 1/sqrt(var+eps)) from part[row][nt][2] (fp64, fixed order), rows = B*C.
 *                   ABI v20: the partial sums are SHIFTED -- slot i holds (sum, sum of squares) of (y - shift_i) over its
 *                   columns, shift_i = the slot's first stored value, written by the producer behind the sums (part = float2
 *                   [rows][nt], then float [rows][nt]); the finaliser combines the slots with Chan's formula (cols = columns
 *                   per slot: d.part_cols or 128 for the convs, 1024 for st2_convt_interleave_stats).  Unshifted E[x^2] - mean^2 from fp32
 *                   partial sums loses rstd of a channel whose mean dominates its variance (|mean| / std = 100: 1e-4 relative,
 *                   three decades above the reference's two-pass fp32 reduction).
 * Replaces the same reference call sites as st2_conv1d_f16s + st2_instnorm_stats. */
int st2_act_split(const float* x, int64_t x_bs, int32_t x_cs, int32_t B, int32_t C, int32_t L,
                  int32_t pro, float slope, const float* stats, const float* gamma, const float* beta, int64_t gb_bs,
                  int32_t gb_seg, int32_t gamma_plus_one, const float* alpha, float x_scale,
                  void* xs, int32_t xs_cg, int32_t Lp, int32_t halo, void* stream);
int st2_conv1d_xs(const st2_conv_desc* d, void* stream);
/* Columns per partial-sum slot (128, 64 or 32) the launch described by *d would use when its caller opts into the small-grid
 * builds (d.part_cols): a function of the geometry alone -- every plan gets the same answer, results are reproducible bit for
 * bit.  k = 3 / 7 / 11 launches of up to three utterances: 32 below ~100 tiles of 128 x 128, 64 up to ~900 (k = 3) / 340 (k = 7,
 * 11); 128 otherwise (y is bitwise the same either way: st2_conv1d_xs_impl.h). */
int st2_conv1d_xs_part_cols(const st2_conv_desc* d);
int st2_stats_finalize(const float* part, int32_t rows, int32_t nt, int32_t L, float eps, float* stats, int32_t cols, void* stream);

/* ---- small direct Conv1d (any stride, tiny C_in): noise convs, F0/N down-convs ------ *
 * y[b,co,l] = bias[co] + sum_{ci,t} w[co,ci,t] * x[b,ci, l*stride + t - pad]   (plain OIK weights)
 * Replaces: Modules/istftnet.py:332-339,361 (noise_convs), :486-488,511-512 (F0_conv/N_conv),
 *           Modules/hifigan.py:296-303,330, models.py:464-465 (F0_proj/N_proj).
 */
int st2_conv1d_direct(const float* x, int64_t x_bs, int32_t x_cs,
                      const float* w, const float* bias,
                      float* y, int64_t y_bs, int32_t y_cs,
                      int32_t B, int32_t C_in, int32_t C_out, int32_t L_in, int32_t L_out,
                      int32_t ks, int32_t stride, int32_t pad, void* stream);

/* ---- polyphase split of a strided Conv1d input (kernel = 2*stride, the noise_convs of both vocoders) -------- *
 * xp[b][ci*stride + r][u] = x[b][ci][u*stride + r - pad]  for u in [0,Lu), r in [0,stride); 0 outside [0,L_in).
 * With weights.polyphase_strided_conv() the strided conv becomes a stride-1, k=2 st2_conv1d over C*stride
 * channels (L_out = Lu - 1) on the matrix pipe.
 * Replaces the im2col inside F.conv1d(stride=s): Modules/istftnet.py:332-336,361, Modules/hifigan.py:296-300,330. */
int st2_phase_split(const float* x, int64_t x_bs, int32_t x_cs, int32_t B, int32_t C, int32_t L_in,
                    int32_t stride, int32_t pad, float* xp, int64_t p_bs, int32_t p_cs, int32_t Lu, void* stream);

/* ---- InstanceNorm1d statistics ------------------------------------------------------ *
 * stats[b][c] = (mean, 1/sqrt(biased_var + eps)) over l in [0,L).  fp64 accumulation in a
 * fixed tree order (bitwise reproducible).  Replaces nn.InstanceNorm1d inside AdaIN1d,
 * Modules/istftnet.py:15-25.
 */
int st2_instnorm_stats(const float* x, int64_t x_bs, int32_t x_cs, int32_t B, int32_t C, int32_t L,
                       float eps, float* stats /* [B][C][2] */, void* stream);

/* LayerNorm statistics over the CHANNEL axis of an NCL tensor: stats[b][l] = (mean, rstd) over c.
 * Replaces F.layer_norm's reduction, Modules/diffusion/modules.py:18-38,556-557. */
int st2_colnorm_stats(const float* x, int64_t x_bs, int32_t x_cs, int32_t B, int32_t C, int32_t L,
                      float eps, float* stats /* [B][L][2] */, void* stream);

/* ---- style FC: h[b][j] = act(bias[j] + sum_k s[b][k] * wt[k*J + j])  (all AdaIN fc's of a module
 * concatenated along j; act is an st2_epilogue_act, NONE or GELU).  Replaces the per-AdaIN nn.Linear,
 * Modules/istftnet.py:19,22-24, and the per-utterance mapping MLPs of the denoiser,
 * Modules/diffusion/modules.py:333-357,363-384. */
int st2_style_fc(const float* s, int32_t B, int32_t K, const float* wt, const float* bias,
                 int32_t J, int32_t act, float* h, void* stream);

/* ---- ConvTranspose1d finishing pass ------------------------------------------------- *
 * phases[b][r*C + co][q] (q in [0,Lq)) is the polyphase GEMM output of st2_conv1d;
 * out[b,co,l] = bias[co] + phases[b][((l+pad)%s)*C + co][(l+pad)/s] + add[b,co,l]
 * for l in [0,L_raw); if reflect_left the result is shifted right by one sample and
 * out[.,.,0] = value at raw index 1 (nn.ReflectionPad1d((1,0))).  L_out = L_raw + reflect_left.
 * Replaces: Modules/istftnet.py:364-368 (ups[i], reflection_pad, + x_source),
 *           Modules/hifigan.py:333-334.
 */
int st2_convt_interleave(const float* phases, int64_t p_bs, int32_t p_cs, int32_t Lq,
                         const float* bias, const float* add, int64_t a_bs, int32_t a_cs,
                         float* out, int64_t o_bs, int32_t o_cs,
                         int32_t B, int32_t C, int32_t stride, int32_t pad, int32_t L_raw,
                         int32_t reflect_left, void* stream);

/* Same, additionally emitting per-tile InstanceNorm partial sums of the stored output (tiles of 1024 positions):
 * part[((b*C + co)*part_nt + l/1024)*2 + {0,1}] = (sum, sum of squares) of (out - shift), shift = out[b][co][1024*(l/1024)] stored at
 * part[B*C*part_nt*2 + (b*C + co)*part_nt + l/1024] (B*C*part_nt*3 floats in all); part may be NULL.  Feed to
 * st2_stats_finalize: the AdaIN that follows the up-sampling needs no pass over the tensor. */
int st2_convt_interleave_stats(const float* phases, int64_t p_bs, int32_t p_cs, int32_t Lq,
                               const float* bias, const float* add, int64_t a_bs, int32_t a_cs,
                               float* out, int64_t o_bs, int32_t o_cs,
                               int32_t B, int32_t C, int32_t stride, int32_t pad, int32_t L_raw,
                               int32_t reflect_left, float* part, int32_t part_nt, void* stream);

/* ---- AdaIN + LeakyReLU + depthwise ConvTranspose1d(k3,s2,p1,op1) ("pool") ------------ *
 * Replaces Modules/istftnet.py:441-444 with upsample=True (weights w[c][3], bias[c]). */
int st2_adain_leaky_pool(const float* x, int64_t x_bs, int32_t x_cs,
                         const float* stats, const float* gamma, const float* beta, int64_t gb_bs,
                         float slope, const float* w, const float* bias,
                         float* y, int64_t y_bs, int32_t y_cs,
                         int32_t B, int32_t C, int32_t L, void* stream);

/* ---- harmonic source (SineGen + SourceModuleHnNSF) ---------------------------------- *
 * f0 [B][F] frame-rate F0 (Hz), U samples per frame, H harmonics (9).
 * noise [B][F*U][H] standard normal draws (the reference's randn_like, istftnet.py:242).
 * lin_w [H], lin_b [1]: l_linear.  out [B][F*U] = tanh(linear(sine_waves)).
 * phase_scratch: [B][H][F] floats.  Bit-faithful to ATen-CPU op order.
 * Replaces Modules/istftnet.py:141-247,283-297,352-354 (identical code hifigan.py:112-268).
 */
int st2_har_source(const float* f0, int32_t B, int32_t F, int32_t U, int32_t H,
                   const float* noise, const float* lin_w, const float* lin_b,
                   float sine_amp, float noise_std, float voiced_threshold, float sample_rate,
                   float* phase_scratch, float* out, void* stream);

/* ---- STFT of the harmonic source (n_fft = win = N, hop, periodic Hann, center/reflect) *
 * har[b][k][m] = |X_k|, har[b][N/2+1+k][m] = atan2(Im, Re), k in [0,N/2], m in [0, L/hop].
 * Replaces Modules/istftnet.py:91-97,355-357. */
int st2_stft_mag_phase(const float* x, int32_t B, int32_t L, int32_t n_fft, int32_t hop,
                       float* har, int64_t har_bs, int32_t har_cs, void* stream);

/* ---- iSTFT synthesis: spec/phase [B][N/2+1][M] each (spec = exp(.), phase = sin(.) already
 * applied by the conv_post epilogue) -> wave [B][hop*(M-1)].
 * Replaces Modules/istftnet.py:99-104,380. */
int st2_istft(const float* sp, int64_t sp_bs, int32_t sp_cs, int32_t B, int32_t M,
              int32_t n_fft, int32_t hop, float* wave, int64_t wave_bs, void* stream);

/* ---- denoiser helpers (tokens channel-major: x[b][c][n]) ------------------------------ */
/* Multi-head attention without mask: q,k,v [B][H*D][N] -> o [B][H*D][N], softmax(q^T k * scale) v.
 * Replaces Modules/diffusion/modules.py:523-535. */
int st2_attention(const float* q, const float* k, const float* v, int64_t bs, int32_t cs,
                  float* o, int64_t o_bs, int32_t o_cs,
                  int32_t B, int32_t H, int32_t D, int32_t N, float scale, void* stream);

/* Same with key padding: keys m >= key_len[b] are excluded from the softmax of every query of utterance b (the HF
 * additive -inf attention mask of a right-padded batch, Utils/PLBERT/util.py:8-11 -> AlbertAttention); key_len may be
 * NULL (= st2_attention).  key_len: int32 [B] on the device, every entry >= 1. */
int st2_attention_keylen(const float* q, const float* k, const float* v, int64_t bs, int32_t cs,
                         float* o, int64_t o_bs, int32_t o_cs,
                         int32_t B, int32_t H, int32_t D, int32_t N, float scale, const int32_t* key_len, void* stream);

/* ---- LayerNorm over channels, applied: y[b,c,l] = act( (x[b,c,l] - mean[b,l]) * rstd[b,l] * G[b,c] + Bt[b,c] ),
 * G = gamma[b*gb_bs + c] (+1 if gamma_plus_one), stats [B][L][2] from st2_colnorm_stats; act NONE or LEAKY(slope);
 * positions l >= len[b] are written as 0 when len != NULL (the masked_fill_ after each block of the text encoders).
 * Replaces nn.LayerNorm / AdaLayerNorm applications whose result is needed as a tensor (residual inputs): ALBERT
 * post-LN blocks, models.py:270-282,308-312 (TextEncoder), :418-438,547-556 (DurationEncoder). */
int st2_colnorm_apply(const float* x, int64_t x_bs, int32_t x_cs, const float* stats, const float* gamma,
                      const float* beta, int64_t gb_bs, int32_t gamma_plus_one, int32_t act, float slope,
                      const int32_t* len, float* y, int64_t y_bs, int32_t y_cs,
                      int32_t B, int32_t C, int32_t L, void* stream);

/* ---- bidirectional LSTM recurrence (hidden size H = 256) ------------------------------------- *
 * G [B][2*4H][N]: input projections W_ih x_t + b_ih + b_hh for both directions (rows 0..4H-1 forward,
 * 4H..8H-1 reverse; PyTorch gate order i,f,g,o), produced by st2_conv1d (k=1).  whh_t [2][H][4H] is
 * W_hh transposed per direction.  lengths [B] int32 (NULL = all N): packed-sequence semantics, outputs
 * beyond the length are zero.  Y [B][2H][N] (rows 0..H-1 forward h_t, H..2H-1 reverse).
 * Replaces nn.LSTM(bidirectional=True) + pack/pad: models.py:300,314-327 (TextEncoder),
 * :450,523-528,545-566 (duration LSTM / DurationEncoder), :453,498 (shared F0/N LSTM). */
int st2_lstm_bidir(const float* G, int64_t g_bs, int32_t g_cs, const float* whh_t, const int32_t* lengths,
                   int32_t B, int32_t H, int32_t N, float* Y, int64_t y_bs, int32_t y_cs, void* stream);

/* Cooperative form of the same recurrence: W_hh stays in registers, split over 8 workgroups (CUs) per group of up to
 * 8 utterances of one direction, which exchange the new hidden state once per step through `scratch` (agent-scope
 * release/acquire on a monotonic counter).  ~5x lower latency per step than st2_lstm_bidir, same results up to fp32
 * summation order.  `scratch` (256-byte aligned, >= st2_lstm_coop_scratch_bytes(B) bytes, contents irrelevant) is
 * zeroed and used by the call; scratch[0..3] holds an int32 status afterwards (0 = ok, 1 = a bounded spin timed out
 * and the outputs are invalid).  st2_lstm_coop_scratch_bytes returns 0 when B is too large for one co-resident
 * launch (B > 48): use st2_lstm_bidir. */
int64_t st2_lstm_coop_scratch_bytes(int32_t B);
/* Hand-off variant (process-wide): 2 (default) = the data is the flag: every new hidden value travels as ONE 8-byte
 * agent-scope store {step tag, value} and the consumers re-read their granules until every tag matches -- one fabric
 * round trip per step, no counter, no cache maintenance; 0 = plain stores/loads bracketed by agent-scope release /
 * acquire fences around a monotonic counter (three round trips); 1 = sc1 atomic stores/loads + the counter.
 * A time-out also raises ST2_STATUS_LSTM_TIMEOUT in st2_status().  st2_lstm_bidir_coop refuses (returns non-zero)
 * when the device cannot hold the launch's workgroups co-resident (occupancy query): use st2_lstm_bidir then. */
int st2_lstm_coop_set_exchange(int mode);
/* Polls a waiting workgroup makes before it gives up (process-wide; <= 0 restores the default of 2^22, seconds).
 * Tests set 1 to provoke ST2_STATUS_LSTM_TIMEOUT. */
int st2_lstm_coop_set_spin_limit(int polls);
/* Measurement hook (process-wide): utterances per cooperative group, 1 / 2 / 4 / 8; 0 (default) = chosen from the batch
 * size (4 up to 32 utterances, 8 up to 48); < 0 = no cooperative launches (st2_lstm_coop_scratch_bytes returns 0, the
 * plans take st2_lstm_bidir).  Smaller blocks trade CUs for less mat-vec work per step and workgroup. */
int st2_lstm_coop_set_block(int utterances);
int st2_lstm_bidir_coop(const float* G, int64_t g_bs, int32_t g_cs, const float* whh_t, const int32_t* lengths,
                        int32_t B, int32_t H, int32_t N, float* Y, int64_t y_bs, int32_t y_cs,
                        void* scratch, int64_t scratch_bytes, void* stream);
/* The same launch with its safety net (ABI v20; what every launch plan issues): the single-CU kernel of st2_lstm_bidir is
 * queued behind the cooperative launch in a CONDITIONAL form -- every workgroup reads scratch[0] and leaves when it is 0 (a
 * few microseconds); when a group was not co-resident in time (another queue held its CUs) it re-runs the whole call into
 * the same Y.  The outputs are then the single-CU kernel's (same results up to fp32 summation order), ST2_STATUS_LSTM_TIMEOUT
 * is NOT raised and ST2_STATUS_LSTM_RECOVERED is.  Same refusal behaviour as st2_lstm_bidir_coop (nothing launched); legal
 * under stream capture. */
int st2_lstm_bidir_coop_recovering(const float* G, int64_t g_bs, int32_t g_cs, const float* whh_t, const int32_t* lengths,
                                   int32_t B, int32_t H, int32_t N, float* Y, int64_t y_bs, int32_t y_cs,
                                   void* scratch, int64_t scratch_bytes, void* stream);

/* generic fused elementwise helpers used by the sampler / denoiser glue */
/* y[b][c][n] = x[b][c][n] + v[b][c]  (x = x + mapping, modules.py:152,394) */
int st2_add_chanvec(const float* x, int64_t x_bs, int32_t x_cs, const float* v, int64_t v_bs,
                    float* y, int64_t y_bs, int32_t y_cs, int32_t B, int32_t C, int32_t N, void* stream);
/* m[b][c] = mean_n x[b][c][n]  (modules.py:155,397) */
int st2_mean_tokens(const float* x, int64_t x_bs, int32_t x_cs, float* m, int64_t m_bs,
                    int32_t B, int32_t C, int32_t N, void* stream);
/* Same over the first len[b] tokens only (int32 [B] on the device, NULL = all N): a right-padded batch then gives
 * every utterance the result of its own un-padded run (the reference runs one utterance at a time,
 * Demo/Inference_LJSpeech.ipynb:268-290). */
int st2_mean_tokens_len(const float* x, int64_t x_bs, int32_t x_cs, float* m, int64_t m_bs,
                        int32_t B, int32_t C, int32_t N, const int32_t* len, void* stream);
/* four[b] = [t, sin(t w_j 2 pi) (j < H2), cos(t w_j 2 pi) (j < H2)], out [B][1 + 2*H2]: LearnedPositionalEmbedding of the
 * denoiser's time input t = c_noise (Modules/diffusion/modules.py:657-671; op order ((t*w)*2)*pi in fp32). */
int st2_time_features(float t, const float* w, int32_t H2, int32_t B, float* out, void* stream);
/* y[b][e][n] = emb[b][n][e]  (b*e_bs + n*E + e): token-major embedding -> channel-major rows of a [B][.][N] tensor
 * (y points at the first destination row); e_bs = 0 broadcasts one [N][E] table (the fixed embedding of the
 * classifier-free-guidance branch, modules.py:412-423) over the batch.  Replaces rearrange / cat, modules.py:388-393. */
int st2_tokens_to_channels(const float* e, int64_t e_bs, int32_t B, int32_t N, int32_t E, float* y, int64_t y_bs,
                           int32_t y_cs, void* stream);
/* y[b][c][n] = x[b*x_bs + c] for n < N  (the noisy style vector repeated over the tokens, modules.py:388-390) */
int st2_broadcast_cols(const float* x, int64_t x_bs, float* y, int64_t y_bs, int32_t y_cs, int32_t B, int32_t C,
                       int32_t N, void* stream);
/* strided NCL copy y[b][c][l] = x[b][c][l] (channel slices of concatenation buffers, tap points) */
int st2_copy_ncl(const float* x, int64_t x_bs, int32_t x_cs, float* y, int64_t y_bs, int32_t y_cs, int32_t B, int32_t C,
                 int32_t L, void* stream);

/* y[b][e][n] = n < len[b] ? (table[tokens[b][n]][e] + add[e]) + pos[n][e] : 0  (tokens int64 [B][N], table [V][E]; add [E]
 * and pos [>= N][E] optional; len int32 [B] or NULL; token ids outside [0, V) contribute 0): nn.Embedding + transpose +
 * masked_fill of TextEncoder.forward (models.py:302-306), and -- with add = token-type row 0, pos = the position table --
 * the embedding sum of the ALBERT model behind PL-BERT (Utils/PLBERT/util.py:6-12; HF AlbertEmbeddings). */
int st2_embed_tokens(const int64_t* tokens, int32_t B, int32_t N, const float* table, int32_t V, int32_t E,
                     const float* add, const float* pos, const int32_t* len, float* y, int64_t y_bs, int32_t y_cs,
                     void* stream);
/* x[b][c][l] = 0 for l >= len[b] (int32 [B] on the device): the masked_fill_ the reference applies after every block of
 * the text-side modules (models.py:308-312, 547-556). */
int st2_mask_tail(float* x, int64_t x_bs, int32_t x_cs, int32_t B, int32_t C, int32_t L, const int32_t* len, void* stream);

/* ---- duration head and alignment expansion (the notebooks' glue between predictor and decoder) ----------------- *
 * st2_duration_head: x [B][K][N] channel-major output of the duration BiLSTM, w [J][K] / bias [J] = duration_proj
 * (models.py:450-451); dur[b][n] (int64) = max(1, round(sum_j sigmoid(w_j . x[b,:,n] + bias_j))), 0 for n >= len[b]
 * (len NULL = no padding), + `tail` frames on token len[b]-1 (5 in the LJSpeech notebook, 0 in the LibriTTS one);
 * dsum (optional, [B][N]) receives the un-rounded sums.  Replaces Demo/Inference_LJSpeech.ipynb:296-301.
 * st2_expand_by_durations: y[b][c][t] = x[b][c][idx(b,t)], idx = the phoneme whose frames cover t (every row of dur
 * sums to T; N <= 512); shift = 1 applies the HiFi-GAN one-frame right shift of Demo/Inference_LibriTTS.ipynb:306-319.
 * Replaces the one-hot alignment matrix + matmul, Demo/Inference_LJSpeech.ipynb:303-312. */
int st2_duration_head(const float* x, int64_t x_bs, int32_t x_cs, const float* w, const float* bias, int32_t B,
                      int32_t K, int32_t J, int32_t N, const int32_t* len, int32_t tail, int64_t* dur, float* dsum,
                      void* stream);
int st2_expand_by_durations(const float* x, int64_t x_bs, int32_t x_cs, const int64_t* dur, int32_t B, int32_t C,
                            int32_t N, int32_t T, int32_t shift, float* y, int64_t y_bs, int32_t y_cs, void* stream);

/* ---- reference-audio style path (compute_style, Demo/Inference_LibriTTS.ipynb:100-111) ------------------------- *
 * The mel front-end (meldataset.py:58-66: torchaudio MelSpectrogram(n_mels 80, n_fft 2048, win 1200, hop 300) ->
 * (log(1e-5 + mel) + 4) / 4) and StyleEncoder (models.py:139-164) run on the conv kernels above: the windowed DFT is a
 * k=1 conv [2*(n_fft/2+1)][n_win] over the frame columns, the mel filter bank a k=1 conv [n_mels][n_fft/2+1], every
 * Conv2d a Conv1d over the width with its kernel rows stacked along the channels (feature maps are stored (h, c, w):
 * element (b,h,c,w) at x + b*x_bs + h*x_hs + c*x_cs + w, so three consecutive rows ARE the 3C-channel input).  These
 * entry points are the remaining non-GEMM steps.
 *   st2_stft_frames:    frames[b][c][m] = wave[b][reflect(m*hop + c - shift)], c < n_win, m < L/hop + 1: the frame
 *                       columns of torch.stft(center=True, pad_mode="reflect") restricted to the taps where the window
 *                       (zero padded to n_fft) is non-zero; shift = n_fft/2 - (n_fft - n_win)/2.
 *   st2_power_spectrum: p[b][k][m] = y[b][k][m]^2 + y[b][K+k][m]^2 (stacked real / imaginary DFT rows).
 *   st2_log_norm:       x = (log(eps + x) - mean) / std in place.
 *   st2_dwconv3x3s2:    depthwise Conv2d(C, C, 3, stride 2, padding 1, groups C), w [C][3][3]: LearnedDownSample('half'),
 *                       models.py:27-42; output map (H-1)/2+1 x (W-1)/2+1.
 *   st2_avgpool2x2:     DownSample('half'), models.py:72-75: replicate the last column of an odd width, 2x2 average;
 *                       H even; output H/2 x (W+1)/2. */
int st2_stft_frames(const float* wave, int64_t w_bs, int32_t B, int32_t L, int32_t n_win, int32_t hop, int32_t shift,
                    float* frames, int64_t f_bs, int32_t f_cs, void* stream);
int st2_power_spectrum(const float* y, int64_t y_bs, int32_t y_cs, int32_t B, int32_t K, int32_t M, float* p,
                       int64_t p_bs, int32_t p_cs, void* stream);
int st2_log_norm(float* x, int64_t n, float eps, float mean, float stdv, void* stream);
int st2_dwconv3x3s2(const float* x, int64_t x_bs, int64_t x_hs, int32_t x_cs, const float* w, const float* bias,
                    int32_t B, int32_t C, int32_t H, int32_t W, float* y, int64_t y_bs, int64_t y_hs, int32_t y_cs,
                    void* stream);
int st2_avgpool2x2(const float* x, int64_t x_bs, int64_t x_hs, int32_t x_cs, int32_t B, int32_t C, int32_t H, int32_t W,
                   float* y, int64_t y_bs, int64_t y_hs, int32_t y_cs, void* stream);

/* out[i] = a*x[i] + b*y[i] + c*z[i] (z may be NULL): sampler updates, sampler.py:184-208,497-510 */
int st2_axpbypcz(const float* x, float a, const float* y, float b, const float* z, float c,
                 float* out, int64_t n, void* stream);

/* ================================================================================================================ *
 * Module-level entry points: one call per reference nn.Module.forward.                          *
 *                                                                                                                      *
 * An engine handle holds one model's packed weights on one device.  The launch plans behind                            *
 *   st2_decoder_forward  ==  Decoder.forward            Modules/istftnet.py:499-528 / Modules/hifigan.py:446-475        *
 *   st2_sampler_run      ==  DiffusionSampler.forward   Modules/diffusion/sampler.py:573-586 (ADPM2 :497-519, KDiffusion *
 *                            :184-208, Transformer1d / StyleTransformer1d Modules/diffusion/modules.py:283-427, 40-185)  *
 * live in C++ (styletts2_amd/csrc/st2_engine.hip): a host in any language synthesises with these calls and nothing    *
 * else.  Memory: the engine owns ONE device allocation for its packed weights (st2_finalize_weights); every forward    *
 * call works inside a caller-owned workspace (size from the *_workspace_bytes query for the same shape), allocates     *
 * nothing, never synchronises and is legal under hipStreamBeginCapture after one eager call of the same shape.         *
 * ================================================================================================================ */
typedef struct st2_engine st2_engine;

typedef struct st2_model_config {
  /* decoder: models.py:617-633, Configs/config.yml:49-57 (istftnet) / Configs/config_libritts.yml:49-55 (hifigan) */
  int32_t decoder_kind;            /* 0 = istftnet, 1 = hifigan */
  int32_t dim_in;                  /* hidden_dim: asr channels (512) */
  int32_t style_dim;               /* acoustic style width (128) */
  int32_t upsample_initial_channel;
  int32_t n_upsamples;  int32_t upsample_rates[4];  int32_t upsample_kernel_sizes[4];
  int32_t n_resblock_kernels;  int32_t resblock_kernel_sizes[4];  int32_t resblock_dilations[4][3];
  int32_t gen_istft_n_fft, gen_istft_hop;   /* istftnet only */
  /* style denoiser: models.py:643-669, Configs/config.yml:66-83 */
  int32_t multispeaker;            /* 0 = Transformer1d (LayerNorm), 1 = StyleTransformer1d (AdaLayerNorm on `features`) */
  int32_t dn_layers, dn_heads, dn_head_features, dn_multiplier;
  int32_t dn_channels;             /* style vector width = 2 * style_dim (256) */
  int32_t dn_embedding;            /* PL-BERT hidden size (768) */
  int32_t dn_context_features;     /* width of `features` (256), multispeaker only */
  int32_t dn_max_length;           /* fixed-embedding table length (512) */
  /* prosody predictor: models.py:440-466, Configs/config.yml `hidden_dim` */
  int32_t pred_hidden;             /* d_hid of ProsodyPredictor (512); 0 = predictor not used */
  /* PL-BERT: Utils/PLBERT/config.yml model_params (an HF AlbertConfig; widths are read from the weights' shapes) */
  int32_t bert_layers;             /* num_hidden_layers (12; ALBERT: one shared weight set); 0 = PL-BERT not used */
  float bert_ln_eps;               /* layer_norm_eps (1e-12) */
} st2_model_config;

int st2_create(const st2_model_config* cfg, st2_engine** out);
int st2_destroy(st2_engine* e);
/* Hands one parameter to the engine (copied): `name` = the reference state_dict key below the module, prefixed by
 * "decoder." / "denoiser." (the Transformer's keys, i.e. `diffusion.unet.*` without that prefix), with weight-norm
 * pairs FOLDED by the caller: `X.weight` = weight_g * weight_v / ||weight_v|| (norm over all dims but 0; dim 0 of a
 * ConvTranspose1d is C_in), replacing X.weight_g / X.weight_v.  `data` is a HOST pointer to fp32, C-contiguous. */
int st2_load_weights(st2_engine* e, const char* name, const float* data, const int64_t* shape, int32_t ndim);
/* Packs everything loaded so far (split-f16 conv layouts, polyphase ConvTranspose / strided-conv forms, concatenated
 * AdaIN fc matrix) and uploads it in one device allocation.  which: bit 0 = decoder, bit 1 = denoiser, bit 2 = prosody
 * predictor (names prefixed "predictor."), bit 3 = text encoder ("text_encoder."), bit 4 = PL-BERT ("bert.", the HF
 * AlbertModel keys) with the `bert_encoder` Linear ("bert_encoder.weight" / ".bias") when it was loaded, bit 5 = the
 * reference-audio style encoders ("style_encoder." / "predictor_encoder.", whichever were loaded; spectral-norm triples
 * folded by the caller: X.weight = weight_orig / (u . (W_mat v)), replacing X.weight_orig / weight_u / weight_v).
 * Synchronous; call once after the last st2_load_weights (again after loading new weights). */
int st2_finalize_weights(st2_engine* e, int32_t which);

/* Optional tap points (NULL = not wanted): device buffers the forward copies intermediates into, for parity work. */
typedef struct st2_decoder_taps {
  float* encode;       /* [B][1024][T]                                       istftnet.py:510 */
  float* front;        /* [B][512][2T]   decoder front = generator input      istftnet.py:513-524 */
  float* har_source;   /* [B][600T]      tanh(l_linear(sine waves))           istftnet.py:352-354 */
  float* har;          /* istftnet [B][n_fft+2][120T+1] |STFT| ++ angle; hifigan: unused (== har_source) */
  float* stage[4];     /* generator stage outputs [B][C_i][L_i]               istftnet.py:359-375 */
  float* spec_phase;   /* istftnet [B][n_fft+2][120T+1] after exp / sin        istftnet.py:378-379 */
} st2_decoder_taps;

int64_t st2_decoder_workspace_bytes(st2_engine* e, int32_t B, int32_t T);
/* wave[B][600*T] = Decoder(asr[B][dim_in][T], F0[B][2T], N[B][2T], s[B][style_dim]).  `sine_noise` [B][600T][9] are
 * the standard-normal draws of SineGen (istftnet.py:242; the reference draws them inside forward); `har_inject`
 * (optional) replaces the harmonic-source features with the caller's (tap-point protocol): istftnet
 * [B][n_fft+2][120T+1], hifigan [B][600T].  All pointers are device pointers, tensors contiguous. */
int st2_decoder_forward(st2_engine* e, const float* asr, const float* f0, const float* n, const float* s,
                        const float* sine_noise, const float* har_inject, int32_t B, int32_t T, float* wave,
                        void* workspace, int64_t workspace_bytes, const st2_decoder_taps* taps, void* stream);

/* Per-step scalars of the ADPM2 loop, all input independent (sampler.py:184-191, 490-495): for step i (sigma_i ->
 * sigma_{i+1}) row i of `table` holds 11 doubles
 *   [0..3]  c_skip, c_out, c_in, c_noise at sigma_i        [4..7]  the same at sigma_mid
 *   [8]     (sigma_mid - sigma_i) / sigma_i                [9]     (sigma_down - sigma_i) / sigma_mid      [10] sigma_up
 * A Python host fills it with the reference's own torch / python-float arithmetic (bit-faithful to the reference);
 * st2_sampler_table does the same with libm for other hosts (KarrasSchedule sigma_min / sigma_max / rho, sampler.py:
 * 328-337; ADPM2 rho = 1).  sigma0 = sigmas[0] (x_0 = sigma0 * noise). */
#define ST2_SAMPLER_TABLE_COLS 11
int st2_sampler_table(int32_t steps, double sigma_min, double sigma_max, double rho, double sigma_data, double* table,
                      double* sigma0);
int64_t st2_sampler_workspace_bytes(st2_engine* e, int32_t B, int32_t N, int32_t steps, double embedding_scale);
/* out[B][C] = DiffusionSampler(noise[B][C], embedding[B][N][E], features[B][Fc] or NULL, num_steps = steps,
 * embedding_scale).  step_noise [steps-1][B][C] are the per-step randn_like draws (sampler.py:509); `lengths` (int32
 * [B] on the device, or NULL) = token counts of a right-padded batch; step_taps (optional) [steps-1][B][C] receives x
 * after every step. */
int st2_sampler_run(st2_engine* e, const float* noise, const float* embedding, const float* features,
                    const float* step_noise, const int32_t* lengths, int32_t B, int32_t N, int32_t steps,
                    double embedding_scale, const double* table, double sigma0, float* out, void* workspace,
                    int64_t workspace_bytes, float* step_taps, void* stream);

/* Alignment expansion + ProsodyPredictor.F0Ntrain (Demo/Inference_LJSpeech.ipynb:303-311, models.py:497-510), i.e. what
 * sits between the duration predictor and Decoder.forward:
 *   asr[B][dim_in][T]  = t_en[B][dim_in][N] expanded by `durations` (int64 [B][N], rows summing to T),
 *   (f0, n)[B][2T]     = F0Ntrain(d expanded the same way, s),  d_cm [B][pred_hidden + style_dim][N] = the duration
 *                        encoder's output channel-major, s [B][style_dim] the prosodic style;
 * shift = 1 applies the HiFi-GAN one-frame right shift (Demo/Inference_LibriTTS.ipynb:306-319).  Same memory / stream
 * contract as st2_decoder_forward; its outputs are exactly that call's inputs. */
int64_t st2_prosody_workspace_bytes(st2_engine* e, int32_t B, int32_t N, int32_t T);
int st2_prosody_forward(st2_engine* e, const float* d_cm, const float* t_en, const int64_t* durations, const float* s,
                        int32_t B, int32_t N, int32_t T, int32_t shift, float* asr, float* f0, float* n, void* workspace,
                        int64_t workspace_bytes, void* stream);

/* TextEncoder.forward (models.py:284-345): tokens int64 [B][N] (id 0 = pad), lengths int32 [B] or NULL ->
 * t_en [B][dim_in][N]: Embedding -> depth x [weight-norm Conv1d k5 -> LayerNorm over channels -> LeakyReLU(0.2) -> mask] ->
 * BiLSTM.  Weights: "text_encoder.*" with the weight-norm pairs folded (st2_finalize_weights bit 3). */
int64_t st2_text_workspace_bytes(st2_engine* e, int32_t B, int32_t N);
int st2_text_forward(st2_engine* e, const int64_t* tokens, const int32_t* lengths, int32_t B, int32_t N, float* t_en,
                     void* workspace, int64_t workspace_bytes, void* stream);

/* DurationEncoder.forward + the duration head (models.py:536-569, 450-451; Demo/Inference_LJSpeech.ipynb:294-301): from
 * d_en [B][pred_hidden][N] (bert_encoder's output, channel-major) and the prosodic style s [B][style_dim] to
 *   d_cm [B][pred_hidden + style_dim][N]  the duration encoder's output (channel-major; st2_prosody_forward's input),
 *   durations int64 [B][N] (may be NULL)  max(1, round(sum sigmoid(duration_proj(lstm(d))))), 0 at pad tokens, + `tail`
 *                                         frames on every utterance's last token (5 in the LJSpeech notebook).
 * `lengths` (int32 [B] on the device, or NULL) = token counts of a right-padded batch (packed-sequence BiLSTMs, masked
 * LayerNorm outputs).  Weights: "predictor.text_encoder.*", "predictor.lstm.*", "predictor.duration_proj.linear_layer.*"
 * (st2_finalize_weights bit 2 packs them with the rest of the predictor when they were loaded). */
int64_t st2_duration_workspace_bytes(st2_engine* e, int32_t B, int32_t N);
int st2_duration_forward(st2_engine* e, const float* d_en, const float* s, const int32_t* lengths, int32_t B, int32_t N,
                         int32_t tail, float* d_cm, int64_t* durations, void* workspace, int64_t workspace_bytes,
                         void* stream);

/* PL-BERT forward (Utils/PLBERT/util.py:6-12: HF AlbertModel(...).last_hidden_state): tokens int64 [B][N], lengths int32
 * [B] or NULL (valid keys of a right-padded batch = the attention mask of utils.py:42-46 `length_to_mask`) ->
 * hidden_cm [B][hidden][N], the last hidden state CHANNEL-major (transpose of the reference's [B][N][hidden]).
 * Token-merged storage inside ([C][B*N]): every Linear is one k = 1 split-f16 conv over B*N columns (q|k|v fused), the
 * embedding LayerNorm rides in the mapping conv's prologue, gelu_new in the FFN conv's epilogue.  Weights: "bert.*"
 * (st2_finalize_weights bit 4); cfg.bert_layers / bert_ln_eps. */
int64_t st2_bert_workspace_bytes(st2_engine* e, int32_t B, int32_t N);
int st2_bert_forward(st2_engine* e, const int64_t* tokens, const int32_t* lengths, int32_t B, int32_t N, float* hidden_cm,
                     void* workspace, int64_t workspace_bytes, void* stream);

/* Everything the notebooks' `inference` cell does between the phoneme ids and the alignment (Demo/Inference_LJSpeech.ipynb:
 * 268-301, Demo/Inference_LibriTTS.ipynb:258-305; LFinference's style carry-over, Inference_LJSpeech.ipynb:409-446) in ONE
 * call: text encoder, PL-BERT, bert_encoder, the style-diffusion sampler (ADPM2 over st2_sampler_table's rows), the style
 * mixing, the duration encoder and (when `durations` is given) the duration head.
 *   s_pred  = sampler(noise, embedding = bert, features = ref_s)            [B][2*style_dim]
 *   s_pred  = t * s_prev + (1 - t) * s_pred                                 when s_prev != NULL
 *             (carry != 0: the B rows are CONSECUTIVE SENTENCES of one passage and row k's s_prev is row k-1's mixed s_pred_out;
 *             row 0 takes `s_prev` [1][2*style_dim] or nothing.  The passage is sequential in this vector only, so its text
 *             encoder / PL-BERT / sampler / duration stages run as ONE right-padded batch and the carry-over as a row scan
 *             of elementwise kernels between them: every row equals the sentence-by-sentence LFinference loop's)
 *   ref | s = s_pred[:, :style_dim] | s_pred[:, style_dim:]
 *   ref     = alpha * ref + (1 - alpha) * ref_s[:, :style_dim];  s = beta * s + (1 - beta) * ref_s[:, style_dim:]   (ref_s)
 * All pointers are device memory; NULL = absent where marked optional.  Needs every weight group of bits 1-4 finalized.
 * Same memory / stream / graph-capture contract as st2_decoder_forward; the outputs feed st2_prosody_forward (after the
 * host has read the frame counts off `durations`) and st2_decoder_forward (`ref`). */
typedef struct st2_front_args {
  const int64_t* tokens;     /* [B][N] */
  const int32_t* lengths;    /* [B] or NULL */
  const float* noise;        /* [B][2*style_dim] */
  const float* step_noise;   /* [steps-1][B][2*style_dim] */
  const float* ref_s;        /* [B][2*style_dim]: reference style (multispeaker models), or NULL */
  const float* s_prev;       /* [B][2*style_dim]: previous sentence's s_pred_out (long-form), or NULL */
  int32_t B, N, steps;
  int32_t tail;              /* frames added to every utterance's last token (5 in the LJSpeech notebook) */
  double embedding_scale;
  const double* table;       /* HOST: st2_sampler_table(steps, ...) rows */
  double sigma0;
  double alpha, beta, t;     /* style mixing weights (0.3 / 0.7 / 0.7 in the notebooks) */
  float* t_en;               /* out [B][dim_in][N] */
  float* d_cm;               /* out [B][pred_hidden + style_dim][N] */
  float* s;                  /* out [B][style_dim]  prosodic style */
  float* ref;                /* out [B][style_dim]  acoustic style (the decoder's `s`) */
  float* s_pred_out;         /* out [B][2*style_dim] = ref | s after mixing (the next sentence's s_prev), or NULL */
  int64_t* durations;        /* out [B][N], or NULL when the caller supplies its own durations */
  int32_t carry;             /* != 0: rows = consecutive sentences, style carried from row k-1 to row k (see above); ABI v21 */
} st2_front_args;
int st2_sizeof_front_args(void);
int64_t st2_front_workspace_bytes(st2_engine* e, const st2_front_args* a);
int st2_front_forward(st2_engine* e, const st2_front_args* a, void* workspace, int64_t workspace_bytes, void* stream);

/* StyleEncoder.forward (models.py:139-164; `compute_style`, Demo/Inference_LibriTTS.ipynb:100-111): mel [B][80][T] (the
 * normalised log-mel of the reference recording, T >= 80 frames) -> style [B][style_dim].  which = 0: `style_encoder`
 * (acoustic half of ref_s), 1: `predictor_encoder` (prosodic half); ref_s = cat(which 0, which 1).  Conv2d layers run as
 * split-f16 Conv1d over the width with their kernel rows stacked along the channels (maps stored (h, c, w)), the
 * depthwise stride-2 conv and the 2x2 average as st2_dwconv3x3s2 / st2_avgpool2x2.  The mel itself is five kernel-level
 * calls (st2_stft_frames, st2_conv1d with the windowed-DFT matrix, st2_power_spectrum, st2_conv1d with the filter bank,
 * st2_log_norm: styletts2_amd/style.py mel_spectrogram_engine).  Same memory / stream contract as st2_decoder_forward. */
int64_t st2_style_workspace_bytes(st2_engine* e, int32_t which, int32_t B, int32_t n_mels, int32_t T);
int st2_style_forward(st2_engine* e, int32_t which, const float* mel, int32_t B, int32_t n_mels, int32_t T, float* style,
                      void* workspace, int64_t workspace_bytes, void* stream);

/* ---- measurement hook (bench.py's roofline leg) ------------------------------------------------------------------- *
 * st2_conv_timing(1) clears and starts, (0) stops recording a HIP event pair around every st2_conv1d_xs launch (C_in >=
 * 64, L_out >= 256) on its launch stream, whichever plan issues it.  st2_conv_timing_read (after stopping) waits for the
 * events and fills rows of 6 doubles {ks, C_in, C_out, L_out, B, milliseconds}; returns the number of launches recorded
 * (rows may be NULL to count), < 0 on error.  Not thread safe; not legal under stream capture. */
int st2_conv_timing(int enable);
int st2_conv_timing_read(double* rows, int32_t cap_rows);

/* ---- start-up autotuner of st2_conv1d_xs (ABI v19) ---------------------------------------------------------------- *
 * A launch of st2_conv1d_xs exists in several BUILDS that issue the same products in the same order and share one
 * epilogue -- results are bitwise identical (tests/test_ops_gpu.py) --: 128 x 128 tiles at 3 workgroups / CU or 128 x 256
 * tiles at 2 (k = 3, 7, 11), in dispatch order or XCD-aware tile order (launches with 2 / 4 / 8 output row blocks: every XCD keeps
 * ONE row block's weights in its L2).  Which is fastest depends on the shape AND on the box by a few percent (and, before
 * round 4's row-end fix of the epilogue, by up to 1.75 x on the C = 256 / L = 8 000 layers of Modules/istftnet.py:358-375),
 * so a serving process measures at start-up:
 *   st2_conv_tune(1)   every FIRST launch of a shape class (device, ks, C_in, C_out, L_out, B) on a non-capturing stream
 *                      times its candidate builds (1 warm-up + 3 x 2 launches each, round-robin, output into a scratch tensor the
 *                      library allocates for the duration of tuning mode -- the one exception to "no allocation" besides
 *                      the status word; the caller's tensors are only read) and records the winner; the call then runs it;
 *   st2_conv_tune(0)   leaves tuning mode (frees the scratch); recorded classes keep their build, others follow the rule;
 *   st2_conv_tune(-1)  also forgets the current device's table.
 * st2_conv_tune_set pins (variant >= 0: bit 0 = 128 x 256 tiles, bit 1 = XCD-aware order) or
 * erases (variant = -1) one class on the current device; st2_conv_tune_read fills rows of 24 doubles {ks, C_in, C_out,
 * L_out, B, device, chosen variant, n candidates, (variant, ms / launch) x 8} for the current device and returns the number
 * of classes (rows may be NULL to count).  Tuning synchronises the stream it measures on; the table is guarded by a mutex and
 * the measurement runs outside it (other threads keep launching, on the rule's build for a class while it is being measured).
 * Tuning mode allocates and waits on events: it must not overlap a stream capture anywhere in the process (a global-mode
 * capture on another stream would be invalidated) -- tune first, record graphs afterwards.  A class is keyed without the
 * dilation: the three dilations of a resblock share one measurement (they differ in halo columns only). */
int st2_conv_tune(int mode);
int st2_conv_tune_set(int32_t ks, int32_t C_in, int32_t C_out, int32_t L_out, int32_t B, int32_t variant);
int st2_conv_tune_read(double* rows, int32_t cap_rows);

/* ---- operand-range telemetry, both ends (ABI v20; debug: allocates, synchronises; not for the serving path) ------------------ *
 * The split-f16 convs carry u = x_scale * pro(x) as f16 hi + lo.  The format has two ends: above 65504 the kernels clamp
 * (ST2_STATUS_F16_RANGE); below ~2^-3 the lo half becomes a subnormal f16 and the operand keeps an ABSOLUTE error of 2^-25
 * instead of 2^-22 relative -- a layer whose scaled input sits at 1e-3 is 100 x less precise than fp32 although nothing
 * overflows.  Where a checkpoint puts each layer (Snake alpha, weight-norm gains, GELU / LeakyReLU outputs are free,
 * Modules/istftnet.py:27-62) is what this hook reports: between st2_debug_headroom(1) and (0) every st2_act_split /
 * st2_conv1d_f16s launch also measures the planes it produced; st2_debug_headroom_read fills rows of ST2_HEADROOM_COLS doubles
 *   [0] kind (0 = activation pass of the xs path, 1 = prologue of the fused conv)   [1] prologue   [2] B  [3] C_in  [4] L
 *   [5] x_scale   [6] max |u|   [7] max |u| / 65504 (>= 1: clamped)
 *   [8] relative RMS error the split adds to the operand: sqrt(sum ulp(lo)^2 / 12 / sum u^2) (fp32 storage itself: 3.4e-8;
 *       every element's lo normal: ~4e-8)
 *   [9] share of the operand's energy carried by elements whose lo half is subnormal or zero
 *   [10] conv site (st2_calibration_read index) when an engine plan issued the launch, else -1   [11] that engine (identity)
 * in launch order and returns their number.  tools/headroom_report.py prints the table for a checkpoint. */
#define ST2_HEADROOM_COLS 12
int st2_debug_headroom(int enable);
int st2_debug_headroom_read(double* rows, int32_t cap_rows);

/* ---- per-layer operand scales (ABI v20) ------------------------------------------------------------------------------------ *
 * Every F.conv1d / nn.Linear of the reference is fp32 (Modules/istftnet.py:68-74,376-377, Modules/diffusion/modules.py:
 * 484-490).  The split-f16 convs match that over the whole fp32 range only if each conv's input is scaled into the part of
 * the f16 range where both halves are normal numbers.  By rule (no calibration) x_scale is 8 after a normalising prologue and
 * 1 otherwise -- right for O(1) tensors, up to 500 x less precise than fp32 for a layer whose input sits at 1e-4.  A serving
 * process therefore calibrates once per checkpoint:
 *     st2_debug_headroom(1);  one or more forward calls on representative inputs;  st2_debug_headroom(0);
 *     st2_calibrate(e, 3, &clamped);            (repeat the three lines while clamped > 0: at most twice)
 * Every conv site of the engine (= every split-f16 packed weight, in packing order) that was launched gets
 *     x_scale = 2^floor(log2(2^(16 - margin_bits) / max |pro(x)|))     (margin_bits = 3: max |u| in (4096, 8192])
 * from the largest operand any of its launches produced (a PL-BERT weight runs 12 times, a denoiser weight once per
 * evaluation: the maximum counts); the clamp and the status bit stay as the safety net for inputs beyond 2^margin_bits x the
 * calibration's.  Scales are powers of two -- the accumulator is rescaled exactly -- and live in the engine: every later
 * forward (eager or captured; capture AFTER calibrating, the scale is a kernel argument) uses them; results stay
 * reproducible bit for bit for a given table.  Returns the number of sites set (< 0 on error).
 *   st2_calibration_read   rows of ST2_CALIBRATION_COLS doubles {C_in, C_out, ks, x_scale (0 = by rule), max |pro(x)| seen};
 *                          returns the number of sites (rows may be NULL to count);
 *   st2_calibration_site_name   the reference state_dict key the site's weight was packed from;
 *   st2_calibration_write  installs a table (n = number of sites, entries 0 or a power of two; n = 0 clears): how rank 0's
 *                          calibration reaches the other ranks and how a table saved beside a checkpoint is restored;
 *   st2_calibration_scale  the formula above for one value (0 for max_abs <= 0).
 * st2_calibrate ACCUMULATES: a site's maximum is the largest over every recorded call since the table was last cleared, and sites whose
 * operand does not come out of a normalising prologue (free-ranging: F0 in Hz, stage outputs, FFN intermediates) get two more bits
 * of headroom than margin_bits (ABI v22).  st2_finalize_weights CLEARS the table (ABI v22; v20-21 kept it for an unchanged conv
 * layout): scales belong to the weights they were measured on. */
#define ST2_CALIBRATION_COLS 5
int st2_calibrate(st2_engine* e, int32_t margin_bits, int32_t* n_clamped);
int st2_calibration_read(st2_engine* e, double* rows, int32_t cap_rows);
int st2_calibration_site_name(st2_engine* e, int32_t site, char* name, int32_t cap);
int st2_calibration_write(st2_engine* e, const float* scales, int32_t n);
float st2_calibration_scale(float max_abs, int32_t margin_bits);

/* ---- box probe (ABI v19; diagnostic: allocates and frees its own device buffers, synchronises the device) ----------- *
 * Writes a JSON object (NUL terminated, <= cap bytes; 4 KB is enough) of micro-measurements of the current device:
 * device properties, sustained matrix-pipe clock and TFLOP/s of a bare v_mfma_f32_32x32x16_f16 loop on random / all-zero
 * operands, dependent-load latency and weight-stream bandwidth for working sets of 0.75 ... 64 MB (level >= 1: also 512
 * MB), a 512 MB HBM copy, and where the 2 048 workgroups of a 2-per-CU launch run (CUs / XCDs seen, workgroups per CU).
 * Takes ~0.5 s; bench.py puts it into its JSON line (`box.probe`) so that a run on a box nobody can log into still says
 * what the box gives the conv path.  No reference call site: measurement only. */
int st2_probe_box(char* json, int32_t cap, int32_t level);

/* ---- CU health probe (ABI v19; diagnostic: allocates ~0.9 GB for its duration, synchronises the device) --------------- *
 * Runs an instrumented copy of the conv kernel (per-workgroup cycle stamps + HW_ID) on the launch class that separated
 * the box classes of rounds 1-3 (k = 7, C = 256, L = 8 000, 128 x 256 tiles), reports as JSON when every XCD finished and the
 * CUs whose median epilogue takes > 3 x the chip's median, finds their CU-mask bits by measurement (8-workgroup probes on
 * single-bit-masked streams: bit i belongs to XCD i % 8) and returns in mask_out (mask_words >= 8 words) the CU mask of the
 * device WITHOUT them and their number in *n_excluded (0 = nothing slow, mask = all CUs).  It is how round 4 found that the
 * "slow CUs" of the slow box class were the row-end tiles of the launch -- since the epilogue fix it
 * reads 0 on every box seen -- and it stays as the check that finds a genuinely degraded CU.  Takes ~0.1 s. */
int st2_probe_cu_health(char* json, int32_t cap, uint32_t* mask_out, int32_t mask_words, int32_t* n_excluded);

/* Matrix-pipe load generator for the co-residency canaries (ABI v22).  Round 6 traced "the BiLSTM returns other bits while
 * narrow-tile convs run on another queue" to the hardware: on gfx950 a packed-f32 VALU op whose op_sel takes the HIGH dword of
 * src1 for the low result lane (`v_pk_fma_f32 ... op_sel:[0,1,0]`) returns a wrong low half in lanes 48-63 while ANOTHER wave of the
 * CU issues MFMAs in certain cadences (tools/simd_hazard_repro.hip).  The library no longer contains that
 * encoding (tools/check_isa.py gates every build); this entry point launches the cadences that provoked it, so that tests and a
 * serving process's start-up check can run ANY kernel of the path next to them and demand bitwise the idle result:
 *   kind 0  v_mfma_f32_16x16x32_f16, one dependent chain per wave (hit rate ~90-100 % on the old BiLSTM kernels)
 *   kind 1  v_mfma_f32_32x32x16_f16 in isolated dependent groups of three with LDS reads between them (what a one-accumulator
 *           32-column conv tile issues per k-step)
 *   kind 2  v_mfma_f32_16x16x32_f16, four independent chains per wave
 * `workgroups` x 4 waves, `iters` MFMA groups per wave (720 / 400: ~60-100 us per launch); nothing is read or written; asynchronous
 * on `stream`.  Returns non-zero on bad arguments. */
int st2_probe_mfma_stream(int32_t kind, int32_t workgroups, int32_t iters, void* stream);

/* ---- CU-partitioned streams (ABI v17) ---------------------------------------------------------------------------- *
 * A HIP stream whose kernels may only be placed on the compute units whose bit is set in `mask` (n_words x 32 bits, bit i
 * = CU i in the driver's numbering, which deals consecutive bits out round-robin over the 8 XCDs and their shader
 * engines: the low 64 bits are 8 CUs of every XCD).  The two-stage pipeline (front of batch k+1 under the decoder of
 * batch k) uses a pair of complementary masks so that the latency-bound front kernels -- among them the spin-waiting
 * cooperative BiLSTM groups -- own their CUs instead of competing with 12 000-workgroup convs for slots.
 * st2_stream_destroy waits for the stream's work.  The handle is a hipStream_t. */
int st2_stream_create_cu_mask(const uint32_t* mask, int32_t n_words, void** stream);
int st2_stream_destroy(void* stream);

/* ---- testing hook ---------------------------------------------------------------------------------------------- *
 * Replaces the kernel / memory entry points the launch plans call by the caller's (an array of ST2_BACKEND_ENTRIES
 * function pointers in the order of `enum st2_backend_slot`; NULL restores the HIP kernels).  tests/ uses it to run the
 * C++ plans on HOST memory against per-kernel CPU contracts, i.e. to validate plan wiring, packing and workspace
 * aliasing without a GPU.  Never used by the product path. */
enum st2_backend_slot {
  ST2_BE_CONV1D_F16S = 0, ST2_BE_CONV1D_XS, ST2_BE_ACT_SPLIT, ST2_BE_STATS_FINALIZE, ST2_BE_CONV1D_DIRECT,
  ST2_BE_PHASE_SPLIT, ST2_BE_INSTNORM_STATS, ST2_BE_COLNORM_STATS, ST2_BE_STYLE_FC, ST2_BE_CONVT_INTERLEAVE_STATS,
  ST2_BE_ADAIN_LEAKY_POOL, ST2_BE_HAR_SOURCE, ST2_BE_STFT_MAG_PHASE, ST2_BE_ISTFT, ST2_BE_ATTENTION_KEYLEN,
  ST2_BE_ADD_CHANVEC, ST2_BE_MEAN_TOKENS_LEN, ST2_BE_AXPBYPCZ, ST2_BE_TIME_FEATURES, ST2_BE_TOKENS_TO_CHANNELS,
  ST2_BE_BROADCAST_COLS, ST2_BE_COPY_NCL, ST2_BE_EXPAND_BY_DURATIONS,
  ST2_BE_LSTM_BIDIR,  /* st2_lstm_bidir's arguments with (void* scratch, int64_t scratch_bytes) inserted before `stream` */
  ST2_BE_COLNORM_APPLY, ST2_BE_DURATION_HEAD, ST2_BE_MASK_TAIL, ST2_BE_EMBED_TOKENS, ST2_BE_DWCONV3X3S2, ST2_BE_AVGPOOL2X2,
  ST2_BE_DEV_ALLOC,   /* void* (*)(int64_t bytes) */
  ST2_BE_DEV_FREE,    /* void (*)(void*) */
  ST2_BE_UPLOAD,      /* int (*)(void* dst, const void* src, int64_t bytes): synchronous host -> device copy */
  ST2_BACKEND_ENTRIES
};
int st2_debug_set_backend(void* const* table, int32_t entries);

#ifdef __cplusplus
}
#endif
#endif /* ST2_H */
