/* dts.h -- C ABI of libdts_hip.so, the MI355X (gfx950) kernels under the noise-trajectory-search hot path.
 *
 * The reference (rvignav/diffusion-tts) is pure Python on PyTorch: it has NO native code and NO FFI
 * (SURVEY.md, fact 1).  Each entry point below therefore replaces a *sequence of PyTorch ops* at the
 * cited reference call site; the Python host (diffusion_tts_amd/) keeps the reference's own module /
 * function surface on top (INTEGRATION.md shows the binding a maintainer would add).
 *
 * Conventions
 *  - every function returns 0 on success, a negative dts_status otherwise; the message is available
 *    (per thread) from dts_last_error().  Nothing throws across the ABI.
 *  - all pointers are DEVICE pointers owned by the caller (PyTorch allocations); the library never
 *    frees or retains them, never allocates on the launch path, never synchronises.
 *  - all work is enqueued on `stream` (a hipStream_t passed as void*; NULL = default stream).
 *  - activations are NHWC ("channels-last"): [n][h][w][c], element type `dtype`
 *    (DTS_F32 parity mode, DTS_BF16 / DTS_F16 throughput modes; accumulation is always f32).
 *  - images/latents at the sampler boundary keep the reference's layout: NCHW, fp64 state
 *    (edm/main.py:99), fp32 denoiser output (networks.py:667), uint8 scorer input (edm/main.py:126).
 */
#ifndef DTS_H_
#define DTS_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* dts_stream;

enum dts_dtype { DTS_F32 = 0, DTS_BF16 = 1, DTS_F16 = 2,
                 /* dts_conv2d only: SPLIT PRECISION on the 16-bit matrix cores -- the DEFAULT compute mode of the Python surfaces.  Activations
                    and the epilogue operands (bias_nc, residual, out) are f32 tensors; the conv's input is their split image from
                    dts_split3_f16 -- x1 = [n][h][w][2*C] f16, per group of 32 channels 128 bytes = hi(32) | lo * 2^11 (32) with hi = f16(x),
                    lo = x - hi (pass c1 = 2*C, c2 = 0) -- and the packed weight is [cout][taps][2*C] f16, per 32 input channels
                    wh(32) | wl(32) = the hi / lo parts of w * 2^k (acc_scale = 2^-k; the powers of two keep every part a NORMAL f16 number:
                    the matrix cores flush subnormal inputs).  One staged 128-byte K step feeds THREE MFMAs per accumulator tile --
                    wh.hi, (wh * 2^-11).(lo * 2^11), wl.hi -- accumulating, in f32, x_hi*w_hi + x_lo*w_hi + x_hi*w_lo: products exact to
                    ~2^-22 (f16 x f16 carries 22 bits) against 2^-24 of the f32 matrix instruction, which runs at 1/16 of the 16-bit
                    rate.  This is the mode that reproduces the reference's fp32 selections (edm/main.py:842: argmax over rewards that
                    differ by 1e-7) at a third of the 16-bit modes' matrix rate.  (ABI 108 laid the image out as three planes hi | lo | hi
                    against hi | hi * 2^-11 | lo: 1.5 x the bytes, LDS-DMA pieces and fragment reads for the same products.) */
                 DTS_F16X3 = 3 };
enum dts_status { DTS_OK = 0, DTS_ERR_ARG = -1, DTS_ERR_LAUNCH = -2, DTS_ERR_UNSUPPORTED = -3 };

#define DTS_ABI_VERSION 111        /* bumped with every change of a signature or of dts_conv_args (101: ev_start/ev_stop; 102: tuning knobs; 103: dts_cosine_rows; 104: dts_nchw_to_nhwc_pad,
                                     105: dts_conv_args.gn_coef / gn_silu, dts_conv_fuses_gn;
                                     head dim 512 in dts_attention; 106: dts_conv_kernel, 128-cout ping-pong blocks; 107: dts_resample_u8, dts_lut_u8_f32; 108: DTS_F16X3, dts_conv_args.acc_scale, dts_split3_f16, dts_gn_apply_x3, dts_split2_f16, dts_attention_x3;
                                     109: dts_candidate_noise_sd; the DTS_F16X3 operand images are 2*C wide, interleaved per 32 channels; dts_gn_apply_x3 raw_out;
                                     110: dts_candidate_noise_sd takes the three scalars of the reference's product separately (scale [n][3]);
                                     111: dts_conv_args.skip_* (a block's 1x1 skip convolution folded into its second 3x3), dts_conv_folds_skip) */
int dts_version(void);            /* == DTS_ABI_VERSION of the build; a binding must refuse any other value */
const char* dts_last_error(void);
/* Tuning knobs (measurement aid; a knob only selects between kernels / block orders / ring depths that give correct results -- the
 * timing-only diagnostic kernels exist only in builds made with -DDTS_DIAG_KERNELS and are refused by the product library).  knob: index of
 * enum dts_knob in csrc/dts_common.h; value -1 = launcher default.  Used by tools/conv_bench.py and tools/att_bench.py to A/B variants in one process. */
int dts_set_tuning(int knob, int value);
int dts_get_tuning(int knob);

/* ---- layout / packing (weight preparation and test plumbing; not on the per-step path) ------------ */
/* NCHW f32 -> NHWC dtype, and back. */
int dts_nchw_to_nhwc(const float* src, void* dst, int dtype, int n, int c, int h, int w, dts_stream s);
int dts_nhwc_to_nchw(const void* src, int dtype, float* dst, int n, int c, int h, int w, dts_stream s);
/* NCHW f32 -> NHWC dtype with the channel count padded to cpad (zeros): SD latents (4 channels) feeding the MFMA conv (cin % 64 == 0). */
int dts_nchw_to_nhwc_pad(const float* src, void* dst, int dtype, int n, int c, int h, int w, int cpad, dts_stream s);
/* OIHW f32 conv weight (torch layout, networks.py:62 / unet.py conv_nd) -> [O][kh][kw][I] dtype.
 * out_perm (device int32[O], nullable): packed row o is taken from source row out_perm[o]; used to
 * regroup the qkv projection's output channels into q|k|v blocks (networks.py:182, unet.py:365). */
int dts_pack_conv_weight(const float* w_oihw, void* dst, int dtype, int O, int I, int kh, int kw,
                         const int32_t* out_perm, dts_stream s);

/* ---- K1/K2/K3: implicit-GEMM convolution on MFMA (networks.py:68-90 Conv2d.forward) --------------- */
typedef struct dts_conv_args {
  const void* x1; int32_t c1;     /* input, NHWC [n][hin][win][c1] */
  const void* x2; int32_t c2;     /* optional 2nd input concatenated on channels (networks.py:458 torch.cat); c2=0 if none */
  const void* w;                  /* packed weight [cout][ksize*ksize][c1+c2] */
  const float* bias;              /* [cout] or NULL */
  const void* bias_nc;            /* per-sample per-channel addend [n][ld_bias_nc] (dtype) or NULL (networks.py:175 x.add_(params)) */
  int32_t ld_bias_nc;
  const void* residual;           /* NHWC [n][hout][wout][cout] or NULL (networks.py:178,185) */
  void* out;                      /* NHWC [n][hout][wout][cout] */
  int32_t n, hin, win;
  int32_t cout;
  int32_t ksize;                  /* 1 or 3 (pad ksize/2, stride 1) */
  int32_t up;                     /* 1: nearest 2x upsample of the input fused into the gather (networks.py:82-83) */
  float out_scale;                /* out = (conv + bias + bias_nc + residual) * out_scale  (networks.py:179,186 skip_scale) */
  int32_t dtype;
  void* workspace;                /* optional scratch (16-byte aligned) for split-K partial sums; NULL disables split-K */
  int64_t workspace_bytes;
  float* stats_out;               /* optional [ceil(P/64)][cout][2] f32: per 64-pixel strip (sum, sumsq) of the stored outputs,
                                     the GroupNorm moments of the NEXT layer fused into this epilogue (needs hout*wout % 64 == 0) */
  int32_t stats_written;          /* OUT: 1 whenever stats_out was filled -- by the fused epilogue, or by the reduce pass of a split-K launch;
                                     0: the statistics were not produced (tile variant without 64-pixel wave strips): run dts_gn_coef */
  void* ev_start; void* ev_stop;  /* optional hipEvent_t pair attached to the conv kernel's own dispatch (start / end of that kernel, as a
                                     kernel trace sees it; no barrier packets between launches).  Measurement only: bench.py's roofline leg */
  const float* gn_coef;           /* optional [n][c1+c2][2] f32 (a, b) from dts_gn_coef / dts_gn_coef_strips: the GroupNorm (+ adaptive
                                     scale/shift) of the INPUT, applied as act(x*a + b) while the conv stages its input tile, so the
                                     normalised tensor is never written (networks.py:168,173-175 feeding conv0 / conv1).  Only launches
                                     for which dts_conv_fuses_gn() returns 1 accept it; padding stays zero (the reference pads AFTER the norm) */
  int32_t gn_silu;                /* 1: act = SiLU, 0: identity */
  float acc_scale;                /* DTS_F16X3 only (0 = 1): out = (conv * acc_scale + bias + bias_nc + residual) * out_scale; undoes the power of
                                     two the packed split-precision weights carry */
  int32_t out_split2;             /* DTS_F16X3 only: 1 = `out` is f16 [n][hout][wout][2*cout] = hi(cout) | lo(cout) of the result * 2^6 per pixel, saturating (the
                                     image dts_split2_f16 would make of it: the qkv projection feeding dts_attention_x3) instead of f32 [..][cout] */
  /* DTS_F16X3 only: the UNetBlock's 1x1 skip convolution (networks.py:164,177: x = conv1(..) + skip(orig)) accumulated by the SAME launch, as a
     second K loop over the block input behind the 3x3 taps: out = ((conv + skip_conv) + bias) * out_scale with `bias` = the two layers' biases
     added by the caller and no `residual`.  Only launches for which dts_conv_folds_skip() returns 1 accept it. */
  int32_t skip_c;                 /* f16 elements per pixel of skip_x (= 2 x the skip conv's input channels), a multiple of 64; 0 = no fold */
  const void* skip_x;             /* split image of the block input, f16 [n][hout >> skip_up][wout >> skip_up][skip_c] */
  const void* skip_w;             /* the 1x1 layer's packed split-precision weight [cout][skip_c] */
  float skip_acc_scale;           /* that weight's power of two (0 = 1), like acc_scale */
  int32_t skip_up;                /* 1: skip_x is at half the output resolution (the nearest-2x upsample of networks.py:82-83 fused into the gather) */
} dts_conv_args;
/* 1 if dts_conv2d would apply a->gn_coef inside the conv for this shape / dtype (3x3, cout % 192 == 0, 16-bit, square power-of-two
 * images >= 16, whole 256-pixel tiles, no fused upsample), else 0: the caller then runs dts_gn_apply first. */
int dts_conv_fuses_gn(const dts_conv_args* a);
/* 1 if dts_conv2d accepts a->skip_* for this launch (DTS_F16X3, 3x3 on the ping-pong kernel with a grid that needs no K split, no residual /
 * out_split2 / upsample of x1), else 0: the caller then runs the 1x1 layer as its own launch and passes its output as `residual`. */
int dts_conv_folds_skip(const dts_conv_args* a);
/* which kernel dts_conv2d takes for these arguments (shape, dtype, residual / gn_coef presence; tuning knobs included): 0 = the 4-wave
 * implicit-GEMM kernel, 6 / 4 = the 8-wave ping-pong / halo kernel with 192- / 128-cout blocks, -1 = invalid arguments.  Measurement aid:
 * bench.py attributes its per-launch times and algorithmic bytes to the kernel name a trace will show. */
int dts_conv_kernel(const dts_conv_args* a);
int dts_conv2d(dts_conv_args* a, dts_stream s);      /* writes a->stats_written; no state is kept between calls (thread-safe) */

/* first / last convolutions of the U-Nets (3 image channels; direct, not MFMA) */
/* x f32 NCHW [n][3][h][w] -> out NHWC [n][h][w][cout]; w f32 OIHW [cout][3][3][3] */
int dts_conv_in3(const float* x, const float* w, const float* bias, void* out, int dtype,
                 int n, int h, int w_, int cout, dts_stream s);
/* x NHWC [n][h][w][c] -> out f32 NCHW [n][3][h][w]; w f32 [3][3][3][c] (O,kh,kw,I) */
int dts_conv_out3(const void* x, int dtype, const float* w, const float* bias, float* out,
                  int n, int h, int w_, int c, dts_stream s);

/* ---- K4/K5: GroupNorm (+ adaptive scale/shift, SiLU, 2x2 avg-pool) (networks.py:104-106,168-175) -- */
/* number of floats of workspace dts_gn_coef needs */
int64_t dts_gn_ws_floats(int n, int groups);
/* coef[n][C][2] = (a, b) such that groupnorm(x)*gamma+beta [*(1+scale)+shift] == x*a + b.
 * x = concat(x1, x2) on channels; scale_shift (dtype, [n][ld_ss], scale = [0,C), shift = [C,2C)) may be NULL. */
int dts_gn_coef(const void* x1, int c1, const void* x2, int c2, int dtype, int n, int hw, int groups, float eps,
                const float* gamma, const float* beta, const void* scale_shift, int ld_ss,
                float* coef, float* ws, dts_stream s);
/* same coefficients from strip statistics emitted by dts_conv2d (stats_out) for each source: st1 [n*hw/64][c1][2],
 * st2 [n*hw/64][c2][2] (NULL when c2 == 0); no pass over x at all. */
int dts_gn_coef_strips(const float* st1, int c1, const float* st2, int c2, int dtype, int n, int hw, int groups, float eps,
                       const float* gamma, const float* beta, const void* scale_shift, int ld_ss, float* coef, dts_stream s);
/* out = act(x*a + b); pool=1 averages 2x2 pixel blocks after the activation (resample filter [1,1],
 * networks.py:84-85; unet.py:213-215 avg_pool) and writes [n][h/2][w/2][C]. */
int dts_gn_apply(const void* x1, int c1, const void* x2, int c2, int dtype, const float* coef,
                 void* out, int n, int h, int w, int silu, int pool, dts_stream s);
/* the same pass in the split-precision mode (DTS_F16X3): x / coef as above in f32, out = the f16 split image [n][h'][w'][2*C] (per 32
 * channels hi | lo * 2^11, the arithmetic and layout of dts_split3_f16) that dts_conv2d(dtype = DTS_F16X3) reads -- the f32 normalised
 * tensor is never written.  C a multiple of 32.  raw_out (optional, same shape as out): the split image of the UN-normalised input rows
 * (2x2-averaged like dts_resample2x when pool = 1) is written by the same pass -- the operand of the block's 1x1 skip convolution
 * (networks.py:177), bit-identical to dts_split3_f16 of the (resampled) input. */
int dts_gn_apply_x3(const float* x1, int c1, const float* x2, int c2, const float* coef, void* out, void* raw_out, int n, int h, int w, int silu,
                    int pool, dts_stream s);
/* single-launch variant for low-resolution levels (hw <= ~256): statistics + apply in one kernel, one block per
 * (group, sample); same result as dts_gn_coef + dts_gn_apply(pool=0).  Channels per group must be even and <= 64. */
int dts_gn_fused(const void* x1, int c1, const void* x2, int c2, int dtype, int n, int hw, int groups, float eps,
                 const float* gamma, const float* beta, const void* scale_shift, int ld_ss, void* out, int silu, dts_stream s);
/* 2x resampling of an NHWC tensor for the skip path: mode 0 = 2x2 average (down), 1 = nearest (up). */
int dts_resample2x(const void* x, void* out, int dtype, int n, int h, int w, int c, int mode, dts_stream s);

/* ---- K6: fused self-attention (networks.py:113-118,181-185; unet.py:355-372,388-407) -------------- */
/* qkv NHWC-flattened [n][t][3*heads*d] laid out q[heads][d] | k[heads][d] | v[heads][d];
 * out [n][t][heads*d]; softmax(q.k * scale) in f32. d in {64,128,256} (and 512 in the 16-bit types: the SD VAE's mid block); any t >= 1. */
int dts_attention(const void* qkv, void* out, int dtype, int n, int t, int heads, int d, float scale, dts_stream s);
/* the same attention in the split-precision mode (DTS_F16X3; d = 64): qkv_split = dts_split2_f16 of the f32 qkv tensor, f16 [n][t][6*heads*d] =
 * hi(3C) | lo(3C) per token (also what dts_conv2d writes with out_split2); out f32 [n][t][heads*d], or with out_split3 = 1 the f16 operand
 * image [n][t][2*heads*d] (per 32 channels hi | lo * 2^11: dts_split3_f16's arithmetic and layout) the proj convolution reads.  Q.K^T and P.V on the 16-bit matrix cores with hi/lo operand pairs (the lo*lo term,
 * 2^-22, dropped), softmax in f32: the f32 kernel's accuracy without the f32 matrix instruction's 1/16 rate. */
int dts_attention_x3(const void* qkv_split, void* out, int out_split3, int n, int t, int heads, int d, float scale, dts_stream s);

/* ---- K7/K8: embedding MLP pieces and EDM preconditioning (networks.py:200-206,437-447,654-668) ---- */
/* y[m][n] = act_out( act_in(x[m][:]) . w[n][:] + bias[n] (+ y[m][n] if accumulate) ); all f32; act: 0 none, 1 SiLU */
int dts_linear(const float* x, int ldx, const float* w, const float* bias, float* y, int ldy,
               int m, int k, int n, int act_in, int act_out, int accumulate, dts_stream s);
/* out[n][0:half] = cos(v[n]*freqs), out[n][half:2half] = sin(..) (swap=1: sin first, networks.py:323) */
int dts_pos_embedding(const float* v, const float* freqs, float* out, int n, int half, int swap, dts_stream s);
/* coef[n][4] = (c_skip, c_out, c_in, c_noise) in f32 from sigma (f64, nsigma = 1 or n); xin = c_in * f32(x) */
int dts_edm_precond_in(const double* x, const double* sigma, int nsigma, float sigma_data,
                       float* xin, float* coef, int n, int chw, dts_stream s);
/* D = c_skip*f32(x) + c_out*F  (f32, NCHW) */
int dts_edm_precond_out(const double* x, const float* F, const float* coef, float* D, int n, int chw, dts_stream s);
/* split-precision operand image for dts_conv2d(dtype = DTS_F16X3): row p of concat(x1, x2) (f32 rows of c1 / c2 channels, x2 may be NULL
 * with c2 = 0; c1, c2 multiples of 8, C = c1 + c2 a multiple of 32) -> out[p][64*g + j] = hi, out[p][64*g + 32 + j] = lo * 2^11 of channel
 * 32*g + j: hi = f16(x) (0 when that would be subnormal; +-65504 beyond the f16 range), lo = f16((x - hi) * 2^11).  out: f16 [rows][2*C]. */
int dts_split3_f16(const float* x1, int c1, const float* x2, int c2, void* out, int64_t rows, dts_stream s);
/* out[p][0:c] = hi, [c:2c] = lo of x[p][:] * 2^6 (hi = f16(y), lo = f16(y - hi); |y| saturates at 65504, i.e. |x| at 1023.5): the operand image of dts_attention_x3 */
int dts_split2_f16(const float* x, int c, void* out, int64_t rows, dts_stream s);
/* f32 -> dtype cast of a dense array (embedding -> activation dtype) and back */
int dts_cast_from_f32(const float* src, void* dst, int dtype, int64_t count, dts_stream s);
int dts_cast_to_f32(const void* src, int dtype, float* dst, int64_t count, dts_stream s);

/* ---- K9: fused Heun / Euler ODE step with churn, fp64 state (edm/main.py:82-96) ------------------- */
/* x_hat[nb] = x_cur[src(i)] + noise_coef * eps[i];  src(i) = i % xb (bcast=0, Tensor.repeat, edm/main.py:803)
 * or i / (nb/xb) (bcast=1, repeat_interleave, edm/main.py:107).  eps is f64 (eps_f32=0) or f32 (edm/main.py:446). */
int dts_heun_xhat(const double* x_cur, int xb, int bcast, const void* eps, int eps_f32, double noise_coef,
                  double* x_hat, int nb, int chw, dts_stream s);
/* d_cur = (x_hat - D)/t_hat ; x_next = x_hat + (t_next - t_hat)*d_cur */
int dts_heun_euler(const double* x_hat, const float* D, double t_hat, double t_next,
                   double* d_cur, double* x_next, int64_t count, dts_stream s);
/* x_next = x_hat + (t_next - t_hat)*(0.5*d_cur + 0.5*(x_next - D2)/t_next)   (in place on x_next) */
int dts_heun_correct(const double* x_hat, const float* D2, const double* d_cur, double t_hat, double t_next,
                     double* x_next, int64_t count, dts_stream s);

/* ---- K10/K11: scorer pre-processing and brightness reward (edm/main.py:126; scorers.py:38-52) ----- */
/* u8 = trunc(clip(x*127.5+128, 0, 255)); x is f64 (is_f32=0) or f32 (1): arithmetic in f64 as the EDM loop's (.to(float64) first);
 * is_f32=2: f32 input AND f32 arithmetic, as the SD loop's (pipeline_stable_diffusion.py:1116) */
int dts_quantize_u8(const void* x, int is_f32, uint8_t* out, int64_t count, dts_stream s);
/* rewards[n] = clamp(mean_hw(0.2126 R + 0.7152 G + 0.0722 B)/1, 0, 1) on u8/255 images NCHW [n][3][h][w] */
int dts_brightness(const uint8_t* img, float* rewards, int n, int hw, dts_stream s);
/* CLIP reward tail (sd/scorers.py:182-183,205-211): out[i] = <a_i/||a_i||, b_i/||b_i||>, f32; b has n rows or 1 (one prompt). */
int dts_cosine_rows(const float* a, const float* b, int b_rows, float* out, int n, int d, dts_stream s);
/* CLIP image pre-processing on the device (sd/scorers.py:166-180: `self.processor(images=...)` = transformers CLIPImageProcessor = Pillow's
 * bicubic resize of the uint8 image + rescale + normalise).  dts_resample_u8: ONE pass of Pillow's separable 8-bit resampling (Resample.c):
 * dst = clip8((2^21 + sum_k src[first + k] * coef[k]) >> 22) along a row (axis 1: [planes][h][w] -> [planes][h][out_len]) or down a column
 * (axis 0: -> [planes][out_len][w]); bounds [out_len][2] = (first, count), coefs [out_len][ksize] = Pillow's integer coefficients (built by
 * clip_preprocess.resample_tables).  dts_lut_u8_f32: out[n][c][hw] = lut[c][img]: the processor's rescale + normalise as a 256-entry table
 * per channel. */
int dts_resample_u8(const uint8_t* src, uint8_t* dst, int planes, int h, int w, int out_len, int axis, const int32_t* bounds,
                    const int32_t* coefs, int ksize, dts_stream s);
int dts_lut_u8_f32(const uint8_t* img, const float* lut, float* out, int n, int c, int hw, dts_stream s);
/* f32 NCHW = u8 / 255.0f (scorers.py:153) */
int dts_u8_to_unit_f32(const uint8_t* img, float* out, int64_t count, dts_stream s);

/* ---- K12 tail: attention pool + softmax-gather (unet.py:61-69; scorers.py:162-172) ---------------- */
/* tokens[n][hw+1][c]: token 0 = mean_hw(x) + pos[:,0]; token 1+p = x[n][p] + pos[:,1+p]; pos f32 [c][hw+1] */
int dts_attnpool_tokens(const void* x, const float* pos, void* tokens, int dtype, int n, int hw, int c, dts_stream s);
/* out f32 [n][c] = src[n][token][:] */
int dts_take_token(const void* src, int dtype, float* out, int n, int t, int c, int token, dts_stream s);
/* rewards[n] = softmax(logits[n][:])[target[n]] */
int dts_softmax_gather(const float* logits, const int32_t* target, float* rewards, int n, int k, dts_stream s);

/* ---- K14: epsilon-greedy / zero-order candidate-noise builder (edm/main.py:749-800) --------------- */
/* g: host-drawn standard normals [nb][chw] f64 (uploaded); for row i (candidate n = i / b, sample = i % b):
 *   mode[n] == 0 : cand[i] = g[i]                                   (fresh Gaussian, edm/main.py:795)
 *   mode[n] == 1 : cand[i] = pivot[i % b] + (double)scale[n] * (g[i] / ||g[i]||_2)   (edm/main.py:767-788) */
int dts_candidate_noise(const double* pivot, const double* g, const int32_t* mode, const float* scale,
                        double* cand, int nb, int b, int chw, dts_stream s);
/* The SD backend's builder (sd/diffusers/.../pipeline_stable_diffusion.py:1371-1379), in the latents' storage type `dtype` with the
 * reference's roundings: u host-drawn normals [n][count]; mode[c] == 0: cand[c] = u[c] (fresh noise, :1375);
 * mode[c] == 1: cand[c] = pivot + ((((u[c] / ||u[c]||) * scale[3c]) * scale[3c+1]) * scale[3c+2]), scale[3c..] = (rand, lambda, sqrt(count)) as f32:
 * the reference's three tensor-by-scalar products, each rounded to `dtype` (:1377-1379).  pivot [count], scale [n][3]. */
int dts_candidate_noise_sd(const void* pivot, const void* u, const int32_t* mode, const float* scale, void* cand, int dtype, int n,
                           int64_t count, dts_stream s);

/* ---- K13: DDIM candidate step, SD backend (scheduling_ddim.py:402-463), eta*std = sigma_t ---------- */
/* For i in [0,count): x0 = (x - sqrt(1-a_t) e)/sqrt(a_t); [e' = (x - sqrt(a_t) x0)/sqrt(1-a_t) == e];
 * prev[cand] = sqrt(a_prev) x0 + sqrt(1 - a_prev - sigma_t^2) e + sigma_t z[cand]; ncand candidates share
 * one (x, e).  f32 math on dtype storage (DTS_F16 as the reference; DTS_F32 for tests). */
/* classifier-free guidance, out = uncond + guidance*(cond - uncond) (pipeline_stable_diffusion.py:1072-1074) */
int dts_cfg_combine(const void* uncond, const void* cond, float guidance, void* out, int dtype, int64_t count, dts_stream s);
int dts_ddim_candidates(const void* x, const void* e, const void* z, void* prev, void* x0_out, int dtype,
                        float alpha_t, float alpha_prev, float sigma_t, int ncand, int64_t count, dts_stream s);

#ifdef __cplusplus
}
#endif
#endif /* DTS_H_ */
