// HBM-bound pieces of the hot path: layout/packing, EDM preconditioning (K8), the fused fp64 Heun/Euler
// step with churn (K9), uint8 quantisation + brightness reward (K10/K11), embedding MLP bits (K7), the
// classifier tail (K12), the epsilon-greedy candidate builder (K14) and the DDIM candidate step (K13).
// Every kernel is a grid-stride loop with consecutive lanes on consecutive addresses.
#include <stdarg.h>
#include "dts_common.h"

// ---- error string (thread local) --------------------------------------------------------------------
static thread_local char g_err[512] = "";
void dts_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
extern "C" const char* dts_last_error(void) { return g_err; }
extern "C" int dts_version(void) { return DTS_ABI_VERSION; }

// ---- tuning knobs (dts_common.h) -------------------------------------------------------------------
static int g_knob[DTS_KNOB_COUNT];
static bool g_knob_init = false;
static const char* const g_knob_env[DTS_KNOB_COUNT] = {"DTS_ATT_XCD", "DTS_ATT_QT", "DTS_CONV_TILE", "DTS_CONV_SPLITS",
                                                       "DTS_CONV_VARIANT", "DTS_GN_FUSE", "DTS_ATT_DB", "DTS_CONV_STAGES", "DTS_CONV_WAVES",
                                                       "DTS_CONV_HALF_ROUND", "DTS_CONV_EPI32", "DTS_CONV_SKIP_FOLD"};
static void knob_init() {
  if (g_knob_init) return;
  for (int i = 0; i < DTS_KNOB_COUNT; ++i) {
    const char* e = g_knob_env[i] ? getenv(g_knob_env[i]) : nullptr;
    g_knob[i] = e ? atoi(e) : -1;
  }
  g_knob_init = true;
}
int dts_knob_get(int knob) {
  knob_init();
  return (knob >= 0 && knob < DTS_KNOB_COUNT) ? g_knob[knob] : -1;
}
extern "C" int dts_set_tuning(int knob, int value) {
  knob_init();
  DTS_CHECK_ARG(knob >= 0 && knob < DTS_KNOB_COUNT, "dts_set_tuning: knob %d", knob);
  g_knob[knob] = value;
  return DTS_OK;
}
extern "C" int dts_get_tuning(int knob) { return dts_knob_get(knob); }

namespace {

inline int grid1d(long long total, int block = 256, int cap = 256 * 8) {
  long long g = (total + block - 1) / block;
  if (g > cap) g = cap;
  if (g < 1) g = 1;
  return (int)g;
}
#define GSL(i, total) for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < (total); i += (long long)gridDim.x * blockDim.x)

// ---- layout ---------------------------------------------------------------------------------------
template <typename T>
__global__ void nchw_to_nhwc_kernel(const float* __restrict__ src, T* __restrict__ dst, int n, int c, int h, int w) {
  const long long total = (long long)n * c * h * w;
  GSL(i, total) {                                           // i indexes dst (NHWC)
    const int ci = (int)(i % c);
    long long r = i / c;
    const int x = (int)(r % w); r /= w;
    const int y = (int)(r % h);
    const int ni = (int)(r / h);
    st1<T>(dst + i, src[(((size_t)ni * c + ci) * h + y) * w + x]);
  }
}
template <typename T>
__global__ void nhwc_to_nchw_kernel(const T* __restrict__ src, float* __restrict__ dst, int n, int c, int h, int w) {
  const long long total = (long long)n * c * h * w;
  GSL(i, total) {                                           // i indexes dst (NCHW)
    const int x = (int)(i % w);
    long long r = i / w;
    const int y = (int)(r % h); r /= h;
    const int ci = (int)(r % c);
    const int ni = (int)(r / c);
    dst[i] = ld1<T>(src + (((size_t)ni * h + y) * w + x) * c + ci);
  }
}
template <typename T>
__global__ void pack_w_kernel(const float* __restrict__ w, T* __restrict__ dst, int O, int I, int kh, int kw,
                              const int32_t* __restrict__ perm) {
  const long long total = (long long)O * I * kh * kw;
  GSL(i, total) {                                           // dst index: [o][kh][kw][i]
    const int ii = (int)(i % I);
    long long r = i / I;
    const int x = (int)(r % kw); r /= kw;
    const int y = (int)(r % kh);
    const int o = (int)(r / kh);
    const int so = perm ? perm[o] : o;
    st1<T>(dst + i, w[(((size_t)so * I + ii) * kh + y) * kw + x]);
  }
}
template <typename T>
__global__ void cast_from_f32_kernel(const float* __restrict__ s, T* __restrict__ d, long long count) {
  GSL(i, count) st1<T>(d + i, s[i]);
}
template <typename T>
__global__ void cast_to_f32_kernel(const T* __restrict__ s, float* __restrict__ d, long long count) {
  GSL(i, count) d[i] = ld1<T>(s + i);
}

// ---- K8: EDM preconditioning (networks.py:655-667; all f32 as the reference) ---------------------------
__global__ void precond_in_kernel(const double* __restrict__ x, const double* __restrict__ sigma, int nsigma, float sd,
                                  float* __restrict__ xin, float* __restrict__ coef, int n, int chw) {
  const long long total = (long long)n * chw;
  GSL(i, total) {
    const int ni = (int)(i / chw);
    const float sg = (float)sigma[nsigma == 1 ? 0 : ni];
    const float s2 = sg * sg, d2 = sd * sd;
    const float c_in = 1.0f / sqrtf(d2 + s2);
    xin[i] = c_in * (float)x[i];
    if (i - (long long)ni * chw == 0) {
      coef[ni * 4 + 0] = d2 / (s2 + d2);
      coef[ni * 4 + 1] = sg * sd / sqrtf(s2 + d2);
      coef[ni * 4 + 2] = c_in;
      coef[ni * 4 + 3] = logf(sg) / 4.0f;
    }
  }
}
__global__ void precond_out_kernel(const double* __restrict__ x, const float* __restrict__ F, const float* __restrict__ coef,
                                   float* __restrict__ D, int n, int chw) {
  const long long total = (long long)n * chw;
  GSL(i, total) {
    const int ni = (int)(i / chw);
    D[i] = coef[ni * 4 + 0] * (float)x[i] + coef[ni * 4 + 1] * F[i];
  }
}

// ---- K9: Heun / Euler step, fp64 state (edm/main.py:82-96) ---------------------------------------------
template <typename E>
__global__ void heun_xhat_kernel(const double* __restrict__ x_cur, int xb, int bcast, const E* __restrict__ eps, double coef,
                                 double* __restrict__ x_hat, int nb, int chw) {
  const long long total = (long long)nb * chw;
  const int rep = nb / xb;
  GSL(i, total) {
    const int row = (int)(i / chw);
    const long long e = i - (long long)row * chw;
    const int src = bcast ? row / rep : row % xb;
    x_hat[i] = x_cur[(long long)src * chw + e] + coef * (double)eps[i];
  }
}
__global__ void heun_euler_kernel(const double* __restrict__ x_hat, const float* __restrict__ D, double t_hat, double t_next,
                                  double* __restrict__ d_cur, double* __restrict__ x_next, long long count) {
  const double dt = t_next - t_hat;
  GSL(i, count) {
    const double xh = x_hat[i];
    const double d = (xh - (double)D[i]) / t_hat;
    d_cur[i] = d;
    x_next[i] = xh + dt * d;
  }
}
__global__ void heun_correct_kernel(const double* __restrict__ x_hat, const float* __restrict__ D2, const double* __restrict__ d_cur,
                                    double t_hat, double t_next, double* __restrict__ x_next, long long count) {
  const double dt = t_next - t_hat;
  GSL(i, count) {
    const double dp = (x_next[i] - (double)D2[i]) / t_next;
    x_next[i] = x_hat[i] + dt * (0.5 * d_cur[i] + 0.5 * dp);
  }
}

// ---- K10/K11 -----------------------------------------------------------------------------------------
template <typename E>
__global__ void quantize_kernel(const E* __restrict__ x, uint8_t* __restrict__ out, long long count) {
  GSL(i, count) {
    // the reference converts the f32 denoiser output to f64 first (edm/main.py:87,92 .to(float64)), so the
    // affine map and the clip run in f64 whatever the storage type
    double v = (double)x[i] * 127.5 + 128.0;
    v = v < 0.0 ? 0.0 : (v > 255.0 ? 255.0 : v);
    out[i] = (uint8_t)v;                                   // truncation, as Tensor.to(torch.uint8)
  }
}
// SD backend: the image is f32 and the reference's affine map runs in f32 (pipeline_stable_diffusion.py:1116)
__global__ void quantize_f32math_kernel(const float* __restrict__ x, uint8_t* __restrict__ out, long long count) {
  GSL(i, count) {
    float v = x[i] * 127.5f + 128.0f;
    v = v < 0.f ? 0.f : (v > 255.f ? 255.f : v);
    out[i] = (uint8_t)v;
  }
}
// classifier-free guidance: out = u + g*(c - u)   (pipeline_stable_diffusion.py:1072-1074)
template <typename T>
__global__ void cfg_combine_kernel(const T* __restrict__ u, const T* __restrict__ c, float g, T* __restrict__ out, long long count) {
  GSL(i, count) {
    const float uv = ld1<T>(u + i), cv = ld1<T>(c + i);
    st1<T>(out + i, uv + g * (cv - uv));
  }
}
// one block per image; u8/255 in f32, weighted channel sum in f32 (as the reference), spatial mean in f64
// ---- CLIP image pre-processing on the device (sd/scorers.py:166-180 -> transformers CLIPImageProcessor -> Pillow) ----------------
// One pass of Pillow's separable 8-bit resampling (src/libImaging/Resample.c, ImagingResampleHorizontal_8bpc / Vertical_8bpc): every
// output sample is  clip8((2^(P-1) + sum_k src[first + k] * coef[k]) >> P)  with P = 22 and the integer coefficients Pillow derives
// from its double-precision filter weights (built on the host, clip_preprocess.py: the table IS Pillow's).  Integer arithmetic end to
// end, so the result is Pillow's bit for bit.  axis 1: along a row ([planes][h][w] -> [planes][h][out_len]); axis 0: down a column
// ([planes][h][w] -> [planes][out_len][w]).  bounds[o] = (first, count), coefs[o][ksize].
__global__ __launch_bounds__(256) void resample_u8_kernel(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, int planes, int h, int w,
                                                           int out_len, int axis, const int* __restrict__ bounds,
                                                           const int* __restrict__ coefs, int ksize) {
  const long long total = axis ? (long long)planes * h * out_len : (long long)planes * out_len * w;
  GSL(i, total) {
    int o, first, cnt;
    long long base; int stride;
    if (axis) {                                      // i = (plane*h + y) * out_len + o
      o = (int)(i % out_len);
      base = (i / out_len) * w; stride = 1;
    } else {                                         // i = (plane*out_len + o) * w + x
      const int x = (int)(i % w);
      const long long r = i / w;
      o = (int)(r % out_len);
      base = (r / out_len) * (long long)h * w + x; stride = w;
    }
    first = bounds[2 * o]; cnt = bounds[2 * o + 1];
    const int* k = coefs + (size_t)o * ksize;
    int ss = 1 << 21;
    for (int t = 0; t < cnt; ++t) ss += (int)src[base + (long long)(first + t) * stride] * k[t];
    ss >>= 22;                                        // arithmetic shift, then Pillow's clip8 lookup: clamp to [0, 255]
    dst[i] = (uint8_t)(ss < 0 ? 0 : (ss > 255 ? 255 : ss));
  }
}

// out[n][c][hw] = lut[c][img[n][c][hw]]: rescale + normalise of the image processor as a table of the 256 values per channel
__global__ __launch_bounds__(256) void lut_u8_f32_kernel(const uint8_t* __restrict__ img, const float* __restrict__ lut, float* __restrict__ out,
                                                          int c, int hw, long long total) {
  GSL(i, total) {
    const int ch = (int)((i / hw) % c);
    out[i] = lut[ch * 256 + img[i]];
  }
}

__global__ __launch_bounds__(256) void brightness_kernel(const uint8_t* __restrict__ img, float* __restrict__ rewards, int hw) {
  __shared__ double red[4];
  const uint8_t* p = img + (size_t)blockIdx.x * 3 * hw;
  double acc = 0.0;
  for (int i = threadIdx.x; i < hw; i += blockDim.x) {
    const float r = (float)p[i] / 255.0f, g = (float)p[hw + i] / 255.0f, b = (float)p[2 * hw + i] / 255.0f;
    const float lum = r * 0.2126f + g * 0.7152f + b * 0.0722f;
    acc += (double)lum;
  }
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    const double tot = red[0] + red[1] + red[2] + red[3];
    float m = (float)(tot / (double)hw);
    rewards[blockIdx.x] = fminf(fmaxf(m, 0.f), 1.f);
  }
}
// CLIP reward tail (sd/scorers.py:182-183,205-211): e = x / ||x||_2 for image and text embeddings, reward = sum(e_img * e_txt).
// One wave per row; the norms and the dot product are accumulated in f32 like the reference's f32 tensors; b has n rows or 1.
__global__ __launch_bounds__(64) void cosine_rows_kernel(const float* __restrict__ a, const float* __restrict__ b, int b_rows,
                                                          float* __restrict__ out, int d) {
  const float* pa = a + (size_t)blockIdx.x * d;
  const float* pb = b + (size_t)(b_rows == 1 ? 0 : blockIdx.x) * d;
  float na = 0.f, nb = 0.f;
  for (int i = threadIdx.x; i < d; i += 64) { na += pa[i] * pa[i]; nb += pb[i] * pb[i]; }
  na = sqrtf(wave_sum(na));
  nb = sqrtf(wave_sum(nb));
  float dot = 0.f;
  for (int i = threadIdx.x; i < d; i += 64) dot += (pa[i] / na) * (pb[i] / nb);
  dot = wave_sum(dot);
  if (threadIdx.x == 0) out[blockIdx.x] = dot;
}
__global__ void u8_to_unit_kernel(const uint8_t* __restrict__ img, float* __restrict__ out, long long count) {
  GSL(i, count) out[i] = (float)img[i] / 255.0f;
}

// ---- K7 ------------------------------------------------------------------------------------------------
// one wave per (output column, block of 8 rows): grid.y walks the row blocks so that enough waves are in flight to
// hide the load latency (a single wave looping over all rows was latency-bound: 200-400 us for a 64x1000x768 problem)
__global__ __launch_bounds__(256) void linear_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ w,
                                                      const float* __restrict__ bias, float* __restrict__ y, int ldy, int m, int k,
                                                      int n, int act_in, int act_out, int accumulate) {
  const int col = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (col >= n) return;
  const float* wr = w + (size_t)col * k;
  constexpr int RB = 8;
  const int row0 = blockIdx.y * RB;
  float acc[RB];
#pragma unroll
  for (int r = 0; r < RB; ++r) acc[r] = 0.f;
#pragma unroll 4
  for (int j = lane; j < k; j += 64) {
    const float wv = wr[j];
#pragma unroll
    for (int r = 0; r < RB; ++r) {
      if (row0 + r < m) {
        float xv = x[(size_t)(row0 + r) * ldx + j];
        if (act_in) xv = silu_f(xv);
        acc[r] += xv * wv;
      }
    }
  }
#pragma unroll
  for (int r = 0; r < RB; ++r) acc[r] = wave_sum(acc[r]);
  if (lane == 0) {
#pragma unroll
    for (int r = 0; r < RB; ++r) {
      if (row0 + r < m) {
        float v = acc[r] + (bias ? bias[col] : 0.f);
        float* o = y + (size_t)(row0 + r) * ldy + col;
        if (accumulate) v += *o;
        if (act_out) v = silu_f(v);
        *o = v;
      }
    }
  }
}
// the same with 16-byte loads (k % 4 == 0, 16-byte aligned rows): a lane takes four consecutive k per trip, a quarter of the load
// instructions of the scalar form (27 -> ~13 us for the 64 x 768 x 768 embedding layers, which are pure load-issue work)
__global__ __launch_bounds__(256) void linear4_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ w,
                                                       const float* __restrict__ bias, float* __restrict__ y, int ldy, int m, int k,
                                                       int n, int act_in, int act_out, int accumulate) {
  const int col = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (col >= n) return;
  const float* wr = w + (size_t)col * k;
  constexpr int RB = 8;
  const int row0 = blockIdx.y * RB;
  float acc[RB];
#pragma unroll
  for (int r = 0; r < RB; ++r) acc[r] = 0.f;
#pragma unroll 2
  for (int j = lane * 4; j < k; j += 256) {
    const float4 wv = *reinterpret_cast<const float4*>(wr + j);
#pragma unroll
    for (int r = 0; r < RB; ++r) {
      if (row0 + r < m) {
        float4 xv = *reinterpret_cast<const float4*>(x + (size_t)(row0 + r) * ldx + j);
        if (act_in) { xv.x = silu_f(xv.x); xv.y = silu_f(xv.y); xv.z = silu_f(xv.z); xv.w = silu_f(xv.w); }
        acc[r] += xv.x * wv.x;
        acc[r] += xv.y * wv.y;
        acc[r] += xv.z * wv.z;
        acc[r] += xv.w * wv.w;
      }
    }
  }
#pragma unroll
  for (int r = 0; r < RB; ++r) acc[r] = wave_sum(acc[r]);
  if (lane == 0) {
#pragma unroll
    for (int r = 0; r < RB; ++r) {
      if (row0 + r < m) {
        float v = acc[r] + (bias ? bias[col] : 0.f);
        float* o = y + (size_t)(row0 + r) * ldy + col;
        if (accumulate) v += *o;
        if (act_out) v = silu_f(v);
        *o = v;
      }
    }
  }
}
__global__ void pos_embedding_kernel(const float* __restrict__ v, const float* __restrict__ freqs, float* __restrict__ out, int n,
                                     int half, int swap) {
  const long long total = (long long)n * half;
  GSL(i, total) {
    const int ni = (int)(i / half), j = (int)(i - (long long)ni * half);
    const float a = v[ni] * freqs[j];
    const float c = cosf(a), s = sinf(a);
    float* o = out + (size_t)ni * 2 * half;
    o[j] = swap ? s : c;
    o[half + j] = swap ? c : s;
  }
}

// ---- K12 tail ------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void attnpool_tokens_kernel(const T* __restrict__ x, const float* __restrict__ pos,
                                                               T* __restrict__ tok, int hw, int c) {
  const int n = blockIdx.x;
  const T* xs = x + (size_t)n * hw * c;
  T* ts = tok + (size_t)n * (hw + 1) * c;
  for (int ch = threadIdx.x; ch < c; ch += blockDim.x) {
    float sum = 0.f;
    for (int p = 0; p < hw; ++p) {
      const float v = ld1<T>(xs + (size_t)p * c + ch);
      sum += v;
      st1<T>(ts + (size_t)(p + 1) * c + ch, v + pos[(size_t)ch * (hw + 1) + p + 1]);
    }
    st1<T>(ts + ch, sum / (float)hw + pos[(size_t)ch * (hw + 1)]);
  }
}
template <typename T>
__global__ void take_token_kernel(const T* __restrict__ src, float* __restrict__ out, int n, int t, int c, int token) {
  const long long total = (long long)n * c;
  GSL(i, total) {
    const int ni = (int)(i / c), ch = (int)(i - (long long)ni * c);
    out[i] = ld1<T>(src + ((size_t)ni * t + token) * c + ch);
  }
}
__global__ __launch_bounds__(256) void softmax_gather_kernel(const float* __restrict__ logits, const int32_t* __restrict__ target,
                                                              float* __restrict__ rewards, int k) {
  __shared__ float red[4];
  __shared__ float bc;
  const float* row = logits + (size_t)blockIdx.x * k;
  float mx = -INFINITY;
  for (int j = threadIdx.x; j < k; j += blockDim.x) mx = fmaxf(mx, row[j]);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mx;
  __syncthreads();
  if (threadIdx.x == 0) bc = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  __syncthreads();
  mx = bc;
  float sum = 0.f;
  for (int j = threadIdx.x; j < k; j += blockDim.x) sum += expf(row[j] - mx);
  sum = wave_sum(sum);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = sum;
  __syncthreads();
  if (threadIdx.x == 0) rewards[blockIdx.x] = expf(row[target[blockIdx.x]] - mx) / (red[0] + red[1] + red[2] + red[3]);
}

// ---- K14 -----------------------------------------------------------------------------------------------
// one block per candidate row; fp64 L2 norm by block reduction, then the axpy
__global__ __launch_bounds__(256) void candidate_noise_kernel(const double* __restrict__ pivot, const double* __restrict__ g,
                                                               const int32_t* __restrict__ mode, const float* __restrict__ scale,
                                                               double* __restrict__ cand, int b, int chw) {
  __shared__ double red[4];
  const int row = blockIdx.x, cn = row / b, sample = row - cn * b;
  const double* gr = g + (size_t)row * chw;
  double* out = cand + (size_t)row * chw;
  if (mode[cn] == 0) {
    for (int i = threadIdx.x; i < chw; i += blockDim.x) out[i] = gr[i];
    return;
  }
  double ss = 0.0;
  for (int i = threadIdx.x; i < chw; i += blockDim.x) ss += gr[i] * gr[i];
  ss = wave_sum(ss);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = ss;
  __syncthreads();
  const double nrm = sqrt(red[0] + red[1] + red[2] + red[3]);
  const double sc = (double)scale[cn];
  const double* pv = pivot + (size_t)sample * chw;
  for (int i = threadIdx.x; i < chw; i += blockDim.x) out[i] = pv[i] + sc * (gr[i] / nrm);
}

// K14 of the SD backend (pipeline_stable_diffusion.py:1371-1379): the same construction in the LATENTS' storage type T, rounding where
// the reference's tensor ops round -- norm (f32 accumulation, result in T), u / norm (in T), * python scalar (f32 math, result in T),
// pivot + ... (in T).  One block per candidate.
template <typename T>
__global__ __launch_bounds__(256) void candidate_noise_sd_kernel(const T* __restrict__ pivot, const T* __restrict__ u,
                                                                  const int32_t* __restrict__ mode, const float* __restrict__ scale,
                                                                  T* __restrict__ cand, int count) {
  __shared__ float red[4];
  const int cn = blockIdx.x;
  const T* ur = u + (size_t)cn * count;
  T* out = cand + (size_t)cn * count;
  if (mode[cn] == 0) {                                        // fresh Gaussian (:1375)
    for (int i = threadIdx.x; i < count; i += blockDim.x) out[i] = ur[i];
    return;
  }
  float ss = 0.f;
  for (int i = threadIdx.x; i < count; i += blockDim.x) { const float v = ld1<T>(ur + i); ss += v * v; }
  ss = wave_sum(ss);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = ss;
  __syncthreads();
  T nrm_t;
  st1<T>(&nrm_t, sqrtf((red[0] + red[1]) + (red[2] + red[3])));
  // `to_add * rand * lambda * sqrt(numel)` (:1379) is ((to_add * rand) * lambda) * sqrt: three tensor-by-scalar products, each rounded to T
  // (torch multiplies in f32 with the scalar cast to f32 and rounds the result to the tensor's type)
  float nrm = ld1<T>(&nrm_t);
  // (hidden from the optimiser: with both operands visibly widened from T it narrows the quotient to a HALF-precision division, which the
  // target lowers through v_rcp_f16 -- one ulp off torch's float division rounded once; tests/test_gpu_sd.py checks bit equality)
  asm volatile("" : "+v"(nrm));
  const float s0 = scale[3 * cn], s1 = scale[3 * cn + 1], s2 = scale[3 * cn + 2];
  for (int i = threadIdx.x; i < count; i += blockDim.x) {
    T a, b, c;
    st1<T>(&a, ld1<T>(ur + i) / nrm);                         // to_add / torch.norm(to_add)
    // each product is rounded to f32 FIRST and then to T, as torch does (f32 arithmetic, result cast to the tensor's type): left to itself the
    // compiler fuses multiply and narrowing into v_fma_mixlo_f16 -- ONE rounding, which differs from the double rounding in ~1e-4 of the values
    float t_ = ld1<T>(&a) * s0;                               // * torch.rand(1).item()
    asm volatile("" : "+v"(t_));
    st1<T>(&b, t_);
    t_ = ld1<T>(&b) * s1;                                     // * params['lambda']
    asm volatile("" : "+v"(t_));
    st1<T>(&a, t_);
    t_ = ld1<T>(&a) * s2;                                     // * np.sqrt(numel)
    asm volatile("" : "+v"(t_));
    st1<T>(&b, t_);
    st1<T>(&c, ld1<T>(pivot + i) + ld1<T>(&b));               // pivot + ...
    out[i] = c;
  }
}

// ---- K13 -----------------------------------------------------------------------------------------------
template <typename T>
__global__ void ddim_candidates_kernel(const T* __restrict__ x, const T* __restrict__ e, const T* __restrict__ z, T* __restrict__ prev,
                                       T* __restrict__ x0_out, float a_t, float a_prev, float sigma_t, int ncand, long long count) {
  const float sa = sqrtf(a_t), sb = sqrtf(1.f - a_t);
  const float sp = sqrtf(a_prev), dirc = sqrtf(1.f - a_prev - sigma_t * sigma_t);
  GSL(i, count) {
    const float xv = ld1<T>(x + i), ev = ld1<T>(e + i);
    const float x0 = (xv - sb * ev) / sa;
    if (x0_out) st1<T>(x0_out + i, x0);
    const float base = sp * x0 + dirc * ev;
    for (int c = 0; c < ncand; ++c) {
      const float zv = z ? ld1<T>(z + (size_t)c * count + i) : 0.f;
      st1<T>(prev + (size_t)c * count + i, base + sigma_t * zv);
    }
  }
}

}  // namespace

// =========================================================================================================
#define ST hipStream_t st = to_stream(s)

extern "C" int dts_nchw_to_nhwc(const float* src, void* dst, int dtype, int n, int c, int h, int w, dts_stream s) {
  DTS_CHECK_ARG(src && dst && n > 0 && c > 0 && h > 0 && w > 0, "dts_nchw_to_nhwc: bad args");
  ST;
  const long long total = (long long)n * c * h * w;
  DTS_DISPATCH_DTYPE(dtype, {
    hipLaunchKernelGGL((nchw_to_nhwc_kernel<T>), dim3(grid1d(total)), dim3(256), 0, st, src, (T*)dst, n, c, h, w);
    DTS_CHECK_LAUNCH("dts_nchw_to_nhwc");
  });
  return DTS_OK;
}
// NCHW f32 [n][c][h][w] -> NHWC [n][h][w][cpad], channels c..cpad-1 zero (SD latents have 4 channels; the MFMA conv wants cin % 64 == 0)
template <typename T>
__global__ void nchw_to_nhwc_pad_kernel(const float* __restrict__ src, T* __restrict__ dst, int n, int c, int h, int w, int cpad) {
  const long long total = (long long)n * h * w * cpad;
  GSL(i, total) {
    const int ch = (int)(i % cpad);
    const long long pix = i / cpad;
    const int x = (int)(pix % w), y = (int)((pix / w) % h), b = (int)(pix / ((long long)w * h));
    st1<T>(dst + i, ch < c ? src[(((size_t)b * c + ch) * h + y) * w + x] : 0.f);
  }
}
extern "C" int dts_nchw_to_nhwc_pad(const float* src, void* dst, int dtype, int n, int c, int h, int w, int cpad, dts_stream s) {
  DTS_CHECK_ARG(src && dst && n > 0 && c > 0 && h > 0 && w > 0 && cpad >= c, "dts_nchw_to_nhwc_pad: bad args");
  ST;
  const long long total = (long long)n * cpad * h * w;
  DTS_DISPATCH_DTYPE(dtype, {
    hipLaunchKernelGGL((nchw_to_nhwc_pad_kernel<T>), dim3(grid1d(total)), dim3(256), 0, st, src, (T*)dst, n, c, h, w, cpad);
    DTS_CHECK_LAUNCH("dts_nchw_to_nhwc_pad");
  });
  return DTS_OK;
}
extern "C" int dts_nhwc_to_nchw(const void* src, int dtype, float* dst, int n, int c, int h, int w, dts_stream s) {
  DTS_CHECK_ARG(src && dst && n > 0 && c > 0 && h > 0 && w > 0, "dts_nhwc_to_nchw: bad args");
  ST;
  const long long total = (long long)n * c * h * w;
  DTS_DISPATCH_DTYPE(dtype, {
    hipLaunchKernelGGL((nhwc_to_nchw_kernel<T>), dim3(grid1d(total)), dim3(256), 0, st, (const T*)src, dst, n, c, h, w);
    DTS_CHECK_LAUNCH("dts_nhwc_to_nchw");
  });
  return DTS_OK;
}
extern "C" int dts_pack_conv_weight(const float* w, void* dst, int dtype, int O, int I, int kh, int kw, const int32_t* perm,
                                    dts_stream s) {
  DTS_CHECK_ARG(w && dst && O > 0 && I > 0 && kh > 0 && kw > 0, "dts_pack_conv_weight: bad args");
  ST;
  const long long total = (long long)O * I * kh * kw;
  DTS_DISPATCH_DTYPE(dtype, {
    hipLaunchKernelGGL((pack_w_kernel<T>), dim3(grid1d(total)), dim3(256), 0, st, w, (T*)dst, O, I, kh, kw, perm);
    DTS_CHECK_LAUNCH("dts_pack_conv_weight");
  });
  return DTS_OK;
}
// split-precision operand image (dts.h DTS_F16X3): row p of concat(x1, x2) -> per 32 channels hi(32) | lo * 2^11 (32); 8 channels (two float4
// in, two 16-byte stores out) per thread.  HBM-bound: 4 bytes read + 4 written per element.
__global__ __launch_bounds__(256) void split3_f16_kernel(const float* __restrict__ x1, int c1, const float* __restrict__ x2, int c2,
                                                          uint4* __restrict__ out, long long rows) {
  const int C = c1 + c2, nch = C / 8;
  const long long total = rows * nch;
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
    const long long row = idx / nch;
    const int ch = (int)(idx - row * nch), c0 = ch * 8;
    const float* src = c0 < c1 ? x1 + row * c1 + c0 : x2 + row * c2 + (c0 - c1);
    const float4 a = reinterpret_cast<const float4*>(src)[0], b = reinterpret_cast<const float4*>(src)[1];
    const float f[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
    float hi[8], lo[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) x3_split(f[e], hi[e], lo[e]);
    uint4* o = out + row * (2 * nch) + ((ch >> 2) << 3) + (ch & 3);          // 16-byte slots: group (ch / 4) holds 4 hi slots, then 4 lo slots
    o[0] = pack16<f16_t>(hi); o[4] = pack16<f16_t>(lo);
  }
}

// split-precision attention input (dts_attention_x3): row p of x (C channels) -> hi(C) | lo(C) of x * 2^6, hi = f16(y), lo = f16(y - hi).
// The factor keeps the lo part a normal f16 number for |x| >= 2^-8 (the matrix cores flush subnormal inputs; below that the lo part,
// <= 2^-20 |x|, is lost); |x| >= 1023.5 saturates (x2_split); the attention kernel takes the powers of two out again exactly.
__global__ __launch_bounds__(256) void split2_f16_kernel(const float* __restrict__ x, int C, uint4* __restrict__ out, long long rows) {
  const int nch = C / 8;
  const long long total = rows * nch;
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
    const long long row = idx / nch;
    const int ch = (int)(idx - row * nch);
    const float* src = x + row * C + ch * 8;
    const float4 a = reinterpret_cast<const float4*>(src)[0], b = reinterpret_cast<const float4*>(src)[1];
    const float f[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
    float hi[8], lo[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) x2_split(f[e], hi[e], lo[e]);
    uint4* o = out + row * (2 * nch) + ch;
    o[0] = pack16<f16_t>(hi); o[nch] = pack16<f16_t>(lo);
  }
}

extern "C" int dts_split2_f16(const float* x, int c, void* out, int64_t rows, dts_stream s) {
  DTS_CHECK_ARG(x && out && rows >= 0 && c > 0 && c % 8 == 0, "dts_split2_f16: bad args (c=%d)", c);
  if (rows == 0) return DTS_OK;
  ST;
  hipLaunchKernelGGL(split2_f16_kernel, dim3(grid1d(rows * (c / 8))), dim3(256), 0, st, x, c, (uint4*)out, (long long)rows);
  DTS_CHECK_LAUNCH("dts_split2_f16");
  return DTS_OK;
}

extern "C" int dts_split3_f16(const float* x1, int c1, const float* x2, int c2, void* out, int64_t rows, dts_stream s) {
  DTS_CHECK_ARG(x1 && out && rows >= 0 && c1 > 0 && c1 % 8 == 0 && c2 >= 0 && c2 % 8 == 0 && (c1 + c2) % 32 == 0,
                "dts_split3_f16: bad args (c1=%d c2=%d: multiples of 8, sum a multiple of 32)", c1, c2);
  DTS_CHECK_ARG(c2 == 0 || x2, "dts_split3_f16: c2 without x2");
  if (rows == 0) return DTS_OK;
  ST;
  hipLaunchKernelGGL(split3_f16_kernel, dim3(grid1d(rows * ((c1 + c2) / 8))), dim3(256), 0, st, x1, c1, x2, c2, (uint4*)out, (long long)rows);
  DTS_CHECK_LAUNCH("dts_split3_f16");
  return DTS_OK;
}

extern "C" int dts_cast_from_f32(const float* src, void* dst, int dtype, int64_t count, dts_stream s) {
  DTS_CHECK_ARG(src && dst && count >= 0, "dts_cast_from_f32: bad args");
  if (count == 0) return DTS_OK;
  ST;
  DTS_DISPATCH_DTYPE(dtype, {
    hipLaunchKernelGGL((cast_from_f32_kernel<T>), dim3(grid1d(count)), dim3(256), 0, st, src, (T*)dst, (long long)count);
    DTS_CHECK_LAUNCH("dts_cast_from_f32");
  });
  return DTS_OK;
}
extern "C" int dts_cast_to_f32(const void* src, int dtype, float* dst, int64_t count, dts_stream s) {
  DTS_CHECK_ARG(src && dst && count >= 0, "dts_cast_to_f32: bad args");
  if (count == 0) return DTS_OK;
  ST;
  DTS_DISPATCH_DTYPE(dtype, {
    hipLaunchKernelGGL((cast_to_f32_kernel<T>), dim3(grid1d(count)), dim3(256), 0, st, (const T*)src, dst, (long long)count);
    DTS_CHECK_LAUNCH("dts_cast_to_f32");
  });
  return DTS_OK;
}

extern "C" int dts_edm_precond_in(const double* x, const double* sigma, int nsigma, float sigma_data, float* xin, float* coef, int n,
                                  int chw, dts_stream s) {
  DTS_CHECK_ARG(x && sigma && xin && coef && n > 0 && chw > 0, "dts_edm_precond_in: bad args");
  DTS_CHECK_ARG(nsigma == 1 || nsigma == n, "dts_edm_precond_in: nsigma=%d n=%d", nsigma, n);
  ST;
  hipLaunchKernelGGL(precond_in_kernel, dim3(grid1d((long long)n * chw)), dim3(256), 0, st, x, sigma, nsigma, sigma_data, xin, coef, n,
                     chw);
  DTS_CHECK_LAUNCH("dts_edm_precond_in");
  return DTS_OK;
}
extern "C" int dts_edm_precond_out(const double* x, const float* F, const float* coef, float* D, int n, int chw, dts_stream s) {
  DTS_CHECK_ARG(x && F && coef && D && n > 0 && chw > 0, "dts_edm_precond_out: bad args");
  ST;
  hipLaunchKernelGGL(precond_out_kernel, dim3(grid1d((long long)n * chw)), dim3(256), 0, st, x, F, coef, D, n, chw);
  DTS_CHECK_LAUNCH("dts_edm_precond_out");
  return DTS_OK;
}

extern "C" int dts_heun_xhat(const double* x_cur, int xb, int bcast, const void* eps, int eps_f32, double noise_coef, double* x_hat,
                             int nb, int chw, dts_stream s) {
  DTS_CHECK_ARG(x_cur && eps && x_hat && nb > 0 && chw > 0, "dts_heun_xhat: bad args");
  DTS_CHECK_ARG(xb > 0 && nb % xb == 0, "dts_heun_xhat: nb=%d not a multiple of xb=%d", nb, xb);
  ST;
  const int g = grid1d((long long)nb * chw);
  if (eps_f32)
    hipLaunchKernelGGL((heun_xhat_kernel<float>), dim3(g), dim3(256), 0, st, x_cur, xb, bcast, (const float*)eps, noise_coef, x_hat, nb,
                       chw);
  else
    hipLaunchKernelGGL((heun_xhat_kernel<double>), dim3(g), dim3(256), 0, st, x_cur, xb, bcast, (const double*)eps, noise_coef, x_hat,
                       nb, chw);
  DTS_CHECK_LAUNCH("dts_heun_xhat");
  return DTS_OK;
}
extern "C" int dts_heun_euler(const double* x_hat, const float* D, double t_hat, double t_next, double* d_cur, double* x_next,
                              int64_t count, dts_stream s) {
  DTS_CHECK_ARG(x_hat && D && d_cur && x_next && count > 0, "dts_heun_euler: bad args");
  DTS_CHECK_ARG(t_hat != 0.0, "dts_heun_euler: t_hat == 0");
  ST;
  hipLaunchKernelGGL(heun_euler_kernel, dim3(grid1d(count)), dim3(256), 0, st, x_hat, D, t_hat, t_next, d_cur, x_next, (long long)count);
  DTS_CHECK_LAUNCH("dts_heun_euler");
  return DTS_OK;
}
extern "C" int dts_heun_correct(const double* x_hat, const float* D2, const double* d_cur, double t_hat, double t_next, double* x_next,
                                int64_t count, dts_stream s) {
  DTS_CHECK_ARG(x_hat && D2 && d_cur && x_next && count > 0, "dts_heun_correct: bad args");
  DTS_CHECK_ARG(t_next != 0.0, "dts_heun_correct: t_next == 0 (last step is Euler only)");
  ST;
  hipLaunchKernelGGL(heun_correct_kernel, dim3(grid1d(count)), dim3(256), 0, st, x_hat, D2, d_cur, t_hat, t_next, x_next,
                     (long long)count);
  DTS_CHECK_LAUNCH("dts_heun_correct");
  return DTS_OK;
}

extern "C" int dts_quantize_u8(const void* x, int is_f32, uint8_t* out, int64_t count, dts_stream s) {
  DTS_CHECK_ARG(x && out && count > 0, "dts_quantize_u8: bad args");
  ST;
  if (is_f32 == 2)
    hipLaunchKernelGGL(quantize_f32math_kernel, dim3(grid1d(count)), dim3(256), 0, st, (const float*)x, out, (long long)count);
  else if (is_f32)
    hipLaunchKernelGGL((quantize_kernel<float>), dim3(grid1d(count)), dim3(256), 0, st, (const float*)x, out, (long long)count);
  else
    hipLaunchKernelGGL((quantize_kernel<double>), dim3(grid1d(count)), dim3(256), 0, st, (const double*)x, out, (long long)count);
  DTS_CHECK_LAUNCH("dts_quantize_u8");
  return DTS_OK;
}
extern "C" int dts_brightness(const uint8_t* img, float* rewards, int n, int hw, dts_stream s) {
  DTS_CHECK_ARG(img && rewards && n > 0 && hw > 0, "dts_brightness: bad args");
  ST;
  hipLaunchKernelGGL(brightness_kernel, dim3(n), dim3(256), 0, st, img, rewards, hw);
  DTS_CHECK_LAUNCH("dts_brightness");
  return DTS_OK;
}
extern "C" int dts_cosine_rows(const float* a, const float* b, int b_rows, float* out, int n, int d, dts_stream s) {
  DTS_CHECK_ARG(a && b && out && n > 0 && d > 0, "dts_cosine_rows: bad args");
  DTS_CHECK_ARG(b_rows == 1 || b_rows == n, "dts_cosine_rows: b has %d rows for %d", b_rows, n);
  ST;
  hipLaunchKernelGGL(cosine_rows_kernel, dim3(n), dim3(64), 0, st, a, b, b_rows, out, d);
  DTS_CHECK_LAUNCH("dts_cosine_rows");
  return DTS_OK;
}
extern "C" int dts_resample_u8(const uint8_t* src, uint8_t* dst, int planes, int h, int w, int out_len, int axis, const int32_t* bounds,
                               const int32_t* coefs, int ksize, dts_stream s) {
  DTS_CHECK_ARG(src && dst && bounds && coefs, "dts_resample_u8: null pointer");
  DTS_CHECK_ARG(planes > 0 && h > 0 && w > 0 && out_len > 0 && ksize > 0 && ksize <= 4096 && (axis == 0 || axis == 1), "dts_resample_u8: bad shape");
  ST;
  const long long total = axis ? (long long)planes * h * out_len : (long long)planes * out_len * w;
  hipLaunchKernelGGL(resample_u8_kernel, dim3(grid1d(total)), dim3(256), 0, st, src, dst, planes, h, w, out_len, axis, bounds, coefs, ksize);
  DTS_CHECK_LAUNCH("dts_resample_u8");
  return DTS_OK;
}
extern "C" int dts_lut_u8_f32(const uint8_t* img, const float* lut, float* out, int n, int c, int hw, dts_stream s) {
  DTS_CHECK_ARG(img && lut && out && n > 0 && c > 0 && hw > 0, "dts_lut_u8_f32: bad args");
  ST;
  const long long total = (long long)n * c * hw;
  hipLaunchKernelGGL(lut_u8_f32_kernel, dim3(grid1d(total)), dim3(256), 0, st, img, lut, out, c, hw, total);
  DTS_CHECK_LAUNCH("dts_lut_u8_f32");
  return DTS_OK;
}
extern "C" int dts_u8_to_unit_f32(const uint8_t* img, float* out, int64_t count, dts_stream s) {
  DTS_CHECK_ARG(img && out && count > 0, "dts_u8_to_unit_f32: bad args");
  ST;
  hipLaunchKernelGGL(u8_to_unit_kernel, dim3(grid1d(count)), dim3(256), 0, st, img, out, (long long)count);
  DTS_CHECK_LAUNCH("dts_u8_to_unit_f32");
  return DTS_OK;
}

extern "C" int dts_linear(const float* x, int ldx, const float* w, const float* bias, float* y, int ldy, int m, int k, int n, int act_in,
                          int act_out, int accumulate, dts_stream s) {
  DTS_CHECK_ARG(x && w && y && m > 0 && k > 0 && n > 0 && ldx >= k && ldy >= n, "dts_linear: bad args");
  ST;
  if (k % 4 == 0 && ldx % 4 == 0 && ((uintptr_t)x & 15) == 0 && ((uintptr_t)w & 15) == 0)
    hipLaunchKernelGGL(linear4_kernel, dim3((n + 3) / 4, (m + 7) / 8), dim3(256), 0, st, x, ldx, w, bias, y, ldy, m, k, n, act_in, act_out, accumulate);
  else
    hipLaunchKernelGGL(linear_kernel, dim3((n + 3) / 4, (m + 7) / 8), dim3(256), 0, st, x, ldx, w, bias, y, ldy, m, k, n, act_in, act_out, accumulate);
  DTS_CHECK_LAUNCH("dts_linear");
  return DTS_OK;
}
extern "C" int dts_pos_embedding(const float* v, const float* freqs, float* out, int n, int half, int swap, dts_stream s) {
  DTS_CHECK_ARG(v && freqs && out && n > 0 && half > 0, "dts_pos_embedding: bad args");
  ST;
  hipLaunchKernelGGL(pos_embedding_kernel, dim3(grid1d((long long)n * half)), dim3(256), 0, st, v, freqs, out, n, half, swap);
  DTS_CHECK_LAUNCH("dts_pos_embedding");
  return DTS_OK;
}

extern "C" int dts_attnpool_tokens(const void* x, const float* pos, void* tokens, int dtype, int n, int hw, int c, dts_stream s) {
  DTS_CHECK_ARG(x && pos && tokens && n > 0 && hw > 0 && c > 0, "dts_attnpool_tokens: bad args");
  ST;
  DTS_DISPATCH_DTYPE(dtype, {
    hipLaunchKernelGGL((attnpool_tokens_kernel<T>), dim3(n), dim3(256), 0, st, (const T*)x, pos, (T*)tokens, hw, c);
    DTS_CHECK_LAUNCH("dts_attnpool_tokens");
  });
  return DTS_OK;
}
extern "C" int dts_take_token(const void* src, int dtype, float* out, int n, int t, int c, int token, dts_stream s) {
  DTS_CHECK_ARG(src && out && n > 0 && c > 0 && token >= 0 && token < t, "dts_take_token: bad args");
  ST;
  DTS_DISPATCH_DTYPE(dtype, {
    hipLaunchKernelGGL((take_token_kernel<T>), dim3(grid1d((long long)n * c)), dim3(256), 0, st, (const T*)src, out, n, t, c, token);
    DTS_CHECK_LAUNCH("dts_take_token");
  });
  return DTS_OK;
}
extern "C" int dts_softmax_gather(const float* logits, const int32_t* target, float* rewards, int n, int k, dts_stream s) {
  DTS_CHECK_ARG(logits && target && rewards && n > 0 && k > 0, "dts_softmax_gather: bad args");
  ST;
  hipLaunchKernelGGL(softmax_gather_kernel, dim3(n), dim3(256), 0, st, logits, target, rewards, k);
  DTS_CHECK_LAUNCH("dts_softmax_gather");
  return DTS_OK;
}

extern "C" int dts_candidate_noise(const double* pivot, const double* g, const int32_t* mode, const float* scale, double* cand, int nb,
                                   int b, int chw, dts_stream s) {
  DTS_CHECK_ARG(pivot && g && mode && scale && cand, "dts_candidate_noise: null pointer");
  DTS_CHECK_ARG(nb > 0 && b > 0 && nb % b == 0 && chw > 0, "dts_candidate_noise: nb=%d b=%d", nb, b);
  ST;
  hipLaunchKernelGGL(candidate_noise_kernel, dim3(nb), dim3(256), 0, st, pivot, g, mode, scale, cand, b, chw);
  DTS_CHECK_LAUNCH("dts_candidate_noise");
  return DTS_OK;
}

extern "C" int dts_candidate_noise_sd(const void* pivot, const void* u, const int32_t* mode, const float* scale, void* cand, int dtype, int n,
                                      int64_t count, dts_stream s) {
  DTS_CHECK_ARG(pivot && u && mode && scale && cand, "dts_candidate_noise_sd: null pointer");
  DTS_CHECK_ARG(n > 0 && count > 0 && count < (1ll << 31), "dts_candidate_noise_sd: n=%d count=%lld", n, (long long)count);
  ST;
  DTS_DISPATCH_DTYPE(dtype, {
    hipLaunchKernelGGL((candidate_noise_sd_kernel<T>), dim3(n), dim3(256), 0, st, (const T*)pivot, (const T*)u, mode, scale, (T*)cand, (int)count);
    DTS_CHECK_LAUNCH("dts_candidate_noise_sd");
  });
  return DTS_OK;
}

extern "C" int dts_cfg_combine(const void* uncond, const void* cond, float guidance, void* out, int dtype, int64_t count, dts_stream s) {
  DTS_CHECK_ARG(uncond && cond && out && count > 0, "dts_cfg_combine: bad args");
  ST;
  DTS_DISPATCH_DTYPE(dtype, {
    hipLaunchKernelGGL((cfg_combine_kernel<T>), dim3(grid1d(count)), dim3(256), 0, st, (const T*)uncond, (const T*)cond, guidance, (T*)out,
                       (long long)count);
    DTS_CHECK_LAUNCH("dts_cfg_combine");
  });
  return DTS_OK;
}

extern "C" int dts_ddim_candidates(const void* x, const void* e, const void* z, void* prev, void* x0_out, int dtype, float alpha_t,
                                   float alpha_prev, float sigma_t, int ncand, int64_t count, dts_stream s) {
  DTS_CHECK_ARG(x && e && prev && ncand > 0 && count > 0, "dts_ddim_candidates: bad args");
  DTS_CHECK_ARG(alpha_t > 0.f && alpha_t <= 1.f && alpha_prev > 0.f && alpha_prev <= 1.f, "dts_ddim_candidates: alphas out of range");
  DTS_CHECK_ARG(1.f - alpha_prev - sigma_t * sigma_t >= 0.f, "dts_ddim_candidates: sigma_t too large");
  ST;
  DTS_DISPATCH_DTYPE(dtype, {
    hipLaunchKernelGGL((ddim_candidates_kernel<T>), dim3(grid1d(count)), dim3(256), 0, st, (const T*)x, (const T*)e, (const T*)z, (T*)prev,
                       (T*)x0_out, alpha_t, alpha_prev, sigma_t, ncand, (long long)count);
    DTS_CHECK_LAUNCH("dts_ddim_candidates");
  });
  return DTS_OK;
}
