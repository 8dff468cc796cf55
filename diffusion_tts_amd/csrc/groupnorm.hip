// K4/K5: GroupNorm statistics -> per-(sample,channel) affine coefficients -> fused apply (+SiLU, +2x2 pool).
//
// Replaces torch.nn.functional.group_norm at edm/training/networks.py:104-106 together with the SiLU and
// the adaptive scale/shift around it (:168, :170-175, :182, :460) and edm/unet.py:254-272 (GroupNorm32,
// scale-shift norm, avg-pool down).  HBM-bound: one read of x for the statistics, one read + one write
// for the apply.  Deterministic (no float atomics): per-(sample, pixel-split, group) partial sums, then a
// fixed-order f64 combine, so identical candidate rows produce bit-identical outputs (ties stay ties).
#include "dts_common.h"

namespace {

constexpr int GN_SPLITS = 32;

// partial[n][split][g][2] = (sum, sumsq) over this split's pixels and the group's channels
template <typename T>
__global__ __launch_bounds__(256) void gn_partial_kernel(const T* __restrict__ x1, int c1, const T* __restrict__ x2, int c2,
                                                          int hw, int groups, int splits, float* __restrict__ partial) {
  constexpr int EPV = ET<T>::EPV;
  extern __shared__ float sm[];                      // [planes][C][2]
  const int C = c1 + c2, cg = C / groups;
  const int nchunk = C / EPV;
  const int slots = nchunk < 256 ? nchunk : 256;     // chunk slots per pixel plane
  const int planes = 256 / slots;
  const int n = blockIdx.y, split = blockIdx.x;
  const int per = (hw + splits - 1) / splits;
  const int p0 = split * per, p1 = min(hw, p0 + per);
  const int tid = threadIdx.x;
  const int plane = tid / slots, slot = tid - plane * slots;
  if (plane < planes) {
    for (int chunk = slot; chunk < nchunk; chunk += slots) {
      float s[EPV], ss[EPV];
#pragma unroll
      for (int e = 0; e < EPV; ++e) s[e] = ss[e] = 0.f;
      const int c0 = chunk * EPV;
      const T* base; int cs, co;
      if (c0 < c1) { base = x1; cs = c1; co = c0; } else { base = x2; cs = c2; co = c0 - c1; }
      for (int p = p0 + plane; p < p1; p += planes) {
        const uint4 v = *reinterpret_cast<const uint4*>(base + ((size_t)n * hw + p) * cs + co);
        float f[EPV];
        unpack16<T>(v, f);
#pragma unroll
        for (int e = 0; e < EPV; ++e) { s[e] += f[e]; ss[e] += f[e] * f[e]; }
      }
#pragma unroll
      for (int e = 0; e < EPV; ++e) {
        sm[(plane * C + c0 + e) * 2 + 0] = s[e];
        sm[(plane * C + c0 + e) * 2 + 1] = ss[e];
      }
    }
  }
  __syncthreads();
  // one thread per group: fixed-order sum over planes and the group's channels
  for (int g = tid; g < groups; g += 256) {
    double a = 0.0, b = 0.0;
    for (int pl = 0; pl < planes; ++pl)
      for (int c = g * cg; c < (g + 1) * cg; ++c) { a += sm[(pl * C + c) * 2]; b += sm[(pl * C + c) * 2 + 1]; }
    float* o = partial + (((size_t)n * splits + split) * groups + g) * 2;
    o[0] = (float)a;     // partial sums over <= a few thousand f32 values; kept in f32 (rel. err ~1e-7)
    o[1] = (float)b;
  }
}

// coef[n][c] = (a, b):  y = x*a + b  ==  ((x-mean)*rstd*gamma + beta) * (1+scale) + shift
template <typename T>
__global__ void gn_coef_kernel(const float* __restrict__ partial, int splits, int groups, int C, int hw, float eps,
                               const float* __restrict__ gamma, const float* __restrict__ beta,
                               const T* __restrict__ ss, int ld_ss, float* __restrict__ coef, int n_total) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= n_total * C) return;
  const int n = idx / C, c = idx - n * C;
  const int cg = C / groups, g = c / cg;
  double a = 0.0, b = 0.0;
  for (int s = 0; s < splits; ++s) {
    const float* q = partial + (((size_t)n * splits + s) * groups + g) * 2;
    a += q[0]; b += q[1];
  }
  const double cnt = (double)hw * cg;
  const double mean = a / cnt;
  double var = b / cnt - mean * mean;
  if (var < 0.0) var = 0.0;
  const float rstd = (float)(1.0 / sqrt(var + (double)eps));
  const float fm = (float)mean;
  float ga = gamma ? gamma[c] : 1.f, be = beta ? beta[c] : 0.f;
  float A = rstd * ga;
  float B = be - fm * A;
  if (ss) {
    const float sc = 1.f + ld1<T>(ss + (size_t)n * ld_ss + c);
    const float sh = ld1<T>(ss + (size_t)n * ld_ss + C + c);
    A = A * sc;
    B = B * sc + sh;
  }
  coef[(size_t)idx * 2 + 0] = A;
  coef[(size_t)idx * 2 + 1] = B;
}

// coefficients from the per-64-pixel-strip statistics the producing convolutions emitted (conv_igemm.hip epilogue).
// One wave per (group, sample): lane-strided sum over the group's strips x channels, fixed-order f64 wave reduction,
// then the group's channels get their (a, b).
// With many strips per sample (SD VAE: 4096 at 512x512) the block is four waves that split the element range and are combined in
// wave order through LDS (still a fixed order); the U-Net levels (<= 64 strips) keep one wave, whose arithmetic is unchanged.
template <typename T>
__global__ __launch_bounds__(256) void gn_coef_strips_kernel(const float* __restrict__ st1, int c1, const float* __restrict__ st2,
                                                             int c2, int strips, int groups, int hw, float eps,
                                                             const float* __restrict__ gamma, const float* __restrict__ beta,
                                                             const T* __restrict__ ss, int ld_ss, float* __restrict__ coef) {
  const int C = c1 + c2, cg = C / groups, g = blockIdx.x, n = blockIdx.y, lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nw = blockDim.x >> 6;
  double a = 0.0, b = 0.0;
  const int total = strips * cg;
  const float inv_cg = 1.0f / (float)cg;
  // the affine / adaptive parameters do not depend on the sums: fetch them first so their latency hides under the strip loads
  float pg[2], pb[2], psc[2], psh[2];
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int ch = lane + 64 * u, c = g * cg + ch;
    const bool ok = ch < cg;
    pg[u] = (ok && gamma) ? gamma[c] : 1.f;
    pb[u] = (ok && beta) ? beta[c] : 0.f;
    psc[u] = (ok && ss) ? ld1<T>(ss + (size_t)n * ld_ss + c) : 0.f;
    psh[u] = (ok && ss) ? ld1<T>(ss + (size_t)n * ld_ss + C + c) : 0.f;
  }
  // eight independent loads per trip: the kernel is one dependent-latency chain per trip, so the trip count is its run time
  for (int e0 = wv * 64 * 8 + lane; e0 < total; e0 += nw * 64 * 8) {
    float2 q[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int e = e0 + 64 * u;
      q[u] = make_float2(0.f, 0.f);
      if (e < total) {
        const int s = (int)(((float)e + 0.5f) * inv_cg), c = g * cg + (e - s * cg);       // exact for e < 2^22
        const float* src; int cs, co;
        if (c < c1) { src = st1; cs = c1; co = c; } else { src = st2; cs = c2; co = c - c1; }
        q[u] = *reinterpret_cast<const float2*>(src + (((size_t)n * strips + s) * cs + co) * 2);
      }
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) { a += q[u].x; b += q[u].y; }
  }
  a = wave_sum(a);
  b = wave_sum(b);
  if (nw > 1) {                                        // block-uniform
    __shared__ double part[4][2];
    if (lane == 0) { part[wv][0] = a; part[wv][1] = b; }
    __syncthreads();
    if (wv != 0) return;
    a = part[0][0]; b = part[0][1];
    for (int k = 1; k < nw; ++k) { a += part[k][0]; b += part[k][1]; }
  }
  const double cnt = (double)hw * cg;
  const double mean = a / cnt;
  double var = b / cnt - mean * mean;
  if (var < 0.0) var = 0.0;
  const float rstd = (float)(1.0 / sqrt(var + (double)eps));
  const float fm = (float)mean;
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int ch = lane + 64 * u, c = g * cg + ch;
    if (ch >= cg) break;
    float A = rstd * pg[u];
    float B = pb[u] - fm * A;
    if (ss) {
      const float sc = 1.f + psc[u];
      A = A * sc;
      B = B * sc + psh[u];
    }
    *reinterpret_cast<float2*>(coef + ((size_t)n * C + c) * 2) = make_float2(A, B);
  }
}

// SiLU: f32 (parity mode) keeps the exact division of the reference's x * sigmoid(x); for bf16/f16 outputs the quotient goes
// through v_rcp_f32 (1 ulp, far below the storage rounding) -- the IEEE division sequence made the apply pass VALU-bound.
template <typename T> __device__ __forceinline__ float silu_t(float x) { return x * __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
template <> __device__ __forceinline__ float silu_t<float>(float x) { return silu_f(x); }

// Split-precision output (dts.h DTS_F16X3, the arithmetic of split3_f16_kernel): 4 consecutive channels c0.. of one pixel row go out as
// f16 hi / lo * 2^11 into the 2C-wide row `orow` of the operand image (per 32 channels: hi(32) | lo(32)) -- what the consuming
// convolution reads, so the f32 tensor in between (one write + one read) and the separate split pass disappear.
__device__ __forceinline__ void store_split4(f16_t* orow, int c0, const float* f) {
  float hi[4], lo[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) x3_split(f[e], hi[e], lo[e]);
  f16_t* d = orow + x3_off(c0);
  *reinterpret_cast<uint2*>(d) = make_uint2(pack2_f16(hi[0], hi[1]), pack2_f16(hi[2], hi[3]));
  *reinterpret_cast<uint2*>(d + 32) = make_uint2(pack2_f16(lo[0], lo[1]), pack2_f16(lo[2], lo[3]));
}

// The same output through 16-byte stores (round 6): lanes 2m / 2m+1 of a wave hold channels 8m'..8m'+3 / 8m'+4..8m'+7 of ONE pixel (a block's
// threads are [pixel row][chunk] with an even chunk count, so lane parity = chunk parity).  The pair swaps halves with one DPP quad
// permute per dword: the even lane stores the hi halves of all 8 channels (16 bytes), the odd lane their lo halves -- one store instruction
// per pixel row and thread that covers WHOLE 128-byte lines (hi(32) | lo(32) of a 32-channel group = 8 lanes x 16 bytes) instead of two
// 8-byte-per-lane instructions that each write half of every line.  Same bits.
__device__ __forceinline__ uint32_t dpp_swap_pair(uint32_t v) {
  return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xf, 0xf, false);         // quad_perm [1,0,3,2]
}
__device__ __forceinline__ void store_split4_paired(f16_t* orow, int c0, const float* f, bool odd) {
  float hi[4], lo[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) x3_split(f[e], hi[e], lo[e]);
  const uint2 h = make_uint2(pack2_f16(hi[0], hi[1]), pack2_f16(hi[2], hi[3])), l = make_uint2(pack2_f16(lo[0], lo[1]), pack2_f16(lo[2], lo[3]));
  // even lane sends its lo half and receives the partner's hi half; odd lane the other way round
  const uint32_t sx = odd ? h.x : l.x, sy = odd ? h.y : l.y;
  const uint32_t rx = dpp_swap_pair(sx), ry = dpp_swap_pair(sy);
  const uint4 v = odd ? make_uint4(rx, ry, l.x, l.y) : make_uint4(h.x, h.y, rx, ry);
  f16_t* d = orow + x3_off(odd ? c0 - 4 : c0) + (odd ? 32 : 0);
  *reinterpret_cast<uint4*>(d) = v;
}

// Apply pass, row form: a block works on a pixel range of ONE sample and every thread keeps ONE 16-byte channel chunk, so
// its (a,b) coefficients are loaded once and the loop body is load -> fma/SiLU -> store with no index arithmetic (the
// grid-stride form below spends more VALU cycles on 64-bit div/mod per element than on the SiLU).  blockDim = k * nchunk.
template <typename T, bool SPLIT = false, bool W16 = false>
__global__ __launch_bounds__(256) void gn_apply_rows_kernel(const T* __restrict__ x1, int c1, const T* __restrict__ x2, int c2,
                                                             const float* __restrict__ coef, T* __restrict__ out, int hw, int ppb,
                                                             int silu, f16_t* __restrict__ raw = nullptr) {
  // raw (SPLIT only, optional): the split image of the UN-normalised input rows is written too -- the operand of the block's 1x1 skip
  // convolution (networks.py:177 `self.skip(orig)`), which would otherwise cost a dts_split3_f16 pass over the same tensor
  static_assert(!SPLIT || sizeof(T) == 4, "split-precision output is the f32 mode's");
  constexpr int EPV = ET<T>::EPV;
  const int C = c1 + c2, nchunk = C / EPV;
  const int k = blockDim.x / nchunk;
  const int pr = threadIdx.x / nchunk, chunk = threadIdx.x - pr * nchunk;
  const int n = blockIdx.y, c0 = chunk * EPV;
  const bool odd = (chunk & 1) != 0;
#define DTS_STORE_SPLIT(orow_, f_) { if constexpr (W16) store_split4_paired(orow_, c0, f_, odd); else store_split4(orow_, c0, f_); }
  float A[EPV], B[EPV];
#pragma unroll
  for (int e = 0; e < EPV; e += 2) {
    const float4 q = *reinterpret_cast<const float4*>(coef + ((size_t)n * C + c0 + e) * 2);
    A[e] = q.x; B[e] = q.y; A[e + 1] = q.z; B[e + 1] = q.w;
  }
  const T* src; int cs;
  if (c0 < c1) { src = x1 + (size_t)n * hw * c1 + c0; cs = c1; } else { src = x2 + (size_t)n * hw * c2 + (c0 - c1); cs = c2; }
  T* dst = out + (size_t)n * hw * C + c0;
  f16_t* const dst3 = reinterpret_cast<f16_t*>(out) + (size_t)n * hw * 2 * C;       // SPLIT: rows of 2C f16
  f16_t* const raw3 = raw ? raw + (size_t)n * hw * 2 * C : nullptr;
  const int p_begin = blockIdx.x * ppb, p_end = min(hw, p_begin + ppb);
  int p = p_begin + pr;
  // two independent pixels per trip keep two loads in flight per thread
  for (; p + k < p_end; p += 2 * k) {
    const uint4 v0 = *reinterpret_cast<const uint4*>(src + (size_t)p * cs);
    const uint4 v1 = *reinterpret_cast<const uint4*>(src + (size_t)(p + k) * cs);
    float f0[EPV], f1[EPV];
    unpack16<T>(v0, f0);
    unpack16<T>(v1, f1);
    if constexpr (SPLIT) {
      if (raw3) { DTS_STORE_SPLIT(raw3 + (size_t)p * 2 * C, f0); DTS_STORE_SPLIT(raw3 + (size_t)(p + k) * 2 * C, f1); }
    }
#pragma unroll
    for (int e = 0; e < EPV; ++e) {
      const float y0 = f0[e] * A[e] + B[e], y1 = f1[e] * A[e] + B[e];
      f0[e] = silu ? silu_t<T>(y0) : y0;
      f1[e] = silu ? silu_t<T>(y1) : y1;
    }
    if constexpr (SPLIT) {
      DTS_STORE_SPLIT(dst3 + (size_t)p * 2 * C, f0);
      DTS_STORE_SPLIT(dst3 + (size_t)(p + k) * 2 * C, f1);
    } else {
      *reinterpret_cast<uint4*>(dst + (size_t)p * C) = pack16<T>(f0);
      *reinterpret_cast<uint4*>(dst + (size_t)(p + k) * C) = pack16<T>(f1);
    }
  }
  if (p < p_end) {
    const uint4 v0 = *reinterpret_cast<const uint4*>(src + (size_t)p * cs);
    float f0[EPV];
    unpack16<T>(v0, f0);
    if constexpr (SPLIT) {
      if (raw3) DTS_STORE_SPLIT(raw3 + (size_t)p * 2 * C, f0);
    }
#pragma unroll
    for (int e = 0; e < EPV; ++e) { const float y0 = f0[e] * A[e] + B[e]; f0[e] = silu ? silu_t<T>(y0) : y0; }
    if constexpr (SPLIT) DTS_STORE_SPLIT(dst3 + (size_t)p * 2 * C, f0)
    else *reinterpret_cast<uint4*>(dst + (size_t)p * C) = pack16<T>(f0);
  }
}
#undef DTS_STORE_SPLIT

template <typename T, bool POOL, bool SPLIT = false>
__global__ __launch_bounds__(256) void gn_apply_kernel(const T* __restrict__ x1, int c1, const T* __restrict__ x2, int c2,
                                                        const float* __restrict__ coef, T* __restrict__ out,
                                                        int n_total, int h, int w, int silu, f16_t* __restrict__ raw = nullptr) {
  // raw (SPLIT only, optional): the split image of the un-normalised [2x2-averaged, with resample_kernel's arithmetic] input rows too
  constexpr int EPV = ET<T>::EPV;
  const int C = c1 + c2, nchunk = C / EPV;
  const int ho = POOL ? h / 2 : h, wo = POOL ? w / 2 : w;
  const long long total = (long long)n_total * ho * wo * nchunk;
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
    const int chunk = (int)(idx % nchunk);
    long long pix = idx / nchunk;
    const int xo = (int)(pix % wo); pix /= wo;
    const int yo = (int)(pix % ho);
    const int n = (int)(pix / ho);
    const int c0 = chunk * EPV;
    const T* base; int cs, co;
    if (c0 < c1) { base = x1; cs = c1; co = c0; } else { base = x2; cs = c2; co = c0 - c1; }
    float A[EPV], B[EPV];
#pragma unroll
    for (int e = 0; e < EPV; e += 2) {
      const float4 q = *reinterpret_cast<const float4*>(coef + ((size_t)n * C + c0 + e) * 2);
      A[e] = q.x; B[e] = q.y; A[e + 1] = q.z; B[e + 1] = q.w;
    }
    float r[EPV], rr[EPV];
    if (!POOL) {
      const uint4 v = *reinterpret_cast<const uint4*>(base + (((size_t)n * h + yo) * w + xo) * cs + co);
      float f[EPV];
      unpack16<T>(v, f);
#pragma unroll
      for (int e = 0; e < EPV; ++e) { rr[e] = f[e]; const float y = f[e] * A[e] + B[e]; r[e] = silu ? silu_t<T>(y) : y; }
    } else {
#pragma unroll
      for (int e = 0; e < EPV; ++e) { r[e] = 0.f; rr[e] = 0.f; }
#pragma unroll
      for (int dy = 0; dy < 2; ++dy)
#pragma unroll
        for (int dx = 0; dx < 2; ++dx) {
          const uint4 v = *reinterpret_cast<const uint4*>(base + (((size_t)n * h + 2 * yo + dy) * w + 2 * xo + dx) * cs + co);
          float f[EPV];
          unpack16<T>(v, f);
#pragma unroll
          for (int e = 0; e < EPV; ++e) { rr[e] += 0.25f * f[e]; const float y = f[e] * A[e] + B[e]; r[e] += 0.25f * (silu ? silu_t<T>(y) : y); }
        }
    }
    if constexpr (SPLIT) {
      store_split4(reinterpret_cast<f16_t*>(out) + (((size_t)n * ho + yo) * wo + xo) * 2 * C, c0, r);
      if (raw) store_split4(raw + (((size_t)n * ho + yo) * wo + xo) * 2 * C, c0, rr);
    } else
      *reinterpret_cast<uint4*>(out + (((size_t)n * ho + yo) * wo + xo) * C + c0) = pack16<T>(r);
  }
}

// ---- single-launch GroupNorm for the low-resolution levels (hw <= 256): one block per (group, sample) computes the
// statistics of its slab (pass 1) and applies x*a+b [+SiLU] (pass 2, served by L1/L2).  Replaces three launches whose
// cost at these sizes is launch latency, not bytes.  Elements are handled in channel pairs (cg is always even).
template <typename T> struct Pair;
template <> struct Pair<float> {
  static __device__ __forceinline__ void load(const float* p, float& a, float& b) { const float2 v = *reinterpret_cast<const float2*>(p); a = v.x; b = v.y; }
  static __device__ __forceinline__ void store(float* p, float a, float b) { *reinterpret_cast<float2*>(p) = make_float2(a, b); }
};
template <> struct Pair<bf16_t> {
  static __device__ __forceinline__ void load(const bf16_t* p, float& a, float& b) {
    const uint32_t v = *reinterpret_cast<const uint32_t*>(p); a = bf16_bits_to_f32(v & 0xffffu); b = bf16_bits_to_f32(v >> 16);
  }
  static __device__ __forceinline__ void store(bf16_t* p, float a, float b) {
    *reinterpret_cast<uint32_t*>(p) = pack2_bf16(a, b);
  }
};
template <> struct Pair<f16_t> {
  static __device__ __forceinline__ void load(const f16_t* p, float& a, float& b) {
    const uint32_t v = *reinterpret_cast<const uint32_t*>(p); a = f16_bits_to_f32(v & 0xffffu); b = f16_bits_to_f32(v >> 16);
  }
  static __device__ __forceinline__ void store(f16_t* p, float a, float b) {
    *reinterpret_cast<uint32_t*>(p) = pack2_f16(a, b);
  }
};

template <typename T>
__global__ __launch_bounds__(256) void gn_fused_kernel(const T* __restrict__ x1, int c1, const T* __restrict__ x2, int c2, int hw,
                                                        int groups, float eps, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, const T* __restrict__ ss, int ld_ss,
                                                        T* __restrict__ out, int silu) {
  __shared__ double red[2][4];
  __shared__ float sAB[2][64];                       // per-channel (a, b) of this group (cg <= 64)
  const int C = c1 + c2, cg = C / groups, hp = cg / 2;
  const int g = blockIdx.x, n = blockIdx.y, tid = threadIdx.x;
  const int npair = hw * hp;
  float s = 0.f, q = 0.f;
  for (int e = tid; e < npair; e += 256) {
    const int p = e / hp, c = g * cg + (e - p * hp) * 2;
    const T* src = c < c1 ? x1 + ((size_t)n * hw + p) * c1 + c : x2 + ((size_t)n * hw + p) * c2 + (c - c1);
    float a, b;
    Pair<T>::load(src, a, b);
    s += a + b;
    q += a * a + b * b;
  }
  double ds = wave_sum((double)s), dq = wave_sum((double)q);
  if ((tid & 63) == 0) { red[0][tid >> 6] = ds; red[1][tid >> 6] = dq; }
  __syncthreads();
  if (tid < cg) {
    const double cnt = (double)hw * cg;
    const double mean = (red[0][0] + red[0][1] + red[0][2] + red[0][3]) / cnt;
    double var = (red[1][0] + red[1][1] + red[1][2] + red[1][3]) / cnt - mean * mean;
    if (var < 0.0) var = 0.0;
    const float rstd = (float)(1.0 / sqrt(var + (double)eps));
    const int c = g * cg + tid;
    float A = rstd * (gamma ? gamma[c] : 1.f);
    float B = (beta ? beta[c] : 0.f) - (float)mean * A;
    if (ss) {
      const float sc = 1.f + ld1<T>(ss + (size_t)n * ld_ss + c);
      const float sh = ld1<T>(ss + (size_t)n * ld_ss + C + c);
      A = A * sc;
      B = B * sc + sh;
    }
    sAB[0][tid] = A;
    sAB[1][tid] = B;
  }
  __syncthreads();
  for (int e = tid; e < npair; e += 256) {
    const int p = e / hp, cl = (e - p * hp) * 2, c = g * cg + cl;
    const T* src = c < c1 ? x1 + ((size_t)n * hw + p) * c1 + c : x2 + ((size_t)n * hw + p) * c2 + (c - c1);
    float a, b;
    Pair<T>::load(src, a, b);
    a = a * sAB[0][cl] + sAB[1][cl];
    b = b * sAB[0][cl + 1] + sAB[1][cl + 1];
    if (silu) { a = silu_t<T>(a); b = silu_t<T>(b); }
    Pair<T>::store(out + ((size_t)n * hw + p) * C + c, a, b);
  }
}

template <typename T>
__global__ __launch_bounds__(256) void resample_kernel(const T* __restrict__ x, T* __restrict__ out, int n_total, int h, int w,
                                                        int c, int mode) {
  constexpr int EPV = ET<T>::EPV;
  const int nchunk = c / EPV;
  const int ho = mode ? 2 * h : h / 2, wo = mode ? 2 * w : w / 2;
  const long long total = (long long)n_total * ho * wo * nchunk;
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
    const int chunk = (int)(idx % nchunk);
    long long pix = idx / nchunk;
    const int xo = (int)(pix % wo); pix /= wo;
    const int yo = (int)(pix % ho);
    const int n = (int)(pix / ho);
    uint4 o;
    if (mode) {
      o = *reinterpret_cast<const uint4*>(x + (((size_t)n * h + (yo >> 1)) * w + (xo >> 1)) * c + chunk * EPV);
    } else {
      float r[EPV];
#pragma unroll
      for (int e = 0; e < EPV; ++e) r[e] = 0.f;
#pragma unroll
      for (int dy = 0; dy < 2; ++dy)
#pragma unroll
        for (int dx = 0; dx < 2; ++dx) {
          const uint4 v = *reinterpret_cast<const uint4*>(x + (((size_t)n * h + 2 * yo + dy) * w + 2 * xo + dx) * c + chunk * EPV);
          float f[EPV];
          unpack16<T>(v, f);
#pragma unroll
          for (int e = 0; e < EPV; ++e) r[e] += 0.25f * f[e];
        }
      o = pack16<T>(r);
    }
    *reinterpret_cast<uint4*>(out + (((size_t)n * ho + yo) * wo + xo) * c + chunk * EPV) = o;
  }
}

inline int gn_splits(int hw) {
  int s = hw / 64;
  if (s < 1) s = 1;
  if (s > GN_SPLITS) s = GN_SPLITS;
  return s;
}

inline int grid_for(long long total, int block) {
  long long g = (total + block - 1) / block;
  if (g > 256 * 16) g = 256 * 16;
  if (g < 1) g = 1;
  return (int)g;
}

}  // namespace

extern "C" int64_t dts_gn_ws_floats(int n, int groups) { return (int64_t)n * GN_SPLITS * groups * 2; }

extern "C" int dts_gn_coef(const void* x1, int c1, const void* x2, int c2, int dtype, int n, int hw, int groups, float eps,
                           const float* gamma, const float* beta, const void* scale_shift, int ld_ss,
                           float* coef, float* ws, dts_stream s) {
  const int C = c1 + c2;
  DTS_CHECK_ARG(x1 && coef && ws, "dts_gn_coef: null pointer");
  DTS_CHECK_ARG(n > 0 && hw > 0 && groups > 0 && C % groups == 0, "dts_gn_coef: C=%d groups=%d", C, groups);
  DTS_CHECK_ARG(c2 == 0 || x2, "dts_gn_coef: c2 without x2");
  const int epv = dtype == DTS_F32 ? 4 : 8;
  DTS_CHECK_ARG(c1 % epv == 0 && c2 % epv == 0, "dts_gn_coef: channels (%d,%d) unsupported", c1, c2);
  DTS_CHECK_ARG(scale_shift == nullptr || ld_ss >= 2 * C, "dts_gn_coef: ld_ss=%d < 2*C=%d", ld_ss, 2 * C);
  const int splits = gn_splits(hw);
  hipStream_t st = to_stream(s);
  DTS_DISPATCH_DTYPE(dtype, {
    const int nchunk = C / ET<T>::EPV;
    const int planes = 256 / (nchunk < 256 ? nchunk : 256);
    const size_t lds = (size_t)planes * C * 2 * sizeof(float);
    hipLaunchKernelGGL((gn_partial_kernel<T>), dim3(splits, n), dim3(256), lds, st, (const T*)x1, c1, (const T*)x2, c2, hw,
                       groups, splits, ws);
    DTS_CHECK_LAUNCH("dts_gn_coef(partial)");
    const int total = n * C;
    hipLaunchKernelGGL((gn_coef_kernel<T>), dim3((total + 255) / 256), dim3(256), 0, st, ws, splits, groups, C, hw, eps, gamma,
                       beta, (const T*)scale_shift, ld_ss, coef, n);
    DTS_CHECK_LAUNCH("dts_gn_coef(coef)");
  });
  return DTS_OK;
}

extern "C" int dts_gn_coef_strips(const float* st1, int c1, const float* st2, int c2, int dtype, int n, int hw, int groups, float eps,
                                  const float* gamma, const float* beta, const void* scale_shift, int ld_ss, float* coef,
                                  dts_stream s) {
  const int C = c1 + c2;
  DTS_CHECK_ARG(st1 && coef, "dts_gn_coef_strips: null pointer");
  DTS_CHECK_ARG(n > 0 && hw > 0 && hw % 64 == 0 && groups > 0 && C % groups == 0, "dts_gn_coef_strips: hw=%d C=%d groups=%d", hw, C,
                groups);
  DTS_CHECK_ARG(c2 == 0 || st2, "dts_gn_coef_strips: c2 without st2");
  DTS_CHECK_ARG(C / groups <= 128, "dts_gn_coef_strips: %d channels per group (max 128)", C / groups);
  DTS_CHECK_ARG(scale_shift == nullptr || ld_ss >= 2 * C, "dts_gn_coef_strips: ld_ss=%d < 2*C=%d", ld_ss, 2 * C);
  DTS_CHECK_ARG(n <= 65535, "dts_gn_coef_strips: n too large for grid.y");
  hipStream_t st = to_stream(s);
  DTS_DISPATCH_DTYPE(dtype, {
    const int threads = (long long)(hw / 64) * (C / groups) > 4096 ? 256 : 64;
    hipLaunchKernelGGL((gn_coef_strips_kernel<T>), dim3(groups, n), dim3(threads), 0, st, st1, c1, st2, c2, hw / 64, groups, hw, eps,
                       gamma, beta, (const T*)scale_shift, ld_ss, coef);
    DTS_CHECK_LAUNCH("dts_gn_coef_strips");
  });
  return DTS_OK;
}

static int gn_apply_impl(const void* x1, int c1, const void* x2, int c2, int dtype, const float* coef, void* out, int n, int h,
                         int w, int silu, int pool, bool split, dts_stream s, void* raw_out = nullptr);
extern "C" int dts_gn_apply(const void* x1, int c1, const void* x2, int c2, int dtype, const float* coef, void* out, int n, int h,
                            int w, int silu, int pool, dts_stream s) {
  return gn_apply_impl(x1, c1, x2, c2, dtype, coef, out, n, h, w, silu, pool, false, s);
}
extern "C" int dts_gn_apply_x3(const float* x1, int c1, const float* x2, int c2, const float* coef, void* out, void* raw_out, int n, int h,
                               int w, int silu, int pool, dts_stream s) {
  DTS_CHECK_ARG((c1 + c2) % 32 == 0, "dts_gn_apply_x3: %d channels are not a multiple of 32", c1 + c2);
  return gn_apply_impl(x1, c1, x2, c2, DTS_F32, coef, out, n, h, w, silu, pool, true, s, raw_out);
}
static int gn_apply_impl(const void* x1, int c1, const void* x2, int c2, int dtype, const float* coef, void* out, int n, int h,
                         int w, int silu, int pool, bool split, dts_stream s, void* raw_out) {
  const int C = c1 + c2;
  DTS_CHECK_ARG(x1 && coef && out, "dts_gn_apply: null pointer");
  DTS_CHECK_ARG(c2 == 0 || x2, "dts_gn_apply: c2 without x2");
  const int epv = dtype == DTS_F32 ? 4 : 8;
  DTS_CHECK_ARG(c1 % epv == 0 && c2 % epv == 0, "dts_gn_apply: channels (%d,%d) unsupported", c1, c2);
  DTS_CHECK_ARG(!pool || (h % 2 == 0 && w % 2 == 0), "dts_gn_apply: pool needs even h,w");
  hipStream_t st = to_stream(s);
  const long long total = (long long)n * (pool ? h / 2 : h) * (pool ? w / 2 : w) * (C / epv);
  if (split) {                                           // f32 in, f16 split image out (dts.h dts_gn_apply_x3)
    using T = float;
    if (pool)
      hipLaunchKernelGGL((gn_apply_kernel<T, true, true>), dim3(grid_for(total, 256)), dim3(256), 0, st, (const T*)x1, c1, (const T*)x2,
                         c2, coef, (T*)out, n, h, w, silu, (f16_t*)raw_out);
    else if (C / epv <= 256 && n <= 65535) {
      const int nchunk = C / epv, k = 256 / nchunk, hw = h * w;
      long long ppb = ((long long)n * hw + 4095) / 4096;
      ppb = ((ppb + 2 * k - 1) / (2 * k)) * (2 * k);
      // (W16: whole-line 16-byte stores through a lane-pair exchange; needs the pair's two chunks on one side of the concat boundary: c1 % 8 == 0.
      //  DTS_GN_FUSE=2 selects the 8-byte-store form for A/B runs)
      if (c1 % 8 == 0 && dts_knob_get(DTS_KNOB_GN_FUSE) != 2)
        hipLaunchKernelGGL((gn_apply_rows_kernel<T, true, true>), dim3((unsigned)((hw + ppb - 1) / ppb), n), dim3(k * nchunk), 0, st, (const T*)x1, c1,
                           (const T*)x2, c2, coef, (T*)out, hw, (int)ppb, silu, (f16_t*)raw_out);
      else
      hipLaunchKernelGGL((gn_apply_rows_kernel<T, true>), dim3((unsigned)((hw + ppb - 1) / ppb), n), dim3(k * nchunk), 0, st, (const T*)x1, c1,
                         (const T*)x2, c2, coef, (T*)out, hw, (int)ppb, silu, (f16_t*)raw_out);
    } else
      hipLaunchKernelGGL((gn_apply_kernel<T, false, true>), dim3(grid_for(total, 256)), dim3(256), 0, st, (const T*)x1, c1, (const T*)x2,
                         c2, coef, (T*)out, n, h, w, silu, (f16_t*)raw_out);
    DTS_CHECK_LAUNCH("dts_gn_apply_x3");
    return DTS_OK;
  }
  DTS_DISPATCH_DTYPE(dtype, {
    if (pool)
      hipLaunchKernelGGL((gn_apply_kernel<T, true>), dim3(grid_for(total, 256)), dim3(256), 0, st, (const T*)x1, c1, (const T*)x2,
                         c2, coef, (T*)out, n, h, w, silu);
    else if (C / epv <= 256 && n <= 65535) {
      const int nchunk = C / epv, k = 256 / nchunk, hw = h * w;
      long long ppb = ((long long)n * hw + 4095) / 4096;              // ~4096 blocks in all
      ppb = ((ppb + 2 * k - 1) / (2 * k)) * (2 * k);                    // whole 2-pixel trips for every thread
      hipLaunchKernelGGL((gn_apply_rows_kernel<T>), dim3((unsigned)((hw + ppb - 1) / ppb), n), dim3(k * nchunk), 0, st, (const T*)x1, c1,
                         (const T*)x2, c2, coef, (T*)out, hw, (int)ppb, silu);
    } else
      hipLaunchKernelGGL((gn_apply_kernel<T, false>), dim3(grid_for(total, 256)), dim3(256), 0, st, (const T*)x1, c1, (const T*)x2,
                         c2, coef, (T*)out, n, h, w, silu);
    DTS_CHECK_LAUNCH("dts_gn_apply");
  });
  return DTS_OK;
}

extern "C" int dts_gn_fused(const void* x1, int c1, const void* x2, int c2, int dtype, int n, int hw, int groups, float eps,
                            const float* gamma, const float* beta, const void* scale_shift, int ld_ss, void* out, int silu,
                            dts_stream s) {
  const int C = c1 + c2;
  DTS_CHECK_ARG(x1 && out, "dts_gn_fused: null pointer");
  DTS_CHECK_ARG(n > 0 && hw > 0 && groups > 0 && C % groups == 0, "dts_gn_fused: C=%d groups=%d", C, groups);
  const int cg = C / groups;
  DTS_CHECK_ARG(cg % 2 == 0 && cg <= 64 && c1 % 2 == 0, "dts_gn_fused: channels per group %d unsupported", cg);
  DTS_CHECK_ARG(c2 == 0 || x2, "dts_gn_fused: c2 without x2");
  DTS_CHECK_ARG(scale_shift == nullptr || ld_ss >= 2 * C, "dts_gn_fused: ld_ss=%d < 2*C=%d", ld_ss, 2 * C);
  DTS_CHECK_ARG(n <= 65535, "dts_gn_fused: n too large for grid.y");
  hipStream_t st = to_stream(s);
  DTS_DISPATCH_DTYPE(dtype, {
    hipLaunchKernelGGL((gn_fused_kernel<T>), dim3(groups, n), dim3(256), 0, st, (const T*)x1, c1, (const T*)x2, c2, hw, groups, eps,
                       gamma, beta, (const T*)scale_shift, ld_ss, (T*)out, silu);
    DTS_CHECK_LAUNCH("dts_gn_fused");
  });
  return DTS_OK;
}

extern "C" int dts_resample2x(const void* x, void* out, int dtype, int n, int h, int w, int c, int mode, dts_stream s) {
  DTS_CHECK_ARG(x && out, "dts_resample2x: null pointer");
  const int epv = dtype == DTS_F32 ? 4 : 8;
  DTS_CHECK_ARG(c % epv == 0, "dts_resample2x: c=%d", c);
  DTS_CHECK_ARG(mode == 1 || (h % 2 == 0 && w % 2 == 0), "dts_resample2x: down needs even h,w");
  hipStream_t st = to_stream(s);
  const long long total = (long long)n * (mode ? 2 * h : h / 2) * (mode ? 2 * w : w / 2) * (c / epv);
  DTS_DISPATCH_DTYPE(dtype, {
    hipLaunchKernelGGL((resample_kernel<T>), dim3(grid_for(total, 256)), dim3(256), 0, st, (const T*)x, (T*)out, n, h, w, c, mode);
    DTS_CHECK_LAUNCH("dts_resample2x");
  });
  return DTS_OK;
}
