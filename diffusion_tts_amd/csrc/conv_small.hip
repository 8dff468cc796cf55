// First and last convolutions of the U-Nets: 3 image channels on one side, so not MFMA-shaped
// (together 0.04 % of the ADM-64 FLOPs).  HBM-bound direct kernels.
//   dts_conv_in3 : edm/training/networks.py:410 (enc '<res>x<res>_conv'), :284 (SongUNet), edm/unet.py:757
//   dts_conv_out3: edm/training/networks.py:433,460 (out_conv), :318,357 (SongUNet aux_conv)
#include "dts_common.h"
#include <type_traits>

namespace {

// x f32 NCHW [n][3][h][w]; w f32 [cout][3][3][3]; out NHWC T
// One thread per output pixel: its 27 inputs are loaded once into registers, then it walks the cout chunks; the
// weight reads are wave-uniform (LDS broadcast), the stores are 16 B per lane.
template <typename T>
__global__ __launch_bounds__(256) void conv_in3_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                        const float* __restrict__ bias, T* __restrict__ out, int n_total, int h,
                                                        int wd, int cout) {
  constexpr int EPV = ET<T>::EPV;
  extern __shared__ float sw[];                       // [27][cout] + [cout] bias
  for (int i = threadIdx.x; i < 27 * cout; i += blockDim.x) {
    const int co = i / 27, k = i - co * 27;
    sw[k * cout + co] = w[i];
  }
  for (int i = threadIdx.x; i < cout; i += blockDim.x) sw[27 * cout + i] = bias ? bias[i] : 0.f;
  __syncthreads();
  const int nchunk = cout / EPV;
  const long long npix = (long long)n_total * h * wd;
  constexpr int GRP = 8;                               // chunks staged per round: 128 B (16-bit) per pixel, 32 KB per block
  char* stage = reinterpret_cast<char*>(sw + 28 * cout);
  for (long long pbase = (long long)blockIdx.x * 256; pbase < npix; pbase += (long long)gridDim.x * 256) {   // block-uniform
    const long long pix0 = pbase + threadIdx.x;
    const bool live = pix0 < npix;
    long long pix = live ? pix0 : 0;
    const int xo = (int)(pix % wd); pix /= wd;
    const int yo = (int)(pix % h);
    const int n = (int)(pix / h);
    float in[27];
#pragma unroll
    for (int ci = 0; ci < 3; ++ci)
#pragma unroll
      for (int kh = 0; kh < 3; ++kh)
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
          const int yy = yo + kh - 1, xx = xo + kw - 1;
          float v = 0.f;
          if (live && (unsigned)yy < (unsigned)h && (unsigned)xx < (unsigned)wd) v = x[(((size_t)n * 3 + ci) * h + yy) * wd + xx];
          in[ci * 9 + kh * 3 + kw] = v;
        }
    // One thread computes all couts of its pixel, so its own 16-byte stores would land 2*cout bytes apart from its
    // neighbours' (a wave-instruction scattering 64 pieces: the kernel ran at 0.7 TB/s).  The block instead stages GRP chunks
    // (8*GRP couts) of its 256 pixels in LDS and writes them out as whole 16*GRP-byte row segments, 16 bytes per lane.
    for (int c0 = 0; c0 < nchunk; c0 += GRP) {
      __syncthreads();                                  // previous group's copy-out done
#pragma unroll
      for (int g = 0; g < GRP; ++g) {
        const int chunk = c0 + g;
        if (chunk >= nchunk) break;
        float acc[EPV];
#pragma unroll
        for (int e = 0; e < EPV; ++e) acc[e] = sw[27 * cout + chunk * EPV + e];
#pragma unroll
        for (int k = 0; k < 27; ++k) {
          const float* wr = sw + k * cout + chunk * EPV;
#pragma unroll
          for (int e = 0; e < EPV; ++e) acc[e] += in[k] * wr[e];
        }
        // chunk slot rotated by the pixel index: the 16-byte slots of 8 neighbouring pixels fall on different banks
        *reinterpret_cast<uint4*>(stage + threadIdx.x * (GRP * 16) + (((g + threadIdx.x) % GRP) << 4)) = pack16<T>(acc);
      }
      __syncthreads();
      const int ng = min(GRP, nchunk - c0);
      for (int t = threadIdx.x; t < 256 * ng; t += 256) {
        const int pl = t / ng, g = t - pl * ng;
        const long long pg = pbase + pl;
        if (pg < npix)
          *reinterpret_cast<uint4*>(out + (size_t)pg * cout + (size_t)(c0 + g) * EPV) =
              *reinterpret_cast<const uint4*>(stage + pl * (GRP * 16) + (((g + pl) % GRP) << 4));
      }
    }
  }
}

// The same convolution on the f32 matrix cores for 16-bit outputs at sizes the U-Nets use (w % 16 == 0, cout = 16*NT): the direct
// kernel above is VALU-bound (27*cout f32 FMAs per pixel: 161 us for 128 x 64 x 64 x 192 against a 40 us output write).  K = 27
// taps padded to 28 = 7 steps of v_mfma_f32_16x16x4_f32, an exact f32 FMA chain like the direct kernel's, on the unrounded f32
// input and weights.  A wave owns 16 consecutive pixels of one image row: A = the weights (kept in registers for the wave's
// whole life, 7 per 16-cout tile), B = the 28 x 16 im2col values (7 gathered loads from the NCHW planes), and a lane's four
// accumulators are four consecutive couts of one pixel.  The 16 x cout tile goes through a wave-private LDS slab (rows padded
// by 16 B) and leaves as ONE contiguous 32*cout-byte NHWC segment, 16 bytes per lane.
template <typename T, int NT>
__global__ __launch_bounds__(256) void conv_in3_mfma_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                             const float* __restrict__ bias, T* __restrict__ out, int nseg, int h,
                                                             int wd) {
  constexpr int ES = sizeof(T) == 4 ? 4 : 2;                            // f32 outputs (round 5: the f32 / split-precision modes took the VALU kernel above, 170 us per 64-row launch)
  constexpr int COUT = NT * 16, ROWB = COUT * ES + 16, TG = 4;          // TG independent accumulator tiles in flight
  static_assert(NT % TG == 0, "tile groups");
  __shared__ __attribute__((aligned(16))) char stage_all[4 * 16 * ROWB];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, p = lane & 15, kq = lane >> 4;
  char* stage = stage_all + wave * 16 * ROWB;
  float a[NT][7];
  int dci[7], dy[7], dx[7];
#pragma unroll
  for (int s = 0; s < 7; ++s) {
    const int k = 4 * s + kq;
    const int kk = k < 27 ? k : 0;
    dci[s] = k < 27 ? kk / 9 : -1;
    dy[s] = (kk % 9) / 3 - 1;
    dx[s] = kk % 3 - 1;
#pragma unroll
    for (int t = 0; t < NT; ++t) a[t][s] = k < 27 ? w[(16 * t + p) * 27 + k] : 0.f;
  }
  f32x4_t b0[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const float4 q = bias ? *reinterpret_cast<const float4*>(bias + 16 * t + 4 * kq) : make_float4(0.f, 0.f, 0.f, 0.f);
    b0[t][0] = q.x; b0[t][1] = q.y; b0[t][2] = q.z; b0[t][3] = q.w;
  }
  const int spr = wd >> 4;
  const size_t plane = (size_t)h * wd;
  for (int seg = blockIdx.x * 4 + wave; seg < nseg; seg += gridDim.x * 4) {
    const int xs = seg % spr, rest = seg / spr;
    const int y = rest % h, n = rest / h;
    float bv[7];
#pragma unroll
    for (int s = 0; s < 7; ++s) {
      const int yy = y + dy[s], xx = xs * 16 + p + dx[s];
      const bool ok = dci[s] >= 0 && (unsigned)yy < (unsigned)h && (unsigned)xx < (unsigned)wd;
      const size_t idx = ok ? ((size_t)n * 3 + dci[s]) * plane + (size_t)yy * wd + xx : 0;
      const float v = x[idx];
      bv[s] = ok ? v : 0.f;
    }
#pragma unroll
    for (int t0 = 0; t0 < NT; t0 += TG) {
      f32x4_t acc[TG];
#pragma unroll
      for (int u = 0; u < TG; ++u) acc[u] = b0[t0 + u];
#pragma unroll
      for (int s = 0; s < 7; ++s)
#pragma unroll
        for (int u = 0; u < TG; ++u) acc[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[t0 + u][s], bv[s], acc[u], 0, 0, 0);
#pragma unroll
      for (int u = 0; u < TG; ++u) {
        if constexpr (std::is_same<T, float>::value) {
          *reinterpret_cast<f32x4_t*>(stage + p * ROWB + (16 * (t0 + u) + 4 * kq) * 4) = acc[u];
        } else {
          uint2 o;
          if (std::is_same<T, bf16_t>::value) { o.x = pack2_bf16(acc[u][0], acc[u][1]); o.y = pack2_bf16(acc[u][2], acc[u][3]); }
          else { o.x = pack2_f16(acc[u][0], acc[u][1]); o.y = pack2_f16(acc[u][2], acc[u][3]); }
          *reinterpret_cast<uint2*>(stage + p * ROWB + (16 * (t0 + u) + 4 * kq) * 2) = o;
        }
      }
    }
    // the slab is wave-private and a wave's LDS operations complete in order: no barrier
    char* dst = reinterpret_cast<char*>(out + (size_t)seg * 16 * COUT);
#pragma unroll
    for (int c = lane; c < COUT * ES; c += 64) {          // 16 pixels x COUT * ES bytes in 16-byte pieces
      const int px = c / (COUT * ES / 16), ch = c - px * (COUT * ES / 16);
      *reinterpret_cast<uint4*>(dst + (size_t)c * 16) = *reinterpret_cast<const uint4*>(stage + px * ROWB + ch * 16);
    }
  }
}

// x NHWC T [n][h][w][c]; w f32 [3][3][3][c] (O, kh, kw, I); out f32 NCHW [n][3][h][w]
// 4 lanes per output pixel split the channel chunks; xor-shuffle reduce.
template <typename T>
__global__ __launch_bounds__(256) void conv_out3_kernel(const T* __restrict__ x, const float* __restrict__ w,
                                                         const float* __restrict__ bias, float* __restrict__ out, int n_total, int h,
                                                         int wd, int c) {
  constexpr int EPV = ET<T>::EPV;
  extern __shared__ float sw[];                       // [3][9][c]
  for (int i = threadIdx.x; i < 27 * c; i += blockDim.x) sw[i] = w[i];
  __syncthreads();
  const int nchunk = c / EPV;
  const long long npix = (long long)n_total * h * wd;
  const int sub = threadIdx.x & 3;
  for (long long pix0 = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 2; pix0 < ((npix + 63) / 64) * 64;
       pix0 += ((long long)gridDim.x * blockDim.x) >> 2) {
    const bool valid = pix0 < npix;
    long long pix = valid ? pix0 : 0;
    const int xo = (int)(pix % wd); pix /= wd;
    const int yo = (int)(pix % h);
    const int n = (int)(pix / h);
    float a0 = 0.f, a1 = 0.f, a2 = 0.f;
    if (valid) {
      for (int kh = 0; kh < 3; ++kh) {
        const int yy = yo + kh - 1;
        if ((unsigned)yy >= (unsigned)h) continue;
        for (int kw = 0; kw < 3; ++kw) {
          const int xx = xo + kw - 1;
          if ((unsigned)xx >= (unsigned)wd) continue;
          const T* row = x + (((size_t)n * h + yy) * wd + xx) * c;
          const float* w0 = sw + (kh * 3 + kw) * c;
          for (int ch = sub; ch < nchunk; ch += 4) {
            const uint4 v = *reinterpret_cast<const uint4*>(row + ch * EPV);
            float f[EPV];
            unpack16<T>(v, f);
#pragma unroll
            for (int e = 0; e < EPV; ++e) {
              a0 += f[e] * w0[ch * EPV + e];
              a1 += f[e] * w0[9 * c + ch * EPV + e];
              a2 += f[e] * w0[18 * c + ch * EPV + e];
            }
          }
        }
      }
    }
    a0 += __shfl_xor(a0, 1, 64); a0 += __shfl_xor(a0, 2, 64);
    a1 += __shfl_xor(a1, 1, 64); a1 += __shfl_xor(a1, 2, 64);
    a2 += __shfl_xor(a2, 1, 64); a2 += __shfl_xor(a2, 2, 64);
    if (valid && sub == 0) {
      const size_t hw = (size_t)h * wd;
      float* o = out + (size_t)n * 3 * hw + (size_t)yo * wd + xo;
      o[0] = a0 + (bias ? bias[0] : 0.f);
      o[hw] = a1 + (bias ? bias[1] : 0.f);
      o[2 * hw] = a2 + (bias ? bias[2] : 0.f);
    }
  }
}

// The same convolution, tiled (round 5): the direct kernel above re-reads every input row nine times through L1 / L2 with four lanes per pixel
// (277 us for 64 x 64 x 64 x 192 f32 against a 40 us read of the tensor).  A block owns a 16 x 16 patch of one image: per chunk of CC
// channels its 18 x 18 halo goes into LDS once (16-byte coalesced loads, pixel pitch padded by 16 bytes so that neighbouring pixels' reads fall
// on different banks), and one thread per pixel accumulates its three outputs against wave-uniform weights (scalar loads).
// Summation order per output: channel chunk outer, tap, channel inner -- fixed, so identical rows give identical results.
template <typename T>
__global__ __launch_bounds__(256) void conv_out3_tiled_kernel(const T* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                                                               float* __restrict__ out, int h, int wd, int c) {
  constexpr int EPV = ET<T>::EPV, VPC = 4, CC = VPC * EPV;               // a chunk = 4 vectors of 16 bytes per pixel: 16 (f32) / 32 (16-bit) channels
  static_assert(CC == VPC * EPV, "chunk = 4 vectors");
  constexpr int PITCH = VPC * 16 + 16;                                  // bytes per halo pixel in LDS
  __shared__ __attribute__((aligned(16))) char halo[18 * 18 * PITCH];
  const int tpr = wd >> 4, tin = blockIdx.x % (tpr * (h >> 4)), n = blockIdx.x / (tpr * (h >> 4));
  const int y0 = (tin / tpr) << 4, x0 = (tin % tpr) << 4;
  const int ty = threadIdx.x >> 4, tx = threadIdx.x & 15;
  float a0 = 0.f, a1 = 0.f, a2 = 0.f;
  for (int c0 = 0; c0 < c; c0 += CC) {
    __syncthreads();                                                    // previous chunk's reads done
    for (int i = threadIdx.x; i < 18 * 18 * VPC; i += 256) {
      const int hp = i / VPC, v = i - hp * VPC;
      const int hy = hp / 18, hx = hp - hy * 18;
      const int yy = y0 + hy - 1, xx = x0 + hx - 1;
      uint4 val = make_uint4(0, 0, 0, 0);
      if ((unsigned)yy < (unsigned)h && (unsigned)xx < (unsigned)wd)
        val = *reinterpret_cast<const uint4*>(x + (((size_t)n * h + yy) * wd + xx) * c + c0 + v * EPV);
      *reinterpret_cast<uint4*>(halo + hp * PITCH + v * 16) = val;
    }
    __syncthreads();
#pragma unroll
    for (int kh = 0; kh < 3; ++kh)
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) {
        const char* hp = halo + ((ty + kh) * 18 + tx + kw) * PITCH;
        // the weights are wave-uniform: read straight from global memory they come in through the scalar cache (s_load) and feed the FMAs as
        // SGPR operands -- staged in LDS they cost one LDS read per three FMAs and the kernel was LDS-issue bound (168 us)
        const float* w0 = w + (size_t)(kh * 3 + kw) * c + c0;
#pragma unroll
        for (int v = 0; v < VPC; ++v) {
          float f[EPV];
          unpack16<T>(*reinterpret_cast<const uint4*>(hp + v * 16), f);
#pragma unroll
          for (int e = 0; e < EPV; ++e) {
            a0 += f[e] * w0[v * EPV + e];
            a1 += f[e] * w0[(size_t)9 * c + v * EPV + e];
            a2 += f[e] * w0[(size_t)18 * c + v * EPV + e];
          }
        }
      }
  }
  const size_t hw = (size_t)h * wd;
  float* o = out + (size_t)n * 3 * hw + (size_t)(y0 + ty) * wd + x0 + tx;
  o[0] = a0 + (bias ? bias[0] : 0.f);
  o[hw] = a1 + (bias ? bias[1] : 0.f);
  o[2 * hw] = a2 + (bias ? bias[2] : 0.f);
}

}  // namespace

extern "C" int dts_conv_in3(const float* x, const float* w, const float* bias, void* out, int dtype, int n, int h, int w_, int cout,
                            dts_stream s) {
  DTS_CHECK_ARG(x && w && out, "dts_conv_in3: null pointer");
  DTS_CHECK_ARG(n > 0 && h > 0 && w_ > 0 && cout % 8 == 0 && cout <= 512, "dts_conv_in3: bad shape (cout=%d)", cout);
  hipStream_t st = to_stream(s);
  const long long nseg = (long long)n * h * (w_ / 16);
  if (w_ % 16 == 0 && (cout == 192 || cout == 128 || cout == 64) && nseg < (1ll << 31) &&
      (bias == nullptr || ((uintptr_t)bias & 15) == 0)) {
    // up to 8 segments per wave (the 7*NT weight registers are loaded once per wave), fewer when that would leave CUs idle
    long long spw = nseg / 2048;
    if (spw < 1) spw = 1;
    if (spw > 8) spw = 8;
    long long g = (nseg + 4 * spw - 1) / (4 * spw);
    if (g > 2048) g = 2048;
#define DTS_IN3_MFMA(TT, NT_)                                                                                                    \
  hipLaunchKernelGGL((conv_in3_mfma_kernel<TT, NT_>), dim3((int)g), dim3(256), 0, st, x, w, bias, (TT*)out, (int)nseg, h, w_)
    if (dtype == DTS_BF16) {
      if (cout == 192) DTS_IN3_MFMA(bf16_t, 12); else if (cout == 128) DTS_IN3_MFMA(bf16_t, 8); else DTS_IN3_MFMA(bf16_t, 4);
    } else if (dtype == DTS_F32) {
      if (cout == 192) DTS_IN3_MFMA(float, 12); else if (cout == 128) DTS_IN3_MFMA(float, 8); else DTS_IN3_MFMA(float, 4);
    } else {
      if (cout == 192) DTS_IN3_MFMA(f16_t, 12); else if (cout == 128) DTS_IN3_MFMA(f16_t, 8); else DTS_IN3_MFMA(f16_t, 4);
    }
#undef DTS_IN3_MFMA
    DTS_CHECK_LAUNCH("dts_conv_in3");
    return DTS_OK;
  }
  DTS_DISPATCH_DTYPE(dtype, {
    const long long total = (long long)n * h * w_;
    long long g = (total + 255) / 256;
    if (g > 4096) g = 4096;
    hipLaunchKernelGGL((conv_in3_kernel<T>), dim3((int)g), dim3(256), 28 * cout * sizeof(float) + 256 * 8 * 16, st, x, w, bias, (T*)out, n,
                       h, w_, cout);
    DTS_CHECK_LAUNCH("dts_conv_in3");
  });
  return DTS_OK;
}

extern "C" int dts_conv_out3(const void* x, int dtype, const float* w, const float* bias, float* out, int n, int h, int w_, int c,
                             dts_stream s) {
  DTS_CHECK_ARG(x && w && out, "dts_conv_out3: null pointer");
  DTS_CHECK_ARG(n > 0 && h > 0 && w_ > 0 && c % 8 == 0 && 27 * c * 4 <= 64 * 1024, "dts_conv_out3: bad shape (c=%d)", c);
  hipStream_t st = to_stream(s);
  if (h % 16 == 0 && w_ % 16 == 0 && c % (dtype == DTS_F32 ? 16 : 32) == 0 && (long long)n * (h / 16) * (w_ / 16) < (1ll << 31)) {
    DTS_DISPATCH_DTYPE(dtype, {
      hipLaunchKernelGGL((conv_out3_tiled_kernel<T>), dim3((unsigned)(n * (h / 16) * (w_ / 16))), dim3(256), 0, st, (const T*)x, w, bias, out, h, w_, c);
      DTS_CHECK_LAUNCH("dts_conv_out3");
    });
    return DTS_OK;
  }
  DTS_DISPATCH_DTYPE(dtype, {
    const long long total = (long long)n * h * w_ * 4;
    long long g = (total + 255) / 256;
    if (g > 8192) g = 8192;
    hipLaunchKernelGGL((conv_out3_kernel<T>), dim3((int)g), dim3(256), 27 * c * sizeof(float), st, (const T*)x, w, bias, out, n, h,
                       w_, c);
    DTS_CHECK_LAUNCH("dts_conv_out3");
  });
  return DTS_OK;
}
