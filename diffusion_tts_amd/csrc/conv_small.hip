// First and last convolutions of the U-Nets: 3 image channels on one side, so not MFMA-shaped
// (together 0.04 % of the ADM-64 FLOPs).  HBM-bound direct kernels.
//   dts_conv_in3 : edm/training/networks.py:410 (enc '<res>x<res>_conv'), :284 (SongUNet), edm/unet.py:757
//   dts_conv_out3: edm/training/networks.py:433,460 (out_conv), :318,357 (SongUNet aux_conv)
#include "dts_common.h"

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
  for (long long pix0 = (long long)blockIdx.x * blockDim.x + threadIdx.x; pix0 < npix; pix0 += (long long)gridDim.x * blockDim.x) {
    long long pix = pix0;
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
          if ((unsigned)yy < (unsigned)h && (unsigned)xx < (unsigned)wd) v = x[(((size_t)n * 3 + ci) * h + yy) * wd + xx];
          in[ci * 9 + kh * 3 + kw] = v;
        }
    T* orow = out + (size_t)pix0 * cout;
    for (int chunk = 0; chunk < nchunk; ++chunk) {
      float acc[EPV];
#pragma unroll
      for (int e = 0; e < EPV; ++e) acc[e] = sw[27 * cout + chunk * EPV + e];
#pragma unroll
      for (int k = 0; k < 27; ++k) {
        const float* wr = sw + k * cout + chunk * EPV;
#pragma unroll
        for (int e = 0; e < EPV; ++e) acc[e] += in[k] * wr[e];
      }
      *reinterpret_cast<uint4*>(orow + chunk * EPV) = pack16<T>(acc);
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

}  // namespace

extern "C" int dts_conv_in3(const float* x, const float* w, const float* bias, void* out, int dtype, int n, int h, int w_, int cout,
                            dts_stream s) {
  DTS_CHECK_ARG(x && w && out, "dts_conv_in3: null pointer");
  DTS_CHECK_ARG(n > 0 && h > 0 && w_ > 0 && cout % 8 == 0 && cout <= 512, "dts_conv_in3: bad shape (cout=%d)", cout);
  hipStream_t st = to_stream(s);
  DTS_DISPATCH_DTYPE(dtype, {
    const long long total = (long long)n * h * w_;
    long long g = (total + 255) / 256;
    if (g > 4096) g = 4096;
    hipLaunchKernelGGL((conv_in3_kernel<T>), dim3((int)g), dim3(256), 28 * cout * sizeof(float), st, x, w, bias, (T*)out, n, h, w_,
                       cout);
    DTS_CHECK_LAUNCH("dts_conv_in3");
  });
  return DTS_OK;
}

extern "C" int dts_conv_out3(const void* x, int dtype, const float* w, const float* bias, float* out, int n, int h, int w_, int c,
                             dts_stream s) {
  DTS_CHECK_ARG(x && w && out, "dts_conv_out3: null pointer");
  DTS_CHECK_ARG(n > 0 && h > 0 && w_ > 0 && c % 8 == 0 && 27 * c * 4 <= 64 * 1024, "dts_conv_out3: bad shape (c=%d)", c);
  hipStream_t st = to_stream(s);
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
