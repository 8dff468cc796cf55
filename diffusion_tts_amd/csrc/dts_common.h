// Shared device/host helpers for libdts_hip.so (gfx950 only; wave = 64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "../../include/dts.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(4))) short s16x4_t;

struct bf16_t { uint16_t v; };
struct f16_t { uint16_t v; };

// ---- error plumbing ---------------------------------------------------------------------------
void dts_set_error(const char* fmt, ...);

#define DTS_CHECK_ARG(cond, ...)                 \
  do {                                           \
    if (!(cond)) {                               \
      dts_set_error(__VA_ARGS__);                \
      return DTS_ERR_ARG;                        \
    }                                            \
  } while (0)

#define DTS_CHECK_LAUNCH(what)                                              \
  do {                                                                      \
    hipError_t e__ = hipGetLastError();                                     \
    if (e__ != hipSuccess) {                                                \
      dts_set_error("%s: launch failed: %s", what, hipGetErrorString(e__)); \
      return DTS_ERR_LAUNCH;                                                \
    }                                                                       \
  } while (0)

static inline hipStream_t to_stream(dts_stream s) { return reinterpret_cast<hipStream_t>(s); }

// ---- tuning knobs (measurement aid: variants are A/B'd interleaved in ONE process, tools/*_bench.py) ------------
// Slot values: -1 = unset (the launcher's default applies).  Initialised from the environment variable of the same name on
// first use; dts_set_tuning() overrides at run time.  Knobs only select between kernels / block orders that give the same
// results.
enum dts_knob {
  DTS_KNOB_ATT_XCD = 0,        // DTS_ATT_XCD      0: plain attention block order, 1: XCD-aware (default)
  DTS_KNOB_ATT_QT = 1,         // DTS_ATT_QT       1|2: query tiles per wave (default: by sequence length)
  DTS_KNOB_CONV_TILE = 2,      // DTS_CONV_TILE    64|128|192|256: cout tile (only honoured when it divides cout)
  DTS_KNOB_CONV_SPLITS = 3,    // DTS_CONV_SPLITS  forced split-K factor (0/unset: heuristic)
  DTS_KNOB_CONV_VARIANT = 4,   // DTS_CONV_VARIANT kernel structure variant (see conv_igemm.hip)
  DTS_KNOB_GN_FUSE = 5,        // DTS_GN_FUSE      (unused on the C side: networks.py reads the environment variable and passes gn_coef)
  DTS_KNOB_ATT_DB = 6,         // DTS_ATT_DB       1: double-buffered attention K/V tiles (one barrier per key tile); default single-buffered.  2: head dim 512 with register staging instead of LDS-DMA (A/B aid)
  DTS_KNOB_CONV_STAGES = 7,    // DTS_CONV_STAGES  2|3|4: LDS ring depth of conv_igemm_kernel (default: by grid size)
  DTS_KNOB_CONV_WAVES = 8,     // DTS_CONV_WAVES   4|8: waves per block of conv_igemm_kernel (default: 8 for grids of <= 256 blocks)
  DTS_KNOB_CONV_HALF_ROUND = 9,// DTS_CONV_HALF_ROUND 0 / 1: half-round grids of the 128-cout ping-pong form at >= 32x32 stay on that kernel / go to the implicit GEMM (A/B aid; default: implicit GEMM in the 16-bit modes, ping-pong in split precision)
  DTS_KNOB_CONV_EPI32 = 10,    // DTS_CONV_EPI32   0 | 1: f32 outputs leave through the accumulator-layout | the row-layout epilogue (default: row layout in the split-precision mode, accumulator layout in the f32 parity mode)
  DTS_KNOB_CONV_SKIP_FOLD = 11,// DTS_CONV_SKIP_FOLD 0: dts_conv_folds_skip answers 0 (the 1x1 skip convolution stays its own launch; A/B aid)
  DTS_KNOB_COUNT = 16
};
int dts_knob_get(int knob);    // defined in elementwise.hip

// ---- element traits -----------------------------------------------------------------------------
template <typename T> struct ET;
template <> struct ET<float> {
  static constexpr int EPV = 4;          // elements per 16-byte vector
  static constexpr int DT = DTS_F32;
};
template <> struct ET<bf16_t> {
  static constexpr int EPV = 8;
  static constexpr int DT = DTS_BF16;
};
template <> struct ET<f16_t> {
  static constexpr int EPV = 8;
  static constexpr int DT = DTS_F16;
};

__device__ __forceinline__ float bf16_bits_to_f32(uint32_t b) { return __builtin_bit_cast(float, b << 16); }
__device__ __forceinline__ uint32_t f32_to_bf16_bits(float f) {
  __bf16 h = (__bf16)f;  // RNE, NaN-preserving (v_cvt_pk_bf16_f32)
  return (uint32_t)__builtin_bit_cast(uint16_t, h);
}
// two floats -> one dword of two 16-bit values, ONE instruction (v_cvt_pk_bf16_f32 / v_cvt_pk_f16_f32, RNE like the scalar
// conversions above).  Converting the halves separately and merging them costs 4 VALU instructions per pair, which was a
// quarter of the conv epilogue's arithmetic and of the attention loop's P packing.
typedef __bf16 dts_bf16x2_t __attribute__((ext_vector_type(2)));
typedef _Float16 dts_f16x2_t __attribute__((ext_vector_type(2)));
typedef float dts_f32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pack2_bf16(float lo, float hi) {
  const dts_f32x2_t v = {lo, hi};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, dts_bf16x2_t));
}
__device__ __forceinline__ uint32_t pack2_f16(float lo, float hi) {
  const dts_f32x2_t v = {lo, hi};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, dts_f16x2_t));
}
__device__ __forceinline__ float f16_bits_to_f32(uint32_t b) {
  return (float)__builtin_bit_cast(_Float16, (uint16_t)b);
}
__device__ __forceinline__ uint32_t f32_to_f16_bits(float f) {
  _Float16 h = (_Float16)f;
  return (uint32_t)__builtin_bit_cast(uint16_t, h);
}

// ---- split-precision operand images (dts.h DTS_F16X3) -----------------------------------------------------------------------------
// x -> (hi, lo * 2^11), both exactly representable in f16: hi = f16(x) (round to nearest even), lo = x - hi (exact in f32: the residual
// of a rounding fits 13 bits).  The matrix cores flush f16 SUBNORMAL inputs (measured: tests/test_gpu_ops.py::test_conv2d_split_precision
// [tiny_values]), so (1) a hi below 2^-14 is dropped and the whole value goes into the lo half, and (2) the lo half carries lo * 2^11 --
// the kernels multiply the weight fragment that meets it by 2^-11 -- which keeps it a normal f16 number down to |x - hi| = 2^-25.
// |x| beyond the f16 range SATURATES to +-65504 (a plain cast would give inf, and lo = x - inf a NaN that poisons every output the
// element touches: ADVICE r4); a NaN stays a NaN.
__device__ __forceinline__ void x3_split(float x, float& hi, float& lo) {
  if (fabsf(x) > 65504.0f) x = copysignf(65504.0f, x);
  float h = (float)(_Float16)x;
  if (fabsf(h) < 6.103515625e-05f) h = 0.f;
  hi = h;
  lo = (x - h) * 2048.0f;
}
// Layout of the image: per pixel row, per group of 32 channels, 128 bytes = hi(32) | lo * 2^11 (32) -- one K step of a split-precision
// convolution.  Element offset (in f16) of channel c's hi half inside a row of C channels; its lo half sits 32 elements further.
__device__ __forceinline__ int x3_off(int c) { return ((c >> 5) << 6) + (c & 31); }
// the attention operand split (dts_split2_f16's arithmetic): y = x * 2^6 -> (hi, lo), saturating like x3_split
__device__ __forceinline__ void x2_split(float x, float& hi, float& lo) {
  float y = x * 64.0f;
  if (fabsf(y) > 65504.0f) y = copysignf(65504.0f, y);
  hi = (float)(_Float16)y;
  lo = y - hi;
}

template <typename T> __device__ __forceinline__ float ld1(const T* p);
template <> __device__ __forceinline__ float ld1<float>(const float* p) { return *p; }
template <> __device__ __forceinline__ float ld1<bf16_t>(const bf16_t* p) { return bf16_bits_to_f32(p->v); }
template <> __device__ __forceinline__ float ld1<f16_t>(const f16_t* p) { return f16_bits_to_f32(p->v); }
template <typename T> __device__ __forceinline__ void st1(T* p, float f);
template <> __device__ __forceinline__ void st1<float>(float* p, float f) { *p = f; }
template <> __device__ __forceinline__ void st1<bf16_t>(bf16_t* p, float f) { p->v = (uint16_t)f32_to_bf16_bits(f); }
template <> __device__ __forceinline__ void st1<f16_t>(f16_t* p, float f) { p->v = (uint16_t)f32_to_f16_bits(f); }

// unpack a 16-byte vector into EPV floats / pack back
template <typename T> __device__ __forceinline__ void unpack16(const uint4& v, float* f);
template <> __device__ __forceinline__ void unpack16<float>(const uint4& v, float* f) {
  f[0] = __builtin_bit_cast(float, v.x); f[1] = __builtin_bit_cast(float, v.y);
  f[2] = __builtin_bit_cast(float, v.z); f[3] = __builtin_bit_cast(float, v.w);
}
template <> __device__ __forceinline__ void unpack16<bf16_t>(const uint4& v, float* f) {
  f[0] = bf16_bits_to_f32(v.x & 0xffffu); f[1] = bf16_bits_to_f32(v.x >> 16);
  f[2] = bf16_bits_to_f32(v.y & 0xffffu); f[3] = bf16_bits_to_f32(v.y >> 16);
  f[4] = bf16_bits_to_f32(v.z & 0xffffu); f[5] = bf16_bits_to_f32(v.z >> 16);
  f[6] = bf16_bits_to_f32(v.w & 0xffffu); f[7] = bf16_bits_to_f32(v.w >> 16);
}
template <> __device__ __forceinline__ void unpack16<f16_t>(const uint4& v, float* f) {
  f[0] = f16_bits_to_f32(v.x & 0xffffu); f[1] = f16_bits_to_f32(v.x >> 16);
  f[2] = f16_bits_to_f32(v.y & 0xffffu); f[3] = f16_bits_to_f32(v.y >> 16);
  f[4] = f16_bits_to_f32(v.z & 0xffffu); f[5] = f16_bits_to_f32(v.z >> 16);
  f[6] = f16_bits_to_f32(v.w & 0xffffu); f[7] = f16_bits_to_f32(v.w >> 16);
}
template <typename T> __device__ __forceinline__ uint4 pack16(const float* f);
template <> __device__ __forceinline__ uint4 pack16<float>(const float* f) {
  return make_uint4(__builtin_bit_cast(uint32_t, f[0]), __builtin_bit_cast(uint32_t, f[1]),
                    __builtin_bit_cast(uint32_t, f[2]), __builtin_bit_cast(uint32_t, f[3]));
}
template <> __device__ __forceinline__ uint4 pack16<bf16_t>(const float* f) {
  return make_uint4(pack2_bf16(f[0], f[1]), pack2_bf16(f[2], f[3]), pack2_bf16(f[4], f[5]), pack2_bf16(f[6], f[7]));
}
template <> __device__ __forceinline__ uint4 pack16<f16_t>(const float* f) {
  return make_uint4(pack2_f16(f[0], f[1]), pack2_f16(f[2], f[3]), pack2_f16(f[4], f[5]), pack2_f16(f[6], f[7]));
}

__device__ __forceinline__ float silu_f(float x) { return x / (1.0f + __expf(-x)); }

// wave-wide (64 lanes) sum
template <typename F> __device__ __forceinline__ F wave_sum(F v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// dtype dispatch for host launchers
#define DTS_DISPATCH_DTYPE(dtype, ...)                                   \
  switch (dtype) {                                                       \
    case DTS_F32: { using T = float; __VA_ARGS__; } break;               \
    case DTS_BF16: { using T = bf16_t; __VA_ARGS__; } break;             \
    case DTS_F16: { using T = f16_t; __VA_ARGS__; } break;               \
    default: dts_set_error("bad dtype %d", (int)(dtype)); return DTS_ERR_ARG; \
  }
