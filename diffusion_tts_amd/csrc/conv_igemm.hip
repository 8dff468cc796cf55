// K1/K2/K3: implicit-GEMM convolution on MFMA for gfx950.
//
// Replaces torch.nn.functional.conv2d at edm/training/networks.py:87 (3x3 pad 1, and the 1x1 skip /
// qkv / proj convs :159,:163,:164), the nearest-2x upsample at :82-83 (fused into the gather), the
// channel concat at :458 (two-source K loop) and the residual add / skip_scale at :178-179,:185-186
// (epilogue).  Same kernel serves the classifier's convs (edm/unet.py:192,228,236).
//
// GEMM view:  D[cout][pixel] = sum_k  W[cout][k] * X[pixel][k],   k = (tap, cin), cin contiguous (NHWC).
//   A operand = packed weights  (row = cout,  K contiguous)
//   B operand = activation rows (row = pixel, K contiguous; zero rows outside the image)
// so a lane's accumulator holds 4 consecutive couts of one pixel -> vector stores into NHWC.
//
// Tile: 256 threads = 4 waves (WM x WN); wave tile (16*MT couts) x (16*NT pixels) of MFMA 16x16 tiles; block tile
// 64x256, 128x128 or 192x128 (couts x pixels) chosen from cout's divisibility; K step = 128 bytes per row (64 bf16/f16,
// 32 f32).  LDS rows are 128 B with the 16-byte chunk index XOR-swizzled by (row & 7): conflict-free ds_read_b128
// fragment reads.  Layers with few pixels (8x8, 16x16 levels; sharded candidate batches) use split-K over grid.y with
// f32 partial slabs and a fixed-order reduce kernel (deterministic: no float atomics), so the grid still fills 256 CUs.
// Staging is LDS-DMA (global_load_lds_dwordx4: 1 KiB per wave-instruction straight into LDS, no VGPR round trip and
// no ds_write -- the register-staged version was LDS-write-bound); the LDS image is lane-linear, so the XOR swizzle is
// applied to the per-lane SOURCE chunk and again on the fragment read.  Out-of-image rows read a 16-byte zero word.
// Double-buffered: tile k+1 is in flight while tile k feeds the MFMAs.
// f32 mode uses v_mfma_f32_16x16x4_f32 (exact f32 FMA chain) = the parity path.
#include "dts_common.h"
#include <hip/hip_ext.h>
#include <type_traits>

static_assert(sizeof(dts_conv_args) == 208, "dts_conv_args layout changed: bump DTS_ABI_VERSION and update the bindings (_lib.py ConvArgs)");

namespace {

// per-call launch state that is not a kernel argument (no globals: dts_conv2d may be called from several host threads)
struct ConvCall {
  hipEvent_t ev_start = nullptr, ev_stop = nullptr;   // optional dispatch-attached timing events (dts_conv_args)
  bool stats_written = false;
};

struct ConvP {
  const char* x1; const char* x2;
  const char* w;
  const float* bias;
  const char* bias_nc;
  const char* residual;
  char* out;
  int c1, c2, cin;
  int ld_bias_nc;
  int n, hin, win, hout, wout, cout;
  int taps;        // 1 or 9
  int up;
  int P;           // n*hout*wout
  int n_ct;        // cout tiles
  int n_pt;        // pixel tiles
  int splits;      // split-K factor (grid.y); > 1 => f32 partial slabs + dts reduce kernel
  int ks_per_split;
  float* partial;  // [splits][P][cout] f32 when splits > 1
  float* stats;    // optional [ceil(P/64)][cout][2]: per 64-pixel strip (sum, sumsq) of the stored outputs (GroupNorm input)
  float out_scale;
  float acc_scale;         // the accumulators are scaled by this before anything is added (split-precision mode: the packed weights carry 2^k)
  int out_split2;          // split-precision mode, f32 epilogue only: the output leaves as the f16 image hi(cout) | lo(cout) of v * 2^6 per pixel
                           // (dts_split2_f16's arithmetic: what dts_attention_x3 reads) instead of as f32 values
  int w_shift, hw_shift;   // log2(wout), log2(hout*wout) when both are powers of two, else -1 (pixel coordinates by division)
  const float* gn_coef;    // optional [n][cin][2] (a, b): GroupNorm of the INPUT applied on the staged halo tile (conv_pp_kernel only)
  int gn_silu;
  int epi_rows;            // f32 outputs: 1 = the row-layout epilogue (conv_epilogue_rows_f32), 0 = the accumulator-layout one (DTS_CONV_EPI32=0)
  // split-precision ping-pong launches only: the block's 1x1 skip convolution as a second K loop (conv_pp_kernel<.., SK = true>)
  const char* sk_x; const char* sk_w;      // split image of the block input [n][hout >> sk_up][wout >> sk_up][sk_c], packed weight [cout][sk_c]
  int sk_c, sk_up;
  float sk_ratio;          // acc_scale of the 3x3 weight / acc_scale of the skip weight (a power of two): the accumulators change units before the second loop
};

template <typename T> struct Mma;
template <> struct Mma<bf16_t> {
  static __device__ __forceinline__ void run(f32x4_t& acc, const uint4& a, const uint4& b) {
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), acc, 0, 0, 0);
  }
};
template <> struct Mma<f16_t> {
  static __device__ __forceinline__ void run(f32x4_t& acc, const uint4& a, const uint4& b) {
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), acc, 0, 0, 0);
  }
};
template <> struct Mma<float> {
  // the 16-byte chunk holds 4 consecutive k of this lane's row; MFMA j consumes element j of A and B
  // (k order inside the 16-wide step is permuted identically for both operands).
  static __device__ __forceinline__ void run(f32x4_t& acc, const uint4& a, const uint4& b) {
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__builtin_bit_cast(float, a.x), __builtin_bit_cast(float, b.x), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__builtin_bit_cast(float, a.y), __builtin_bit_cast(float, b.y), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__builtin_bit_cast(float, a.z), __builtin_bit_cast(float, b.z), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__builtin_bit_cast(float, a.w), __builtin_bit_cast(float, b.w), acc, 0, 0, 0);
  }
};

// the 8 f16 values of a fragment register quad times 2^-11 (4 x v_pk_mul_f16; exact while the result stays a normal number): the
// split-precision mode's weight operand for the activations' lo * 2^11 half (see conv_igemm_kernel's X3I note)
__device__ __forceinline__ uint4 f16x8_mul_2m11(const uint4& v) {
  const f16x8_t h = __builtin_bit_cast(f16x8_t, v) * (_Float16)0x1p-11f;
  return __builtin_bit_cast(uint4, h);
}

template <typename T> struct Vec4;
template <> struct Vec4<float> {
  using type = float4;
  static __device__ __forceinline__ void unpack(const float4& v, float* f) { f[0] = v.x; f[1] = v.y; f[2] = v.z; f[3] = v.w; }
  static __device__ __forceinline__ float4 pack(const float* f) { return make_float4(f[0], f[1], f[2], f[3]); }
  static __device__ __forceinline__ void load(const float* p, float* f) {
    const float4 v = *reinterpret_cast<const float4*>(p);
    f[0] = v.x; f[1] = v.y; f[2] = v.z; f[3] = v.w;
  }
  static __device__ __forceinline__ void store(float* p, const float* f) {
    *reinterpret_cast<float4*>(p) = make_float4(f[0], f[1], f[2], f[3]);
  }
};
template <> struct Vec4<bf16_t> {
  using type = uint2;
  static __device__ __forceinline__ uint2 pack(const float* f) {
    return make_uint2(pack2_bf16(f[0], f[1]), pack2_bf16(f[2], f[3]));
  }
  static __device__ __forceinline__ void unpack(const uint2& v, float* f) {          // one instruction per element: shift / mask
    f[0] = __builtin_bit_cast(float, v.x << 16); f[1] = __builtin_bit_cast(float, v.x & 0xffff0000u);
    f[2] = __builtin_bit_cast(float, v.y << 16); f[3] = __builtin_bit_cast(float, v.y & 0xffff0000u);
  }
  static __device__ __forceinline__ void load(const bf16_t* p, float* f) {
    const uint2 v = *reinterpret_cast<const uint2*>(p);
    f[0] = bf16_bits_to_f32(v.x & 0xffffu); f[1] = bf16_bits_to_f32(v.x >> 16);
    f[2] = bf16_bits_to_f32(v.y & 0xffffu); f[3] = bf16_bits_to_f32(v.y >> 16);
  }
  static __device__ __forceinline__ void store(bf16_t* p, const float* f) {
    *reinterpret_cast<uint2*>(p) = make_uint2(pack2_bf16(f[0], f[1]), pack2_bf16(f[2], f[3]));
  }
};
template <> struct Vec4<f16_t> {
  using type = uint2;
  static __device__ __forceinline__ uint2 pack(const float* f) {
    return make_uint2(pack2_f16(f[0], f[1]), pack2_f16(f[2], f[3]));
  }
  static __device__ __forceinline__ void unpack(const uint2& v, float* f) {
    f[0] = f16_bits_to_f32(v.x & 0xffffu); f[1] = f16_bits_to_f32(v.x >> 16);
    f[2] = f16_bits_to_f32(v.y & 0xffffu); f[3] = f16_bits_to_f32(v.y >> 16);
  }
  static __device__ __forceinline__ void load(const f16_t* p, float* f) {
    const uint2 v = *reinterpret_cast<const uint2*>(p);
    f[0] = f16_bits_to_f32(v.x & 0xffffu); f[1] = f16_bits_to_f32(v.x >> 16);
    f[2] = f16_bits_to_f32(v.y & 0xffffu); f[3] = f16_bits_to_f32(v.y >> 16);
  }
  static __device__ __forceinline__ void store(f16_t* p, const float* f) {
    *reinterpret_cast<uint2*>(p) = make_uint2(pack2_f16(f[0], f[1]), pack2_f16(f[2], f[3]));
  }
};

#ifdef DTS_STAMPS
// diagnostic build only (tools/conv_stamps.py builds a second library with -DDTS_STAMPS; the product library carries none of this):
// per block, wave 0 records s_memtime at entry / after the prologue / after the K loop / at exit, and s_memrealtime at entry / exit
__device__ unsigned long long g_stamps[8192 * 8];
#define DTS_STAMP(slot_)                                                                                       \
  if (threadIdx.x == 0 && (blockIdx.x + blockIdx.y * gridDim.x) < 8192) {                                       \
    unsigned long long t_;                                                                                      \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                                 \
    g_stamps[(blockIdx.x + blockIdx.y * gridDim.x) * 8 + (slot_)] = t_;                                         \
  }
#define DTS_STAMP_RT(slot_)                                                                                    \
  if (threadIdx.x == 0 && (blockIdx.x + blockIdx.y * gridDim.x) < 8192) {                                       \
    unsigned long long t_;                                                                                      \
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                             \
    g_stamps[(blockIdx.x + blockIdx.y * gridDim.x) * 8 + (slot_)] = t_;                                         \
  }
// finer: per-wave cycle sums of the sections of conv_pp_kernel's tap loop (wave 0 = group 0, wave 4 = group 1), kept in SGPRs
#define DTS_SEG_DECL unsigned long long seg_t_ = 0, seg_acc_[6] = {0, 0, 0, 0, 0, 0};
#define DTS_SEG_START { asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(seg_t_)::"memory"); }
#define DTS_SEG_MARK(i_)                                                                                       \
  {                                                                                                            \
    __builtin_amdgcn_sched_barrier(0);                                                                         \
    unsigned long long n_;                                                                                     \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(n_)::"memory");                                \
    seg_acc_[i_] += n_ - seg_t_;                                                                               \
    seg_t_ = n_;                                                                                               \
    __builtin_amdgcn_sched_barrier(0);                                                                         \
  }
#define DTS_SEG_STORE                                                                                          \
  if ((threadIdx.x & 255) == 0 && (blockIdx.x + blockIdx.y * gridDim.x) < 4096) {                              \
    unsigned long long* d_ = g_seg + ((blockIdx.x + blockIdx.y * gridDim.x) * 2 + (threadIdx.x >> 8)) * 8;     \
    for (int i_ = 0; i_ < 6; ++i_) d_[i_] = seg_acc_[i_];                                                      \
  }
__device__ unsigned long long g_seg[4096 * 2 * 8];
#else
#define DTS_STAMP(slot_)
#define DTS_STAMP_RT(slot_)
#define DTS_SEG_DECL
#define DTS_SEG_START
#define DTS_SEG_MARK(i_)
#define DTS_SEG_STORE
#endif

__device__ uint4 g_zero16[1024];   // 16 KiB of zeros: source of padded (out-of-image) rows; a row pointer into it is advanced
                                   // along K like a real one (cin * element size <= 15 KiB: 3 x 1536 channels of the split-precision mode included), so no per-step select is needed

// LDS-DMA through inline asm: hipcc does not count an asm memory op in its s_waitcnt bookkeeping, so it does not
// drain the in-flight tile in front of the (non-aliasing) ds_reads of the other buffer, as it does for the builtin.
// Completion is awaited explicitly (s_waitcnt vmcnt(0) before the barrier that publishes the tile).
// lds_off must be wave-uniform (LDS byte address of this wave-instruction's 1 KiB destination).
__device__ __forceinline__ void glds16(const char* g, uint32_t lds_off) {
  // M0 carries the LDS destination; it is written in the same statement that reads it and declared clobbered (the
  // compiler keeps nothing live in M0 in these kernels), which saves the save/restore pair per load.
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" : : "v"(g), "s"(lds_off) : "memory", "m0");
}

// LDS-DMA through a buffer descriptor: per-lane 32-bit byte offset + wave-uniform SGPR offset.  Two things the flat form above
// cannot do: (1) the K advance is ONE scalar add per tile for all rows of an operand instead of a 64-bit vector add per row
// (the LOAD segment of the ping-pong kernel competes with its SIMD partner's MFMAs for vector issue slots, so every VALU
// instruction removed from it counts double); (2) an offset at or beyond num_records reads as zero, so out-of-image (padding)
// rows need no zero page and no pointer select: their lane offset is simply out of range.
typedef int dts_i32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ dts_i32x4 make_rsrc(const void* base, uint32_t bytes) {
  const uint64_t a = (uint64_t)(uintptr_t)base;
  return dts_i32x4{(int)(uint32_t)a, (int)(uint32_t)(a >> 32), (int)bytes, 0x00020000};     // raw buffer: stride 0, num_records in bytes
}
__device__ __forceinline__ void bdma16(uint32_t voff, dts_i32x4 rsrc, uint32_t soff, uint32_t lds_off) {
  asm volatile("s_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %2 offen lds"
               : : "v"(voff), "s"(rsrc), "s"(soff), "s"(lds_off) : "memory", "m0");
}
#ifndef PP_LOAD_ORDER
#define PP_LOAD_ORDER 2
#endif
constexpr uint32_t DTS_OOR = 0x80000000u;      // lane offset beyond any tensor here (< 2 GiB each): reads as zeros

__device__ __forceinline__ int swz(int row, int chunk) { return row * 128 + ((chunk ^ (row & 7)) << 4); }

// Pixel rows of a block's output tile.  Linear tiles (conv_igemm_kernel; ping-pong kernel at W <= 16): row r is pixel pn0 + r of the
// NHWC tensor.  Patch tiles (ping-pong kernel at W >= 32): the tile is a 16 x 16 pixel patch of one image, row r is patch pixel
// (r >> 4, r & 15).  A 64-row wave strip of either kind lies inside one sample, which is all the GroupNorm strip statistics need.
struct TileMap {
  int pn0;          // first pixel (linear) -- also the patch's tile index * 256, for "whole tile inside P" tests
  int patch;        // 0 linear, 1 patch
  int porg;         // patch: pixel index of the patch's top-left corner
  int w;            // patch: image width
  int strip0;       // index of the tile's first 64-pixel strip in the statistics buffer
  __device__ __forceinline__ int pix(int r) const { return patch ? porg + (r >> 4) * w + (r & 15) : pn0 + r; }
  __device__ __forceinline__ int strip(int wn) const { return strip0 + wn; }
};
__device__ __forceinline__ TileMap linear_tile(int pn0) { return TileMap{pn0, 0, 0, 0, pn0 >> 6}; }

// LDS-DMA pieces [u0, u1) of the residual tile into the staged tile at LDS byte address `stage` (wave-uniform): piece u is the
// 16-byte slots u*NTHR .. u*NTHR+NTHR-1 of the dense, XOR-swizzled pixel-major tile described in conv_epilogue_fast.
template <int BM, int NTHR>
__device__ __forceinline__ void issue_residual_pieces(const ConvP& kp, int cm0, const TileMap& tm, uint32_t stage, int tid, int u0, int u1) {
  constexpr int CPR = BM / 8;
  const char* resb = kp.residual + (size_t)cm0 * 2;
  const size_t rstride = (size_t)kp.cout * 2;
  const uint32_t wave_dst = __builtin_amdgcn_readfirstlane(stage + (tid >> 6) * 1024);
  for (int u = u0; u < u1; ++u) {
    const int slot = u * NTHR + tid, row = slot / CPR, c = slot - row * CPR;
    glds16(resb + (size_t)tm.pix(row) * rstride + ((c ^ (row & 7)) << 4), wave_dst + u * (NTHR * 16));
  }
}

// sum over the 16 lanes of a DPP row (fixed order; every lane ends with the total)
__device__ __forceinline__ float row16_sum(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xf, 0xf, false));    // quad_perm [1,0,3,2]
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xf, 0xf, false));    // quad_perm [2,3,0,1]
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xf, 0xf, false));   // row_half_mirror
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xf, 0xf, false));   // row_mirror
  return v;
}
// The same reduction for EIGHT values at once with the DPP operand folded into the add (v_add_f32_dpp): 32 instructions.  Through the
// builtin above hipcc emits v_mov_b32_dpp + v_pk_add_f32 + the v_movs that pair up the packed operands -- 3.5 instructions per step and
// value, 680 of the 1410 instructions of a wave's epilogue (the epilogue is vector-issue bound: tools/conv_stamps.py).  Same steps, same
// order, same sums.  Interleaving the eight chains keeps >= 7 instructions between a write and the DPP read of the same register (the
// hazard needs 2 wait states, and nothing pads inside an asm statement).
__device__ __forceinline__ void row16_sum8(float (&a)[4], float (&b)[4]) {
#define DTS_DPP_STEP(ctrl_)                                                                     \
  "v_add_f32_dpp %0, %0, %0 " ctrl_ " row_mask:0xf bank_mask:0xf\n\t"                           \
  "v_add_f32_dpp %1, %1, %1 " ctrl_ " row_mask:0xf bank_mask:0xf\n\t"                           \
  "v_add_f32_dpp %2, %2, %2 " ctrl_ " row_mask:0xf bank_mask:0xf\n\t"                           \
  "v_add_f32_dpp %3, %3, %3 " ctrl_ " row_mask:0xf bank_mask:0xf\n\t"                           \
  "v_add_f32_dpp %4, %4, %4 " ctrl_ " row_mask:0xf bank_mask:0xf\n\t"                           \
  "v_add_f32_dpp %5, %5, %5 " ctrl_ " row_mask:0xf bank_mask:0xf\n\t"                           \
  "v_add_f32_dpp %6, %6, %6 " ctrl_ " row_mask:0xf bank_mask:0xf\n\t"                           \
  "v_add_f32_dpp %7, %7, %7 " ctrl_ " row_mask:0xf bank_mask:0xf\n\t"
  asm volatile("s_nop 1\n\t"                                   // the operands may have been written by the instruction just before
               DTS_DPP_STEP("quad_perm:[1,0,3,2]") DTS_DPP_STEP("quad_perm:[2,3,0,1]") DTS_DPP_STEP("row_half_mirror") DTS_DPP_STEP("row_mirror")
               : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]));
#undef DTS_DPP_STEP
}

// ---- epilogue fast path: 16-bit output, whole pixel tile inside the image batch (always, for the U-Net levels at the batch sizes
// of the search loop).  No per-lane predication, every global/LDS address is one base per nt plus compile-time offsets:
// the generic epilogue below executes ~1.5k instructions per wave (a predicated branch per access, a 64-bit address per
// vector), which in-kernel stamps put at 9k cycles per block -- as long as 4 K steps -- before the first byte is stored.
// Arithmetic and its order are the generic path's: ((acc + bias) + bias_nc + residual) * out_scale, rounded once.
template <typename T, int MT, int NT, int BM, int BN, bool RES, bool BNC, bool STATS, int NTHR>
__device__ __forceinline__ void conv_epilogue_fast(const ConvP& kp, f32x4_t (&acc)[MT][NT], int cm0, const TileMap& tm, int wm, int wn, int lrow,
                                                   int lq, char* smem_ring, bool bias_in_acc, int stage_off, int early_u0,
                                                   int early_u1) {
  char* smem = smem_ring + stage_off;                    // staged tile: placed so that the early residual pieces fit the free buffer
  using V4 = typename Vec4<T>::type;
  // Staged tile in the idle LDS ring: pixel-major rows of BM*2 bytes, dense (LDS-DMA writes 1 KiB contiguous per
  // wave-instruction, so no padding is possible); the 16-byte chunk c of row r sits at chunk slot c ^ (r & 7), which keeps
  // the accumulator-layout 8-byte accesses (16 rows per lane group) at 2-way bank conflicts instead of 8-way.
  constexpr int ROWB = BM * 2, CPR = BM / 8, DR = NTHR / CPR, DC = NTHR - DR * CPR, ITERS = BN * CPR / NTHR;
  static_assert(BN * CPR % NTHR == 0 && CPR % 8 == 0, "staged tile trips / swizzle groups");
  const int p_cout = kp.cout;
  const int prow0 = wn * 16 * NT + lrow;                 // tile-local pixel of nt = 0; (row & 7) == (lrow & 7) for every nt
  const int col0 = wm * 16 * MT + lq * 4;                // tile-local cout of mt = 0
  const T* __restrict__ bnc = reinterpret_cast<const T*>(kp.bias_nc);
  const size_t rstride = (size_t)p_cout * 2;
  const int tid = threadIdx.x;
  if constexpr (RES) {
    // residual tile -> LDS by LDS-DMA, whole rows, 16 bytes per lane (the accumulator-layout loads it replaces were 24
    // scattered 8-byte reads per lane: ~8k cycles per block by the in-kernel stamps).  The pieces [early_u0, early_u1) were
    // already issued during the last K step into the ring buffer that step did not read (and are complete: that step's
    // vmcnt(0) + barrier covered them); only the rest is fetched here.
    const uint32_t stage = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
    issue_residual_pieces<BM, NTHR>(kp, cm0, tm, stage, tid, 0, early_u0);
    issue_residual_pieces<BM, NTHR>(kp, cm0, tm, stage, tid, early_u1, ITERS);
  }
  float4 bv[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) bv[mt] = make_float4(0.f, 0.f, 0.f, 0.f);
  // (the bias is already in the accumulators here: the fast path is 16-bit and un-split, exactly the bias_in_acc condition)
  (void)bias_in_acc; (void)bv;
  const T* np[NT];
  if constexpr (BNC) {
    const int hw = kp.hout * kp.wout;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const int pp = tm.pix(prow0 + nt * 16);
      np[nt] = bnc + (size_t)(kp.hw_shift >= 0 ? pp >> kp.hw_shift : pp / hw) * kp.ld_bias_nc + cm0 + col0;
    }
  }
  constexpr bool want_stats = STATS && NT == 4;
  float* sp = want_stats ? kp.stats + ((size_t)tm.strip(wn) * p_cout + cm0 + col0) * 2 : nullptr;
  // this lane's slot of (mt, nt): row prow0 + 16 nt, chunk (col0 / 8 + 2 mt) ^ (lrow & 7), half lq & 1
  char* sw = smem + prow0 * ROWB + (lq & 1) * 8;
  int coff[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) coff[mt] = (((wm * 2 * MT + 2 * mt + (lq >> 1)) ^ (lrow & 7)) << 4);
  const float osc = kp.out_scale;
  if constexpr (RES) {
    if (early_u0 > 0 || early_u1 < ITERS) {            // block-uniform: some pieces were fetched just now
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
    }
  }
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    V4 nv[NT], rv[NT];
    if constexpr (BNC) {
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) nv[nt] = *reinterpret_cast<const V4*>(np[nt] + mt * 16);
    }
    if constexpr (RES) {
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) rv[nt] = *reinterpret_cast<const V4*>(sw + nt * 16 * ROWB + coff[mt]);
    }
    // the element arithmetic runs on PAIRS (v_pk_add_f32 / v_pk_mul_f32 / v_pk_fma_f32: two f32 per lane and issue slot, same IEEE
    // results): the epilogue is vector-issue bound (two waves per SIMD, ~1.1k vector instructions each per tile)
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    f32x2 ssa = {0.f, 0.f}, ssb = {0.f, 0.f}, sqa = {0.f, 0.f}, sqb = {0.f, 0.f};
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      f32x2 va = {acc[mt][nt][0], acc[mt][nt][1]}, vb = {acc[mt][nt][2], acc[mt][nt][3]};
      if constexpr (BNC) {
        float f[4];
        Vec4<T>::unpack(nv[nt], f);
        va += f32x2{f[0], f[1]}; vb += f32x2{f[2], f[3]};
      }
      if constexpr (RES) {
        float f[4];
        Vec4<T>::unpack(rv[nt], f);
        va += f32x2{f[0], f[1]}; vb += f32x2{f[2], f[3]};
      }
      va *= osc; vb *= osc;
      const float v[4] = {va.x, va.y, vb.x, vb.y};
      const V4 pk = Vec4<T>::pack(v);
      *reinterpret_cast<V4*>(sw + nt * 16 * ROWB + coff[mt]) = pk;       // the slot this lane read its residual from
      if constexpr (want_stats) {                      // moments of the values as stored (rounded to T)
        float f[4];
        Vec4<T>::unpack(pk, f);
        const f32x2 fa = {f[0], f[1]}, fb = {f[2], f[3]};
        ssa += fa; ssb += fb;
        sqa = __builtin_elementwise_fma(fa, fa, sqa); sqb = __builtin_elementwise_fma(fb, fb, sqb);
      }
    }
    if constexpr (want_stats) {
      float ss4[4] = {ssa.x, ssa.y, ssb.x, ssb.y}, sq4[4] = {sqa.x, sqa.y, sqb.x, sqb.y};
      // 16-lane (pixel) reduction on DPP row operations -- quad swaps, half-row mirror, row mirror: four v_add_f32_dpp per
      // value, no LDS crossbar (__shfl_xor is a ds_bpermute here: 192 of them per wave and tile, plus their waits)
      row16_sum8(ss4, sq4);
      if (lrow == 0) {
        float4* d = reinterpret_cast<float4*>(sp + mt * 32);
        d[0] = make_float4(ss4[0], sq4[0], ss4[1], sq4[1]);
        d[1] = make_float4(ss4[2], sq4[2], ss4[3], sq4[3]);
      }
    }
  }
  DTS_STAMP(6);
  __syncthreads();
  DTS_STAMP(7);
  // copy-out: 16 bytes per lane, whole rows; (row, chunk slot) advance incrementally (NTHR threads = DR rows + DC chunks)
  int row = tid / CPR, c = tid - row * CPR;
  char* outb = kp.out + (size_t)cm0 * 2;
  typedef unsigned int u32x4_nt __attribute__((ext_vector_type(4)));
#pragma unroll
  for (int u = 0; u < ITERS; ++u) {
    const u32x4_nt v_ = *reinterpret_cast<const u32x4_nt*>(smem + row * ROWB + c * 16);
    __builtin_nontemporal_store(v_, reinterpret_cast<u32x4_nt*>(outb + (size_t)tm.pix(row) * rstride + ((c ^ (row & 7)) << 4)));
    row += DR; c += DC;
    if (c >= CPR) { c -= CPR; ++row; }
  }
}

// ---- epilogue for f32 outputs (the f32 parity mode and the split-precision mode), whole pixel tiles: WAVE-PRIVATE transpose through LDS.
// The accumulator layout gives a lane 4 couts of one pixel: a store instruction of the accumulator-layout epilogue below covers 16 pixel
// rows x 64 bytes, its residual loads likewise (three dependent round trips: 96 residual registers do not fit beside 96 accumulators), and
// its strip statistics cost 32 ds_bpermute per 16-cout slice.  In-kernel stamps (tools/conv_stamps.py --dtype f16x3, round 5): 19.7k cycles
// per ping-pong block without a residual, 33k..45k with one (20 % of a 64x64 192 -> 192 block), and 20k..29k of a 1x1 block whose whole K
// loop is 32k..49k.  Here each wave transposes ITS OWN 16*MT couts x 64 pixels through its private LDS slice in two halves of 32 pixels --
// no block barrier: a wave's LDS operations execute in order -- and works in ROW layout: a lane holds 4 consecutive couts (fixed for the
// lane: bias and per-sample bias live in registers) of RPI pixels per trip, residual loads and output stores are 16 bytes per lane over
// whole 64*MT-byte row segments, all residual loads of a half are in flight before its first use, and the strip moments are summed per
// lane (one cross-lane step at the end).  Arithmetic and order per element are the accumulator-layout path's: outputs are bit-identical;
// the strip moments are summed in another (fixed) order, so EVERY tile of a launch must take the same path -- identical candidate rows
// must give identical outputs wherever they sit in the batch (tests: rows independent of batch position): the path is chosen by the
// layer shape (hout * wout a multiple of 64: strips never straddle samples or the end of the batch), never by the tile's position.
// What it does NOT buy (measured, round 5): with one block per CU the blocks of a launch reach their epilogues together, and 196 KB of f32
// output (+ 196 KB of residual) per block at a 256th of the HBM rate is ~20k (+ 20k) cycles whatever the instruction stream does: -13 % on
// the 1x1 layers, -3 % on the ping-pong residual layers, +1.8 % on the N = 64 step (profiles/r05_experiments.txt).
template <int MT, int NT>
__device__ __forceinline__ void conv_epilogue_rows_f32(const ConvP& kp, f32x4_t (&acc)[MT][NT], int cm0, const TileMap& tm, int wm, int wn,
                                                       int lrow, int lq, char* smem, int wave) {
  static_assert(NT == 4, "64-pixel wave strips");
  constexpr int CW = 16 * MT;                          // couts of this wave's rows
  constexpr int LPR = 4 * MT;                          // lanes per row (one float4 each)
  constexpr int RPI = LPR <= 8 ? 8 : (LPR <= 16 ? 4 : 2);      // rows per trip (a power of two: divides the 32 rows of a half)
  constexpr int ITERS = 32 / RPI;
  constexpr int ROWB = CW * 4 + 16;                    // LDS row pitch: +4 dwords, so the 8 rows of a ds_write_b128 lane group start 4-bank groups apart
  char* const ws = smem + (size_t)wave * (32 * ROWB);
  const int lane = threadIdx.x & 63;
  const int rsub = lane / LPR, c4 = lane - rsub * LPR;
  const bool act = rsub < RPI;
  const int p_cout = kp.cout;
  const int co = cm0 + wm * CW + c4 * 4;
  const float* __restrict__ res = reinterpret_cast<const float*>(kp.residual);
  if (tm.pix(wn * 64) >= kp.P) return;                 // this wave's strip lies beyond the batch (ragged last tile): whole strips only, see above
  float4 b4 = make_float4(0.f, 0.f, 0.f, 0.f), n4 = make_float4(0.f, 0.f, 0.f, 0.f);
  if (act) {
    if (kp.bias) b4 = *reinterpret_cast<const float4*>(kp.bias + co);
    if (kp.bias_nc) {                                  // a 64-pixel strip lies inside one sample
      const int pp0 = tm.pix(wn * 64), hw = kp.hout * kp.wout;
      n4 = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(kp.bias_nc) + (size_t)(kp.hw_shift >= 0 ? pp0 >> kp.hw_shift : pp0 / hw) * kp.ld_bias_nc + co);
    }
  }
  const float asc = kp.acc_scale, osc = kp.out_scale;
  const bool want_stats = kp.stats != nullptr;
  float ss4[4] = {0.f, 0.f, 0.f, 0.f}, sq4[4] = {0.f, 0.f, 0.f, 0.f};
  typedef float f32x4_nt __attribute__((ext_vector_type(4)));
  constexpr int CH = ITERS < 8 ? ITERS : 8;            // trips per residual batch (all CH loads in flight; 16 at once spilled the 96-cout forms)
#pragma unroll
  for (int half = 0; half < 2; ++half) {
#pragma unroll
    for (int n2 = 0; n2 < 2; ++n2)
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
        *reinterpret_cast<f32x4_t*>(ws + (n2 * 16 + lrow) * ROWB + (mt * 16 + lq * 4) * 4) = acc[mt][half * 2 + n2];
#pragma unroll
   for (int u0 = 0; u0 < ITERS; u0 += CH) {
    size_t prow[CH];                                   // element offset of this lane's pixel row of each trip (x cout)
    float4 rv[CH];
#pragma unroll
    for (int u = 0; u < CH; ++u) {
      prow[u] = (size_t)tm.pix(wn * 64 + half * 32 + (u0 + u) * RPI + (act ? rsub : 0)) * p_cout + co;
      rv[u] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (res && act) rv[u] = *reinterpret_cast<const float4*>(res + prow[u]);
    }
#pragma unroll
    for (int u = 0; u < CH; ++u) {
      if (!act) continue;
      const f32x4_t a = *reinterpret_cast<const f32x4_t*>(ws + ((u0 + u) * RPI + rsub) * ROWB + c4 * 16);
      float v[4] = {a[0] * asc + b4.x, a[1] * asc + b4.y, a[2] * asc + b4.z, a[3] * asc + b4.w};      // (acc_scale is 1 outside the split-precision mode)
      if (kp.bias_nc) { v[0] += n4.x; v[1] += n4.y; v[2] += n4.z; v[3] += n4.w; }
      if (res) { v[0] += rv[u].x; v[1] += rv[u].y; v[2] += rv[u].z; v[3] += rv[u].w; }
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] *= osc;
      if (kp.out_split2) {                             // the qkv projection of the split-precision mode: the attention's operand image
        float hi[4], lo[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) x2_split(v[r], hi[r], lo[r]);
        // lanes c4 = 2m / 2m + 1 hold couts 8m .. 8m+3 / 8m+4 .. 8m+7 of the same pixel (LPR is even, so lane parity = c4 parity): the pair swaps
        // halves (one DPP quad permute per dword) and the even lane stores 16 bytes of the hi plane, the odd lane 16 bytes of the lo plane --
        // half the store instructions of the two 8-byte stores per lane (the epilogue is store-issue bound), same bits
        const uint2 h = make_uint2(pack2_f16(hi[0], hi[1]), pack2_f16(hi[2], hi[3])), l = make_uint2(pack2_f16(lo[0], lo[1]), pack2_f16(lo[2], lo[3]));
        const bool odd = (c4 & 1) != 0;
        const uint32_t rx = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(odd ? h.x : l.x), 0xB1, 0xf, 0xf, false);      // quad_perm [1,0,3,2]
        const uint32_t ry = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(odd ? h.y : l.y), 0xB1, 0xf, 0xf, false);
        if (kp.epi_rows == 2) {                        // (A/B aid, DTS_CONV_EPI32=2: the former two 8-byte stores per lane)
          f16_t* o8 = reinterpret_cast<f16_t*>(kp.out) + 2 * (prow[u] - co) + co;
          *reinterpret_cast<uint2*>(o8) = h;
          *reinterpret_cast<uint2*>(o8 + p_cout) = l;
          continue;
        }
        f16_t* orow = reinterpret_cast<f16_t*>(kp.out) + 2 * (prow[u] - co) + (odd ? co - 4 + p_cout : co);
        *reinterpret_cast<uint4*>(orow) = odd ? make_uint4(rx, ry, l.x, l.y) : make_uint4(h.x, h.y, rx, ry);
      } else {
        const f32x4_nt o = {v[0], v[1], v[2], v[3]};
        __builtin_nontemporal_store(o, reinterpret_cast<f32x4_nt*>(reinterpret_cast<float*>(kp.out) + prow[u]));
      }
      if (want_stats) {
#pragma unroll
        for (int r = 0; r < 4; ++r) { ss4[r] += v[r]; sq4[r] += v[r] * v[r]; }
      }
    }
   }
  }
  if (want_stats) {
    // the RPI lanes that hold the same couts (lane = rsub * LPR + c4) are summed in a fixed order: deterministic, no atomics
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float s_ = ss4[r], q_ = sq4[r];
#pragma unroll
      for (int k = 1; k < RPI; ++k) { s_ += __shfl(ss4[r], c4 + k * LPR, 64); q_ += __shfl(sq4[r], c4 + k * LPR, 64); }
      ss4[r] = s_; sq4[r] = q_;
    }
    if (rsub == 0) {
      float4* d = reinterpret_cast<float4*>(kp.stats + ((size_t)tm.strip(wn) * p_cout + co) * 2);
      d[0] = make_float4(ss4[0], sq4[0], ss4[1], sq4[1]);
      d[1] = make_float4(ss4[2], sq4[2], ss4[3], sq4[3]);
    }
  }
}

// ---- epilogue of one (cout tile, pixel tile[, K split]): lane holds couts co..co+3 of pixel pp for each (mt, nt)
// RING: bytes of the caller's (idle) LDS ring, which the staged output tile reuses
template <typename T, int MT, int NT, int BM, int BN, int NTHR, int RING>
__device__ __forceinline__ void conv_epilogue(const ConvP& kp, f32x4_t (&acc)[MT][NT], int cm0, const TileMap& tm, int split, int wm, int wn,
                                              int lrow, int lq, char* smem, bool bias_in_acc, int stage_off, int early_u0, int early_u1) {
  const int p_P = kp.P, p_cout = kp.cout, p_hout = kp.hout, p_wout = kp.wout;
  // ---- epilogue: lane holds couts co..co+3 of pixel pp for each (mt, nt): one 4-element vector load/store
  if (kp.splits > 1) {
    float* part = kp.partial + (size_t)split * p_P * p_cout;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const int pp = tm.pix(wn * 16 * NT + nt * 16 + lrow);
      if (pp >= p_P) continue;
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        const int co = cm0 + wm * 16 * MT + mt * 16 + lq * 4;
        *reinterpret_cast<float4*>(part + (size_t)pp * p_cout + co) =
            make_float4(acc[mt][nt][0], acc[mt][nt][1], acc[mt][nt][2], acc[mt][nt][3]);
      }
    }
    return;
  }
  if constexpr (sizeof(T) == 4 && NT == 4) {
    constexpr int NWV = NTHR / 64;
    static_assert(NWV * 32 * (16 * MT * 4 + 16) <= RING, "the waves' private transpose slices must fit the caller's LDS ring");
    if (kp.epi_rows && (p_hout * p_wout) % 64 == 0) {  // launch-uniform (by layer shape, not by tile position: see conv_epilogue_rows_f32)
      conv_epilogue_rows_f32<MT, NT>(kp, acc, cm0, tm, wm, wn, lrow, lq, smem, (int)(threadIdx.x >> 6));
      return;
    }
  }
  if constexpr (sizeof(T) == 2) {
    if (tm.pn0 + BN <= p_P) {                          // block-uniform
      const bool r_ = kp.residual != nullptr, b_ = kp.bias_nc != nullptr, s_ = NT == 4 && kp.stats != nullptr;
#define DTS_EPI(R_, B_, S_) conv_epilogue_fast<T, MT, NT, BM, BN, R_, B_, S_, NTHR>(kp, acc, cm0, tm, wm, wn, lrow, lq, smem, bias_in_acc, stage_off, early_u0, early_u1)
      if (r_) { if (b_) { if (s_) DTS_EPI(true, true, true); else DTS_EPI(true, true, false); }
                else    { if (s_) DTS_EPI(true, false, true); else DTS_EPI(true, false, false); } }
      else    { if (b_) { if (s_) DTS_EPI(false, true, true); else DTS_EPI(false, true, false); }
                else    { if (s_) DTS_EPI(false, false, true); else DTS_EPI(false, false, false); } }
#undef DTS_EPI
      return;
    }
  }
  // Epilogue, one 16-cout slice (mt) at a time; interleaved load->store pairs would serialise a memory round trip per
  // (mt, nt) because the compiler must assume `out` aliases the inputs, so the reads are hoisted by hand.
  const T* __restrict__ res = reinterpret_cast<const T*>(kp.residual);
  const T* __restrict__ bnc = reinterpret_cast<const T*>(kp.bias_nc);
  T* __restrict__ out = reinterpret_cast<T*>(kp.out);
  const int hw = p_hout * p_wout;
  using V4 = typename Vec4<T>::type;
  int ppv[NT], nsv[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const int pp = tm.pix(wn * 16 * NT + nt * 16 + lrow);
    ppv[nt] = pp < p_P ? pp : -1;
    nsv[nt] = pp < p_P ? pp / hw : 0;
  }
  // residual vectors: all reads of a slice group in flight before its first store (a memory round trip per slice otherwise); the group
  // is all MT slices, except for f32 outputs of the wide tiles (MT > 4: 96 residual registers beside 96 accumulators do not fit), which
  // go two slices at a time; the rarer per-sample bias (SongUNet's conv0) is fetched per slice to keep the register budget under 256.
  constexpr int RG = (sizeof(T) == 4 && MT > 4) ? 2 : MT;
  static_assert(MT % RG == 0, "residual slice groups");
  V4 rv[RG][NT], nv[NT];
  float4 bv[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
    bv[mt] = (kp.bias && !bias_in_acc) ? *reinterpret_cast<const float4*>(kp.bias + cm0 + wm * 16 * MT + mt * 16 + lq * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
  const float asc = kp.acc_scale;
  // 16-bit outputs leave through the (now idle) LDS tile ring: the accumulator layout gives each lane 4 couts of one pixel,
  // i.e. 8-byte stores scattered over 16 pixel rows per instruction, and a timing-only build without the epilogue showed
  // those stores costing a quarter of the whole conv time (2x on the 1x1 layers).  Transposed through LDS, every lane
  // stores 16 contiguous bytes and a wave-instruction covers 1 KiB of whole output rows.
  constexpr bool VIA_LDS = sizeof(T) == 2;
  constexpr int ROWP = BM * 2 + 16;                    // LDS row pitch of the staged tile (pixel-major), bytes
  static_assert(!VIA_LDS || BN * ROWP <= RING, "staged tile must fit the caller's LDS ring");
  const bool want_stats = NT == 4 && kp.stats != nullptr;
  float* sp = want_stats ? kp.stats + ((size_t)tm.strip(wn) * p_cout) * 2 : nullptr;
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const int co = cm0 + wm * 16 * MT + mt * 16 + lq * 4;
    const float4 bcur = bv[mt];
    if (res && mt % RG == 0) {
#pragma unroll
      for (int m2 = 0; m2 < RG; ++m2) {
        const int co_ = cm0 + wm * 16 * MT + (mt + m2) * 16 + lq * 4;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
          if (ppv[nt] >= 0) rv[m2][nt] = *reinterpret_cast<const V4*>(res + (size_t)ppv[nt] * p_cout + co_);
      }
    }
    if (bnc) {
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
        if (ppv[nt] >= 0) nv[nt] = *reinterpret_cast<const V4*>(bnc + (size_t)nsv[nt] * kp.ld_bias_nc + co);
    }
    float ss4[4] = {0.f, 0.f, 0.f, 0.f}, sq4[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      if (ppv[nt] < 0) continue;
      float v[4];
      // (acc_scale is 1 outside the split-precision mode: x * 1 + b rounds once, like x + b)
      v[0] = acc[mt][nt][0] * asc + bcur.x; v[1] = acc[mt][nt][1] * asc + bcur.y;
      v[2] = acc[mt][nt][2] * asc + bcur.z; v[3] = acc[mt][nt][3] * asc + bcur.w;
      if (bnc) {
        float f[4];
        Vec4<T>::unpack(nv[nt], f);
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] += f[r];
      }
      if (res) {
        float f[4];
        Vec4<T>::unpack(rv[mt % RG][nt], f);
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] += f[r];
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] *= kp.out_scale;
      const V4 pk = Vec4<T>::pack(v);
      if constexpr (VIA_LDS)
        *reinterpret_cast<V4*>(smem + (wn * 16 * NT + nt * 16 + lrow) * ROWP + (wm * 16 * MT + mt * 16 + lq * 4) * 2) = pk;
      else if (sizeof(T) == 4 && kp.out_split2) {
        // the qkv projection of the split-precision mode: its only reader is the split-precision attention, so the f32 tensor and the
        // separate split pass (4 B written + 4 B read per element) are skipped
        float hi[4], lo[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) x2_split(v[r], hi[r], lo[r]);
        f16_t* orow = reinterpret_cast<f16_t*>(kp.out) + (size_t)ppv[nt] * 2 * p_cout + co;
        *reinterpret_cast<uint2*>(orow) = make_uint2(pack2_f16(hi[0], hi[1]), pack2_f16(hi[2], hi[3]));
        *reinterpret_cast<uint2*>(orow + p_cout) = make_uint2(pack2_f16(lo[0], lo[1]), pack2_f16(lo[2], lo[3]));
      } else
        *reinterpret_cast<V4*>(out + (size_t)ppv[nt] * p_cout + co) = pk;
      if (want_stats) {                                // moments of the values as stored (rounded to T)
        float f[4];
        Vec4<T>::unpack(pk, f);
#pragma unroll
        for (int r = 0; r < 4; ++r) { ss4[r] += f[r]; sq4[r] += f[r] * f[r]; }
      }
    }
    if (want_stats) {
      // Fused GroupNorm statistics (input moments of the next networks.py:104-106 norm): this wave covers one 64-pixel
      // strip; fixed-order reduction over the 16 pixel lanes, (sum, sumsq) per (strip, cout).  No atomics.
#pragma unroll
      for (int r = 0; r < 4; ++r) {
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) { ss4[r] += __shfl_xor(ss4[r], o, 64); sq4[r] += __shfl_xor(sq4[r], o, 64); }
      }
      if (lrow == 0 && tm.pix(wn * 64) < p_P) {
        float4* d = reinterpret_cast<float4*>(sp + (size_t)co * 2);
        d[0] = make_float4(ss4[0], sq4[0], ss4[1], sq4[1]);
        d[1] = make_float4(ss4[2], sq4[2], ss4[3], sq4[3]);
      }
    }
  }
  if constexpr (VIA_LDS) {
    __syncthreads();
    constexpr int CPR = BM / 8;                          // 16-byte chunks per staged row
    char* outb = kp.out + (size_t)cm0 * 2;
    const int rows = min(BN, p_P - tm.pn0);
#pragma unroll 4
    for (int t = threadIdx.x; t < rows * CPR; t += NTHR) {
      const int row = t / CPR, c = t - row * CPR;
      typedef unsigned int u32x4_nt __attribute__((ext_vector_type(4)));
      const u32x4_nt v_ = *reinterpret_cast<const u32x4_nt*>(smem + row * ROWP + c * 16);
      __builtin_nontemporal_store(v_, reinterpret_cast<u32x4_nt*>(outb + (size_t)tm.pix(row) * p_cout * 2 + c * 16));
    }
  }
}

// PF: both k-substeps' fragments are read ahead of the first MFMA (40 more VGPRs; the second set lands while the first
// set's MFMAs run).  Measured on one box against the read-as-you-go order: +3..11 % on the 3x3 layers, -4..8 % on the
// short-K 1x1 layers, so the launcher picks it by kernel size.
// STAGES: depth of the LDS tile ring.  2 = tile k+1 in flight while tile k feeds the MFMAs (two resident blocks per CU cover each other's
// waits: the throughput configuration).  3 / 4 = two / three tiles in flight behind a COUNTED vmcnt: for launches whose grid leaves one
// block per CU (the small per-GPU batches of a sharded search, MCTS groups, the 8x8 level), where a 2-deep ring makes every K step one
// full L2 round trip (~2.1k cycles against 768 cycles of MFMAs).
// OT: element type of the epilogue's operands and of the output (bias_nc, residual, out).  OT = T everywhere except the split-precision
// mode (T = f16, OT = float: dts.h DTS_F16X3), whose epilogue is the f32 one and whose K loop is the X3I form:
// X3I (round 5).  An f32 activation x = hi + lo and an f32 weight w = wh + wl (f16 parts) give x.w ~ hi.wh + lo.wh + hi.wl.  The operand
// images interleave the parts per 32 channels: a 128-byte K-step row of the activations is [hi(32) | lo * 2^11 (32)], of the weights
// [wh(32) | wl(32)], so ONE staged K step (the same LDS-DMA pieces and fragment reads as a 16-bit step) feeds THREE MFMAs per accumulator
// tile: (wh, hi), (wh * 2^-11, lo * 2^11), (wl, hi) -- the scaled copy of the wh fragment is made in registers (4 v_pk_mul_f16).  Round 4
// staged three separate K steps (planes hi | lo | hi against hi | hi * 2^-11 | lo): 1.5 x the LDS-DMA pieces, fragment reads and HBM bytes
// per product for the same MFMAs.
template <typename T, int MT, int NT, int WM, int WN, bool PF, int STAGES = 2, typename OT = T>
__global__ __launch_bounds__(64 * WM * WN, (WM * WN == 8 || STAGES == 2 ? 2 : 1)) void conv_igemm_kernel(const ConvP kp) {
  const char* const p_x1 = kp.x1; const char* const p_x2 = kp.x2; const char* const p_w = kp.w;
  const int p_c1 = kp.c1, p_c2 = kp.c2, p_cin = kp.cin, p_hin = kp.hin, p_win = kp.win, p_hout = kp.hout, p_wout = kp.wout;
  const int p_taps = kp.taps, p_up = kp.up, p_P = kp.P, p_n_ct = kp.n_ct, p_n_pt = kp.n_pt;
  // 4 waves per block and two blocks per CU is the throughput configuration (8-wave blocks with one block per CU measured slower
  // when the grid fills the chip twice over: profiles/r01_conv_variants.txt).  8 waves (WM x WN = 4 x 2, half the couts per wave) is
  // for grids of at most one block per CU: a lone 4-wave block has ONE wave per SIMD, which issues its 10 LDS-DMA pieces (~110 cycles
  // each) and THEN its 48 MFMAs -- 1900 cycles per K step, the matrix pipe idle 60 % of it (tools/conv_stamps.py); with two waves per
  // SIMD each issues 5 pieces and 24 MFMAs and one wave's issue stalls sit under the other's MFMAs.
  constexpr int NW = WM * WN;
  constexpr bool X3I = !std::is_same<T, OT>::value;
  static_assert(!X3I || std::is_same<T, f16_t>::value, "split precision runs on the f16 matrix instruction");
  static_assert(NW == 4 || NW == 8, "4 or 8 waves");
  static_assert(STAGES >= 2 && STAGES <= 4, "ring depth");
  constexpr int SLAB = 8 * NW;             // rows staged by one wave-instruction round of the whole block
  constexpr int BM = 16 * MT * WM;         // couts per block
  constexpr int BN = 16 * NT * WN;         // pixels per block
  constexpr int EPV = ET<T>::EPV;
  constexpr int BKE = 8 * EPV;             // K elements per step (128 bytes)
  constexpr int ES = 16 / EPV;             // element size
  constexpr int RA = BM / SLAB;            // A wave-instructions (8 rows each) per wave
  constexpr int RB = BN / SLAB;            // B wave-instructions per wave
  static_assert(BM % SLAB == 0 && BN % SLAB == 0, "tile rows must split over the waves x 8 rows");
  constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128, STAGE_BYTES = A_BYTES + B_BYTES;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  DTS_STAMP_RT(4);
  DTS_STAMP(0);

  // ---- XCD-aware tile mapping: blocks b, b+8, b+16.. (same XCD under round-robin dispatch) walk the
  // cout tiles of one pixel tile consecutively, so the activation rows are re-read from that XCD's L2.
  const int nblk = p_n_ct * p_n_pt;
  int bid = blockIdx.x;
  {
    const int q = nblk >> 3, r = nblk & 7, xcd = bid & 7, j = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
  }
  const int ct = bid % p_n_ct, pt = bid / p_n_ct;
  const int cm0 = ct * BM, pn0 = pt * BN;

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm = wid % WM, wn = wid / WM;
  const int chunk = tid & 7, r0 = tid >> 3;          // staging: 16-byte chunk, first row

  int pix_n[RB], pix_hw[RB];
  const int K = p_taps * p_cin;
  const int steps_per_tap = p_cin / BKE;
  const int nk_all = p_taps * steps_per_tap;
  const int ks_begin = blockIdx.y * kp.ks_per_split;
  const int ks_end = min(nk_all, ks_begin + kp.ks_per_split);

  // Issue-side addressing is incremental: per-row source pointers are rebuilt only when the tap (or the concat source)
  // changes and otherwise just advance by one K step (one 64-bit add per load).  The per-load address arithmetic of the
  // first version (64-bit multiply-add, select, M0 save/restore: ~14 instructions x 10 loads) was costing each wave about
  // as many issue cycles per K step as its 48 MFMAs.
  const int schunk = (chunk ^ (r0 & 7)) * 16;          // source chunk of this lane's (linear) LDS slot
  const char* zsrc = reinterpret_cast<const char*>(g_zero16);
  const char* arow[RA];
  const char* brow[RB];
  // LDS byte address of this wave's 8 rows inside each slab (wave-uniform by construction)
  const uint32_t lds_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  const uint32_t wave_rows = __builtin_amdgcn_readfirstlane(lds_base + (tid >> 6) * 8 * 128);

#define SET_ROWS(tap_, ci0_)                                                                                  \
  {                                                                                                           \
    const int dh_ = (p_taps == 9) ? (tap_) / 3 - 1 : 0, dw_ = (p_taps == 9) ? (tap_) % 3 - 1 : 0;             \
    const char* xb; int cs, cofs;                                                                             \
    if ((ci0_) < p_c1) { xb = p_x1; cs = p_c1; cofs = (ci0_); } else { xb = p_x2; cs = p_c2; cofs = (ci0_) - p_c1; } \
    _Pragma("unroll") for (int j = 0; j < RB; ++j) {                                                          \
      const int hu = (pix_hw[j] >> 16) + dh_, wu = (pix_hw[j] & 0xffff) + dw_;                                \
      const bool ok = pix_n[j] >= 0 && (unsigned)hu < (unsigned)p_hout && (unsigned)wu < (unsigned)p_wout;    \
      const int hs = p_up ? (hu >> 1) : hu, ws = p_up ? (wu >> 1) : wu;                                       \
      brow[j] = ok ? xb + ((size_t)(pix_n[j] + hs * p_win + ws) * cs + cofs) * ES + schunk : zsrc;            \
    }                                                                                                         \
  }
#define ISSUE_TILE(buf_)                                                                                      \
  {                                                                                                           \
    const uint32_t sa_ = wave_rows + (buf_) * STAGE_BYTES;                                                    \
    const uint32_t sb_ = sa_ + A_BYTES;                                                                       \
    _Pragma("unroll") for (int j = 0; j < RA; ++j) glds16(arow[j], sa_ + j * SLAB * 128);                     \
    _Pragma("unroll") for (int j = 0; j < RB; ++j) glds16(brow[j], sb_ + j * SLAB * 128);                     \
  }
#define ADVANCE_K()                                                                                           \
  {                                                                                                           \
    ci0 += BKE;                                                                                               \
    _Pragma("unroll") for (int j = 0; j < RA; ++j) arow[j] += BKE * ES;                                       \
    if (ci0 == p_cin) { ci0 = 0; ++tap; SET_ROWS(tap, 0); }                                                   \
    else if (ci0 == p_c1) { SET_ROWS(tap, ci0); }                                                             \
    else { _Pragma("unroll") for (int j = 0; j < RB; ++j) brow[j] += BKE * ES; }                              \
  }

  const bool bias_in_acc = sizeof(OT) == 2 && kp.splits == 1 && kp.bias != nullptr;

  // issue-side K state (tap, cin offset) runs STAGES-1 tiles ahead of the compute side
  int tap = ks_begin / steps_per_tap, ci0 = (ks_begin - tap * steps_per_tap) * BKE;
#pragma unroll
  for (int j = 0; j < RA; ++j) arow[j] = p_w + ((size_t)(cm0 + r0 + SLAB * j) * K + (size_t)ks_begin * BKE) * ES + schunk;
  // first tile: the weight rows go out before the pixel coordinates of the activation rows are worked out, so that
  // arithmetic runs under the first loads' latency (the prologue is ~5k exposed cycles per block)
  {
    const uint32_t sa_ = wave_rows;
    _Pragma("unroll") for (int j = 0; j < RA; ++j) glds16(arow[j], sa_ + j * SLAB * 128);
  }
  // ---- per-thread pixel rows of the B tile (fixed for the whole K loop)
#pragma unroll
  for (int j = 0; j < RB; ++j) {
    const int pp = pn0 + r0 + SLAB * j;
    if (pp < p_P) {
      int n, ho, wo;
      if (kp.hw_shift >= 0) {                            // block-uniform: every U-Net level here is a power of two
        n = pp >> kp.hw_shift;
        const int rem = pp & ((1 << kp.hw_shift) - 1);
        ho = rem >> kp.w_shift; wo = rem & ((1 << kp.w_shift) - 1);
      } else {
        const int hw = p_hout * p_wout;
        n = pp / hw;
        const int rem = pp - n * hw;
        ho = rem / p_wout; wo = rem - ho * p_wout;
      }
      pix_n[j] = n * p_hin * p_win;
      pix_hw[j] = (ho << 16) | wo;
    } else {
      pix_n[j] = -1;
      pix_hw[j] = 0;
    }
  }
  SET_ROWS(tap, ci0);
  {
    const uint32_t sb_ = wave_rows + A_BYTES;
    _Pragma("unroll") for (int j = 0; j < RB; ++j) glds16(brow[j], sb_ + j * SLAB * 128);
  }
  // 16-bit modes start the accumulators at the bias instead of adding it in the epilogue, where the load latency (~1.5k cycles by the
  // in-kernel stamps) sits on every block's critical path; f32 (parity) adds it last like the reference, and split-K adds it once in the
  // reduce pass.  The bias is fetched HERE, behind the first tile's LDS-DMA issue: in front of it, its round trip delayed the first
  // issue by ~1k cycles in every block (the accumulator initialisation waits for it).
  f32x4_t acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i) {
    float4 b0 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (bias_in_acc) b0 = *reinterpret_cast<const float4*>(kp.bias + cm0 + wm * 16 * MT + i * 16 + (lane >> 4) * 4);
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[i][j] = f32x4_t{b0.x, b0.y, b0.z, b0.w};
  }
  constexpr int LPT = RA + RB;                         // LDS-DMA loads per wave and tile (vmcnt counts them in issue order)
  if constexpr (STAGES > 2) {
    // tiles 1 .. STAGES-2 go out behind the first; only tile 0 has to have landed before the loop starts
    int ahead = 0;
#pragma unroll
    for (int q = 1; q < STAGES - 1; ++q)
      if (ks_begin + q < ks_end) { ADVANCE_K(); ISSUE_TILE(q); ++ahead; }
    if (ahead == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * LPT) : "memory");
    else if (ahead == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LPT) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  } else {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __syncthreads();
  DTS_STAMP(1);

  const int lrow = lane & 15, lq = lane >> 4;
  // early residual fetch (whole 16-bit tiles only: the ones that take conv_epilogue_fast)
  constexpr int RING_BYTES = STAGES * STAGE_BYTES, STAGED_BYTES = BN * BM * 2, PIECE = 64 * NW * 16, PIECES = STAGED_BYTES / PIECE;
  static_assert(STAGED_BYTES % PIECE == 0 && STAGED_BYTES <= RING_BYTES, "staged tile pieces");
  const bool res_early = sizeof(OT) == 2 && kp.residual != nullptr && kp.splits == 1 && pn0 + BN <= p_P;
  int stage_off = 0, early_u0 = 0, early_u1 = 0;
  int buf = 0;
  for (int ks = ks_begin; ks < ks_end; ++ks) {
    const char* sa = smem + buf * STAGE_BYTES + (wm * 16 * MT) * 128;
    const char* sb = smem + buf * STAGE_BYTES + A_BYTES + (wn * 16 * NT) * 128;
    uint4 fa[MT], fb[NT], ga[MT], gb[NT];
    // fragment reads go out first; the LDS-DMA issue for the next tile (address arithmetic + M0 writes) then overlaps
    // their latency instead of delaying the first MFMA
#pragma unroll
    for (int i = 0; i < NT; ++i) fb[i] = *reinterpret_cast<const uint4*>(sb + swz(i * 16 + lrow, lq));
#pragma unroll
    for (int i = 0; i < MT; ++i) fa[i] = *reinterpret_cast<const uint4*>(sa + swz(i * 16 + lrow, lq));
    if (PF) {
#pragma unroll
      for (int i = 0; i < NT; ++i) gb[i] = *reinterpret_cast<const uint4*>(sb + swz(i * 16 + lrow, lq + 4));
#pragma unroll
      for (int i = 0; i < MT; ++i) ga[i] = *reinterpret_cast<const uint4*>(sa + swz(i * 16 + lrow, lq + 4));
    }
    // the buffer written here was last read in iteration ks-1; every wave has passed that iteration's barrier
    if (ks + STAGES - 1 < ks_end) {
      ADVANCE_K();
      int wr = buf + STAGES - 1;
      if (wr >= STAGES) wr -= STAGES;
      ISSUE_TILE(wr);
    } else if (res_early && ks + 1 == ks_end) {
      // last K step: nothing is in flight any more and every ring buffer but `buf` is free, so the larger free side of the ring takes
      // the residual pieces that fit it; they land under this step's MFMAs instead of an exposed fetch in the epilogue
      const int lo_free = buf * STAGE_BYTES, hi_free = (STAGES - 1 - buf) * STAGE_BYTES;      // bytes below / above the buffer being read
      int f0, f1;                                                                             // the free interval used (ring coordinates)
      if (lo_free >= hi_free) { f0 = 0; f1 = lo_free; stage_off = 0; }
      else { f0 = (buf + 1) * STAGE_BYTES; f1 = RING_BYTES; stage_off = min(f0, RING_BYTES - STAGED_BYTES); }
      early_u0 = max(0, (f0 - stage_off + PIECE - 1) / PIECE);
      early_u1 = min(PIECES, max(0, (f1 - stage_off) / PIECE));
      if (early_u1 < early_u0) early_u1 = early_u0;
      issue_residual_pieces<BM, 64 * NW>(kp, cm0, linear_tile(pn0), lds_base + stage_off, tid, early_u0, early_u1);
    }
    uint4 fs[X3I ? MT : 1];
    if constexpr (X3I) {
#pragma unroll
      for (int i = 0; i < MT; ++i) fs[i] = f16x8_mul_2m11(fa[i]);                     // wh * 2^-11 (meets lo * 2^11)
    }
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j) Mma<T>::run(acc[i][j], fa[i], fb[j]);              // (X3I: wh . hi)
    if (!PF) {
      __builtin_amdgcn_s_setprio(0);
#pragma unroll
      for (int i = 0; i < NT; ++i) gb[i] = *reinterpret_cast<const uint4*>(sb + swz(i * 16 + lrow, lq + 4));
#pragma unroll
      for (int i = 0; i < MT; ++i) ga[i] = *reinterpret_cast<const uint4*>(sa + swz(i * 16 + lrow, lq + 4));
      __builtin_amdgcn_s_setprio(1);
    }
    if constexpr (X3I) {
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) Mma<T>::run(acc[i][j], fs[i], gb[j]);            // wh * 2^-11 . lo * 2^11
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) Mma<T>::run(acc[i][j], ga[i], fb[j]);            // wl . hi
    } else {
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) Mma<T>::run(acc[i][j], ga[i], gb[j]);
    }
    __builtin_amdgcn_s_setprio(0);
    // next tile has landed (LDS-DMA completion is tracked by vmcnt, in issue order: the tiles issued after it may still fly) and
    // this wave's LDS reads of the current tile have returned
    if constexpr (STAGES > 2) {
      const int later = min(STAGES - 2, ks_end - 2 - ks);                  // tiles issued beyond tile ks+1 (wave-uniform)
      if (later >= 2) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(2 * LPT) : "memory");
      else if (later == 1) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(LPT) : "memory");
      else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    }
    __syncthreads();
    if (++buf == STAGES) buf = 0;
  }

  DTS_STAMP(2);
  conv_epilogue<OT, MT, NT, BM, BN, 64 * NW, RING_BYTES>(kp, acc, cm0, linear_tile(pn0), (int)blockIdx.y, wm, wn, lrow, lq, smem, bias_in_acc, stage_off, early_u0, early_u1);
  DTS_STAMP(3);
  DTS_STAMP_RT(5);
}
#undef SET_ROWS
#undef ISSUE_TILE
#undef ADVANCE_K

// act(x*a + b) on the 8 channels of one 16-byte LDS slot: the arithmetic of gn_apply_rows_kernel (groupnorm.hip) -- one fma, SiLU
// through v_rcp for the 16-bit types, one rounding -- so fusing the apply into the conv does not change a bit of its input.
template <typename T>
__device__ __forceinline__ uint4 gn_act8(const uint4 v, const float (&ca)[8], const float (&cb)[8], int silu) {
  float f[8];
  unpack16<T>(v, f);
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const float y = f[e] * ca[e] + cb[e];
    f[e] = silu ? y * __builtin_amdgcn_rcpf(1.0f + __expf(-y)) : y;
  }
  return pack16<T>(f);
}

// ------------------------------------------------------------------------------------------------------------------------------
// Ping-pong + halo-tile variant (16-bit types, cout % 192 == 0, square power-of-two images >= 16, no fused upsample):
// ONE 8-wave block per CU on a 192-cout x 256-pixel tile.  Two ideas, each answering a measurement (profiles/r02_conv_variants.txt):
//
// (1) ANTI-PHASE WAVE GROUPS.  In conv_igemm_kernel every wave runs [fragment reads -> 10 LDS-DMA issues -> 48 MFMAs -> vmcnt(0) +
//     barrier] as one serial chain of ~2.1k cycles per K step and the two co-resident blocks of a CU move in lockstep, so the matrix
//     pipe idles while both waves of a SIMD read and issue loads (73 % MFMA-busy inside the K loop).  Here waves 0-3 (group 0) and
//     waves 4-7 (group 1; wave w and w+4 share a SIMD) alternate between a LOAD segment (fragment reads of K tile t, LDS-DMA issue)
//     and a COMPUTE segment (the 48 MFMAs of tile t), group 1 one segment behind group 0, one workgroup barrier per segment: while
//     one wave of a SIMD issues MFMAs its partner reads LDS and issues DMA (MI355X_MICROARCH.md, "Two waves per SIMD").
// (2) HALO TILES.  Timing-only builds of the first ping-pong version showed the L2 -> LDS fill as the limit, not its latency and not
//     its instruction count: with the activation pieces out of range (issued, written to LDS as zeros, nothing fetched) the loop ran
//     20-25 % faster, with the weight pieces out of range 14 %, and neither a longer prefetch distance nor cheaper addressing nor an
//     L2-friendly K order changed that.  A 3x3 conv staged as 9 shifted copies of the activation tile moves every activation byte
//     nine times from L2 into LDS.  Here K runs channel chunk OUTER, tap INNER, and per 64-channel chunk the block stages the HALO of
//     its 16 x 16 pixel patch once -- 18 x 18 = 324 rows of 128 bytes -- and the nine taps read their B fragments from it at shifted
//     rows: 1.27x one tile instead of 9x.  The fill per K tile drops from 56 KB to ~29 KB (24 KB of it weights).
//
// Work split: wave = (group g, index i): couts g*96.., pixels i*64.. of the 16 x 16 patch (the image itself at W = 16).  The weight
// halves are therefore PRIVATE to a group (each group stages its own 96 rows, 3 slots, two tiles ahead = 3-4 segments of flight);
// the halo tile (18 x 18 = 324 rows of 128 bytes for a 3x3 conv, the 256 pixel rows themselves for a 1x1) is shared by the two
// groups and double-buffered per chunk.  [An earlier arrangement -- groups split by pixel half, private halos, shared weights --
// left group 0's weight rows one segment of flight and ran 13 % below its own no-weight-fetch build.]
//   LDS: A 2 groups x 3 slots x 12 KB = 72 KB | halo 2 buffers x 328 rows x 128 B = 82 KB      (154 KB)
// Ordering (group 0: LOAD(t) = segment 2t, COMPUTE(t) = 2t+1; group 1 one later):
//   * fragment reads complete (lgkmcnt(0)) before the barrier that ends a LOAD segment;
//   * A_g(t+2) is issued in LOAD(t) into the slot of A_g(t-1), last read in LOAD(t-1) by the same group; first read in LOAD(t+2);
//   * the halo of chunk c+1 goes into the buffer chunk c-1 used (last read: group 1's last LOAD of chunk c-1, one segment before
//     group 0's first LOAD of chunk c, where the first piece is issued): one 8-row piece per wave per LOAD segment for the first six
//     tiles of chunk c (all four in the one LOAD of a 1x1 conv);
//   * COMPUTE(t) ends with a counted wait that leaves only the pieces issued in LOAD(t) in flight (vmcnt(3 + #halo pieces)): A_g(t+1)
//     and every older halo piece have landed; a 3x3 chunk issues no halo piece in its last three tiles, so the next chunk's halo is
//     complete two tiles before its first LOAD; a 1x1 chunk is one tile, so there the wait is vmcnt(0).
// Padding: halo rows outside the image get an out-of-range lane offset, which the buffer load turns into zeros.
// FUSED GROUPNORM APPLY (kp.gn_coef, 3x3 only): the conv's input is the un-normalised tensor; every wave turns the halo pieces IT
// staged into act(x*a + b) in place -- a piece is one 16-byte slot per lane, whose 8 channels are fixed for the lane, so its 16
// coefficients sit in registers for the whole chunk -- two tiles after issuing them (the counted wait at the end of COMPUTE(j+1) has
// retired piece j), between the MFMAs of COMPUTE(j+2), where the wave's vector issue slots are otherwise idle.  Padding rows are
// skipped and stay zero, as in the reference (norm, then pad).  The halo is staged once per chunk, so this is one pass over the
// input, not nine; the separate apply pass (read + write of the whole tensor, 3.6 ms of a 38 ms step) disappears.
// The summation order over K differs from conv_igemm_kernel's (same terms, other f32 rounding order); it is fixed, so identical
// inputs still give identical outputs (ties stay ties).
// DBG != 0: timing-only diagnostic builds (outputs wrong by construction; tools/conv_bench.py conv_variant=11/21/41/51):
//   1 = no LDS-DMA after the prologue, 2 = no MFMAs, 4 = every halo piece out of range, 5 = every A piece out of range
template <typename T, int TAPS, int DBG = 0, bool GN = false, int MT = 6, typename OT = T, bool SK = false>
__global__ __launch_bounds__(512, 2) void conv_pp_kernel(const ConvP kp) {
  // MT = M tiles (of 16 couts) per wave: 6 -> 192-cout blocks (the EDM U-Net widths), 4 -> 128-cout blocks (classifier, SD VAE widths)
  static_assert(MT == 6 || MT == 4, "cout tile");
  // X3I: the split-precision K loop (see conv_igemm_kernel): a K tile's rows are [hi | lo * 2^11] / [wh | wl] of 32 channels, and COMPUTE
  // issues 3 x MT x NT MFMAs per tile instead of 2 x MT x NT -- a 1.5 x longer matrix segment per barrier pair for the same LOAD work
  constexpr bool X3I = !std::is_same<T, OT>::value;
  static_assert(!X3I || (std::is_same<T, f16_t>::value && DBG == 0 && !GN), "split precision: the shipped f16 form only");
  constexpr int NT = 4, GM = 16 * MT, BM = 2 * GM, BN = 256, NTHR = 512, AJ = MT / 2;      // AJ: A pieces (8 rows x 4 waves) per wave and tile
  constexpr int BKE = 64, ES = 2;
  constexpr int A_HALF = GM * 128, A_RING = 3 * A_HALF, H_OFF = 2 * A_RING, H_BUF = 328 * 128;       // see the LDS map above
  extern __shared__ __attribute__((aligned(16))) char smem[];
  DTS_STAMP_RT(4);
  DTS_STAMP(0);
  const char* const p_x1 = kp.x1; const char* const p_x2 = kp.x2; const char* const p_w = kp.w;
  // W, H: OUTPUT image size (= input size, or twice it when the nearest-2x upsample of the input is fused into the halo gather)
  const int p_c1 = kp.c1, p_c2 = kp.c2, p_cin = kp.cin, W = kp.wout, H = kp.hout, Wi = kp.win, Hi = kp.hin, ups = kp.up ? 1 : 0;
  constexpr int p_taps = TAPS;                           // 9 (3x3) or 1 (1x1): the tap loop is unrolled
  const int p_n_ct = kp.n_ct, p_n_pt = kp.n_pt;

  const int nblk = p_n_ct * p_n_pt;
  int bid = blockIdx.x;
  {
    const int q = nblk >> 3, r = nblk & 7, xcd = bid & 7, j = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
  }
  const int ct = bid % p_n_ct, pt = bid / p_n_ct;
  const int cm0 = ct * BM;

  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = w >> 2, wi = w & 3;
  const int wm = grp, wn = wi;
  const int chunk = lane & 7, r0 = lane >> 3;
  const int lrow = lane & 15, lq = lane >> 4;

  // ---- tile geometry (uniform).  W >= 32: 16 x 16 patch of one image; W <= 16: 256 consecutive pixels (one 16 x 16 image / four 8 x 8)
  const int hw = H * W;
  TileMap tm;
  tm.pn0 = pt * BN; tm.w = W;
  int img0, y0 = 0, x0 = 0;                              // first image of the tile, patch origin
  if (W >= 32) {
    const int tpr = W >> 4, tpi = tpr * (H >> 4);        // patches per row / per image (powers of two)
    const int tin = pt & (tpi - 1);
    img0 = pt / tpi;
    y0 = (tin / tpr) << 4; x0 = (tin & (tpr - 1)) << 4;
    tm.patch = 1; tm.porg = img0 * hw + y0 * W + x0;
    tm.strip0 = img0 * (hw >> 6) + tin * 4;
  } else {
    img0 = tm.pn0 / hw;
    tm.patch = 0; tm.porg = 0; tm.strip0 = tm.pn0 >> 6;
  }
  constexpr int bd = (p_taps == 9) ? 1 : 0;              // halo border
  constexpr int hwid = 16 + 2 * bd;                      // halo row length in pixels: 18 / 16
  constexpr int hr_total = hwid * hwid;                  // 324 / 256 halo rows
  constexpr int npieces = (hr_total + 7) / 8;            // 41 / 32 LDS-DMA pieces of 8 rows, dealt round-robin to the 8 waves
  constexpr int NHP = (npieces + 7) / 8;                 // pieces per wave: 6 / 4

  const int K = p_taps * p_cin;
  const int nk_all = p_taps * (p_cin / BKE);
  const int ks_begin = blockIdx.y * kp.ks_per_split;
  const int ks_end = min(nk_all, ks_begin + kp.ks_per_split);
  const int nk = ks_end - ks_begin;

  const int schunk = (chunk ^ (r0 & 7)) * 16;
  const uint32_t lds_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;

  // ---- LDS-DMA addressing: (descriptor, 32-bit lane offset, scalar offset).  A share = rows grp*GM + 8*(wi + 4j) + r0 (j < AJ) of the
  // weight tile; weights are stored [cout][tap][cin]: the A tile of (chunk, tap) is at byte (tap*cin + chunk*64) * ES of each row.
  const dts_i32x4 rs_w = make_rsrc(p_w, (uint32_t)((size_t)kp.cout * K * ES));
  uint32_t avo[AJ];
#pragma unroll
  for (int j = 0; j < AJ; ++j) avo[j] = DBG == 5 ? DTS_OOR : (uint32_t)((cm0 + grp * GM + 8 * (wi + 4 * j) + r0) * K) * ES + schunk;
  const uint32_t a_dst = lds_base + grp * A_RING + (8 * wi) * 128;               // + j*32*128 + slot*A_HALF (private ring of group g)
  const uint32_t h_dst = lds_base + H_OFF + (8 * w) * 128;                       // + j*64*128 + buffer*H_BUF  (piece w + 8j)

  // halo rows of this lane: piece w + 8j (j < NHP), row 8*(w + 8j) + r0 -> source pixel (or -1: padding / beyond the halo)
  int hpix[NHP];
#pragma unroll
  for (int j = 0; j < NHP; ++j) {
    const int hr = 8 * (w + 8 * j) + r0;
    hpix[j] = -1;
    if (hr < hr_total) {
      const int hy = hr / hwid, hx = hr - hy * hwid;
      const int y = y0 + hy - bd, x = x0 + hx - bd;
      // networks.py:82-83: the upsampled pixel (y, x) is input pixel (y >> 1, x >> 1); padding is applied AFTER the upsample
      if ((unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W) hpix[j] = (img0 * Hi + (y >> ups)) * Wi + (x >> ups);
    }
  }
  uint32_t hvo[NHP];                                   // lane offsets of the halo rows in the current source
  uint32_t hval = 0;                                   // bit j: this lane's row of piece j is inside the image
#pragma unroll
  for (int j = 0; j < NHP; ++j) hval |= (hpix[j] >= 0 ? 1u : 0u) << j;
  constexpr bool fuse_gn = GN && p_taps == 9;          // compile-time: the plain instantiation carries none of this
  // GroupNorm coefficients of this lane's 8 channels (source chunk chunk ^ r0 of the 64-channel K chunk) for the halo being staged
  float gca[8], gcb[8];
  const float* const gn_base = fuse_gn ? kp.gn_coef + ((size_t)img0 * p_cin + 8 * (chunk ^ (r0 & 7))) * 2 : nullptr;
  (void)gn_base; (void)hval;
#define PP_LOAD_COEF(ci0_)                                                                                    \
  {                                                                                                           \
    const float4* q_ = reinterpret_cast<const float4*>(gn_base + (size_t)(ci0_) * 2);                         \
    _Pragma("unroll") for (int e = 0; e < 4; ++e) {                                                           \
      const float4 t_ = q_[e];                                                                                \
      gca[2 * e] = t_.x; gcb[2 * e] = t_.y; gca[2 * e + 1] = t_.z; gcb[2 * e + 1] = t_.w;                     \
    }                                                                                                         \
  }
  // in-place act(x*a+b) of this wave's piece j_ of the halo buffer at LDS byte offset hb_ (generic pointer smem + hb_)
#define PP_GN_PIECE(j_, hb_)                                                                                  \
  if ((hval >> (j_)) & 1u) {                                                                                  \
    uint4* s_ = reinterpret_cast<uint4*>(smem + (hb_) + (size_t)(8 * (w + 8 * (j_))) * 128 + lane * 16);      \
    *s_ = gn_act8<T>(*s_, gca, gcb, kp.gn_silu);                                                              \
  }
  dts_i32x4 rs_x;                                      // descriptor of the current source (x1, or x2 past the concat boundary)
#define PP_SET_SRC(ci0_)                                                                                      \
  {                                                                                                           \
    const bool s2_ = (ci0_) >= p_c1;                                                                          \
    const int cs_ = s2_ ? p_c2 : p_c1;                                                                        \
    rs_x = make_rsrc(s2_ ? p_x2 : p_x1, (uint32_t)((size_t)kp.n * Hi * Wi * cs_ * ES));                       \
    _Pragma("unroll") for (int j = 0; j < NHP; ++j)                                                           \
      hvo[j] = (hpix[j] >= 0 && DBG != 4) ? (uint32_t)(hpix[j] * cs_) * ES + schunk : DTS_OOR;                \
  }
  int a_so = ((ks_begin % p_taps) * p_cin + (ks_begin / p_taps) * BKE) * ES;     // scalar offset of the NEXT A tile this wave issues
  int a_tap = ks_begin % p_taps;
  int a_slot = 0;                                                                 // its ring slot
#define PP_ISSUE_A_PIECES(J0_, J1_)     /* pieces [J0_, J1_) of this group's GM rows of the next A tile */       \
  {                                                                                                           \
    const uint32_t d_ = a_dst + a_slot * A_HALF;                                                              \
    _Pragma("unroll") for (int j = (J0_); j < (J1_); ++j) bdma16(avo[j], rs_w, (uint32_t)a_so, d_ + j * (32 * 128)); \
  }
#define PP_ADVANCE_A()                                                                                        \
  {                                                                                                           \
    if (++a_tap == p_taps) { a_tap = 0; a_so += (BKE - (p_taps - 1) * p_cin) * ES; } else a_so += p_cin * ES; \
    if (++a_slot == 3) a_slot = 0;                                                                            \
  }
#define PP_ISSUE_A() { PP_ISSUE_A_PIECES(0, AJ); PP_ADVANCE_A(); }
  // All AJ weight pieces of tile t+2 go out in LOAD(t).  Issuing some from COMPUTE(t) instead (one after each dozen MFMAs, to even the
  // ~1100-cycle LOAD and the 768-cycle COMPUTE segments out: profiles/r02_conv_sq_counters.txt) was built twice and measured slower both
  // times (-3.5 % over the 3x3 ADM shapes, r02_conv_variants.txt items 2b and 13): an issue cycle in COMPUTE is a lost matrix cycle.
  // DBG 8 builds that split form (one piece in LOAD) for the A/B: tools/conv_bench.py conv_variant=81.
  constexpr int A_IN_LOAD = DBG == 8 ? 1 : AJ;

  // ---- B fragments come from the halo tile: centre row of this lane's pixel for each n tile; a tap adds a uniform row delta
  int hc[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const int pl = wn * 64 + nt * 16 + lrow;           // pixel of the patch: (pl >> 4, pl & 15)
    hc[nt] = ((pl >> 4) + bd) * hwid + (pl & 15) + bd;
  }

  const bool bias_in_acc = sizeof(OT) == 2 && kp.splits == 1 && kp.bias != nullptr;

  // ---- prologue: A_g(0), A_g(1), the halo of the first chunk (all pieces).  The launcher makes every K split a whole number of
  // chunks (nk % TAPS == 0, ks_begin % TAPS == 0).
  int ci0 = (ks_begin / p_taps) * BKE;                   // channel offset of the chunk being computed
  const int nchunks = nk / p_taps;
  PP_ISSUE_A();
  if (nk > 1) PP_ISSUE_A();
  PP_SET_SRC(ci0);
  {
    const uint32_t so_ = (uint32_t)(ci0 >= p_c1 ? ci0 - p_c1 : ci0) * ES;
#pragma unroll
    for (int j = 0; j < NHP; ++j)
      if (w + 8 * j < npieces) bdma16(hvo[j], rs_x, so_, h_dst + j * (64 * 128));
  }
  // accumulators start at the bias; fetched behind the prologue's LDS-DMA issue (in front of it the round trip delayed the first issue)
  f32x4_t acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i) {
    float4 b0 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (bias_in_acc) b0 = *reinterpret_cast<const float4*>(kp.bias + cm0 + wm * GM + i * 16 + lq * 4);
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[i][j] = f32x4_t{b0.x, b0.y, b0.z, b0.w};
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if constexpr (fuse_gn) {                               // the first chunk's halo: every wave normalises the pieces it staged
    PP_LOAD_COEF(ci0);
#pragma unroll
    for (int j = 0; j < NHP; ++j)
      if (w + 8 * j < npieces) PP_GN_PIECE(j, H_OFF);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  asm volatile("s_barrier" ::: "memory");
  DTS_STAMP(1);
  if (grp == 1) asm volatile("s_barrier" ::: "memory");          // the stagger: group 1 runs one segment behind group 0

  DTS_SEG_DECL
  DTS_SEG_START
  int t = 0;                                             // K tile index inside this block's range
  uint32_t a_rd = grp * A_RING;                          // read slot of this group's A ring (byte offset)
  uint32_t h_rd = H_OFF;                                 // halo buffer being read (byte offset)
  for (int c = 0; c < nchunks; ++c) {
    const bool more = c + 1 < nchunks;
    const int next_ci0 = ci0 + BKE;
    const uint32_t h_wr = (h_rd == H_OFF) ? h_dst + H_BUF : h_dst;                         // LDS-DMA destination: the other buffer
    uint32_t h_so = 0;
    if (more) {
      // the next chunk's halo goes out during this chunk; all of this chunk's pieces were issued during the previous one, so the
      // source may be switched now
      if ((next_ci0 >= p_c1) != (ci0 >= p_c1)) PP_SET_SRC(next_ci0);
      h_so = (uint32_t)(next_ci0 >= p_c1 ? next_ci0 - p_c1 : next_ci0) * ES;
      if constexpr (fuse_gn) PP_LOAD_COEF(next_ci0);
    }
    const uint32_t h_wr_off = (h_rd == H_OFF) ? H_OFF + H_BUF : H_OFF;                     // the same buffer as a byte offset in smem
#pragma unroll 1
    for (int tap = 0; tap < p_taps; ++tap) {
      // ---------------- LOAD(t): fragments of tile t -> registers; DMA: this group's A rows, then (at most) its halo piece(s)
      const char* sa = smem + a_rd;
      const char* sh = smem + h_rd;
      // (the tap loop is NOT unrolled: unrolled, hipcc hoists the 9 x 4 fragment addresses and spills -- 256 VGPRs + scratch)
      const int dh3 = (tap * 11) >> 5;                   // tap / 3 for tap < 9
      const int dlt = (p_taps == 9) ? (dh3 - 1) * hwid + (tap - 3 * dh3 - 1) : 0;
      uint4 fa[MT], fb[NT], ga[MT], gb[NT];
      // LOAD order, measured on the nine 3x3 ADM shapes (profiles/r02_conv_variants.txt item 12; outputs identical in all three):
      //   0: fragment reads, then the LDS-DMA pieces, then lgkmcnt(0)                                   1104 TF/s (unweighted mean)
      //   1: fragment reads, lgkmcnt(0), then the pieces (no reads in the LDS queue while they issue)   1086
      //   2: the pieces first, then the fragment reads (shipped)                                        1109
      // DBG 6 / 7 build orders 1 / 2 whatever the default (tools/conv_bench.py conv_variant=61 / 71).
      constexpr int LOAD_ORDER = DBG == 6 ? 1 : (DBG == 7 ? 2 : PP_LOAD_ORDER);
      // DBG 9 = the former order (all 20 fragment reads in LOAD), kept for the A/B: conv_variant=91.  (X3I, round 5: with 72 MFMAs per tile the
      // six late reads were moved back into LOAD as an A/B: 1 374 vs 1 376 evals/s on one box, 228 instead of 206 VGPRs -- no difference, not kept.)
      constexpr bool GA_LATE = DBG != 9;
      if constexpr (LOAD_ORDER != 2) {
#pragma unroll
      for (int i = 0; i < NT; ++i) {
        const int row = hc[i] + dlt;
        const int off = row * 128 + ((lq ^ (row & 7)) << 4);
        fb[i] = *reinterpret_cast<const uint4*>(sh + off);
        gb[i] = *reinterpret_cast<const uint4*>(sh + (off ^ 64));              // chunk lq + 4: slot index with bit 2 flipped
      }
#pragma unroll
      for (int i = 0; i < MT; ++i) fa[i] = *reinterpret_cast<const uint4*>(sa + swz(i * 16 + lrow, lq));
#pragma unroll
      for (int i = 0; i < MT; ++i) ga[i] = *reinterpret_cast<const uint4*>(sa + swz(i * 16 + lrow, lq + 4));
      }
      if constexpr (LOAD_ORDER == 1) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_sched_barrier(0); }
      uint4 gv = make_uint4(0, 0, 0, 0);                 // fused GroupNorm: this wave's piece tap-2 of the next halo, read back here
      const bool do_piece = fuse_gn && more && tap >= 2 && tap < NHP + 2 && w + 8 * (tap - 2) < npieces;   // wave-uniform
      uint4* const gp = reinterpret_cast<uint4*>(smem + h_wr_off + (size_t)(8 * (w + 8 * (do_piece ? tap - 2 : 0))) * 128 + lane * 16);
      if (do_piece) gv = *gp;
      int inflight = 0;                                  // pieces issued in this LOAD: they may still be flying when COMPUTE(t) ends
      if constexpr (DBG != 1) {
        if (t + 2 < nk) {                                   // A_g(t+2)
          PP_ISSUE_A_PIECES(0, A_IN_LOAD); inflight = AJ;
          // the ring bookkeeping for the next issue belongs HERE, in LOAD: at the end of COMPUTE its scalar instructions sat behind the
          // last MFMA on the side of the loop that sets the pace (tools/conv_stamps.py: COMPUTE 1040 vs LOAD 950 cycles per tile)
          if constexpr (A_IN_LOAD == AJ && DBG != 2) PP_ADVANCE_A();
        }
        if (more) {
          if constexpr (p_taps == 9) {
            if (tap < NHP) {
              // piece `tap` of the next halo: the lane offsets rotate through hvo[0] (NHP rotations = identity by the chunk's end)
              if (w + 8 * tap < npieces) { bdma16(hvo[0], rs_x, h_so, h_wr + tap * (64 * 128)); ++inflight; }
              const uint32_t h0 = hvo[0];
#pragma unroll
              for (int j = 0; j < NHP - 1; ++j) hvo[j] = hvo[j + 1];
              hvo[NHP - 1] = h0;
            }
          } else {
#pragma unroll
            for (int j = 0; j < NHP; ++j) bdma16(hvo[j], rs_x, h_so, h_wr + j * (64 * 128));
            inflight = 0;                                      // a 1x1 chunk is one tile: everything must land before the next LOAD
          }
        }
      }
      DTS_SEG_MARK(0)                                      // section 0: barrier exit -> all LDS-DMA pieces issued
      if constexpr (LOAD_ORDER == 2) {
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < NT; ++i) {
        const int row = hc[i] + dlt;
        const int off = row * 128 + ((lq ^ (row & 7)) << 4);
        fb[i] = *reinterpret_cast<const uint4*>(sh + off);
        gb[i] = *reinterpret_cast<const uint4*>(sh + (off ^ 64));              // chunk lq + 4: slot index with bit 2 flipped
      }
#pragma unroll
      for (int i = 0; i < MT; ++i) fa[i] = *reinterpret_cast<const uint4*>(sa + swz(i * 16 + lrow, lq));
      if constexpr (!GA_LATE) {
#pragma unroll
      for (int i = 0; i < MT; ++i) ga[i] = *reinterpret_cast<const uint4*>(sa + swz(i * 16 + lrow, lq + 4));
      }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          // every fragment is in registers: slots / buffers may be refilled
      uint4 fs[X3I ? MT : 1];
      if constexpr (X3I) {                                 // wh * 2^-11, made HERE in LOAD (vector issue slots in COMPUTE are lost matrix cycles)
#pragma unroll
        for (int i = 0; i < MT; ++i) fs[i] = f16x8_mul_2m11(fa[i]);
      }
      DTS_SEG_MARK(1)                                      // section 1: fragment reads issued and returned
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("s_barrier" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      DTS_SEG_MARK(2)                                      // section 2: wait in the barrier that ends LOAD
      // ---------------- COMPUTE(t)
      __builtin_amdgcn_s_setprio(1);
      if constexpr (DBG == 2) {
        if (t + 2 < nk) { PP_ISSUE_A_PIECES(A_IN_LOAD, AJ); PP_ADVANCE_A(); }
#pragma unroll
        for (int i = 0; i < MT; ++i) asm volatile("" ::"v"(fa[i].x), "v"(ga[i].w));
#pragma unroll
        for (int i = 0; i < NT; ++i) asm volatile("" ::"v"(fb[i].x), "v"(gb[i].w));
      } else {
        // piece tap-2 of the next halo (issued in LOAD(tap-2), retired by the counted wait that ended COMPUTE(tap-1); read back in
        // LOAD(tap) above) is normalised here: ~60 VALU instructions, 16 of them transcendental.  Branch-free per lane (a padding
        // row selects its old zeros).  MEASURED (profiles/r02_gn_fusion.txt): hipcc schedules them as a block after the MFMAs, the
        // matrix pipe idles meanwhile, and the conv loses more (+3.4 ms per step) than the separate apply pass cost (2.5 ms): the
        // fused path is correct (bit-identical, tested) but OFF by default (DTS_GN_FUSE=1 turns it on in networks.py).
        const bool a_more = DBG != 1 && t + 2 < nk;             // wave-uniform: the condition LOAD(t) issued its pieces under
#pragma unroll
        for (int i = 0; i < MT; ++i) {
#pragma unroll
          for (int j = 0; j < NT; ++j) Mma<T>::run(acc[i][j], fa[i], fb[j]);
          // GA_LATE: the weight fragments of the second K half are read HERE, one per four MFMAs, instead of in LOAD (6 of its 20 reads): the
          // A ring is private to the group and its slot is refilled by this wave's own later LOAD, so no other wave depends on when it is read
          if constexpr (GA_LATE) ga[i] = *reinterpret_cast<const uint4*>(sa + swz(i * 16 + lrow, lq + 4));
          if constexpr (A_IN_LOAD < AJ) {
            if (i == MT / 2 - 1) {
              __builtin_amdgcn_sched_barrier(0);
              if (a_more) PP_ISSUE_A_PIECES(A_IN_LOAD, A_IN_LOAD + 1);
              __builtin_amdgcn_sched_barrier(0);
            }
          }
        }
        if constexpr (fuse_gn) {
          if (do_piece) {
            const uint4 gr = gn_act8<T>(gv, gca, gcb, kp.gn_silu);
            const bool ok = (hval >> (tap - 2)) & 1u;
            *gp = make_uint4(ok ? gr.x : gv.x, ok ? gr.y : gv.y, ok ? gr.z : gv.z, ok ? gr.w : gv.w);
            // (forcing one MFMA : two VALU with sched_group_barrier made hipcc spill 342 VGPRs -- 835 evals/s; left to its scheduler)
          }
        }
        if constexpr (X3I) {
#pragma unroll
          for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j) Mma<T>::run(acc[i][j], fs[i], gb[j]);       // wh * 2^-11 . lo * 2^11
        }
#pragma unroll
        for (int i = 0; i < MT; ++i) {
#pragma unroll
          for (int j = 0; j < NT; ++j) Mma<T>::run(acc[i][j], ga[i], X3I ? fb[j] : gb[j]);      // (X3I: wl . hi)
          if constexpr (A_IN_LOAD + 1 < AJ) {
            if (i == MT / 2 - 1) {
              __builtin_amdgcn_sched_barrier(0);
              if (a_more) PP_ISSUE_A_PIECES(A_IN_LOAD + 1, AJ);
              __builtin_amdgcn_sched_barrier(0);
            }
          }
        }
        if constexpr (A_IN_LOAD != AJ) { if (a_more) PP_ADVANCE_A(); }
      }
      __builtin_amdgcn_s_setprio(0);
      if constexpr (fuse_gn) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");        // the normalised piece is in LDS
      DTS_SEG_MARK(3)                                      // section 3: the MFMAs (issue; the last ones still drain)
      __builtin_amdgcn_sched_barrier(0);
      // everything issued before LOAD(t) has landed -- A_g(t+1), the older halo pieces; only LOAD(t)'s own pieces may fly on.  ONE branch:
      // with weight pieces issued in LOAD(t) the wait leaves AJ loads outstanding -- its AJ weight pieces, or AJ - 1 of them plus the halo
      // piece, i.e. at most one piece of this segment (the first one issued, a whole segment old) is waited for early -- and the last two
      // tiles of a block drain everything.  [The exact count as a 5-way if-chain compiled to six scalar branches: ~150 cycles per tile
      // on the COMPUTE side, which paces the loop since GA_LATE (tools/conv_stamps.py).]
      (void)inflight;
      if (DBG != 1 && t + 2 < nk) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(AJ) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      DTS_SEG_MARK(4)                                      // section 4: counted vmcnt wait
      asm volatile("s_barrier" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      DTS_SEG_MARK(5)                                      // section 5: wait in the barrier that ends COMPUTE
      ++t;
      a_rd = (a_rd == grp * A_RING + 2 * A_HALF) ? grp * A_RING : a_rd + A_HALF;
    }
    ci0 = next_ci0;
    h_rd = (h_rd == H_OFF) ? H_OFF + H_BUF : H_OFF;
  }
  if (grp == 0) asm volatile("s_barrier" ::: "memory");          // matches group 1's last COMPUTE barrier
  DTS_SEG_STORE
#undef PP_SET_SRC
#undef PP_ISSUE_A
#undef PP_ISSUE_A_PIECES
#undef PP_ADVANCE_A
#undef PP_LOAD_COEF
#undef PP_GN_PIECE
  if constexpr (SK) {
    // ---- FOLDED 1x1 SKIP CONVOLUTION (split precision, splits == 1; dts_conv_args.skip_*): out = conv3x3(h) + conv1x1(block input), one
    // epilogue.  A second, plain K loop over the skip operand's 32-channel steps: the whole LDS is free now (the barrier above: every fragment of
    // the last 3x3 tile is in registers, every LDS-DMA piece has landed).  One barrier per step, all 8 waves in step, each keeps its accumulator
    // tiles.  The weight rows (BM x 128 bytes, L2-resident) are double-buffered and issued one step ahead; the tile's 256 pixel rows (32 KB per
    // step, every byte of them from HBM: this loop is the launch's memory phase) go through a ring of THREE slots, issued two steps ahead, behind
    // a counted wait that leaves the newest four pieces in flight.
    // The accumulators change units first: the two packed weights carry different powers of two (exact: a power-of-two ratio).
    static_assert(X3I, "skip fold: split precision only");
    const int sC = kp.sk_c, nks = sC / BKE, su = kp.sk_up & 1, Hs = H >> su, Ws = W >> su;
    const float ratio = kp.sk_ratio;
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j) acc[i][j] *= ratio;
    constexpr int S_A = BM * 128, S_B = BN * 128, B_OFF = 2 * S_A;
    static_assert(B_OFF + 3 * S_B <= H_OFF + 2 * H_BUF, "skip fold: the rings fit the kernel's LDS");
    // (readfirstlane: behind the loop hipcc keeps these uniform values in vector registers, which the LDS-DMA statement cannot take)
    dts_i32x4 rs_sw = make_rsrc(kp.sk_w, (uint32_t)((size_t)kp.cout * sC * ES));
    dts_i32x4 rs_sx = make_rsrc(kp.sk_x, (uint32_t)((size_t)kp.n * Hs * Ws * sC * ES));
#pragma unroll
    for (int e = 0; e < 4; ++e) { rs_sw[e] = __builtin_amdgcn_readfirstlane(rs_sw[e]); rs_sx[e] = __builtin_amdgcn_readfirstlane(rs_sx[e]); }
    uint32_t savo[AJ], sbvo[4];
#pragma unroll
    for (int j = 0; j < AJ; ++j) savo[j] = (uint32_t)((cm0 + 8 * (w + 8 * j) + r0) * sC) * ES + schunk;       // weight rows: piece w + 8j of BM / 8
#pragma unroll
    for (int j = 0; j < 4; ++j) {                                                                             // pixel rows: piece w + 8j of 32
      const int pl = 8 * (w + 8 * j) + r0, y = y0 + (pl >> 4), x = x0 + (pl & 15);
      sbvo[j] = (uint32_t)(((img0 * Hs + (y >> su)) * Ws + (x >> su)) * sC) * ES + schunk;
    }
    const uint32_t sa_dst = lds_base + (8 * w) * 128, sb_dst = lds_base + B_OFF + (8 * w) * 128;
#define SK_ISSUE_A(kt_, slot_)                                                                                 \
    {                                                                                                         \
      const uint32_t so_ = (uint32_t)(kt_) * 128u, d_ = sa_dst + (uint32_t)(slot_) * S_A;                      \
      _Pragma("unroll") for (int j = 0; j < AJ; ++j) bdma16(savo[j], rs_sw, so_, d_ + j * (64 * 128));         \
    }
#define SK_ISSUE_B(kt_, slot_)                                                                                 \
    {                                                                                                         \
      const uint32_t so_ = (uint32_t)(kt_) * 128u, d_ = sb_dst + (uint32_t)(slot_) * S_B;                      \
      _Pragma("unroll") for (int j = 0; j < 4; ++j) bdma16(sbvo[j], rs_sx, so_, d_ + j * (64 * 128));          \
    }
    SK_ISSUE_A(0, 0);
    SK_ISSUE_B(0, 0);
    if (nks > 1) SK_ISSUE_B(1, 1);
    int bs = 0;                                            // pixel ring slot of step kt (step kt + 2 goes into the slot step kt - 1 read)
#pragma unroll 1
    for (int kt = 0; kt < nks; ++kt) {
      // landed: the weight rows and pixel rows of this step; in flight at most the four pixel pieces of the next one (the newest issued)
      if (kt + 1 < nks) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      asm volatile("s_barrier" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      const int as = kt & 1;
      if (kt + 1 < nks) SK_ISSUE_A(kt + 1, as ^ 1);
      if (kt + 2 < nks) SK_ISSUE_B(kt + 2, bs == 0 ? 2 : bs - 1);
      const char* sa = smem + as * S_A + (grp * GM) * 128;
      const char* sb = smem + B_OFF + bs * S_B;
      bs = bs == 2 ? 0 : bs + 1;
      uint4 fa[MT], fb[NT], ga[MT], gb[NT], fs[MT];
      // all 20 fragment reads go out at once, in the order the three MFMA groups need them (wh . hi first); no full wait: the compiler's counted
      // lgkmcnt waits let the first group start when its operands are back
#pragma unroll
      for (int i = 0; i < NT; ++i) fb[i] = *reinterpret_cast<const uint4*>(sb + swz(wn * 64 + i * 16 + lrow, lq));
#pragma unroll
      for (int i = 0; i < MT; ++i) fa[i] = *reinterpret_cast<const uint4*>(sa + swz(i * 16 + lrow, lq));
#pragma unroll
      for (int i = 0; i < NT; ++i) gb[i] = *reinterpret_cast<const uint4*>(sb + (swz(wn * 64 + i * 16 + lrow, lq) ^ 64));
#pragma unroll
      for (int i = 0; i < MT; ++i) ga[i] = *reinterpret_cast<const uint4*>(sa + (swz(i * 16 + lrow, lq) ^ 64));
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < MT; ++i) fs[i] = f16x8_mul_2m11(fa[i]);
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) Mma<T>::run(acc[i][j], fa[i], fb[j]);
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) Mma<T>::run(acc[i][j], fs[i], gb[j]);
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) Mma<T>::run(acc[i][j], ga[i], fb[j]);
      __builtin_amdgcn_s_setprio(0);
      __builtin_amdgcn_sched_barrier(0);
    }
#undef SK_ISSUE_A
#undef SK_ISSUE_B
  }
  __syncthreads();
  DTS_STAMP(2);

  conv_epilogue<OT, MT, NT, BM, BN, NTHR, H_OFF + 2 * H_BUF>(kp, acc, cm0, tm, (int)blockIdx.y, wm, wn, lrow, lq, smem, bias_in_acc, 0, 0, 0);
  DTS_STAMP(3);
  DTS_STAMP_RT(5);
}

// split-K second pass: fixed-order sum of the f32 slabs + the conv epilogue.
// SPLITS is a template parameter so that ALL slab loads of an element are in flight at once: with a run-time trip count hipcc emits
// one load + wait per split, i.e. `splits` dependent L2 round trips per element -- 8 us for a 12 MB reduce that moves 3 us of bytes
// (tools/conv_stamps.py: the 8x8-level launches spent more time in this pass and its gaps than in their conv kernel).
template <typename T, int SPLITS>
__global__ __launch_bounds__(256) void conv_splitk_reduce_kernel(const ConvP kp) {
  const long long total = (long long)kp.P * (kp.cout / 4);
  const T* res = reinterpret_cast<const T*>(kp.residual);
  const T* bnc = reinterpret_cast<const T*>(kp.bias_nc);
  T* out = reinterpret_cast<T*>(kp.out);
  const int hw = kp.hout * kp.wout, c4 = kp.cout / 4;
  const size_t slab = (size_t)kp.P * kp.cout;
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
    const int pp = (int)(idx / c4), co = (int)(idx - (long long)pp * c4) * 4;
    const float* base = kp.partial + (size_t)pp * kp.cout + co;
    float4 q[SPLITS];
#pragma unroll
    for (int s = 0; s < SPLITS; ++s) q[s] = *reinterpret_cast<const float4*>(base + (size_t)s * slab);
    float v[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < SPLITS; ++s) { v[0] += q[s].x; v[1] += q[s].y; v[2] += q[s].z; v[3] += q[s].w; }      // fixed order
    for (int r = 0; r < 4; ++r) v[r] *= kp.acc_scale;                                                          // (1 outside the split-precision mode)
    if (kp.bias) {
      const float4 bv = *reinterpret_cast<const float4*>(kp.bias + co);
      v[0] += bv.x; v[1] += bv.y; v[2] += bv.z; v[3] += bv.w;
    }
    if (bnc) {
      float f[4];
      Vec4<T>::load(bnc + (size_t)(pp / hw) * kp.ld_bias_nc + co, f);
      for (int r = 0; r < 4; ++r) v[r] += f[r];
    }
    if (res) {
      float f[4];
      Vec4<T>::load(res + (size_t)pp * kp.cout + co, f);
      for (int r = 0; r < 4; ++r) v[r] += f[r];
    }
    for (int r = 0; r < 4; ++r) v[r] *= kp.out_scale;
    Vec4<T>::store(out + (size_t)pp * kp.cout + co, v);
  }
}

// split-K second pass that ALSO emits the GroupNorm strip statistics the fused epilogue would have written (per 64-pixel strip and
// channel: sum and sum of squares of the stored, i.e. rounded, outputs).  Without it every split-K layer followed by a GroupNorm
// costs two more passes (gn_partial + gn_coef): at the 8-candidates-per-GPU batch of a sharded search those were 2.6 ms of a
// 15.9 ms iteration.  Block = one strip x 64 channels, then a fixed-order LDS reduction over the sixteen pixel lanes
// (deterministic, no atomics).  Same arithmetic and order per element as conv_splitk_reduce_kernel.
template <typename T, int SPLITS>
__global__ __launch_bounds__(256) void conv_splitk_reduce_stats_kernel(const ConvP kp) {
  // thread = 4 channels x 4 pixels (pixel lane pl, pixels pl + 16 i of the strip): 16 quads x 16 B = one 256-byte segment per
  // pixel row, four independent load chains per thread over the splits
  __shared__ float red[16][16][8];
  const int strip = blockIdx.x, q = threadIdx.x & 15, pl = threadIdx.x >> 4;
  const int co = (blockIdx.y * 16 + q) * 4;
  const bool live = co < kp.cout;
  const T* res = reinterpret_cast<const T*>(kp.residual);
  const T* bnc = reinterpret_cast<const T*>(kp.bias_nc);
  T* out = reinterpret_cast<T*>(kp.out);
  const int hw = kp.hout * kp.wout;
  float ss[4] = {0.f, 0.f, 0.f, 0.f}, sq[4] = {0.f, 0.f, 0.f, 0.f};
  if (live) {
    const int p0 = strip * 64 + pl;
    float v[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { v[i][0] = v[i][1] = v[i][2] = v[i][3] = 0.f; }
    {                                                    // every slab load of the thread in flight at once (SPLITS x 4)
      const float* base = kp.partial + (size_t)p0 * kp.cout + co;
      const size_t slab = (size_t)kp.P * kp.cout;
      float4 t4[SPLITS][4];
#pragma unroll
      for (int s_ = 0; s_ < SPLITS; ++s_)
#pragma unroll
        for (int i = 0; i < 4; ++i) t4[s_][i] = *reinterpret_cast<const float4*>(base + (size_t)s_ * slab + (size_t)(16 * i) * kp.cout);
#pragma unroll
      for (int s_ = 0; s_ < SPLITS; ++s_)                // fixed order over the splits (as conv_splitk_reduce_kernel)
#pragma unroll
        for (int i = 0; i < 4; ++i) { v[i][0] += t4[s_][i].x; v[i][1] += t4[s_][i].y; v[i][2] += t4[s_][i].z; v[i][3] += t4[s_][i].w; }
    }
    float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
    if (kp.bias) bv = *reinterpret_cast<const float4*>(kp.bias + co);
    float nb[4] = {0.f, 0.f, 0.f, 0.f};
    if (bnc) Vec4<T>::load(bnc + (size_t)((strip * 64) / hw) * kp.ld_bias_nc + co, nb);    // a strip never straddles two samples
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int pp = p0 + 16 * i;
      for (int r = 0; r < 4; ++r) v[i][r] *= kp.acc_scale;
      v[i][0] += bv.x; v[i][1] += bv.y; v[i][2] += bv.z; v[i][3] += bv.w;
      if (bnc) { for (int r = 0; r < 4; ++r) v[i][r] += nb[r]; }
      if (res) {
        float f[4];
        Vec4<T>::load(res + (size_t)pp * kp.cout + co, f);
        for (int r = 0; r < 4; ++r) v[i][r] += f[r];
      }
      for (int r = 0; r < 4; ++r) v[i][r] *= kp.out_scale;
      const typename Vec4<T>::type pk = Vec4<T>::pack(v[i]);
      *reinterpret_cast<typename Vec4<T>::type*>(out + (size_t)pp * kp.cout + co) = pk;
      float f[4];
      Vec4<T>::unpack(pk, f);                            // moments of the values as stored
      for (int r = 0; r < 4; ++r) { ss[r] += f[r]; sq[r] += f[r] * f[r]; }
    }
  }
  for (int r = 0; r < 4; ++r) { red[pl][q][2 * r] = ss[r]; red[pl][q][2 * r + 1] = sq[r]; }
  __syncthreads();
  if (pl == 0 && live) {
    float o[8];
    for (int j = 0; j < 8; ++j) {
      float a_ = red[0][q][j];
      for (int k = 1; k < 16; ++k) a_ += red[k][q][j];   // fixed order: deterministic
      o[j] = a_;
    }
    float4* d = reinterpret_cast<float4*>(kp.stats + ((size_t)strip * kp.cout + co) * 2);
    d[0] = make_float4(o[0], o[1], o[2], o[3]);
    d[1] = make_float4(o[4], o[5], o[6], o[7]);
  }
}

constexpr int DTS_MAX_DEVICES = 64;
inline int current_device() {
  int d = 0;
  if (hipGetDevice(&d) != hipSuccess || d < 0 || d >= DTS_MAX_DEVICES) d = 0;
  return d;
}

// launches the split-K second pass (with the strip statistics when `stats_req` is set and the shape allows)
template <typename T, int SPLITS>
int launch_reduce_n(const ConvP& q, bool with_stats, hipStream_t st) {
  if (with_stats) {
    hipLaunchKernelGGL((conv_splitk_reduce_stats_kernel<T, SPLITS>), dim3(q.P / 64, (q.cout + 63) / 64), dim3(256), 0, st, q);
    DTS_CHECK_LAUNCH("dts_conv2d(split-K reduce + statistics)");
  } else {
    long long g = ((long long)q.P * (q.cout / 4) + 255) / 256;
    if (g > 2048) g = 2048;
    hipLaunchKernelGGL((conv_splitk_reduce_kernel<T, SPLITS>), dim3((int)g), dim3(256), 0, st, q);
    DTS_CHECK_LAUNCH("dts_conv2d(split-K reduce)");
  }
  return DTS_OK;
}
template <typename T>
int launch_reduce(const ConvP& q, bool with_stats, hipStream_t st) {
  switch (q.splits) {
    case 2: return launch_reduce_n<T, 2>(q, with_stats, st);
    case 3: return launch_reduce_n<T, 3>(q, with_stats, st);
    case 4: return launch_reduce_n<T, 4>(q, with_stats, st);
    case 5: return launch_reduce_n<T, 5>(q, with_stats, st);
    case 6: return launch_reduce_n<T, 6>(q, with_stats, st);
    case 7: return launch_reduce_n<T, 7>(q, with_stats, st);
    case 8: return launch_reduce_n<T, 8>(q, with_stats, st);
    default: dts_set_error("dts_conv2d: split-K factor %d outside 2..8", q.splits); return DTS_ERR_ARG;
  }
}

template <typename T, int MT, int NT, int WM, int WN, bool PF, int STAGES, typename OT = T>
int launch_conv_staged(const ConvP& q, int nblk, int splits, hipStream_t st, ConvCall& call) {
  constexpr int NW = WM * WN;
  constexpr int BM = 16 * MT * WM, BN = 16 * NT * WN;
  const size_t lds = (size_t)STAGES * (BM + BN) * 128;
  static bool attr_done[DTS_MAX_DEVICES] = {};            // the attribute is per DEVICE: a process that drives several GPUs sets it on each
  const int dev = current_device();
  if (!attr_done[dev]) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_igemm_kernel<T, MT, NT, WM, WN, PF, STAGES, OT>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_done[dev] = true;
  }
  if (call.ev_start != nullptr && call.ev_stop != nullptr)
    hipExtLaunchKernelGGL((conv_igemm_kernel<T, MT, NT, WM, WN, PF, STAGES, OT>), dim3(nblk, splits), dim3(64 * NW), lds, st, call.ev_start, call.ev_stop, 0, q);
  else
    hipLaunchKernelGGL((conv_igemm_kernel<T, MT, NT, WM, WN, PF, STAGES, OT>), dim3(nblk, splits), dim3(64 * NW), lds, st, q);
  DTS_CHECK_LAUNCH("dts_conv2d");
  return DTS_OK;
}

template <typename T, int MT, int NT, int WM, int WN, bool PF, typename OT = T>
int launch_conv(const ConvP& p, hipStream_t st, float* ws, long long ws_bytes, ConvCall& call) {
  constexpr int BM = 16 * MT * WM, BN = 16 * NT * WN;
  constexpr int BKE = 8 * ET<T>::EPV;
  ConvP q = p;
  q.n_ct = p.cout / BM;
  q.n_pt = (p.P + BN - 1) / BN;
  const int nblk = q.n_ct * q.n_pt;
  const int nk = p.taps * (p.cin / BKE);
  // split-K when the tile grid cannot fill the chip.  Two launch forms follow from the grid (see below): at most 256 blocks run as 8 waves
  // with a 3-deep ring, one block per CU; more run as 4 waves with two co-resident blocks per CU (512 slots).  Estimated microseconds per
  // form, from tools/conv_stamps.py and the split sweep of tools/conv_bench.py (conv_splits = 1..8 at 4, 8 and 16 rows,
  // profiles/r03_small_batch_experiments.txt item 9): ~5 us of prologue + epilogue; 0.75 us per K step for the 8-wave form, 1.25 us per step
  // and round of resident blocks for the 4-wave form; the second pass moves `splits` f32 slabs of P x cout (~3 us + slab bytes at ~4 MB/us),
  // which for many-pixel layers at a small batch costs more than the shorter K loop saves.
  const int slots = 512;                                     // 256 CUs x 2 resident blocks
  int splits = 1;
  if (ws != nullptr && nblk < (slots * 3) / 4 && nk >= 16 && !p.out_split2) {      // (the split-image output is written by the conv epilogue only)
    int smax = 8;
    if (smax > nk / 8) smax = nk / 8;
    while (smax > 1 && (long long)smax * p.P * p.cout * 4 > ws_bytes) --smax;
    if (smax < 1) smax = 1;
    const bool can8 = sizeof(T) == 2 && MT % 2 == 0;
    constexpr double KS = std::is_same<T, OT>::value ? 1.0 : 1.5;        // a split-precision K step carries 3 MFMAs per tile instead of 2
    auto est = [&](int s_) {
      const int steps = (nk + s_ - 1) / s_;
      double t;
      if (can8 && (long long)nblk * s_ <= 256) t = 5.0 + steps * 0.75 * KS;
      else t = (double)((nblk * s_ + slots - 1) / slots) * (5.5 + steps * 1.25 * KS);
      if (s_ > 1) t += 3.0 + (s_ + 0.5) * (double)p.P * p.cout * 4.0 / 4.0e6;
      return t;
    };
    int best = 1;
    for (int s_ = 2; s_ <= smax; ++s_)
      if (est(s_) < est(best)) best = s_;
    splits = best;
  }
  {
    const int forced = dts_knob_get(DTS_KNOB_CONV_SPLITS);    // DTS_CONV_SPLITS: tuning aid, as in the ping-pong launcher
    // (the reduce pass is instantiated for 2..8 slabs: a larger forced value is clamped here, as in the ping-pong launcher, instead of
    // launching the conv kernel and then failing in launch_reduce with the output never produced)
    const int f8 = (forced > 8 ? 8 : forced) * (p.out_split2 ? 0 : 1);
    if (f8 > 0 && ws != nullptr && nk >= 2 * f8 && (long long)f8 * p.P * p.cout * 4 <= ws_bytes) splits = f8;
  }
  q.ks_per_split = (nk + splits - 1) / splits;
  splits = (nk + q.ks_per_split - 1) / q.ks_per_split;       // no empty split
  q.splits = splits;
  q.partial = ws;
  // statistics come from the fused epilogue (whole launches, 64-pixel wave strips) or, under split-K, from the reduce pass
  float* const stats_req = p.stats;
  const bool stats_in_reduce = splits > 1 && stats_req != nullptr && (p.hout * p.wout) % 64 == 0 && p.cout % 4 == 0;
  if (splits > 1 || NT != 4) q.stats = nullptr;
  call.stats_written = q.stats != nullptr || stats_in_reduce;
  // ring depth: a grid of at most one block per CU cannot hide a tile's round trip behind a co-resident block, so it keeps two
  // tiles in flight (3 stages; DTS_CONV_STAGES = 2 | 3 | 4 forces a depth for A/B runs)
  int stages = ((long long)nblk * splits <= 256 && q.ks_per_split >= 3) ? 3 : 2;
  const int forced_stages = dts_knob_get(DTS_KNOB_CONV_STAGES);
  if (forced_stages >= 2 && forced_stages <= 4) stages = forced_stages;
  if (std::is_same<T, float>::value) stages = 2;            // parity mode: one configuration
  if ((size_t)stages * (BM + BN) * 128 > 160 * 1024) stages = 3;
  // 8 waves per block (the same tile, half the couts per wave) when the grid leaves one block per CU; DTS_CONV_WAVES = 4 | 8 forces it
  bool waves8 = sizeof(T) == 2 && MT % 2 == 0 && (long long)nblk * splits <= 256;
  const int forced_waves = dts_knob_get(DTS_KNOB_CONV_WAVES);
  if (forced_waves == 4) waves8 = false;
  if (forced_waves == 8) waves8 = sizeof(T) == 2 && MT % 2 == 0;
  int rc;
  if constexpr (sizeof(T) == 2 && MT % 2 == 0) {
    if (waves8) {
      if (stages >= 3) rc = launch_conv_staged<T, MT / 2, NT, WM * 2, WN, PF, 3, OT>(q, nblk, splits, st, call);
      else rc = launch_conv_staged<T, MT / 2, NT, WM * 2, WN, PF, 2, OT>(q, nblk, splits, st, call);
      if (rc != DTS_OK) return rc;
      if (splits > 1) {
        if (stats_in_reduce) q.stats = stats_req;
        return launch_reduce<OT>(q, stats_in_reduce, st);
      }
      return DTS_OK;
    }
  }
  // (the 4-deep ring exists for the A/B knob only, and only for OT = T)
  if (stages == 4 && std::is_same<T, OT>::value) rc = launch_conv_staged<T, MT, NT, WM, WN, PF, (sizeof(T) == 2 ? 4 : 2)>(q, nblk, splits, st, call);
  else if (stages >= 3) rc = launch_conv_staged<T, MT, NT, WM, WN, PF, (sizeof(T) == 2 ? 3 : 2), OT>(q, nblk, splits, st, call);
  else rc = launch_conv_staged<T, MT, NT, WM, WN, PF, 2, OT>(q, nblk, splits, st, call);
  if (rc != DTS_OK) return rc;
  if (splits > 1) {
    if (stats_in_reduce) q.stats = stats_req;
    return launch_reduce<OT>(q, stats_in_reduce, st);
  }
  return DTS_OK;
}

// ---- ping-pong launcher: one block per (192- or 128-cout tile, 256-pixel tile[, K split]); 512 threads, 154 KB (MT = 6) / 130 KB (MT = 4) of LDS
template <typename T, int TAPS, int DBG = 0, bool GN = false, int MT = 6, typename OT = T, bool SK = false>
int launch_conv_pp(const ConvP& p, hipStream_t st, float* ws, long long ws_bytes, ConvCall& call) {
  // 3x3 only: the kernel's TAPS == 1 form has a landing race with more than one channel chunk (see conv_pick_pp) and must not be launched
  static_assert(TAPS == 9, "conv_pp_kernel: only the 3x3 form is safe to launch");
  if constexpr (sizeof(T) != 2) {
    return DTS_ERR_UNSUPPORTED;
  } else {
    constexpr int BM = 32 * MT, BN = 256, BKE = 64;
    ConvP q = p;
    q.n_ct = p.cout / BM;
    q.n_pt = (p.P + BN - 1) / BN;
    const int nblk = q.n_ct * q.n_pt;
    const int nk = p.taps * (p.cin / BKE);
    int splits = 1;
    const int forced = dts_knob_get(DTS_KNOB_CONV_SPLITS);
    if (ws != nullptr && nk >= 16 && !p.out_split2 && !SK) {      // (SK: the second K loop needs the complete accumulators; dts_conv_folds_skip only admits grids that would not split)
      if (forced > 0) splits = forced;
      else if (nblk < 192) {
        // One resident block per CU.  Splitting K fills idle CUs but pays a second pass over `splits` f32 slabs of P x cout, which for
        // the many-pixel layers of a SMALL batch costs more than it saves (64x64 level at 8 rows: 2 x 25 MB of slabs for a 10 us
        // shorter K loop -- tools/conv_stamps.py).  Estimated microseconds, from the in-kernel stamps: ~5 us of prologue + epilogue,
        // ~1.05 us per K tile (two ~1050-cycle segments), reduce = ~3 us + slab bytes at ~4 MB/us.
        const int nchunk = nk / TAPS;
        double best = 1e30;
        for (int s_ = 1; s_ <= 8 && s_ <= nchunk; ++s_) {
          const int cps = (nchunk + s_ - 1) / s_, se = (nchunk + cps - 1) / cps;
          if (se != s_ || nblk * se > 256) continue;
          double t = 5.0 + cps * TAPS * (std::is_same<T, OT>::value ? 1.05 : 1.4);      // (split precision: 72 MFMAs per tile and wave instead of 48)
          if (se > 1) t += 3.0 + (se + 0.5) * (double)p.P * p.cout * 4.0 / 4.0e6;
          if (t < best) { best = t; splits = se; }
        }
      }
      if (splits > 8) splits = 8;
      if (forced > 0 && splits > nk / 8) splits = nk / 8;
      if (splits > nk / TAPS) splits = nk / TAPS;
      while (splits > 1 && (long long)splits * p.P * p.cout * 4 > ws_bytes) --splits;
      if (splits < 1) splits = 1;
    }
    {   // every split a whole number of channel chunks (TAPS tiles each)
      const int nchunk = nk / TAPS;
      const int cps = (nchunk + splits - 1) / splits;
      q.ks_per_split = cps * TAPS;
      splits = (nchunk + cps - 1) / cps;
    }
    q.splits = splits;
    q.partial = ws;
    float* const stats_req = p.stats;
    const bool stats_in_reduce = splits > 1 && stats_req != nullptr && (p.hout * p.wout) % 64 == 0 && p.cout % 4 == 0;
    if (splits > 1) q.stats = nullptr;
    call.stats_written = q.stats != nullptr || stats_in_reduce;
    constexpr size_t lds = (size_t)(6 * 16 * MT + 2 * 328) * 128;      // A: 2 groups x 3 slots x 16*MT rows; halo: 2 buffers x 328 rows
    static bool attr_done[DTS_MAX_DEVICES] = {};
    const int dev = current_device();
    if (!attr_done[dev]) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_pp_kernel<T, TAPS, DBG, GN, MT, OT, SK>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      attr_done[dev] = true;
    }
    if (call.ev_start != nullptr && call.ev_stop != nullptr)
      hipExtLaunchKernelGGL((conv_pp_kernel<T, TAPS, DBG, GN, MT, OT, SK>), dim3(nblk, splits), dim3(512), lds, st, call.ev_start, call.ev_stop, 0, q);
    else
      hipLaunchKernelGGL((conv_pp_kernel<T, TAPS, DBG, GN, MT, OT, SK>), dim3(nblk, splits), dim3(512), lds, st, q);
    DTS_CHECK_LAUNCH("dts_conv2d(ping-pong)");
    if (splits > 1) {
      if (stats_in_reduce) q.stats = stats_req;
      return launch_reduce<OT>(q, stats_in_reduce, st);
    }
    return DTS_OK;
  }
}

template <typename T, bool PF, typename OT = T>
int conv_dispatch_tile(const ConvP& p, int tile, hipStream_t st, float* ws, long long ws_bytes, ConvCall& call) {
  switch (tile) {
    case 192: return launch_conv<T, 6, 4, 2, 2, PF, OT>(p, st, ws, ws_bytes, call);
    case 128: return launch_conv<T, 4, 4, 2, 2, PF, OT>(p, st, ws, ws_bytes, call);
    default: return launch_conv<T, 4, 4, 1, 4, PF, OT>(p, st, ws, ws_bytes, call);
  }
}

// the shapes the ping-pong / halo kernel takes (and with it the fused GroupNorm apply)
bool conv_pp_eligible(int dtype, int ksize, int cout, int c1, int c2, int n, int hin, int win, int up) {
  const int ho = up ? 2 * hin : hin, wo = up ? 2 * win : win;      // the tile geometry lives in OUTPUT coordinates
  const long long P = (long long)n * ho * wo;
  return dtype != DTS_F32 && cout % 192 == 0 && ho == wo && wo >= 16 && (wo & (wo - 1)) == 0 && P % 256 == 0 && P < (1ll << 30) &&
         (long long)n * hin * win * (c1 > c2 ? c1 : c2) * 2 < (1ll << 31) && (long long)cout * ksize * ksize * (c1 + c2) * 2 < (1ll << 31) &&
         (c1 % 64 == 0) && (c2 % 64 == 0);
}

// Which kernel a launch takes: 0 = conv_igemm_kernel (4 waves), 6 / 4 = conv_pp_kernel with 192- / 128-cout blocks.  One rule for
// conv_dispatch and for dts_conv_kernel (what bench.py attributes its per-launch times to).
// Measured (tools/conv_bench.py --variants, profiles/r02_conv_variants.txt): 3x3 layers +3..11 %, 1x1 layers -10..15 % (their 6-12 K
// tiles do not amortise the exposed prologue/epilogue of a one-block-per-CU kernel): 3x3 only.  With a residual input the epilogue's
// residual fetch is exposed in the ping-pong kernel (no early fetch): short-K layers (cin < 384) then lose 2-3 % to the 4-wave kernel.
int conv_pick_pp(bool f32, const ConvP& p, bool x3 = false) {
  const int variant = dts_knob_get(DTS_KNOB_CONV_VARIANT);      // 1 = ping-pong wherever it applies, 0 = never, unset = by shape
  if (f32 || variant == 0 || dts_knob_get(DTS_KNOB_CONV_TILE) > 0) return 0;
  // 3x3 only.  The kernel's 1x1 form (TAPS = 1) was 5-20 % slower than the 4-wave kernel on every 1x1 layer, and with more than one
  // channel chunk it had a landing race: a 1x1 chunk is ONE tile, so the other group's pieces of the next halo are issued one segment
  // before they are read, but only waited for at the end of that group's COMPUTE, a segment later (3x3 issues them >= 3 segments
  // ahead).  tools/pp1x1_bisect.py showed it (first launch wrong, later ones often right); the 1x1 dispatch is removed.
  if (p.taps != 9) return 0;
  const int mt = p.cout % 192 == 0 ? 6 : (p.cout % 128 == 0 ? 4 : 0);
  // square power-of-two images of 16..., whole 256-pixel tiles, 32-bit lane offsets, no fused upsample (cout passed as 192: the
  // divisibility is handled here)
  if (mt == 0 || !conv_pp_eligible(DTS_BF16, p.taps == 9 ? 3 : 1, 192, p.c1, p.c2, p.n, p.hin, p.win, p.up)) return 0;
  if ((long long)p.cout * p.taps * p.cin * 2 >= (1ll << 31)) return 0;
  if (p.gn_coef != nullptr) return (mt == 6 && !p.up) ? 6 : 0;   // dts_conv2d has checked dts_conv_fuses_gn
  if (variant >= 1) return mt;                                   // forced (1) or a timing-only diagnostic build (11/21/41/51)
  const long long blocks_pp = (long long)(p.cout / (32 * mt)) * ((p.P + 255) / 256);
  // a short-K layer whose ping-pong grid fills at most half the chip (64x64 level, 192 -> 192, at the 8 rows per forward of a sharded
  // search: 128 blocks of 27 tiles) is faster on the 8-wave implicit-GEMM kernel, whose 192 x 128 tiles give twice the blocks and need no
  // K split: 32 vs 41 us (profiles/r03_conv_variants.txt)
  const bool short_small = blocks_pp <= 128 && p.taps * (p.cin / 64) <= 27;
  // a HALF round of the 128-cout form at 32x32 (DDPM++ CIFAR-32, 256 -> 256 at the 16 rows of BASELINE config 2: 128 blocks on 256 CUs)
  // runs 38 us against 31 on the implicit-GEMM kernel, whose 128 x 128 tiles give twice the blocks: -4 % on that workload's step.  Keyed on
  // the image size: the classifier's 16x16 layers are the same GEMM at 64 rows, but routed the same way they cost the ADM headline 0.5 %
  // (their neighbours keep the operands of the ping-pong form warm; profiles/r03_small_batch_experiments.txt item 9c).
  // NOT in the split-precision mode: with three MFMAs per staged K step the ping-pong form's half round (+ its K split) beats the
  // 128 x 128 implicit-GEMM grid there -- DDPM++-32 rejection step 9.07 -> 8.82 ms, same box, alternating (profiles/r05_experiments.txt item 16).
  const int hr_knob = dts_knob_get(DTS_KNOB_CONV_HALF_ROUND);     // unset: by mode; 1 / 0 force the rule on / off (A/B aid)
  const bool half_round_128 = mt == 4 && blocks_pp > 64 && blocks_pp <= 128 && p.wout >= 32 && p.cout <= 256 && (x3 ? hr_knob == 1 : hr_knob != 0);
  const bool auto_pp = p.taps == 9 && !short_small && !half_round_128;     // (round 2 excluded residual layers with cin < 384: since the COMPUTE-side trims of round 3 the ping-pong kernel wins there too, profiles/r03_conv_variants.txt)
  return (auto_pp && blocks_pp >= 64) ? mt : 0;
}

template <typename T, typename OT = T>
int conv_dispatch(const ConvP& p, hipStream_t st, float* ws, long long ws_bytes, ConvCall& call) {
  const int g_tile_override = dts_knob_get(DTS_KNOB_CONV_TILE);     // DTS_CONV_TILE=64|128|192 (tuning aid; only honoured when it divides cout)
  int tile = (p.cout % 192 == 0) ? 192 : (p.cout % 128 == 0 ? 128 : 64);    // measured: tools/conv_bench.py
  // f32 (parity mode): the 192-cout tile spilled until round 4 (its epilogue held 96 residual registers beside 96 accumulators; the residual
  // slices now come two at a time: 204-220 VGPRs, no scratch).  DTS_F32_TILE192=0 restores the smaller tile (A/B aid).
  static const bool f32_192 = !(getenv("DTS_F32_TILE192") && atoi(getenv("DTS_F32_TILE192")) == 0);
  if (std::is_same<T, float>::value && tile == 192 && !f32_192) tile = (p.cout % 128 == 0) ? 128 : 64;
  const int pp = conv_pick_pp(std::is_same<T, float>::value, p, !std::is_same<T, OT>::value);
  if constexpr (!std::is_same<T, OT>::value) {       // split-precision mode: the two shipped ping-pong forms and the implicit-GEMM forms, no knob variants
    if (p.sk_w != nullptr) {                         // dts_conv2d has checked dts_conv_folds_skip
      if (pp == 6) return launch_conv_pp<T, 9, 0, false, 6, OT, true>(p, st, ws, ws_bytes, call);
      if (pp == 4) return launch_conv_pp<T, 9, 0, false, 4, OT, true>(p, st, ws, ws_bytes, call);
      return DTS_ERR_UNSUPPORTED;
    }
    if (pp == 6) return launch_conv_pp<T, 9, 0, false, 6, OT>(p, st, ws, ws_bytes, call);
    if (pp == 4) return launch_conv_pp<T, 9, 0, false, 4, OT>(p, st, ws, ws_bytes, call);
    if (g_tile_override > 0 && p.cout % g_tile_override == 0) tile = g_tile_override;
    if (p.taps == 9) return conv_dispatch_tile<T, true, OT>(p, tile, st, ws, ws_bytes, call);
    return conv_dispatch_tile<T, false, OT>(p, tile, st, ws, ws_bytes, call);
  }
  if (pp == 6) {
    const int variant = dts_knob_get(DTS_KNOB_CONV_VARIANT);
#define DTS_PP(DBG_) launch_conv_pp<T, 9, DBG_>(p, st, ws, ws_bytes, call)
    if (p.gn_coef != nullptr) return launch_conv_pp<T, 9, 0, true>(p, st, ws, ws_bytes, call);
#ifdef DTS_DIAG_KERNELS      // diagnostic builds only (tools/conv_stamps.py --diag): 11/21/41/51 are TIMING-ONLY (outputs wrong by construction)
    if (variant == 11) return DTS_PP(1);
    if (variant == 21) return DTS_PP(2);
    if (variant == 41) return DTS_PP(4);
    if (variant == 51) return DTS_PP(5);
    if (variant == 61) return DTS_PP(6);     // LOAD-order A/B builds (correct outputs)
    if (variant == 71) return DTS_PP(7);
    if (variant == 81) return DTS_PP(8);     // weight pieces split between LOAD and COMPUTE (measured slower)
#else
    if (variant == 91) return DTS_PP(9);     // EXPERIMENT: second-half weight fragments read inside COMPUTE
    if (variant > 1) {
      dts_set_error("dts_conv2d: DTS_CONV_VARIANT=%d selects a diagnostic kernel that this library was built without (-DDTS_DIAG_KERNELS)", variant);
      return DTS_ERR_ARG;
    }
#endif
    return DTS_PP(0);
#undef DTS_PP
  }
  if (pp == 4)     // 128-cout blocks: the widths that are no multiple of 192 (classifier 128/256/512, SD VAE 128/256/512)
    return launch_conv_pp<T, 9, 0, false, 4>(p, st, ws, ws_bytes, call);
  if (g_tile_override > 0 && p.cout % g_tile_override == 0) tile = g_tile_override;
  // fragment prefetch pays on the long K loops of the 3x3 layers; f32 (parity mode) keeps the lean order: its 192-cout
  // tile is already at the 256-VGPR limit
  if (p.taps == 9 && !std::is_same<T, float>::value) return conv_dispatch_tile<T, true>(p, tile, st, ws, ws_bytes, call);
  return conv_dispatch_tile<T, false>(p, tile, st, ws, ws_bytes, call);
}

// the shape part of ConvP from the C arguments (what the kernel choice depends on)
void conv_shape_from_args(const dts_conv_args* a, ConvP& p) {
  p.c1 = a->c1; p.c2 = a->c2; p.cin = a->c1 + a->c2;
  p.n = a->n; p.hin = a->hin; p.win = a->win;
  p.hout = a->up ? 2 * a->hin : a->hin; p.wout = a->up ? 2 * a->win : a->win;
  p.cout = a->cout; p.taps = a->ksize * a->ksize; p.up = a->up;
  p.P = (int)((long long)p.n * p.hout * p.wout);
  p.residual = (const char*)a->residual; p.gn_coef = a->gn_coef;
}

}  // namespace

#ifdef DTS_STAMPS
extern "C" int dts_debug_read_stamps(unsigned long long* host, int count) {
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_stamps), sizeof(unsigned long long) * (size_t)count);
}
extern "C" int dts_debug_clear_stamps(void) {
  void* p = nullptr;
  if (hipGetSymbolAddress(&p, HIP_SYMBOL(g_stamps)) != hipSuccess) return -1;
  if (hipMemset(p, 0, sizeof(g_stamps)) != hipSuccess) return -1;
  if (hipGetSymbolAddress(&p, HIP_SYMBOL(g_seg)) != hipSuccess) return -1;
  return (int)hipMemset(p, 0, sizeof(g_seg));
}
extern "C" int dts_debug_read_segments(unsigned long long* host, int count) {
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_seg), sizeof(unsigned long long) * (size_t)count);
}
#endif

extern "C" int dts_conv_kernel(const dts_conv_args* a) {
  if (a == nullptr || (a->ksize != 1 && a->ksize != 3) || a->n <= 0 || a->hin <= 0 || a->win <= 0 || a->cout <= 0) return -1;
  if ((long long)a->n * a->hin * a->win * (a->up ? 4 : 1) >= (1ll << 30)) return -1;
  ConvP p;
  conv_shape_from_args(a, p);
  return conv_pick_pp(a->dtype == DTS_F32, p, a->dtype == DTS_F16X3);
}

extern "C" int dts_conv_fuses_gn(const dts_conv_args* a) {
  if (a == nullptr || a->ksize != 3) return 0;
  const int variant = dts_knob_get(DTS_KNOB_CONV_VARIANT);
  if (variant == 0 || dts_knob_get(DTS_KNOB_CONV_TILE) > 0) return 0;
  if (a->up || !conv_pp_eligible(a->dtype, a->ksize, a->cout, a->c1, a->c2, a->n, a->hin, a->win, a->up)) return 0;
  const long long blocks_pp = (long long)(a->cout / 192) * (((long long)a->n * a->hin * a->win + 255) / 256);
  return (variant == 1 || blocks_pp >= 64) ? 1 : 0;              // with gn_coef the kernel choice does not depend on the residual
}

extern "C" int dts_conv_folds_skip(const dts_conv_args* a) {
  if (a == nullptr || a->dtype != DTS_F16X3 || a->ksize != 3 || a->up || a->residual != nullptr || a->out_split2 || a->gn_coef != nullptr) return 0;
  if (a->skip_c <= 0 || a->skip_c % 64 != 0 || a->n <= 0 || a->hin <= 0 || a->win <= 0 || a->cout <= 0) return 0;
  if (dts_knob_get(DTS_KNOB_CONV_SKIP_FOLD) == 0 || dts_knob_get(DTS_KNOB_CONV_SPLITS) > 0) return 0;
  if ((long long)a->n * a->hin * a->win >= (1ll << 30)) return 0;
  if (a->skip_up && ((a->hin | a->win) & 1)) return 0;
  ConvP p;
  conv_shape_from_args(a, p);
  const int mt = conv_pick_pp(false, p, true);
  if (mt == 0) return 0;
  const long long nblk = (long long)(p.cout / (32 * mt)) * ((p.P + 255) / 256);
  // a grid below 192 blocks would take a K split (launch_conv_pp), which the fold excludes: there the separate 1x1 launch stays.  (Forced
  // ping-pong, DTS_CONV_VARIANT=1: any grid folds, unsplit -- the tests' small shapes.)
  if (nblk < 192 && dts_knob_get(DTS_KNOB_CONV_VARIANT) != 1) return 0;
  const long long px = (long long)a->n * (a->hin >> (a->skip_up ? 1 : 0)) * (a->win >> (a->skip_up ? 1 : 0));
  return (px * a->skip_c * 2 < (1ll << 31) && (long long)a->cout * a->skip_c * 2 < (1ll << 31)) ? 1 : 0;
}

extern "C" int dts_conv2d(dts_conv_args* a, dts_stream s) {
  DTS_CHECK_ARG(a != nullptr, "dts_conv2d: null args");
  DTS_CHECK_ARG(a->x1 && a->w && a->out, "dts_conv2d: null tensor");
  DTS_CHECK_ARG(a->ksize == 1 || a->ksize == 3, "dts_conv2d: ksize %d", a->ksize);
  DTS_CHECK_ARG(a->n > 0 && a->hin > 0 && a->win > 0, "dts_conv2d: bad shape");
  DTS_CHECK_ARG(a->cout > 0 && a->cout % 64 == 0, "dts_conv2d: cout %d must be a multiple of 64", a->cout);
  DTS_CHECK_ARG(a->c2 == 0 || a->x2 != nullptr, "dts_conv2d: c2 without x2");
  const int bke = (a->dtype == DTS_F32) ? 32 : 64;
  DTS_CHECK_ARG(a->c1 > 0 && a->c1 % bke == 0 && a->c2 % bke == 0, "dts_conv2d: channels (%d,%d) must be multiples of %d",
                a->c1, a->c2, bke);
  DTS_CHECK_ARG(a->bias_nc == nullptr || a->ld_bias_nc >= a->cout, "dts_conv2d: ld_bias_nc");
  // out-of-image rows are read from g_zero16 with a pointer that advances along K like a real row: it must stay inside the page
  DTS_CHECK_ARG((long long)(a->c1 + a->c2) * (a->dtype == DTS_F32 ? 4 : 2) + 128 <= (long long)sizeof(g_zero16),
                "dts_conv2d: %d input channels exceed the zero-row page (%d bytes)", a->c1 + a->c2, (int)sizeof(g_zero16));
  ConvP p;
  p.x1 = (const char*)a->x1; p.x2 = (const char*)a->x2; p.w = (const char*)a->w;
  p.bias = a->bias; p.bias_nc = (const char*)a->bias_nc; p.residual = (const char*)a->residual; p.out = (char*)a->out;
  p.c1 = a->c1; p.c2 = a->c2; p.cin = a->c1 + a->c2; p.ld_bias_nc = a->ld_bias_nc;
  p.n = a->n; p.hin = a->hin; p.win = a->win;
  p.hout = a->up ? 2 * a->hin : a->hin; p.wout = a->up ? 2 * a->win : a->win;
  DTS_CHECK_ARG(p.hout < 32768 && p.wout < 32768, "dts_conv2d: spatial size too large");
  p.cout = a->cout; p.taps = a->ksize * a->ksize; p.up = a->up;
  const long long P = (long long)p.n * p.hout * p.wout;
  DTS_CHECK_ARG(P < (1ll << 30), "dts_conv2d: too many pixels");
  p.P = (int)P; p.out_scale = a->out_scale; p.acc_scale = a->acc_scale == 0.f ? 1.f : a->acc_scale; p.n_ct = p.n_pt = 0;
  p.out_split2 = a->out_split2;
  DTS_CHECK_ARG(!a->out_split2 || (a->dtype == DTS_F16X3 && a->stats_out == nullptr), "dts_conv2d: out_split2 is the split-precision mode's, without strip statistics");
  p.splits = 1; p.ks_per_split = 0; p.partial = nullptr;
  p.w_shift = p.hw_shift = -1;
  {
    const int hw = p.hout * p.wout;
    if ((p.wout & (p.wout - 1)) == 0 && (hw & (hw - 1)) == 0) { p.w_shift = __builtin_ctz(p.wout); p.hw_shift = __builtin_ctz(hw); }
  }
  p.gn_coef = a->gn_coef; p.gn_silu = a->gn_silu;
  {   // the row-layout f32 epilogue: the split-precision mode's default; the f32 parity mode keeps the accumulator-layout epilogue (the mode is the
      // REFERENCE the others are compared with: its summation orders stay what rounds 2-4 validated against the CPU oracle) unless DTS_CONV_EPI32=1
    const int k_ = dts_knob_get(DTS_KNOB_CONV_EPI32);
    p.epi_rows = k_ < 0 ? (a->dtype == DTS_F16X3 ? 1 : 0) : (k_ == 2 ? 2 : (k_ != 0));
  }
  DTS_CHECK_ARG(a->gn_coef == nullptr || dts_conv_fuses_gn(a), "dts_conv2d: gn_coef given for a launch that cannot fuse the GroupNorm apply "
                "(ask dts_conv_fuses_gn first)");
  p.sk_x = p.sk_w = nullptr; p.sk_c = p.sk_up = 0; p.sk_ratio = 1.f;
  if (a->skip_c != 0 || a->skip_w != nullptr || a->skip_x != nullptr) {
    DTS_CHECK_ARG(a->skip_w != nullptr && a->skip_x != nullptr && dts_conv_folds_skip(a), "dts_conv2d: skip_* given for a launch that cannot fold the 1x1 "
                  "skip convolution (ask dts_conv_folds_skip first)");
    const float ss = a->skip_acc_scale == 0.f ? 1.f : a->skip_acc_scale;
    p.sk_x = (const char*)a->skip_x; p.sk_w = (const char*)a->skip_w; p.sk_c = a->skip_c; p.sk_up = a->skip_up ? 1 : 0;
    p.sk_ratio = p.acc_scale / ss; p.acc_scale = ss;
  }
  p.stats = a->stats_out;
  DTS_CHECK_ARG(a->stats_out == nullptr || ((p.hout * p.wout) % 64 == 0 && (uintptr_t)a->stats_out % 16 == 0),
                "dts_conv2d: strip statistics need hout*wout to be a multiple of 64 and a 16-byte aligned buffer");
  DTS_CHECK_ARG(a->workspace == nullptr || ((uintptr_t)a->workspace % 16 == 0 && a->workspace_bytes >= 0), "dts_conv2d: workspace");
  hipStream_t st = to_stream(s);
  int rc = DTS_OK;
  ConvCall call;
  call.ev_start = (hipEvent_t)a->ev_start; call.ev_stop = (hipEvent_t)a->ev_stop;
  DTS_CHECK_ARG(p.acc_scale == 1.f || a->dtype == DTS_F16X3, "dts_conv2d: acc_scale is the split-precision mode's (DTS_F16X3)");
  if (a->dtype == DTS_F16X3) {
    // split precision: f16 operand planes on the 16-bit MFMA, f32 accumulate, f32 epilogue operands and output
    DTS_CHECK_ARG(a->gn_coef == nullptr, "dts_conv2d: gn_coef is not available in the split-precision mode");
    rc = conv_dispatch<f16_t, float>(p, st, (float*)a->workspace, (long long)a->workspace_bytes, call);
  } else {
    DTS_DISPATCH_DTYPE(a->dtype, rc = conv_dispatch<T>(p, st, (float*)a->workspace, (long long)a->workspace_bytes, call));
  }
  a->stats_written = call.stats_written ? 1 : 0;
  return rc;
}
