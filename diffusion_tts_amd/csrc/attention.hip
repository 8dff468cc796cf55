// K6: fused self-attention (flash-style, online softmax in f32) on MFMA for gfx950.
//
// Replaces edm/training/networks.py:182-184 (qkv split, AttentionOp = softmax(q^T k / sqrt(d)) in f32, einsum with v)
// and edm/unet.py:355-372 / :388-407 (QKVAttentionLegacy / QKVAttention).  The projections' output channels were
// regrouped at weight-pack time so qkv is [n][t][ q(heads*d) | k(heads*d) | v(heads*d) ].
//
// One block = 4 waves = 64 queries of one (sample, head); K/V tiles of 64 keys staged in LDS (row stride d*es+32 B:
// conflict-free for both the ds_read_b128 row reads of K and the ds_read_b64_tr_b16 transposed reads of V).
// Per wave (16 queries), with the query on the MFMA column (lane & 15):
//   S^T[key][q] = K . Q^T        A = K rows (LDS), B = Q rows (registers, loaded once)
//   online softmax over keys     lane-local over its 16 accumulators + 2 cross-lane maxima/sums
//   O^T[d][q]  += V^T . P^T      A = V^T via transposed LDS reads, B = P straight from the S^T accumulators
// so P never leaves registers and each lane ends up with 4 consecutive channels of its query (vector store).
#include "dts_common.h"
#include <mutex>
#include <unordered_map>

namespace {

template <typename T> struct AttMma;
template <> struct AttMma<bf16_t> {
  static __device__ __forceinline__ f32x4_t run(const uint4& a, const uint4& b, f32x4_t c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
  }
  static __device__ __forceinline__ uint32_t pack2(float lo, float hi) { return pack2_bf16(lo, hi); }
  static constexpr uint32_t ONES2 = 0x3F803F80u;            // two 1.0
};
template <> struct AttMma<f16_t> {
  static __device__ __forceinline__ f32x4_t run(const uint4& a, const uint4& b, f32x4_t c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0);
  }
  static __device__ __forceinline__ uint32_t pack2(float lo, float hi) { return pack2_f16(lo, hi); }
  static constexpr uint32_t ONES2 = 0x3C003C00u;
};

__device__ __forceinline__ float vmax3(float a, float b, float c) {
  float r;
  asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}
__device__ __forceinline__ float vmax2(float a, float b) {
  float r;
  asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}

__device__ uint4 g_att_zero[64];      // 1 KiB of zeros: LDS-DMA source of key rows past the end of a ragged sequence

// LDS-DMA (global -> LDS, no registers), 16 bytes per active lane; lds_off = wave-uniform LDS byte address of lane 0's slot.  Inline asm:
// the compiler's s_waitcnt bookkeeping does not see it, completion is awaited with the counted vmcnt below (conv_igemm.hip has the same).
__device__ __forceinline__ void att_glds16(const char* g, uint32_t lds_off) {
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" : : "v"(g), "s"(lds_off) : "memory", "m0");
}

struct AttP {
  const char* qkv; char* out;
  int n, t, heads, d;
  float scale_log2e;
  int qblocks;                   // query blocks per (sample, head)
  int xcd_remap;                 // 1: XCD-aware block order (default); 0: plain order (A/B aid, DTS_ATT_XCD=0)
  int out_split3;                // attention_x3_kernel: the output leaves as the split-precision conv operand image, per 32 channels hi | lo * 2^11 (f16 [n][t][2C])
};

// Block order.  The grid is 1-D over (sample*head, query block), (sample, head)-major.  Hardware deals consecutive block ids
// round-robin over the 8 XCDs, each with a private 4 MB L2: in plain order the query blocks of one (sample, head) land on 8
// different XCDs and every L2 fetches that head's K and V again (rocprofv3: 472 MB per launch against ~200 MB algorithmic at
// T=1024, the kernel ran fabric-bound).  Remapped, the ids b, b+8, b+16.. that share an XCD walk a CONTIGUOUS range of the
// (sample, head)-major order, so all query blocks of a head run back to back on one XCD and K/V are fetched once.
// Bijective for any grid size (conv_igemm.hip uses the same map); placement only changes speed, never results.
__device__ __forceinline__ void att_block(const AttP& p, int& nh, int& qb) {
  int bid = blockIdx.x;
  if (p.xcd_remap) {
    const int nblk = gridDim.x, q = nblk >> 3, r = nblk & 7, xcd = bid & 7, j = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
  }
  nh = bid / p.qblocks;
  qb = bid - nh * p.qblocks;
}

// ------------------------------------------------------------------------------------------------
// 16-bit element types
// QT = query tiles (of 16) per wave: a block covers 64*QT queries.  QT = 2 halves the K/V staging, the barriers and the K / V^T
// fragment reads per query (each fragment feeds both query tiles); used for long sequences, where the grid stays large.
// DV < D (head dim 512, the SD VAE's single-head mid-block attention): a block produces only DV of the D output channels (value
// slice vs = slot % (D/DV)): the 128 accumulator registers a 512-wide output would need do not fit beside the 64 of the query
// fragments.  Q.K^T is recomputed per slice (1.5x the FLOPs of the unsplit product); the K/V tiles then go to LDS without the
// register prefetch (PREF = false: 96 more registers).
// DB (with PREF): the K/V tiles alternate between two LDS buffers, so a key tile costs ONE block barrier: tile k+1 is written to
// the other buffer while tile k is still being read (the barrier at the top of the next trip publishes it), and the global
// loads of tile k+2 go out right behind.  Available where two buffers still leave >= 2 blocks per CU (head dims 64 and 128).
// DMA (head dim 512 only): K rows (1 KiB = one wave-instruction each) and V-slice rows (512 B: the lower half-wave) go global -> LDS by
// LDS-DMA instead of through registers (the 256 VGPRs of this instantiation leave no room for a register prefetch, and without one
// every key tile paid six dependent global-load round trips: the kernel ran 8x below its MFMA time).  K and V have separate buffers
// and separate phases: K(t+1) flies while the softmax and P.V of tile t run, V(t+1) while Q.K^T of tile t+1 runs; four barriers per tile.
template <typename T, int D, int QT, int DV = D, bool PREF = true, bool DB = false, bool DMA = false>
__global__ __launch_bounds__(256) void attention16_kernel(const AttP p) {
  static_assert(!DB || PREF, "double buffering rides on the register prefetch");
  static_assert(!DMA || (D == 512 && DV == 256 && QT == 1 && !PREF && !DB), "LDS-DMA staging: the head-dim-512 form only");
  constexpr int ES = 2, ROWB = D * ES + 32;        // LDS row stride of K in bytes
  constexpr int VROWB = DV * ES + 32;              // LDS row stride of the V slice
  constexpr int CH = D / 8;                        // 16-byte chunks per K row
  constexpr int CHV = DV / 8;                      // ... per V-slice row
  constexpr int KSTEPS = D / 32;                   // MFMA k-steps for Q.K
  constexpr int DT = DV / 16;                      // output d tiles
  constexpr int NSL = D / DV;                      // value slices
  constexpr bool WIDE = DMA;                       // fragment reads issued four at a time (see the Q.K^T loop); at D = 256, with the register
                                                   // prefetch at the 256-VGPR limit, hipcc serialises them again and the DDPM++ step measured 1-3 % slower
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int BUFB = 64 * ROWB + 64 * VROWB;      // one K + V tile
  char* sK = smem;
  char* sV = smem + 64 * ROWB;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int lq = lane & 15, lg = lane >> 4;
  int nh, qblk;
  att_block(p, nh, qblk);
  int vs = 0;
  if constexpr (NSL > 1) { vs = nh % NSL; nh /= NSL; }
  const int n = nh / p.heads, head = nh - n * p.heads;
  const int C = p.heads * D;
  const size_t rowstride = (size_t)3 * C * ES;
  const char* base = p.qkv + (size_t)n * p.t * rowstride + (size_t)head * D * ES;
  const int q0 = qblk * (64 * QT) + wid * (16 * QT);

  // Q fragments: B operand, lane holds Q[q][8*(lg + 4s) .. +8]
  uint4 qf[QT][KSTEPS];
#pragma unroll
  for (int qt = 0; qt < QT; ++qt)
#pragma unroll
    for (int s = 0; s < KSTEPS; ++s) {
      const int qrow = q0 + qt * 16 + lq;
      qf[qt][s] = make_uint4(0, 0, 0, 0);
      if (qrow < p.t) qf[qt][s] = *reinterpret_cast<const uint4*>(base + (size_t)qrow * rowstride + (lg + 4 * s) * 16);
    }
  // softmax(q.k*scale) == softmax2((q*scale*log2e).k): one multiply per S element saved in the tile loop.  For exact
  // powers of two (d = 64, 256: scale = 1/8, 1/16) the product q*scale is exact in bf16/f16; log2e is applied in f32 below.
  const float sc2 = p.scale_log2e;
  f32x4_t o[QT][DT];
  // The softmax denominator comes off the matrix core too: an all-ones A operand against P^T gives, in every row of a 16x16 tile,
  // the sum over the 32 keys of the k-step, i.e. l accumulates exactly like O (same rounded P, rescaled by the same alpha) and
  // the 16 adds + 2 cross-lane shuffles per query tile and key tile leave the VALU, which is this kernel's bound at d = 64
  // (~7 VALU issue slots per S element against 4 MFMAs per 16x16 S tile).
  f32x4_t ol[QT];
  const uint4 ones = make_uint4(AttMma<T>::ONES2, AttMma<T>::ONES2, AttMma<T>::ONES2, AttMma<T>::ONES2);
  float m_run[QT];
#pragma unroll
  for (int qt = 0; qt < QT; ++qt) {
    m_run[qt] = -INFINITY; ol[qt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < DT; ++i) o[qt][i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  }

  const int ntiles = (p.t + 63) / 64;
  constexpr int NCH = (64 * CH) / 256;             // 16-byte chunks of K staged per thread per tile
  constexpr int NCHV = (64 * CHV) / 256;           // ... of the V slice
  constexpr int NPF = PREF ? NCH : 1, NPFV = PREF ? NCHV : 1;
  uint4 pk[NPF], pv[NPFV];
  const char* const vbase = base + (size_t)2 * C * ES + (size_t)vs * DV * ES;
  // register prefetch: the next tile's K/V rows are in flight while the current tile feeds the MFMAs
#define ATT_LOAD_TILE(key0_)                                                                     \
  _Pragma("unroll") for (int u = 0; u < NCH; ++u) {                                               \
    const int idx = tid + 256 * u, r = idx / CH, c = idx - r * CH;                                \
    pk[u] = make_uint4(0, 0, 0, 0);                                                               \
    if ((key0_) + r < p.t) pk[u] = *reinterpret_cast<const uint4*>(base + (size_t)((key0_) + r) * rowstride + c * 16 + (size_t)C * ES); \
  }                                                                                               \
  _Pragma("unroll") for (int u = 0; u < NCHV; ++u) {                                              \
    const int idx = tid + 256 * u, r = idx / CHV, c = idx - r * CHV;                              \
    pv[u] = make_uint4(0, 0, 0, 0);                                                               \
    if ((key0_) + r < p.t) pv[u] = *reinterpret_cast<const uint4*>(vbase + (size_t)((key0_) + r) * rowstride + c * 16); \
  }
#define ATT_STAGE_TILE(boff_)                                                                     \
  _Pragma("unroll") for (int u = 0; u < NCH; ++u) {                                               \
    const int idx = tid + 256 * u, r = idx / CH, c = idx - r * CH;                                \
    *reinterpret_cast<uint4*>(sK + (boff_) + r * ROWB + c * 16) = pk[u];                          \
  }                                                                                               \
  _Pragma("unroll") for (int u = 0; u < NCHV; ++u) {                                              \
    const int idx = tid + 256 * u, r = idx / CHV, c = idx - r * CHV;                              \
    *reinterpret_cast<uint4*>(sV + (boff_) + r * VROWB + c * 16) = pv[u];                         \
  }
  // LDS-DMA staging (DMA): wave w stages key rows 16w .. 16w+15 of a tile; vmcnt counts 16 K + 16 V wave-instructions per tile, in issue order
  const uint32_t lds_k = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)sK;
  const uint32_t lds_v = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)sV;
  const char* const zsrc = reinterpret_cast<const char*>(g_att_zero) + lane * 16;
#define ATT_DMA_K(key0_)                                                                         \
  _Pragma("unroll") for (int i = 0; i < 16; ++i) {                                                \
    const int r = wid * 16 + i;                                                                   \
    const char* src = ((key0_) + r < p.t) ? base + (size_t)((key0_) + r) * rowstride + (size_t)C * ES + lane * 16 : zsrc; \
    att_glds16(src, __builtin_amdgcn_readfirstlane(lds_k + r * ROWB));                            \
  }
#define ATT_DMA_V(key0_)                                                                         \
  if (lane < 32) {                                                                                \
    _Pragma("unroll") for (int i = 0; i < 16; ++i) {                                              \
      const int r = wid * 16 + i;                                                                 \
      const char* src = ((key0_) + r < p.t) ? vbase + (size_t)((key0_) + r) * rowstride + lane * 16 : zsrc; \
      att_glds16(src, __builtin_amdgcn_readfirstlane(lds_v + r * VROWB));                         \
    }                                                                                             \
  }
  if constexpr (DMA) { ATT_DMA_K(0); ATT_DMA_V(0); }
  if constexpr (PREF) { ATT_LOAD_TILE(0); }
  if constexpr (DB) {                                // tile 0 -> buffer 0, tile 1 on its way
    ATT_STAGE_TILE(0);
    if (ntiles > 1) { ATT_LOAD_TILE(64); }
  }
  for (int kt = 0; kt < ntiles; ++kt) {
    const int key0 = kt * 64;
    const int boff = DB ? (kt & 1) * BUFB : 0;       // this tile's buffer
    if constexpr (DMA) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");       // K(kt) has landed; V(kt), issued after it, may still fly
    __syncthreads();                                 // DB: tile kt published, tile kt-1 fully consumed; else: previous tile consumed; DMA: K(kt) published
    if constexpr (DMA) {
    } else if constexpr (DB) {
      if (kt + 1 < ntiles) {
        ATT_STAGE_TILE(BUFB - boff);                 // tile kt+1 into the buffer tile kt-1 occupied
        if (kt + 2 < ntiles) { ATT_LOAD_TILE(key0 + 128); }
      }
    } else if constexpr (PREF) {
      ATT_STAGE_TILE(0);
    } else {
      // no register prefetch: global -> LDS in groups of four chunks (the loads of a group are in flight together)
#pragma unroll
      for (int u0 = 0; u0 < NCH; u0 += 4) {
        uint4 tk[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int idx = tid + 256 * (u0 + u), r = idx / CH, c = idx - r * CH;
          tk[u] = make_uint4(0, 0, 0, 0);
          if (key0 + r < p.t) tk[u] = *reinterpret_cast<const uint4*>(base + (size_t)(key0 + r) * rowstride + c * 16 + (size_t)C * ES);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int idx = tid + 256 * (u0 + u), r = idx / CH, c = idx - r * CH;
          *reinterpret_cast<uint4*>(sK + r * ROWB + c * 16) = tk[u];
        }
      }
#pragma unroll
      for (int u0 = 0; u0 < NCHV; u0 += 4) {
        uint4 tv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int idx = tid + 256 * (u0 + u), r = idx / CHV, c = idx - r * CHV;
          tv[u] = make_uint4(0, 0, 0, 0);
          if (key0 + r < p.t) tv[u] = *reinterpret_cast<const uint4*>(vbase + (size_t)(key0 + r) * rowstride + c * 16);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int idx = tid + 256 * (u0 + u), r = idx / CHV, c = idx - r * CHV;
          *reinterpret_cast<uint4*>(sV + r * VROWB + c * 16) = tv[u];
        }
      }
    }
    if constexpr (!DB && !DMA) {
      __syncthreads();
      if constexpr (PREF) { if (kt + 1 < ntiles) { ATT_LOAD_TILE(key0 + 64); } }
    }

    // ---- S^T tiles: 4 x (16 keys x 16 queries) per query tile; a K fragment feeds every query tile
    f32x4_t sacc[QT][4];
    if constexpr (WIDE) {
      // long K dimension (16 k-steps): k-step outer, key tile inner -- four independent accumulator chains and four fragment reads in
      // flight per step.  In the order below (one chain at a time) hipcc, at the register limit of this instantiation, emitted
      // read -> wait -> MFMA 64 times per key tile: one LDS round trip per MFMA.
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) sacc[qt][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s = 0; s < KSTEPS; ++s) {
        uint4 ka[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) ka[j] = *reinterpret_cast<const uint4*>(sK + boff + (j * 16 + lq) * ROWB + (lg + 4 * s) * 16);
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int qt = 0; qt < QT; ++qt) sacc[qt][j] = AttMma<T>::run(ka[j], qf[qt][s], sacc[qt][j]);
      }
    } else {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
#pragma unroll
      for (int qt = 0; qt < QT; ++qt) sacc[qt][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s = 0; s < KSTEPS; ++s) {
        const uint4 ka = *reinterpret_cast<const uint4*>(sK + boff + (j * 16 + lq) * ROWB + (lg + 4 * s) * 16);
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) sacc[qt][j] = AttMma<T>::run(ka, qf[qt][s], sacc[qt][j]);
      }
    }
    }
    if constexpr (DMA) {
      __syncthreads();                               // every wave has read its K fragments: the K buffer is free
      if (kt + 1 < ntiles) { ATT_DMA_K(key0 + 64); }
    }
    // ---- online softmax; lane holds keys key0 + j*16 + lg*4 + r of query lq.  The running max is kept in the
    // un-scaled domain and the scale (incl. log2 e) is folded into the exponent: exp2(s*sc2 - m*sc2), one FMA per element.
    uint4 pb[QT][2];
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
      if (key0 + 64 > p.t) {                         // only the last tile of a ragged sequence needs the mask
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (key0 + j * 16 + lg * 4 + r >= p.t) sacc[qt][j][r] = -INFINITY;
      }
      // raw v_max3_f32 / v_max_f32: fmaxf() makes hipcc canonicalise every operand first (IEEE sNaN quieting, one extra v_max per
      // value: 46 vector instructions per key tile instead of 20 in a loop that is vector-issue bound); no NaN can occur here
      float tmax = vmax3(sacc[qt][0][0], sacc[qt][0][1], sacc[qt][0][2]);
      tmax = vmax3(tmax, sacc[qt][0][3], sacc[qt][1][0]);
      tmax = vmax3(tmax, sacc[qt][1][1], sacc[qt][1][2]);
      tmax = vmax3(tmax, sacc[qt][1][3], sacc[qt][2][0]);
      tmax = vmax3(tmax, sacc[qt][2][1], sacc[qt][2][2]);
      tmax = vmax3(tmax, sacc[qt][2][3], sacc[qt][3][0]);
      tmax = vmax3(tmax, sacc[qt][3][1], sacc[qt][3][2]);
      tmax = vmax2(tmax, sacc[qt][3][3]);
      {   // the other three lane groups' maxima in ONE crossbar round trip (three independent permutes) instead of two dependent ones
        const float t16 = __shfl_xor(tmax, 16, 64), t32 = __shfl_xor(tmax, 32, 64), t48 = __shfl_xor(tmax, 48, 64);
        tmax = vmax2(vmax3(tmax, t16, t32), t48);
      }
      const float m_new = vmax2(m_run[qt], tmax);    // finite: every tile has >= 1 valid key
      const float alpha = __builtin_amdgcn_exp2f((m_run[qt] - m_new) * sc2);
      const float mb = m_new * sc2;
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) sacc[qt][j][r] = __builtin_amdgcn_exp2f(fmaf(sacc[qt][j][r], sc2, -mb));
      m_run[qt] = m_new;
      if (!__all(alpha == 1.0f)) {                   // running max unchanged for the whole wave: nothing to rescale
#pragma unroll
        for (int i = 0; i < DT; ++i) o[qt][i] *= alpha;
        ol[qt][0] *= alpha;                          // rows of the ones-tile are all equal; only element 0 is read
      }
      // P^T as the B operand; k-slot (lg, e): e<4 -> key 16*(2kk) + 4lg + e ; e>=4 -> key 16*(2kk+1) + 4lg + e-4
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        pb[qt][kk].x = AttMma<T>::pack2(sacc[qt][2 * kk][0], sacc[qt][2 * kk][1]);
        pb[qt][kk].y = AttMma<T>::pack2(sacc[qt][2 * kk][2], sacc[qt][2 * kk][3]);
        pb[qt][kk].z = AttMma<T>::pack2(sacc[qt][2 * kk + 1][0], sacc[qt][2 * kk + 1][1]);
        pb[qt][kk].w = AttMma<T>::pack2(sacc[qt][2 * kk + 1][2], sacc[qt][2 * kk + 1][3]);
      }
    }

    if constexpr (DMA) {                             // V(kt) has landed (K(kt+1), issued after it, may still fly) and is published
      if (kt + 1 < ntiles) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
    }
    // ---- O^T += V^T . P^T ; a V^T fragment feeds every query tile
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      // transposed read: lane 4q'+p' of each 16-lane group addresses row q', columns 4p'..4p'+3 of a 4x16 block
      const int rq = (lane & 15) >> 2, rp = lane & 3;
      const char* va = sV + boff + (32 * kk + 4 * lg + rq) * VROWB + rp * 8;
#pragma unroll
      for (int qt = 0; qt < QT; ++qt) ol[qt] = AttMma<T>::run(ones, pb[qt][kk], ol[qt]);
      const char* vb = va + 16 * VROWB;
      if constexpr (WIDE) {
        // four output tiles per round: eight transposed reads in flight, then four independent MFMAs (see the Q.K^T loop above)
#pragma unroll
        for (int dt0 = 0; dt0 < DT; dt0 += 4) {
          uint4 av[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(va + (dt0 + u) * 32));
            const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(vb + (dt0 + u) * 32));
            const uint2 lo2 = __builtin_bit_cast(uint2, lo), hi2 = __builtin_bit_cast(uint2, hi);
            av[u] = make_uint4(lo2.x, lo2.y, hi2.x, hi2.y);
          }
#pragma unroll
          for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int qt = 0; qt < QT; ++qt) o[qt][dt0 + u] = AttMma<T>::run(av[u], pb[qt][kk], o[qt][dt0 + u]);
        }
      } else {
#pragma unroll
      for (int dt = 0; dt < DT; ++dt) {
        const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(va + dt * 32));
        const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(vb + dt * 32));
        // the two transposed 8-byte reads ARE the operand's four dwords (element-wise repacking compiled to 16 shift/or ops)
        const uint2 lo2 = __builtin_bit_cast(uint2, lo), hi2 = __builtin_bit_cast(uint2, hi);
        const uint4 av = make_uint4(lo2.x, lo2.y, hi2.x, hi2.y);
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) o[qt][dt] = AttMma<T>::run(av, pb[qt][kk], o[qt][dt]);
      }
      }
    }
    if constexpr (DMA) {
      __syncthreads();                               // every wave has read its V^T fragments: the V buffer is free
      if (kt + 1 < ntiles) { ATT_DMA_V(key0 + 64); }
    }
  }
  // ---- store: lane holds channels dt*16 + lg*4 + r of query lq
#pragma unroll
  for (int qt = 0; qt < QT; ++qt) {
    const int qrow = q0 + qt * 16 + lq;
    if (qrow < p.t) {
      const float inv = 1.f / ol[qt][0];
      T* orow = reinterpret_cast<T*>(p.out) + ((size_t)n * p.t + qrow) * C + head * D + vs * DV;
#pragma unroll
      for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int r = 0; r < 4; ++r) st1<T>(orow + dt * 16 + lg * 4 + r, o[qt][dt][r] * inv);
    }
  }
}
#undef ATT_LOAD_TILE
#undef ATT_STAGE_TILE
#undef ATT_DMA_K
#undef ATT_DMA_V

// ------------------------------------------------------------------------------------------------
// Split-precision attention (dts.h DTS_F16X3; head dim 64): the f32 attention's accuracy on the 16-bit matrix cores.  attention32_kernel
// is bound by the f32 matrix instruction (1/16 of the f16 rate): 0.66 ms at best for the ADM 32x32 level at 64 rows, 17 % of a
// split-precision search iteration.  Here every operand is an f16 pair hi + lo (hi = f16(y), lo = f16(y - hi)) of a value y = x * 2^k, and
// every product x*y ~ xh*yh + xl*yh + xh*yl (the lo*lo term, 2^-22, is dropped) accumulates in ONE f32 accumulator.  The powers of two keep
// the lo parts NORMAL f16 numbers -- the matrix cores flush subnormal inputs -- and come out again exactly:
//   q, k, v arrive as the image dts_split2_f16 makes of the f32 qkv tensor: hi(3C) | lo(3C) of x * 2^6 per token (|x| < 1023; the lo part
//     of an |x| below 2^-8 is lost: <= 2^-20 |x|, a few such elements per dot product);
//   S^T * 2^12 = Kh.Qh^T + Kh.Ql^T + Kl.Qh^T, the 2^-12 folded into the softmax scale;
//   P (f32 in [0,1] after exp2) * 2^14 -> ph, pl in registers;  O^T * 2^20 = Vh^T.ph + Vh^T.pl + Vl^T.ph,  l * 2^14 = ones.ph + ones.pl.
// Structure, LDS rows (K' = Kh | Kl, V' = Vh | Vl: 256 B + 32 B pad), register prefetch, QT query tiles per wave and the transposed V
// reads are attention16_kernel's; three MFMAs per fragment pair instead of one, and the fragment reads are what bounds it (QT = 2 halves them).
template <int QT, int VAR = 7>
__global__ __launch_bounds__(256) void attention_x3_kernel(const AttP p) {
  using T = f16_t;
  // VAR (A/B aid, DTS_ATT_DB = 16 + VAR): bit 0 = K fragments of group j + 1 requested before the MFMAs of group j, bit 1 = both query tiles' maxima
  // exchanged in one round, bit 2 = the first V^T fragments requested before the softmax
  constexpr bool KPRE = (VAR & 1) != 0, SM2 = (VAR & 2) != 0;
  constexpr int D = 64, ES = 2, ROWB = 2 * D * ES + 32, CH = 2 * D / 8, KSTEPS = D / 32, DT = D / 16, NCH = (64 * CH) / 256;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* sK = smem;
  char* sV = smem + 64 * ROWB;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int lq = lane & 15, lg = lane >> 4;
  int nh, qblk;
  att_block(p, nh, qblk);
  const int n = nh / p.heads, head = nh - n * p.heads;
  const int C = p.heads * D;
  const size_t rowstride = (size_t)6 * C * ES;           // hi(3C) | lo(3C)
  const size_t LO = (size_t)3 * C * ES;                  // byte offset of the lo planes inside a token row
  const char* base = p.qkv + (size_t)n * p.t * rowstride + (size_t)head * D * ES;
  const int q0 = qblk * (64 * QT) + wid * (16 * QT);
  uint4 qh[QT][KSTEPS], ql[QT][KSTEPS];
#pragma unroll
  for (int qt = 0; qt < QT; ++qt)
#pragma unroll
    for (int s = 0; s < KSTEPS; ++s) {
      const int qrow = q0 + qt * 16 + lq;
      qh[qt][s] = ql[qt][s] = make_uint4(0, 0, 0, 0);
      if (qrow < p.t) {
        qh[qt][s] = *reinterpret_cast<const uint4*>(base + (size_t)qrow * rowstride + (lg + 4 * s) * 16);
        ql[qt][s] = *reinterpret_cast<const uint4*>(base + (size_t)qrow * rowstride + LO + (lg + 4 * s) * 16);
      }
    }
  const float sc2 = p.scale_log2e * (1.0f / 4096.0f);    // the accumulators hold S * 2^12 (q and k carry 2^6 each)
  f32x4_t o[QT][DT], ol[QT];
  const uint4 ones = make_uint4(AttMma<T>::ONES2, AttMma<T>::ONES2, AttMma<T>::ONES2, AttMma<T>::ONES2);
  float m_run[QT];
#pragma unroll
  for (int qt = 0; qt < QT; ++qt) {
    m_run[qt] = -INFINITY; ol[qt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < DT; ++i) o[qt][i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  }
  const int ntiles = (p.t + 63) / 64;
  uint4 pk[NCH], pv[NCH];
  // chunk c of a K' / V' row: c < 8 -> hi plane chunk c, else lo plane chunk c - 8.  Thread tid stages chunk c = tid & 15 of rows (tid >> 4) + 16 u.
  // Addressing (round 6): buffer loads -- the block's descriptor (scalar registers), per-lane 32-bit byte offsets that never change, and the
  // key tile as the instruction's SCALAR offset -- so a tile's eight loads cost no vector instruction at all; the former per-load 64-bit
  // row * stride products, plane selects and row bound checks were ~100 of the loop's ~350 vector instructions per key tile (against 104
  // MFMAs: the kernel is vector-issue bound).  A ragged last tile gives the rows past the end an out-of-range lane offset, which the
  // buffer load turns into zeros (the scalar offset is not part of the range check, so the whole-tile form needs none).
  static_assert(CH == 16 && NCH == 4, "tile staging layout");
  typedef unsigned int att_u32x4 __attribute__((ext_vector_type(4)));
  const __amdgpu_buffer_rsrc_t rs_kv = __builtin_amdgcn_make_buffer_rsrc((void*)const_cast<char*>(base), 0, 0x7fff0000, 0x00020000);
  uint32_t koff[NCH];
  {
    const int r = tid >> 4, c = tid & 15;
#pragma unroll
    for (int u = 0; u < NCH; ++u)
      koff[u] = (uint32_t)((size_t)(r + 16 * u) * rowstride + (c < 8 ? 0 : LO) + (size_t)(c & 7) * 16 + (size_t)C * ES);
  }
  const uint32_t v_plane = (uint32_t)((size_t)C * ES);                                  // v sits one C-wide block behind k
#define X3_LD(voff_, soff_) __builtin_bit_cast(uint4, (att_u32x4)__builtin_amdgcn_raw_buffer_load_b128(rs_kv, (int)(voff_), (int)(soff_), 0))
#define X3_LOAD_TILE(key0_)                                                                      \
  {                                                                                               \
    const uint32_t so_ = (uint32_t)((size_t)(key0_) * rowstride);                                 \
    if ((key0_) + 64 <= p.t) {                                                                    \
      _Pragma("unroll") for (int u = 0; u < NCH; ++u) { pk[u] = X3_LD(koff[u], so_); pv[u] = X3_LD(koff[u], so_ + v_plane); }      \
    } else {                                                                                      \
      _Pragma("unroll") for (int u = 0; u < NCH; ++u) {                                           \
        const uint32_t vo_ = ((key0_) + (tid >> 4) + 16 * u < p.t) ? koff[u] : 0x80000000u;       \
        pk[u] = X3_LD(vo_, so_); pv[u] = X3_LD(vo_, so_ + v_plane);                               \
      }                                                                                           \
    }                                                                                             \
  }
  const uint32_t st_off = (uint32_t)((tid >> 4) * ROWB + (tid & 15) * 16);               // LDS slot of chunk (row tid >> 4, c); + 16 u rows
#define X3_STAGE_TILE()                                                                           \
  _Pragma("unroll") for (int u = 0; u < NCH; ++u) {                                               \
    *reinterpret_cast<uint4*>(sK + st_off + u * (16 * ROWB)) = pk[u];                             \
    *reinterpret_cast<uint4*>(sV + st_off + u * (16 * ROWB)) = pv[u];                             \
  }
  X3_LOAD_TILE(0);
  for (int kt = 0; kt < ntiles; ++kt) {
    const int key0 = kt * 64;
    __syncthreads();                                   // the previous tile is consumed
    X3_STAGE_TILE();
    __syncthreads();
    if (kt + 1 < ntiles) { X3_LOAD_TILE(key0 + 64); }
    // ---- S^T * 2^12: a K fragment pair feeds every query tile
    f32x4_t sacc[QT][4];
    // the fragments of key group j + 1 are requested BEFORE the MFMAs of group j are issued (two register sets): in the former order the
    // four reads of a group went out behind the previous group's MFMAs and the wave then waited a whole LDS round trip for them, four
    // times per key tile (ISA trace, round 6: RRRR wait MMMM.. RRRR wait ..) -- with two waves per SIMD nothing else covers that wait
    uint4 kf[2][KSTEPS][2];
#pragma unroll
    for (int s = 0; s < KSTEPS; ++s) {
      kf[0][s][0] = *reinterpret_cast<const uint4*>(sK + lq * ROWB + (lg + 4 * s) * 16);
      kf[0][s][1] = *reinterpret_cast<const uint4*>(sK + lq * ROWB + D * ES + (lg + 4 * s) * 16);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
#pragma unroll
      for (int qt = 0; qt < QT; ++qt) sacc[qt][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
      if (KPRE && j < 3) {
#pragma unroll
        for (int s = 0; s < KSTEPS; ++s) {
          kf[(j + 1) & 1][s][0] = *reinterpret_cast<const uint4*>(sK + ((j + 1) * 16 + lq) * ROWB + (lg + 4 * s) * 16);
          kf[(j + 1) & 1][s][1] = *reinterpret_cast<const uint4*>(sK + ((j + 1) * 16 + lq) * ROWB + D * ES + (lg + 4 * s) * 16);
        }
      }
      if constexpr (KPRE) __builtin_amdgcn_sched_barrier(0);               // (keep the requests in front of this group's MFMAs)
#pragma unroll
      for (int s = 0; s < KSTEPS; ++s) {
        uint4 kh = kf[j & 1][s][0], kl = kf[j & 1][s][1];
        if (!KPRE && j > 0) {
          kh = *reinterpret_cast<const uint4*>(sK + (j * 16 + lq) * ROWB + (lg + 4 * s) * 16);
          kl = *reinterpret_cast<const uint4*>(sK + (j * 16 + lq) * ROWB + D * ES + (lg + 4 * s) * 16);
        }
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) {
          sacc[qt][j] = AttMma<T>::run(kh, qh[qt][s], sacc[qt][j]);
          sacc[qt][j] = AttMma<T>::run(kh, ql[qt][s], sacc[qt][j]);
          sacc[qt][j] = AttMma<T>::run(kl, qh[qt][s], sacc[qt][j]);
        }
      }
    }
    // V^T fragment pair (hi, lo) of key half kk, value tile dt: four transposing reads
#define X3_VFRAG(kk_, dt_, avh_, avl_)                                                                                                    \
  {                                                                                                                                       \
    const char* va_ = sV + (32 * (kk_) + 4 * lg + ((lane & 15) >> 2)) * ROWB + (lane & 3) * 8 + (dt_) * 32;                                \
    const uint2 h0_ = __builtin_bit_cast(uint2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(va_)));               \
    const uint2 h1_ = __builtin_bit_cast(uint2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(va_ + 16 * ROWB)));   \
    const uint2 l0_ = __builtin_bit_cast(uint2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(va_ + D * ES)));      \
    const uint2 l1_ = __builtin_bit_cast(uint2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(va_ + 16 * ROWB + D * ES))); \
    avh_ = make_uint4(h0_.x, h0_.y, h1_.x, h1_.y); avl_ = make_uint4(l0_.x, l0_.y, l1_.x, l1_.y);                                          \
  }
    // the first V^T fragments do not depend on the softmax: requested HERE, their LDS round trip runs under its vector work instead of in
    // front of the first P.V MFMA
    constexpr int VPRE = (VAR & 4) ? 2 : 0;
    uint4 vpre[VPRE > 0 ? VPRE : 1][2];
#pragma unroll
    for (int dt = 0; dt < VPRE; ++dt) X3_VFRAG(0, dt, vpre[dt][0], vpre[dt][1]);
    // ---- online softmax (attention16_kernel's), then P * 2^14 as (hi, lo) operand pairs.  Both query tiles' row maxima first: their
    // cross-lane exchanges (three independent permutes per tile) are all in flight before the first is awaited -- one crossbar round trip
    // per key tile instead of four dependent ones
    uint4 pbh[QT][2], pbl[QT][2];
    float tmx[QT], t16[QT], t32[QT], t48[QT];
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
      if (key0 + 64 > p.t) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (key0 + j * 16 + lg * 4 + r >= p.t) sacc[qt][j][r] = -INFINITY;
      }
      float tmax = vmax3(sacc[qt][0][0], sacc[qt][0][1], sacc[qt][0][2]);
      tmax = vmax3(tmax, sacc[qt][0][3], sacc[qt][1][0]);
      tmax = vmax3(tmax, sacc[qt][1][1], sacc[qt][1][2]);
      tmax = vmax3(tmax, sacc[qt][1][3], sacc[qt][2][0]);
      tmax = vmax3(tmax, sacc[qt][2][1], sacc[qt][2][2]);
      tmax = vmax3(tmax, sacc[qt][2][3], sacc[qt][3][0]);
      tmax = vmax3(tmax, sacc[qt][3][1], sacc[qt][3][2]);
      tmx[qt] = vmax2(tmax, sacc[qt][3][3]);
    }
    if constexpr (SM2) {
#pragma unroll
      for (int qt = 0; qt < QT; ++qt) {
        t16[qt] = __shfl_xor(tmx[qt], 16, 64); t32[qt] = __shfl_xor(tmx[qt], 32, 64); t48[qt] = __shfl_xor(tmx[qt], 48, 64);
      }
    }
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
      float tmax;
      if constexpr (SM2) tmax = vmax2(vmax3(tmx[qt], t16[qt], t32[qt]), t48[qt]);
      else { tmax = vmax2(tmx[qt], __shfl_xor(tmx[qt], 16, 64)); tmax = vmax2(tmax, __shfl_xor(tmax, 32, 64)); }
      const float m_new = vmax2(m_run[qt], tmax);
      const float alpha = __builtin_amdgcn_exp2f((m_run[qt] - m_new) * sc2);
      const float mb = m_new * sc2 - 14.0f;             // exp2(.. + 14): P * 2^14 straight out of the exponential
      m_run[qt] = m_new;
      if (!__all(alpha == 1.0f)) {
#pragma unroll
        for (int i = 0; i < DT; ++i) o[qt][i] *= alpha;
        ol[qt][0] *= alpha;
      }
      // P * 2^14 = hi + lo, two values at a time: the hi halves come out of ONE packed conversion (round to nearest even, as the scalar one), the
      // lo halves are e - hi with the f16 operand widened inside the instruction (v_fma_mix_f32: exact, the product and the sum are
      // representable) and packed.  The former scalar form converted every hi twice and cost 4.1 vector instructions per value, this costs 2.5.
      uint32_t hw_[4][2], lw_[4][2];
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 4; r += 2) {
          const float e0 = __builtin_amdgcn_exp2f(fmaf(sacc[qt][j][r], sc2, -mb));
          const float e1 = __builtin_amdgcn_exp2f(fmaf(sacc[qt][j][r + 1], sc2, -mb));
          const dts_f32x2_t ev = {e0, e1};
          const dts_f16x2_t hv = __builtin_convertvector(ev, dts_f16x2_t);
          const float l0 = __builtin_fmaf((float)hv[0], -1.0f, e0), l1 = __builtin_fmaf((float)hv[1], -1.0f, e1);
          hw_[j][r >> 1] = __builtin_bit_cast(uint32_t, hv);
          lw_[j][r >> 1] = pack2_f16(l0, l1);
        }
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        pbh[qt][kk] = make_uint4(hw_[2 * kk][0], hw_[2 * kk][1], hw_[2 * kk + 1][0], hw_[2 * kk + 1][1]);
        pbl[qt][kk] = make_uint4(lw_[2 * kk][0], lw_[2 * kk][1], lw_[2 * kk + 1][0], lw_[2 * kk + 1][1]);
      }
    }
    // ---- O^T * 2^20 and l * 2^14; a V^T fragment pair feeds every query tile.  (The first VPRE fragment pairs were requested before the
    // softmax, see above.)
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
      for (int qt = 0; qt < QT; ++qt) {
        ol[qt] = AttMma<T>::run(ones, pbh[qt][kk], ol[qt]);
        ol[qt] = AttMma<T>::run(ones, pbl[qt][kk], ol[qt]);
      }
#pragma unroll
      for (int dt = 0; dt < DT; ++dt) {
        uint4 avh, avl;
        if (kk == 0 && dt < VPRE) { avh = vpre[dt][0]; avl = vpre[dt][1]; }
        else X3_VFRAG(kk, dt, avh, avl);
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) {
          o[qt][dt] = AttMma<T>::run(avh, pbh[qt][kk], o[qt][dt]);
          o[qt][dt] = AttMma<T>::run(avh, pbl[qt][kk], o[qt][dt]);
          o[qt][dt] = AttMma<T>::run(avl, pbh[qt][kk], o[qt][dt]);
        }
      }
    }
  }
#undef X3_VFRAG
#undef X3_LOAD_TILE
#undef X3_LD
#undef X3_STAGE_TILE
#pragma unroll
  for (int qt = 0; qt < QT; ++qt) {
    const int qrow = q0 + qt * 16 + lq;
    if (qrow < p.t) {
      const float inv = (1.0f / 64.0f) / ol[qt][0];      // (O * 2^20) / (l * 2^14) = 2^6 * O  (v carries 2^6)
      if (p.out_split3) {
        // the only reader is the proj convolution of the split-precision mode: write its operand image (dts_split3_f16's arithmetic)
        // instead of the f32 tensor + a split pass
        f16_t* orow3 = reinterpret_cast<f16_t*>(p.out) + ((size_t)n * p.t + qrow) * 2 * C;      // rows of 2C f16: per 32 channels hi(32) | lo(32)
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
          float hi[4], lo[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float x = o[qt][dt][r] * inv;
            asm volatile("" : "+v"(x));                    // the ROUNDED product (no FMA contraction into the subtraction below): the value the f32 output holds
            x3_split(x, hi[r], lo[r]);
          }
          f16_t* d_ = orow3 + x3_off(head * D + dt * 16 + lg * 4);
          *reinterpret_cast<uint2*>(d_) = make_uint2(pack2_f16(hi[0], hi[1]), pack2_f16(hi[2], hi[3]));
          *reinterpret_cast<uint2*>(d_ + 32) = make_uint2(pack2_f16(lo[0], lo[1]), pack2_f16(lo[2], lo[3]));
        }
      } else {
        float* orow = reinterpret_cast<float*>(p.out) + ((size_t)n * p.t + qrow) * C + head * D;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
          *reinterpret_cast<float4*>(orow + dt * 16 + lg * 4) = make_float4(o[qt][dt][0] * inv, o[qt][dt][1] * inv, o[qt][dt][2] * inv, o[qt][dt][3] * inv);
      }
    }
  }
}


// ------------------------------------------------------------------------------------------------
// f32 (parity) path: same structure on v_mfma_f32_16x16x4_f32; V needs no transpose (one k per lane group).
template <int D>
__global__ __launch_bounds__(256) void attention32_kernel(const AttP p) {
  constexpr int ES = 4, ROWB = D * ES + 32;
  constexpr int CH = D / 4;
  constexpr int KCH = D / 16;                      // 16-byte chunk groups along d for Q.K
  constexpr int DT = D / 16;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* sK = smem;
  char* sV = smem + 64 * ROWB;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int lq = lane & 15, lg = lane >> 4;
  int nh, qblk;
  att_block(p, nh, qblk);
  const int n = nh / p.heads, head = nh - n * p.heads;
  const int C = p.heads * D;
  const size_t rowstride = (size_t)3 * C * ES;
  const char* base = p.qkv + (size_t)n * p.t * rowstride + (size_t)head * D * ES;
  const int q0 = qblk * 64 + wid * 16;
  const int qrow = q0 + lq;

  f32x4_t o[DT];
#pragma unroll
  for (int i = 0; i < DT; ++i) o[i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  float m_run = -INFINITY, l_run = 0.f;
  const int ntiles = (p.t + 63) / 64;
  for (int kt = 0; kt < ntiles; ++kt) {
    const int key0 = kt * 64;
    __syncthreads();
    for (int idx = tid; idx < 64 * CH; idx += 256) {
      const int r = idx / CH, c = idx - r * CH;
      uint4 kv = make_uint4(0, 0, 0, 0), vv = make_uint4(0, 0, 0, 0);
      if (key0 + r < p.t) {
        const char* g = base + (size_t)(key0 + r) * rowstride + c * 16;
        kv = *reinterpret_cast<const uint4*>(g + (size_t)C * ES);
        vv = *reinterpret_cast<const uint4*>(g + (size_t)2 * C * ES);
      }
      *reinterpret_cast<uint4*>(sK + r * ROWB + c * 16) = kv;
      *reinterpret_cast<uint4*>(sV + r * ROWB + c * 16) = vv;
    }
    __syncthreads();
    f32x4_t sacc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) sacc[j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < KCH; ++s) {
      float4 qv = make_float4(0.f, 0.f, 0.f, 0.f);
      if (qrow < p.t) qv = *reinterpret_cast<const float4*>(base + (size_t)qrow * rowstride + (lg + 4 * s) * 16);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float4 ka = *reinterpret_cast<const float4*>(sK + (j * 16 + lq) * ROWB + (lg + 4 * s) * 16);
        sacc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(ka.x, qv.x, sacc[j], 0, 0, 0);
        sacc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(ka.y, qv.y, sacc[j], 0, 0, 0);
        sacc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(ka.z, qv.z, sacc[j], 0, 0, 0);
        sacc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(ka.w, qv.w, sacc[j], 0, 0, 0);
      }
    }
    float tmax = -INFINITY;
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float v = sacc[j][r] * p.scale_log2e;
        if (key0 + j * 16 + lg * 4 + r >= p.t) v = -INFINITY;
        sacc[j][r] = v;
        tmax = fmaxf(tmax, v);
      }
    tmax = fmaxf(tmax, __shfl_xor(tmax, 16, 64));
    tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
    const float m_new = fmaxf(m_run, tmax);
    const float alpha = exp2f(m_run - m_new);
    float psum = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float e = exp2f(sacc[j][r] - m_new);
        sacc[j][r] = e;
        psum += e;
      }
    psum += __shfl_xor(psum, 16, 64);
    psum += __shfl_xor(psum, 32, 64);
    l_run = l_run * alpha + psum;
    m_run = m_new;
#pragma unroll
    for (int i = 0; i < DT; ++i) o[i] *= alpha;
    // O^T[d][q] += sum_key V[key][d] P[key][q]; MFMA step r of key tile j: lane group lg supplies key j*16 + 4lg + r
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const char* vrow = sV + (j * 16 + 4 * lg + r) * ROWB + lq * 4;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
          const float va = *reinterpret_cast<const float*>(vrow + dt * 64);
          o[dt] = __builtin_amdgcn_mfma_f32_16x16x4f32(va, sacc[j][r], o[dt], 0, 0, 0);
        }
      }
  }
  if (qrow < p.t) {
    const float inv = 1.f / l_run;
    float* orow = reinterpret_cast<float*>(p.out) + ((size_t)n * p.t + qrow) * C + head * D;
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) {
      const float4 v = make_float4(o[dt][0] * inv, o[dt][1] * inv, o[dt][2] * inv, o[dt][3] * inv);
      *reinterpret_cast<float4*>(orow + dt * 16 + lg * 4) = v;
    }
  }
}

template <typename K>
int launch_att(K kernel, const AttP& p0, size_t lds, hipStream_t st, int qblock = 64, int slices = 1) {
  {   // hipFuncSetAttribute once per (device, kernel): the attribute is per device, and every instantiation has the same pointer
      // TYPE, so the key is (device, pointer value)
    static std::mutex mu;
    static std::unordered_map<uintptr_t, size_t> done;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) dev = 0;
    std::lock_guard<std::mutex> lk(mu);
    size_t& have = done[reinterpret_cast<uintptr_t>(reinterpret_cast<const void*>(kernel)) ^ ((uintptr_t)(dev + 1) << 56)];
    if (lds > have) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      have = lds;
    }
  }
  AttP p = p0;
  p.qblocks = (p.t + qblock - 1) / qblock;
  p.xcd_remap = dts_knob_get(DTS_KNOB_ATT_XCD) != 0;        // DTS_ATT_XCD=0 restores the plain block order (A/B aid)
  const long long nblk = (long long)p.qblocks * p.n * p.heads * slices;
  DTS_CHECK_ARG(nblk < (1ll << 31), "dts_attention: grid too large");
  hipLaunchKernelGGL(kernel, dim3((unsigned)nblk), dim3(256), lds, st, p);
  DTS_CHECK_LAUNCH("dts_attention");
  return DTS_OK;
}

template <typename T>
int att16(const AttP& p, hipStream_t st) {
  const int g_att_qt = dts_knob_get(DTS_KNOB_ATT_QT) > 0 ? dts_knob_get(DTS_KNOB_ATT_QT) : 0;    // DTS_ATT_QT=1|2 forces the query tiles per wave
  const size_t lds = (size_t)2 * 64 * (p.d * 2 + 32);
  // double-buffered tiles measured no faster (T=1024: 167.6 vs 166.3 us; T=256: 27.0 vs 25.7 us, profiles/r02_attention_variants.txt):
  // the barrier was not what the waves wait on.  Kept behind DTS_ATT_DB=1 for the record; default single-buffered.
  const bool db = dts_knob_get(DTS_KNOB_ATT_DB) == 1 && p.t > 64;
  switch (p.d) {
    case 64:
      // two query tiles per wave once the sequence is long enough to keep >= 2 blocks per CU in the grid
      if (g_att_qt == 2 || (g_att_qt == 0 && p.t >= 256 && (long long)((p.t + 127) / 128) * p.n * p.heads >= 512))
        return db ? launch_att(attention16_kernel<T, 64, 2, 64, true, true>, p, 2 * lds, st, 128)
                  : launch_att(attention16_kernel<T, 64, 2>, p, lds, st, 128);
      return db ? launch_att(attention16_kernel<T, 64, 1, 64, true, true>, p, 2 * lds, st) : launch_att(attention16_kernel<T, 64, 1>, p, lds, st);
    case 128:
      return db ? launch_att(attention16_kernel<T, 128, 1, 128, true, true>, p, 2 * lds, st) : launch_att(attention16_kernel<T, 128, 1>, p, lds, st);
    case 256: return launch_att(attention16_kernel<T, 256, 1>, p, lds, st);
    case 512:   // two 256-wide value slices per (sample, head, query block); K rows 1056 B + V-slice rows 544 B per key
      if (dts_knob_get(DTS_KNOB_ATT_DB) == 2)     // DTS_ATT_DB=2: the register-staged form (A/B aid)
        return launch_att(attention16_kernel<T, 512, 1, 256, false>, p, (size_t)64 * (512 * 2 + 32) + (size_t)64 * (256 * 2 + 32), st, 64, 2);
      return launch_att(attention16_kernel<T, 512, 1, 256, false, false, true>, p, (size_t)64 * (512 * 2 + 32) + (size_t)64 * (256 * 2 + 32), st, 64, 2);
  }
  return DTS_ERR_UNSUPPORTED;
}

int att32(const AttP& p, hipStream_t st) {
  const size_t lds = (size_t)2 * 64 * (p.d * 4 + 32);
  switch (p.d) {
    case 64: return launch_att(attention32_kernel<64>, p, lds, st);
    case 128: return launch_att(attention32_kernel<128>, p, lds, st);
    case 256: return launch_att(attention32_kernel<256>, p, lds, st);
  }
  return DTS_ERR_UNSUPPORTED;
}

}  // namespace

extern "C" int dts_attention_x3(const void* qkv_split, void* out, int out_split3, int n, int t, int heads, int d, float scale, dts_stream s) {
  DTS_CHECK_ARG(qkv_split && out, "dts_attention_x3: null pointer");
  DTS_CHECK_ARG(n > 0 && t > 0 && heads > 0, "dts_attention_x3: bad shape");
  DTS_CHECK_ARG(d == 64, "dts_attention_x3: head dim %d unsupported (64; other sizes take dts_attention in DTS_F32)", d);
  // the key / value tiles are fetched with buffer loads: 32-bit byte offsets inside one sample's rows of 6 * heads * d f16
  DTS_CHECK_ARG((long long)t * 6 * heads * d * 2 < (1ll << 31), "dts_attention_x3: %d tokens x %d heads exceed the 2 GiB a sample's rows may span", t, heads);
  AttP p{(const char*)qkv_split, (char*)out, n, t, heads, d, scale * 1.4426950408889634f, 0, 1, out_split3 ? 1 : 0};
  const size_t lds = (size_t)2 * 64 * (2 * 64 * 2 + 32);
  // two query tiles per wave once the sequence is long enough to keep >= 2 blocks per CU in the grid (the rule of the 16-bit kernel)
  const int qt = dts_knob_get(DTS_KNOB_ATT_QT);            // DTS_ATT_QT = 1 | 2 forces a form (A/B aid, tools/att_bench.py --x3-kernel)
  // (round 6: one query tile per wave below 512 tokens -- 140 registers, three waves per SIMD: T = 256 at 64 rows 49.3 -> 43.8 us, tools/att_bench.py --x3-kernel)
  const bool two = qt == 2 || (qt != 1 && t >= 512 && (long long)((t + 127) / 128) * n * heads >= 512);
  const int var = (dts_knob_get(DTS_KNOB_ATT_DB) >= 16 && dts_knob_get(DTS_KNOB_ATT_DB) < 32) ? dts_knob_get(DTS_KNOB_ATT_DB) - 16 : 7;
#define X3_LAUNCH(V_) (two ? launch_att(attention_x3_kernel<2, V_>, p, lds, to_stream(s), 128) : launch_att(attention_x3_kernel<1, V_>, p, lds, to_stream(s)))
  if (var == 0) return X3_LAUNCH(0);      // DTS_ATT_DB=16: without the three latency changes of round 6 (A/B aid, tools/att_bench.py --x3-kernel --variants 0)
  return X3_LAUNCH(7);
#undef X3_LAUNCH
}

extern "C" int dts_attention(const void* qkv, void* out, int dtype, int n, int t, int heads, int d, float scale, dts_stream s) {
  DTS_CHECK_ARG(qkv && out, "dts_attention: null pointer");
  DTS_CHECK_ARG(n > 0 && t > 0 && heads > 0, "dts_attention: bad shape");
  DTS_CHECK_ARG(d == 64 || d == 128 || d == 256 || (d == 512 && dtype != DTS_F32),
                "dts_attention: head dim %d unsupported (64/128/256; 512 in the 16-bit types)", d);
  AttP p{(const char*)qkv, (char*)out, n, t, heads, d, scale * 1.4426950408889634f, 0, 1, 0};
  hipStream_t st = to_stream(s);
  switch (dtype) {
    case DTS_F32: return att32(p, st);
    case DTS_BF16: return att16<bf16_t>(p, st);
    case DTS_F16: return att16<f16_t>(p, st);
  }
  dts_set_error("dts_attention: bad dtype %d", dtype);
  return DTS_ERR_ARG;
}
