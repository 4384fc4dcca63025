// dsx_conv.hip — fused GroupNorm-apply + Swish + KxK convolution (+bias +FiLM
// +residual) as an implicit GEMM on gfx950 MFMA; NHWC activations in HBM in the MFMA operand type
// (fp32 in the parity build, bf16 in the bf16 build).
//
// Replaces the ATen op chain of Block / ResnetBlock / Upsample / Downsample /
// 1x1 convs of the reference UNets (model/sr3_modules/unet.py:58-110,
// model/ddpm_modules/unet.py:42-96): group_norm -> sigmoid -> mul -> conv2d ->
// add (FiLM) -> add (residual), and torch.cat / upsample_nearest2d in front of it.
//
// Work decomposition (one 256-thread workgroup = 4 wave64):
//   output tile  : BM = TB x TH x TW output pixels  x  BN output channels
//   K loop       : chunks of 64 B of input channels per pixel (16 fp32 / 32 bf16)
//   A operand    : the (TH*S+KS-S) x (TW*S+KS-S) input halo patch of the chunk is
//                  loaded once (coalesced 16-byte units along C), normalised + activated
//                  in registers, converted, and parked in LDS with an 80-B pixel
//                  stride (conflict-free ds_read_b128); all KS*KS taps re-read it
//                  at constant LDS offsets -> 9x fewer global reads than im2col.
//   B operand    : weights pre-packed on the host in MFMA fragment order, so each
//                  lane's 16-B fragment is one fully coalesced global load (1 KiB
//                  per wave instruction), software-prefetched one step ahead; no
//                  LDS traffic for weights.
//   MFMA         : bf16  v_mfma_f32_32x32x16_bf16 (1 per 16-B fragment pair)
//                  fp32  v_mfma_f32_32x32x2_f32   (4 per 16-B fragment pair; exact
//                        fp32 FMA chain -> the <=1e-3 parity path)
//                  k_conv_ws (the persistent kernel most launches run on): the 16 x 16 shapes on the SAME packed
//                  weights -- v_mfma_f32_16x16x32_bf16 / _f16, v_mfma_f32_16x16x4_f32 (see mfma16_step: the chip holds
//                  a higher clock under them)
//   epilogue     : weights are the MFMA A operand (rows packed permuted), pixels the B operand: a lane
//                  holds one pixel x 16 consecutive channels (k_conv_ws: 8 channels of one pixel per 16-pixel block)
//                  -> 16-byte loads / NHWC stores only.
#include "dsx_kernels.h"
#include <algorithm>
#include <cstdlib>
#include <type_traits>
#include <utility>

#ifndef DSX_WS_DEPTH_EXPR
#define DSX_WS_DEPTH_EXPR (bm == 64 ? 5 : (bm == 256 ? 2 : (ks == 1 ? 3 : 4)))   // measured: one more group in flight than the HBM latency strictly needs (256-pixel tile: 2, its third MFMA image takes the LDS of the fourth ring slot)
#endif
// residual prefetch one tile ahead: only with one N block per wave.  With two, the register demand passes 256 and
// hipcc (ROCm 7.2) fails in its spill path ("Illegal instruction detected: Operand has incorrect register class
// V_CMP_NE_U32_e32 0, $src_private_base"); the same happens for a 64 x 256 tile (MB 2, WM 1, WN 4, NB 2).
#ifndef DSX_PRE_RESID_EXPR
#define DSX_PRE_RESID_EXPR (NB == 1 && MB <= 2)   // (MB 4 with 16-bit storage fits the registers but measured 3 % slower)
#endif
#ifndef DSX_PFF_EXPR
#define DSX_PFF_EXPR (NB == 2 ? 3 : 6)   // k_conv_ws: pixel fragments in flight ahead of the MFMAs (~190 cycles of MFMA work)
#endif
#ifndef DSX_WS_K0_EXPR
#define DSX_WS_K0_EXPR 64   // raw groups requested before the first wait (>= P + 1: the whole ring, the round-2 behaviour)
#endif
#ifndef DSX_EPI_PRIO_COND
#define DSX_EPI_PRIO_COND false
#endif
#ifndef DSX_STAMP_CVT_COND
#define DSX_STAMP_CVT_COND (tiC == 2 && gC == 0)   // which item the loader-conversion stamps 120-123 record (diagnostic builds)
#endif
// 1 x 1 convs: weight fragments in flight per wave.  A group is only 4 steps (8-16 MFMAs), so a ring of one group
// (round 1) exposed a whole L2 round trip per group: 1800-2500 cycles for 256 cycles of MFMA work.
#ifndef DSX_RING_1X1_NB1
#define DSX_RING_1X1_NB1 16
#endif
#ifndef DSX_RING_1X1_NB2
#define DSX_RING_1X1_NB2 4
#endif
#ifndef DSX_LOADER_PRIO
#define DSX_LOADER_PRIO 3   // (1 until the 16 x 16 MFMA shape made the loaders the bound of most items; A/B in one call: 3 -0.3 %, 0 +0.5 %)
#endif
#ifndef DSX_LOADER_PRIO_EXPR
#define DSX_LOADER_PRIO_EXPR DSX_LOADER_PRIO   // (may name MB, KS, NB: per-tile experiments)
#endif
#ifndef DSX_RING_DEPTH
#define DSX_RING_DEPTH 6  // weight-fragment prefetch ring depth for 3x3 (divides 18)
#endif

// DSX_ABLATE (phase-skipping timing experiments, results wrong) exists only in the diagnostic build
// (DSX_EXTRA_FLAGS=-DDSX_DIAG ./build.sh); the product binary has no such switch.
#ifdef DSX_DIAG
#define DSX_ABLATED(bit) ((a.ablate & (bit)) != 0)
#else
#define DSX_ABLATED(bit) false
#endif

namespace dsx {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;


// x * sigmoid(x).  v_exp + v_rcp (1 ulp each) instead of an IEEE division sequence (10 more VALU
// instructions per element on the loaders' critical path); far inside the 1e-3 parity budget.
__device__ __forceinline__ float swish_f(float v) {
  return v * __builtin_amdgcn_rcpf(1.0f + __expf(-v));
}
// The same arithmetic on element pairs, written with two-element vectors so that hipcc selects the packed fp32
// instructions (v_pk_mul_f32 / v_pk_add_f32 / v_pk_fma_f32: one issue slot for two elements; the scalar spelling gets
// one v_mul / v_add per element).  Results are bit-identical to swish_f / x * s + h: the same operations in the same
// order (v_exp_f32 is 2^x: __expf(-v) is the product with -log2(e), then v_exp_f32).
typedef __attribute__((ext_vector_type(2))) float f32x2_t;
template <int N> __device__ __forceinline__ void swish_vec(float (&v)[N]) {
  static_assert(N % 2 == 0, "pairs");
#pragma unroll
  for (int j = 0; j < N; j += 2) {
    f32x2_t x = {v[j], v[j + 1]};
    const f32x2_t k = {-1.44269504088896340736f, -1.44269504088896340736f};
    const f32x2_t t = x * k;
    f32x2_t e = {__builtin_amdgcn_exp2f(t.x), __builtin_amdgcn_exp2f(t.y)};
    e = e + 1.0f;
    const f32x2_t r = {__builtin_amdgcn_rcpf(e.x), __builtin_amdgcn_rcpf(e.y)};
    x = x * r;
    v[j] = x.x; v[j + 1] = x.y;
  }
}
template <int N> __device__ __forceinline__ void affine_vec(float (&v)[N], const float (&sc)[N], const float (&sh)[N]) {
  static_assert(N % 2 == 0, "pairs");
#pragma unroll
  for (int j = 0; j < N; j += 2) {
    const f32x2_t x = {v[j], v[j + 1]}, s = {sc[j], sc[j + 1]}, h = {sh[j], sh[j + 1]};
    const f32x2_t y = x * s + h;               // contracted to v_pk_fma_f32 (as the scalar form is to v_fma_f32)
    v[j] = y.x; v[j + 1] = y.y;
  }
}

// 16 per-lane registers -> one total per lane: lane i of a DPP row ends up with the row's sum of register
// r = 8*bit0(i) + 4*bit1(i) + 2*bit2(i) + bit3(i).  Each butterfly stage halves the register count (the
// lane keeps the half its bit selects and receives the partner's copy of that half): 15 adds instead of
// the 64 of sixteen full 16-lane reductions.
template <int CTRL, int BANKS>
__device__ __forceinline__ float dpp_take(float old, float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old), __builtin_bit_cast(int, v),
                                                               CTRL, 0xF, BANKS, false));
}
__device__ __forceinline__ float row16_fold(const float (&s)[16], int lane) {
  const bool b0 = lane & 1, b1 = lane & 2, b2 = lane & 4, b3 = lane & 8;
  float a8[8], a4[4], a2[2];
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const float keep = b0 ? s[k + 8] : s[k], send = b0 ? s[k] : s[k + 8];
    a8[k] = keep + dpp_take<0xB1, 0xF>(0.f, send);                       // lane ^ 1
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const float keep = b1 ? a8[k + 4] : a8[k], send = b1 ? a8[k] : a8[k + 4];
    a4[k] = keep + dpp_take<0x4E, 0xF>(0.f, send);                       // lane ^ 2
  }
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const float keep = b2 ? a4[k + 2] : a4[k], send = b2 ? a4[k] : a4[k + 2];
    float t = dpp_take<0x104, 0x5>(0.f, send);                           // banks 0,2 <- lane + 4
    t = dpp_take<0x114, 0xA>(t, send);                                   // banks 1,3 <- lane - 4
    a2[k] = keep + t;
  }
  const float keep = b3 ? a2[1] : a2[0], send = b3 ? a2[0] : a2[1];
  return keep + dpp_take<0x128, 0xF>(0.f, send);                         // lane ^ 8 (row_ror:8)
}
__device__ __forceinline__ int row16_fold_reg(int lane) {
  return 8 * (lane & 1) + 4 * ((lane >> 1) & 1) + 2 * ((lane >> 2) & 1) + ((lane >> 3) & 1);
}

// (vector conversions: one v_cvt_pk_* per pair; converted one by one each element costs a convert, and the pair a shift and an or)
typedef __attribute__((ext_vector_type(2))) float f32x2_cvt;
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2_cvt;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_cvt;
__device__ __forceinline__ unsigned pack_f16x2(float lo, float hi) {
  const f32x2_cvt v = {lo, hi};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, f16x2_cvt));   // RNE
}
__device__ __forceinline__ float f16_lo(unsigned w) { return (float)__builtin_bit_cast(_Float16, (unsigned short)(w & 0xffffu)); }
__device__ __forceinline__ float f16_hi(unsigned w) { return (float)__builtin_bit_cast(_Float16, (unsigned short)(w >> 16)); }
__device__ __forceinline__ unsigned pack_bf16x2(float lo, float hi) {
  const f32x2_cvt v = {lo, hi};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_cvt));  // RNE (v_cvt_pk_bf16_f32)
}

// diagnostic stamps (DSX_STAMP_OP): wave 0 of one chosen workgroup records s_memtime at phase
// boundaries into a debug buffer that nothing else reads; off (nullptr) in normal runs
#ifdef DSX_STAMPS   // diagnostic build (DSX_EXTRA_FLAGS=-DDSX_STAMPS ./build.sh): the checks cost scalar work in hot loops
#define DSX_STAMP(i)                                                                        \
  do {                                                                                      \
    if (a.stamp != nullptr && blockIdx.x == (unsigned)a.stamp_block && tid == 0 && (i) < 120) \
      a.stamp[(i)] = __builtin_amdgcn_s_memtime();                                          \
  } while (0)

#define DSX_STAMP_T(i, cond)                                                               \
  do {                                                                                      \
    if (a.stamp != nullptr && blockIdx.x == (unsigned)a.stamp_block && (cond) && (i) < 128)  \
      a.stamp[(i)] = __builtin_amdgcn_s_memtime();                                          \
  } while (0)
#else
#define DSX_STAMP(i) do { } while (0)
#define DSX_STAMP_T(i, cond) do { } while (0)
#endif

template <typename DT> struct Chunk;
template <> struct Chunk<float> { static constexpr int KC = 16; };
template <> struct Chunk<__bf16> { static constexpr int KC = 32; };
template <> struct Chunk<_Float16> { static constexpr int KC = 32; };
// storage kind of an activation tensor as the host passes it (ConvArgs::act_bf16 / out_bf16): 0 fp32, 1 bf16, 2 fp16
template <typename DT> struct Kind;
template <> struct Kind<float> { static constexpr int value = 0; };
template <> struct Kind<__bf16> { static constexpr int value = 1; };
template <> struct Kind<_Float16> { static constexpr int value = 2; };

// Activations live in HBM in the MFMA operand type (fp32 or bf16); everything is staged in 16-byte units:
// Unit<DT>::N consecutive channels of one pixel (4 fp32 / 8 bf16), which is also 16 bytes of the LDS image.
template <typename DT> struct Unit;
template <> struct Unit<float> {
  static constexpr int N = 4;
  static __device__ __forceinline__ void unpack(const uint4 r, float (&v)[4]) {
    v[0] = __builtin_bit_cast(float, r.x); v[1] = __builtin_bit_cast(float, r.y);
    v[2] = __builtin_bit_cast(float, r.z); v[3] = __builtin_bit_cast(float, r.w);
  }
  static __device__ __forceinline__ uint4 pack(const float (&v)[4]) {
    return make_uint4(__builtin_bit_cast(unsigned, v[0]), __builtin_bit_cast(unsigned, v[1]),
                      __builtin_bit_cast(unsigned, v[2]), __builtin_bit_cast(unsigned, v[3]));
  }
};
template <> struct Unit<__bf16> {
  static constexpr int N = 8;
  static __device__ __forceinline__ void unpack(const uint4 r, float (&v)[8]) {
    v[0] = __builtin_bit_cast(float, r.x << 16); v[1] = __builtin_bit_cast(float, r.x & 0xffff0000u);
    v[2] = __builtin_bit_cast(float, r.y << 16); v[3] = __builtin_bit_cast(float, r.y & 0xffff0000u);
    v[4] = __builtin_bit_cast(float, r.z << 16); v[5] = __builtin_bit_cast(float, r.z & 0xffff0000u);
    v[6] = __builtin_bit_cast(float, r.w << 16); v[7] = __builtin_bit_cast(float, r.w & 0xffff0000u);
  }
  static __device__ __forceinline__ uint4 pack(const float (&v)[8]) {
    return make_uint4(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]),
                      pack_bf16x2(v[6], v[7]));
  }
};
template <> struct Unit<_Float16> {
  static constexpr int N = 8;
  static __device__ __forceinline__ void unpack(const uint4 r, float (&v)[8]) {
    v[0] = f16_lo(r.x); v[1] = f16_hi(r.x); v[2] = f16_lo(r.y); v[3] = f16_hi(r.y);
    v[4] = f16_lo(r.z); v[5] = f16_hi(r.z); v[6] = f16_lo(r.w); v[7] = f16_hi(r.w);
  }
  static __device__ __forceinline__ uint4 pack(const float (&v)[8]) {
    return make_uint4(pack_f16x2(v[0], v[1]), pack_f16x2(v[2], v[3]), pack_f16x2(v[4], v[5]), pack_f16x2(v[6], v[7]));
  }
};
// one activation element -> float (scalar fallback paths)
template <typename DT> __device__ __forceinline__ float act_load(const void* p, size_t i) {
  if constexpr (Kind<DT>::value == 1) return __builtin_bit_cast(float, (unsigned)((const unsigned short*)p)[i] << 16);
  else if constexpr (Kind<DT>::value == 2) return (float)((const _Float16*)p)[i];
  else return ((const float*)p)[i];
}
__device__ __forceinline__ float act_load_kind(const void* p, size_t i, int kind) {
  return kind == 1 ? act_load<__bf16>(p, i) : (kind == 2 ? act_load<_Float16>(p, i) : act_load<float>(p, i));
}
// kind: 0 fp32, 1 bf16, 2 fp16
__device__ __forceinline__ void act_store(void* p, size_t i, float v, int kind) {
  if (kind == 1) {
    const __bf16 h = (__bf16)v;
    ((unsigned short*)p)[i] = __builtin_bit_cast(unsigned short, h);
  } else if (kind == 2) {
    ((_Float16*)p)[i] = (_Float16)v;
  } else {
    ((float*)p)[i] = v;
  }
}
// Accumulator layout of every conv kernel here (weights are the MFMA A operand, packed with permuted
// rows, see pack_conv): lane (li = lane & 31, lh = lane >> 5) holds pixel li of its row block, register r
// holds channel 16*lh + r of the wave's 32-channel block -> 16 consecutive channels per lane.
// x[16] -> out (+ optional 16-B vector stores); `n0` = first channel, `valid` = channels that exist
template <bool VEC>
__device__ __forceinline__ void store16(void* out, size_t elem, const float (&x)[16], int kind, int valid) {
  if (VEC) {
    if (kind == 1) {
      uint4* q = (uint4*)((unsigned short*)out + elem);
      q[0] = make_uint4(pack_bf16x2(x[0], x[1]), pack_bf16x2(x[2], x[3]), pack_bf16x2(x[4], x[5]), pack_bf16x2(x[6], x[7]));
      q[1] = make_uint4(pack_bf16x2(x[8], x[9]), pack_bf16x2(x[10], x[11]), pack_bf16x2(x[12], x[13]), pack_bf16x2(x[14], x[15]));
    } else if (kind == 2) {
      uint4* q = (uint4*)((unsigned short*)out + elem);
      q[0] = make_uint4(pack_f16x2(x[0], x[1]), pack_f16x2(x[2], x[3]), pack_f16x2(x[4], x[5]), pack_f16x2(x[6], x[7]));
      q[1] = make_uint4(pack_f16x2(x[8], x[9]), pack_f16x2(x[10], x[11]), pack_f16x2(x[12], x[13]), pack_f16x2(x[14], x[15]));
    } else {
      float4* q = (float4*)((float*)out + elem);
#pragma unroll
      for (int j = 0; j < 4; ++j) q[j] = make_float4(x[4 * j], x[4 * j + 1], x[4 * j + 2], x[4 * j + 3]);
    }
  } else {
#pragma unroll
    for (int r = 0; r < 16; ++r)
      if (r < valid) act_store(out, elem + r, x[r], kind);
  }
}

// LDS accesses of the loader waves go through inline asm: for a ds_read that may alias an LDS-DMA
// destination hipcc would insert s_waitcnt vmcnt(0) and drain the whole DMA ring; ordering is done by
// the counted vmcnt waits (wait_young) instead.
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
static __device__ __forceinline__ void lds_write_b128_asm(unsigned addr, f32x4_t w) {
  asm volatile("ds_write_b128 %0, %1" ::"v"(addr), "v"(w) : "memory");
}

static __device__ __forceinline__ void ws_barrier() {
  __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0) only: own LDS writes visible, DMAs stay in flight
  __builtin_amdgcn_s_barrier();
}
template <int N> static __device__ __forceinline__ void wait_vmcnt() {
  static_assert(N >= 0 && N < 64, "vmcnt is 6 bits");
  __builtin_amdgcn_s_waitcnt((N & 0xF) | ((N >> 4) << 14) | 0x0F70);  // vmcnt(N) only
}


// compile-time loop: f(std::integral_constant<int, 0>) ... f(std::integral_constant<int, N-1>)
template <class F, int... I>
static __device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, I...>) {
  (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F> static __device__ __forceinline__ void static_for(F&& f) {
  static_for_impl(f, std::make_integer_sequence<int, N>{});
}
// MFMA operand fragments of the compute waves are read with explicit ds_read_b128 + counted lgkmcnt waits,
// PF steps ahead of their MFMAs: left to itself the register-starved scheduler puts every read right before
// its use ("ds_read; s_waitcnt lgkmcnt(0); v_mfma"), which exposes the LDS latency on every MFMA.
template <int OFF> static __device__ __forceinline__ void lds_read_frag(f32x4_t& d, unsigned addr) {
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d) : "v"(addr), "n"(OFF) : "memory");
}
static __device__ __forceinline__ void touch_frag(f32x4_t& d, unsigned addr) { asm volatile("" : "+v"(d) : "v"(addr)); }
template <int N> static __device__ __forceinline__ void wait_frags(f32x4_t& a) {
  asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(a) : "n"(N) : "memory");
}
template <int N> static __device__ __forceinline__ void wait_frags(f32x4_t& a, f32x4_t& b) {
  asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(a), "+v"(b) : "n"(N) : "memory");
}
template <int N> static __device__ __forceinline__ void wait_frags(f32x4_t& a, f32x4_t& b, f32x4_t& c, f32x4_t& d) {
  asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "n"(N) : "memory");
}


// one 16-byte fragment pair on the matrix core in the operand type DT (weights = A, pixels = B)
template <typename DT>
__device__ __forceinline__ f32x16 mfma_step(const uint4 w, const f32x4_t px, f32x16 acc) {
  if constexpr (Kind<DT>::value == 1) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, w), __builtin_bit_cast(bf16x8, px), acc, 0, 0, 0);
  } else if constexpr (Kind<DT>::value == 2) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, w), __builtin_bit_cast(f16x8, px), acc, 0, 0, 0);
  } else {
    const float4 af = __builtin_bit_cast(float4, px);
    const float4 bf = __builtin_bit_cast(float4, w);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(bf.x, af.x, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(bf.y, af.y, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(bf.z, af.z, acc, 0, 0, 0);
    return __builtin_amdgcn_mfma_f32_32x32x2f32(bf.w, af.w, acc, 0, 0, 0);
  }
}

// k_conv_ws multiplies with the 16 x 16 MFMA shapes (v_mfma_f32_16x16x32_bf16 / _f16, 16x16x4 f32): the same FLOP per
// cycle, operand bytes and accumulator registers as 32x32x16, but the chip holds a higher clock under them
// (MI355X_MICROARCH.md, DVFS give-back 7; measured here with a probe build before the re-layout: every 3 x 3 launch
// -3.6 .. -4.2 %, the 2000-step loop -4.1 %, profiles/r03_ab_experiments.md).
//   A (weights): lane (i = lane & 15, kq = lane >> 4) = output channel row i, K slice kq (16 bytes);
//   B (pixels):  lane (c, kq) = pixel column c, K slice kq;   D: lane (c, kq), register j = row 4 kq + j of column c.
// The 16-byte K slice of a 64-byte chunk that lane group kq multiplies is slice ws16_slice(kq) = {0, 2, 1, 3}[kq] and
// column c is pixel ws16_pixel(c) of a 16-pixel block -- the b128 read groups of the LDS ({0-3,12-15,20-27}, ...) then
// pair 8 even pixels at one slice with the 8 odd pixels two slices on: 16 distinct 16-byte bank groups for the pixel
// pitches used here (80 / 144 bytes; the row pitches of conv_lds_row keep the parity).
// One fragment per lane pair (ch = 0, 1) covers a 32-channel N block: row i of fragment ch is channel
// 8 (i / 4) + 4 ch + i % 4, so a lane ends up with 8 consecutive channels 8 kq .. 8 kq + 7 of ONE pixel per 16-pixel
// block (registers 8 ph + 4 ch + j of the f32x16: ph = which half of the 32-pixel row block).
__host__ __device__ constexpr int ws16_slice(int kq) { return ((kq & 1) << 1) | (kq >> 1); }
__device__ __forceinline__ int ws16_pixel(int c) { return c < 4 ? 2 * c : (c < 12 ? 2 * (c - 4) + 1 : 2 * (c - 8)); }
// byte offset of lane (i, kq)'s 16 bytes of fragment ch inside the 2 KiB (tap, chunk) pair of pack_conv's stream
// ([half fs][lane (i_old, h)][16 B]: K slice s is half s / 2, lane half s % 2; channel 16 hh + 4 jj + q is row q + 8 jj + 4 hh)
__device__ __forceinline__ int ws16_woff(int lane, int ch) {
  const int i = lane & 15, sl = ws16_slice(lane >> 4);
  const int i_old = (i & 3) + 16 * ((i >> 2) & 1) + 8 * ch + 4 * (i >> 3);
  return (sl >> 1) * 1024 + ((sl & 1) * 32 + i_old) * 16;
}
template <typename DT>
__device__ __forceinline__ f32x4_t mfma16_step(const uint4 w, const f32x4_t px, f32x4_t acc) {
  if constexpr (Kind<DT>::value == 1) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, w), __builtin_bit_cast(bf16x8, px), acc, 0, 0, 0);
  } else if constexpr (Kind<DT>::value == 2) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, w), __builtin_bit_cast(f16x8, px), acc, 0, 0, 0);
  } else {
    const float4 af = __builtin_bit_cast(float4, px);
    const float4 bf = __builtin_bit_cast(float4, w);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(bf.x, af.x, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(bf.y, af.y, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(bf.z, af.z, acc, 0, 0, 0);
    return __builtin_amdgcn_mfma_f32_16x16x4f32(bf.w, af.w, acc, 0, 0, 0);
  }
}
// 8 per-lane registers -> the 16-lane DPP row's total of ONE register per lane: lane i ends up with register
// row8_fold_reg(i) summed over the row (three halving butterfly stages, then lane ^ 8): 7 + 1 adds
__device__ __forceinline__ float row8_fold(const float (&s)[8], int lane) {
  const bool b0 = lane & 1, b1 = lane & 2, b2 = lane & 4;
  float a4[4], a2[2];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const float keep = b0 ? s[k + 4] : s[k], send = b0 ? s[k] : s[k + 4];
    a4[k] = keep + dpp_take<0xB1, 0xF>(0.f, send);                       // lane ^ 1
  }
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const float keep = b1 ? a4[k + 2] : a4[k], send = b1 ? a4[k] : a4[k + 2];
    a2[k] = keep + dpp_take<0x4E, 0xF>(0.f, send);                       // lane ^ 2
  }
  const float keep = b2 ? a2[1] : a2[0], send = b2 ? a2[0] : a2[1];
  float t = dpp_take<0x104, 0x5>(0.f, send);                             // banks 0,2 <- lane + 4
  t = dpp_take<0x114, 0xA>(t, send);                                     // banks 1,3 <- lane - 4
  const float a1 = keep + t;
  return a1 + dpp_take<0x128, 0xF>(0.f, a1);                             // lane ^ 8 (row_ror:8)
}
__device__ __forceinline__ int row8_fold_reg(int lane) { return 4 * (lane & 1) + 2 * ((lane >> 1) & 1) + ((lane >> 2) & 1); }
// x[8] = 8 consecutive channels of one pixel -> out in the storage type (16-byte vector stores)
__device__ __forceinline__ void store8(void* out, size_t elem, const float (&x)[8], int kind) {
  if (kind == 1) {
    *(uint4*)((unsigned short*)out + elem) =
        make_uint4(pack_bf16x2(x[0], x[1]), pack_bf16x2(x[2], x[3]), pack_bf16x2(x[4], x[5]), pack_bf16x2(x[6], x[7]));
  } else if (kind == 2) {
    *(uint4*)((unsigned short*)out + elem) =
        make_uint4(pack_f16x2(x[0], x[1]), pack_f16x2(x[2], x[3]), pack_f16x2(x[4], x[5]), pack_f16x2(x[6], x[7]));
  } else {
    float4* q = (float4*)((float*)out + elem);
    q[0] = make_float4(x[0], x[1], x[2], x[3]);
    q[1] = make_float4(x[4], x[5], x[6], x[7]);
  }
}

// Loader -> compute hand-off of k_conv_ws.  NBUF == 2: two MFMA images, one workgroup barrier per (tile, group) item
// (strict alternation).  NBUF == 3 / 4: three / four images and LDS counters instead of the barrier -- FULL (one per
// loader wave: items converted) and FREE (one per compute wave: items consumed): the loaders may run two items
// ahead, i.e. they convert during the compute waves' epilogue instead of parking at the barrier, and the compute waves
// find the next tile's first groups ready.  Spins are bounded (a lost hand-off gives wrong pixels, never a hung GPU).
// Measured (round 3, A/B inside one gpurun call, three alternations): counters for the 256-pixel 3 x 3 tile only (the
// 64-channel layers of the 128^2 level: two groups per tile, so the loaders used to idle through every epilogue and
// the compute waves then waited 3500 cycles for the second group) -0.9 % on the step, those layers -10 %; for every
// 3 x 3 tile -0.5 % (deep-K layers lose 2 us each to the polling); four buffers no better than three.
#ifndef DSX_WS_NBUF_EXPR
#define DSX_WS_NBUF_EXPR(bm, ks, nb, cpg) ((bm) == 256 && (ks) == 3 ? 3 : 2)
#endif
static constexpr int ws_nbuf(int bm, int ks, int nb, int cpg) { return DSX_WS_NBUF_EXPR(bm, ks, nb, cpg); }
// a counter PER WAVE (four words, read with one ds_read_b128): a single shared counter would let three fast waves
// stand in for a slow one, and the image of the item that one is still reading would be overwritten
static __device__ __forceinline__ void lds_wait_all_ge(unsigned addr, int target, unsigned* timeouts) {
  for (int spin = 0; spin < (1 << 18); ++spin) {   // ~30 ms: a hand-off normally takes microseconds
    f32x4_t v;
    asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
    const int4 c = __builtin_bit_cast(int4, v);
    const int m = min(min(c.x, c.y), min(c.z, c.w));
    if (__builtin_amdgcn_readfirstlane(m) >= target) return;
    __builtin_amdgcn_s_sleep(1);
  }
  // gave up: the pixels of this launch are wrong -- say so (the host reads the counter: dsx_exec_handoff_timeouts)
  if (timeouts != nullptr && (threadIdx.x & 63) == 0) atomicAdd(timeouts, 1u);
}
static __device__ __forceinline__ void lds_signal(unsigned addr, int wave, int count) {   // this wave's word := count, behind its own LDS traffic
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  if ((threadIdx.x & 63) == 0) asm volatile("ds_write_b32 %0, %1" ::"v"(addr + 4u * (unsigned)wave), "v"(count) : "memory");
}

// MB   : 32-row M blocks per wave;  WM x WN waves (WM*WN == 4); every wave owns ONE
//        32-channel N block, so with WM == 1 no weight fragment is loaded twice.
// CPG  : channel chunks staged per barrier ("group"); 1 for 3x3, 2 for 1x1 (few steps per chunk)
// D    : depth of the weight-fragment prefetch ring (global -> VGPR), steps ahead
//
// Addressing is done by the memory pipeline, not the VALU: both operands come through
// buffer_load with a wave-uniform SGPR offset; out-of-range offsets (zero padding, channel
// tails, prefetch past the end) return 0 from the hardware bounds check, so the staging
// code has no predicates.  LDS image: pixel stride PIXB, row pitch `a.lds_row` chosen by the
// host so that ds_read_b128 of a 32-row fragment is bank-conflict-free (see conv_lds_row).
template <typename DT, int MB, int WM, int WN, int KS, int S, int CPG, int D, int MAX_IT>
__global__ __launch_bounds__(256, ((KS == 3 && CPG == 2) ? 2   // two-chunk 3 x 3 variant: up to 11 staging units per thread in flight (spills at 168 VGPRs)
                                  : (MB == 1 ? (S == 2 ? 3 : 4) : (MB == 2 ? 3 : (MB == 4 ? 2 : 1))))) void k_conv_mfma(const ConvArgs a) {
  constexpr int KC = Chunk<DT>::KC;
  constexpr int CPU = Unit<DT>::N;            // channels per 16-byte staging unit
  constexpr int ES = (int)sizeof(DT);         // bytes per activation element in HBM
  constexpr int UPP = KC / CPU;               // staging units per pixel per chunk (4)
  constexpr int UPG = UPP * CPG;              // ... per group
  constexpr int UPG_LOG2 = UPG == 16 ? 4 : (UPG == 8 ? 3 : 2);
  constexpr int UB = 16;                      // LDS bytes per staging unit
  constexpr int PIXB = 64 * CPG + 16;         // LDS bytes per patch pixel
  constexpr int TAPS = KS * KS;
  constexpr int PAD = KS / 2;
  constexpr int NSTEP = CPG * TAPS * 2;       // MFMA steps per group: (chunk, tap, 32-B half)
  constexpr bool IS_BF16 = sizeof(DT) == 2;
  static_assert(NSTEP % D == 0, "ring depth must divide the steps per group");
  static_assert(WM * WN == 4 && UPP == 4, "4 waves; 64-byte chunks");

  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int li = lane & 31, lh = lane >> 5;

  const int TW = 1 << a.tw_log2, TH = 1 << a.th_log2;
  const int PW = (TW - 1) * S + KS;
  const int PH = (TH - 1) * S + KS;
  const int PPI = PH * PW;                       // patch pixels per image
  const int PP = PPI << a.tb_log2;               // patch pixels per tile
  const int RB = a.lds_row;                      // LDS bytes per patch row
  const int BUFB = (PH << a.tb_log2) * RB;       // one LDS buffer
  const bool multi_img = a.tb_log2 != 0;

  // ---- tile coordinates: blockIdx.x = (split * n_tiles + nt) * m_tiles + mt
  const int mt = blockIdx.x % a.m_tiles;
  const int rest = blockIdx.x / a.m_tiles;
  const int nt = rest % a.n_tiles;
  const int split = rest / a.n_tiles;
  const int txi = mt % a.tiles_x;
  const int tyi = (mt / a.tiles_x) % a.tiles_y;
  const int bg = mt / (a.tiles_x * a.tiles_y);
  const int oy0 = tyi << a.th_log2, ox0 = txi << a.tw_log2, b0 = bg << a.tb_log2;
  const int iy0 = oy0 * S - PAD, ix0 = ox0 * S - PAD;
  const int Hi = a.up ? a.Hs * 2 : a.Hs;
  const int Wi = a.up ? a.Ws * 2 : a.Ws;
  const int C = a.C0 + a.C1;
  const int kgroups = a.kchunks / CPG;
  const int g0 = split * a.groups_per_split;
  const int g1 = min(kgroups, g0 + a.groups_per_split);

  // ---- staging plan: source pixel and LDS slot of each of this thread's units.
  // Unit u = tid + 256*it covers patch pixel u >> UPG_LOG2; (tb, py, px) advance by a fixed stride
  // per `it`, so only the first unit needs integer divisions.
  const int nunits = PP << UPG_LOG2;
  const int cvg = tid & (UPG - 1);  // this thread's unit inside a group (same for all its units)
  int soff[MAX_IT];   // source pixel index, or -1 (zero padding / outside batch / no such unit)
  int loff[MAX_IT];   // LDS byte offset of the unit inside a buffer, or -1
  int simg[MAX_IT];   // image index (GroupNorm scale/shift lookup when a tile spans images)
  {
    constexpr int PSTEP = 256 >> UPG_LOG2;           // pixels between consecutive units of a thread
    const int pix0 = tid >> UPG_LOG2;
    int tb = pix0 / PPI;
    int rem = pix0 - tb * PPI;
    int py = rem / PW;
    int px = rem - py * PW;
    const int dpy = PSTEP / PW, dpx = PSTEP - dpy * PW;   // uniform
#pragma unroll
    for (int it = 0; it < MAX_IT; ++it) {
      int so = -1, lo = -1;
      const int b = b0 + tb;
      if (tid + it * 256 < nunits) {
        const int iy = iy0 + py, ix = ix0 + px;
        lo = (tb * PH + py) * RB + px * PIXB + cvg * UB;
        if (b < a.B && iy >= 0 && iy < Hi && ix >= 0 && ix < Wi) {
          const int sy = a.up ? (iy >> 1) : iy;
          const int sx = a.up ? (ix >> 1) : ix;
          so = (b * a.Hs + sy) * a.Ws + sx;
        }
      }
      soff[it] = so;
      loff[it] = lo;
      simg[it] = b;
      px += dpx; py += dpy;
      if (px >= PW) { px -= PW; py += 1; }
      while (py >= PH) { py -= PH; tb += 1; }
    }
  }

  // ---- A-fragment LDS addresses: one per (row block, tap row); tap column / chunk / half are immediates
  int abase[MB][KS];
#pragma unroll
  for (int mb = 0; mb < MB; ++mb) {
    const int m = (wm * MB + mb) * 32 + li;
    const int tx = m & (TW - 1);
    const int ty = (m >> a.tw_log2) & (TH - 1);
    const int tb = m >> (a.tw_log2 + a.th_log2);
#pragma unroll
    for (int dy = 0; dy < KS; ++dy)
      abase[mb][dy] = (tb * PH + ty * S + dy) * RB + tx * S * PIXB + lh * 16;
  }

  // ---- buffer descriptors (wave-uniform): activations (two sources) and this wave's weight stream
  const __amdgpu_buffer_rsrc_t rs0 = __builtin_amdgcn_make_buffer_rsrc(
      (void*)a.src0, 0, (int)min((long long)a.B * a.Hs * a.Ws * a.C0 * ES, 0x7fffffffLL), 0x00020000);
  const __amdgpu_buffer_rsrc_t rs1 = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(a.src1 ? a.src1 : a.src0), 0,
      a.src1 ? (int)min((long long)a.B * a.Hs * a.Ws * a.C1 * ES, 0x7fffffffLL) : 0, 0x00020000);
  int blk = nt * WN + wn;
  if (blk >= a.nblocks) blk = a.nblocks - 1;  // results of a clamped block are never stored
  const long long wblock = (long long)kgroups * (NSTEP * 1024);  // bytes of one N block's fragment stream
  const __amdgpu_buffer_rsrc_t rsw = __builtin_amdgcn_make_buffer_rsrc(
      (void*)((const unsigned char*)a.wpack + (size_t)blk * wblock), 0, (int)wblock, 0x00020000);
  const int wlane = lane * 16;
  auto load_b = [&](int q) -> uint4 {  // step q of the stream; past the end -> zeros (bounds check)
    return __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rsw, wlane, q * 1024, 0));
  };
  uint4 bq[D];
#pragma unroll
  for (int j = 0; j < D; ++j) bq[j] = load_b(g0 * NSTEP + j);

  f32x16 acc[MB];
#pragma unroll
  for (int mb = 0; mb < MB; ++mb)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[mb][r] = 0.0f;

  uint4 stg[MAX_IT];   // raw 16-byte units (CPU channels in the storage type)
#pragma unroll
  for (int it = 0; it < MAX_IT; ++it) stg[it] = make_uint4(0u, 0u, 0u, 0u);

  // issue the global loads of group g into registers
  auto stage_load = [&](int g) {
    const int c = g * (CPG * KC) + cvg * CPU;
#ifdef DSX_DIAG
    if (a.ablate & 2) return;   // timing experiments only (DSX_ABLATE, diagnostic build): skip activation loads
#endif
    if (a.stage_mode == 0) {
      // fast path (both channel counts multiples of the group width): the whole workgroup reads ONE
      // source in this group -> wave-uniform descriptor, offsets by the memory pipeline, OOB -> 0
      const bool first = g * (CPG * KC) < a.C0;            // uniform
      const int cs = first ? a.C0 : a.C1;
      const unsigned coff = c < C ? (unsigned)((first ? c : c - a.C0) * ES) : 0x80000000u;
      if (first) {
#pragma unroll
        for (int it = 0; it < MAX_IT; ++it) {
          const unsigned vo = soff[it] >= 0 ? (unsigned)soff[it] * (unsigned)(cs * ES) + coff : 0x80000000u;
          stg[it] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rs0, vo, 0, 0));
        }
      } else {
#pragma unroll
        for (int it = 0; it < MAX_IT; ++it) {
          const unsigned vo = soff[it] >= 0 ? (unsigned)soff[it] * (unsigned)(cs * ES) + coff : 0x80000000u;
          stg[it] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rs1, vo, 0, 0));
        }
      }
    } else if (a.stage_mode == 1) {
      // channel counts multiples of the unit but a group may straddle the two sources: per-lane source select
#pragma unroll
      for (int it = 0; it < MAX_IT; ++it) {
        uint4 v = make_uint4(0u, 0u, 0u, 0u);
        const int so = soff[it];
        if (so >= 0) {
          if (c < a.C0) v = *(const uint4*)((const DT*)a.src0 + (size_t)so * a.C0 + c);
          else if (c < C) v = *(const uint4*)((const DT*)a.src1 + (size_t)so * a.C1 + (c - a.C0));
        }
        stg[it] = v;
      }
    } else {
#pragma unroll
      for (int it = 0; it < MAX_IT; ++it) {
        float e[CPU];
#pragma unroll
        for (int j = 0; j < CPU; ++j) e[j] = 0.f;
        const int so = soff[it];
        if (so >= 0) {
#pragma unroll
          for (int j = 0; j < CPU; ++j) {
            const int cc = c + j;
            if (cc < a.C0) e[j] = act_load<DT>(a.src0, (size_t)so * a.C0 + cc);
            else if (cc < C) e[j] = act_load<DT>(a.src1, (size_t)so * a.C1 + (cc - a.C0));
          }
        }
        stg[it] = Unit<DT>::pack(e);   // exact: the values came from the storage type
      }
    }
  };

  // GroupNorm affine + Swish in registers, convert, park in LDS buffer `buf`
  auto stage_store = [&](int g, int buf) {
    const int c = g * (CPG * KC) + cvg * CPU;
    unsigned char* dst = lds + buf * BUFB;
    const bool has_gn = a.gn_scale != nullptr && c < C;
    float sc[CPU], sh[CPU];
#pragma unroll
    for (int j = 0; j < CPU; ++j) { sc[j] = 1.f; sh[j] = 0.f; }
    auto load_affine = [&](int b) {
      const size_t gi = (size_t)b * C + c;
      if (!(a.stage_mode == 2)) {
#pragma unroll
        for (int q = 0; q < CPU / 4; ++q) {
          const float4 s4 = *(const float4*)(a.gn_scale + gi + 4 * q);
          const float4 h4 = *(const float4*)(a.gn_shift + gi + 4 * q);
          sc[4 * q] = s4.x; sc[4 * q + 1] = s4.y; sc[4 * q + 2] = s4.z; sc[4 * q + 3] = s4.w;
          sh[4 * q] = h4.x; sh[4 * q + 1] = h4.y; sh[4 * q + 2] = h4.z; sh[4 * q + 3] = h4.w;
        }
      } else {
#pragma unroll
        for (int j = 0; j < CPU; ++j) {
          sc[j] = (c + j < C) ? a.gn_scale[gi + j] : 0.f;
          sh[j] = (c + j < C) ? a.gn_shift[gi + j] : 0.f;
        }
      }
    };
    if (has_gn && !multi_img) load_affine(b0);  // one image per tile: same (b, c) for every unit
#pragma unroll
    for (int it = 0; it < MAX_IT; ++it) {
      if (loff[it] >= 0) {
        float v[CPU];
        Unit<DT>::unpack(stg[it], v);
        if (soff[it] >= 0 && c < C && !DSX_ABLATED(1)) {   // padding pixels stay exactly 0 (padded AFTER the activation)
          if (has_gn) {
            if (multi_img) load_affine(simg[it]);
            affine_vec(v, sc, sh);
          }
          if (a.swish) swish_vec(v);
          if ((a.stage_mode == 2)) {  // channels past C inside the last unit must stay 0
#pragma unroll
            for (int j = 1; j < CPU; ++j) if (c + j >= C) v[j] = 0.f;
          }
        }
        *(uint4*)(dst + loff[it]) = Unit<DT>::pack(v);
      }
    }
  };

  // ---- main loop over channel groups (double-buffered LDS, one barrier per group)
  DSX_STAMP(0);
  stage_load(g0);
  DSX_STAMP(1);
  stage_store(g0, 0);
  DSX_STAMP(2);
  __syncthreads();
  DSX_STAMP(3);

  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)lds;  // LDS byte address of lds[0]
  constexpr int PF = 1;   // operand fragments are read one step ahead of their MFMAs (see lds_read_frag)
  for (int g = g0; g < g1; ++g) {
    const bool more = (g + 1) < g1;
    const int qbase = g * NSTEP;

    if (more) stage_load(g + 1);
    DSX_STAMP(8 + 4 * (g - g0));

    unsigned aaddr[MB][KS];
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
      for (int dy = 0; dy < KS; ++dy) aaddr[mb][dy] = lds0 + ((g - g0) & 1) * BUFB + abase[mb][dy];
    f32x4_t fb[PF + 1][MB];
    auto read_step = [&](auto sc) {
      constexpr int s = decltype(sc)::value;
      constexpr int kTapSteps = TAPS * 2;
      constexpr int cg = s / kTapSteps, tap = (s >> 1) % TAPS, fs = s & 1;
      constexpr int dy = tap / KS, dx = tap % KS;
      constexpr int imm = dx * PIXB + cg * 64 + fs * 32;
#pragma unroll
      for (int mb = 0; mb < MB; ++mb) lds_read_frag<imm>(fb[s % (PF + 1)][mb], aaddr[mb][dy]);
    };
    static_for<PF>(read_step);
    static_for<NSTEP>([&](auto sc) {
      constexpr int s = decltype(sc)::value;
      const uint4 bcur = bq[s % D];
      bq[s % D] = load_b(qbase + s + D);
      if constexpr (s + PF < NSTEP) read_step(std::integral_constant<int, s + PF>{});
      constexpr int ahead = (NSTEP - 1 - s < PF ? NSTEP - 1 - s : PF) * MB;   // younger reads that may stay in flight
      constexpr int cb = s % (PF + 1);
      static_assert(PF * MB <= 15, "lgkmcnt is 4 bits");
      if constexpr (MB == 1) wait_frags<ahead>(fb[cb][0]);
      else if constexpr (MB == 2) wait_frags<ahead>(fb[cb][0], fb[cb][1]);
      else if constexpr (MB == 4) wait_frags<ahead>(fb[cb][0], fb[cb][1], fb[cb][2], fb[cb][3]);
      else {
        wait_frags<ahead>(fb[cb][0], fb[cb][1], fb[cb][2], fb[cb][3]);   // the wait; the rest only need the ordering
#pragma unroll
        for (int mb = 4; mb < MB; ++mb) asm volatile("" : "+v"(fb[cb][mb]));
      }
#pragma unroll
      for (int mb = 0; mb < MB; ++mb) {
        // weights as the A operand, pixels as B: the accumulator then holds, per lane, one pixel's channels
        acc[mb] = mfma_step<DT>(bcur, fb[cb][mb], acc[mb]);
      }
    });

    DSX_STAMP(9 + 4 * (g - g0));
    if (more && !DSX_ABLATED(32)) stage_store(g + 1, (g + 1 - g0) & 1);
    DSX_STAMP(10 + 4 * (g - g0));
    __syncthreads();
    DSX_STAMP(11 + 4 * (g - g0));
  }
  DSX_STAMP(4);

  // ---- epilogue.  Accumulator layout (operands swapped): column = lane&31 = pixel of the row block,
  // register r = channel 16*(lane>>5) + r of this wave's 32-channel block (the weight rows are packed
  // permuted): 16 consecutive channels per lane -> 16-byte bias / FiLM / residual loads and NHWC stores.
  // Split-K slices write raw partial sums to their fp32 slab.  The GroupNorm statistics of the tensor being
  // written (sum, sum of squares per channel over this wave's pixels) are accumulated in the same pass.
  if (DSX_ABLATED(16)) return;
  const int nbase = (nt * WN + wn) * 32 + 16 * lh;
  const bool partial = a.ksplit > 1;
  const int okind = (IS_BF16 && a.out_bf16) ? Kind<DT>::value : 0;   // storage kind of `out`
  const bool obf = okind != 0;
  void* outp = partial ? (void*)((float*)a.out + (size_t)split * a.slab_stride) : a.out;
  const int oal = obf ? 7 : 3, ral = IS_BF16 ? 7 : 3;   // 16-byte alignment of rows, in elements
  const bool vec = (a.Cout & 15) == 0 && (a.out_ld & oal) == 0 && (!a.resid || (a.resid_ld & ral) == 0);
  // fused statistics are compiled only into the small-MB tiles: in the MB >= 4 tiles their registers
  // would cost a wave of occupancy (the host then falls back to k_chan_stats)
  constexpr bool STATS = MB <= 2;
  const bool do_stats = STATS && a.stat_part != nullptr;   // host guarantees: vec, one image per tile, no split-K
  float s1[16], s2[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) { s1[r] = 0.f; s2[r] = 0.f; }
#pragma unroll
  for (int mb = 0; mb < MB; ++mb) {
    // one row block at a time: keeps the epilogue's loads from being hoisted across blocks,
    // which would cost registers (and a wave of occupancy) in the main loop's favour
    __builtin_amdgcn_sched_barrier(0);
    const int m = (wm * MB + mb) * 32 + li;
    const int tx = m & (TW - 1);
    const int ty = (m >> a.tw_log2) & (TH - 1);
    const int b = b0 + (m >> (a.tw_log2 + a.th_log2));
    if (b >= a.B || nbase >= a.Cout) continue;
    const size_t opix = ((size_t)b * a.Ho + (oy0 + ty)) * a.Wo + (ox0 + tx);
    float x[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) x[r] = acc[mb][r];
    if (vec) {
      if (!partial) {
        if (a.bias) {
#pragma unroll
          for (int j = 0; j < 4; ++j) { const float4 t = *(const float4*)(a.bias + nbase + 4 * j); x[4 * j] += t.x; x[4 * j + 1] += t.y; x[4 * j + 2] += t.z; x[4 * j + 3] += t.w; }
        }
        if (a.film) {
#pragma unroll
          for (int j = 0; j < 4; ++j) { const float4 t = *(const float4*)(a.film + (size_t)b * a.film_bs + nbase + 4 * j); x[4 * j] += t.x; x[4 * j + 1] += t.y; x[4 * j + 2] += t.z; x[4 * j + 3] += t.w; }
        }
        if (a.resid) {
          const DT* rp = (const DT*)a.resid + opix * a.resid_ld + nbase;
#pragma unroll
          for (int q = 0; q < 16 / CPU; ++q) {
            float t[CPU];
            Unit<DT>::unpack(*(const uint4*)(rp + CPU * q), t);
#pragma unroll
            for (int j = 0; j < CPU; ++j) x[CPU * q + j] += t[j];
          }
        }
      }
      store16<true>(outp, opix * a.out_ld + nbase, x, okind, 16);
      if (do_stats) {
#pragma unroll
        for (int r = 0; r < 16; ++r) { s1[r] += x[r]; s2[r] += x[r] * x[r]; }
      }
    } else {
      const int valid = min(16, a.Cout - nbase);
      if (!partial) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          if (r >= valid) break;
          if (a.bias) x[r] += a.bias[nbase + r];
          if (a.film) x[r] += a.film[(size_t)b * a.film_bs + nbase + r];
          if (a.resid) x[r] += act_load<DT>(a.resid, opix * a.resid_ld + nbase + r);
        }
      }
      store16<false>(outp, opix * a.out_ld + nbase, x, okind, valid);
    }
  }
  DSX_STAMP(5);

  // ---- statistics: reduce over the 32 pixel lanes of each half-wave (DPP butterfly inside the 16-lane
  // rows, one cross-row exchange; fixed order -> bitwise reproducible).  One partial row per (tile, wm):
  // [b][chunk][channel][2] fp32.
  if (do_stats) {
    float w1 = row16_fold(s1, lane), w2 = row16_fold(s2, lane);
    w1 += __shfl_xor(w1, 16, 64);
    w2 += __shfl_xor(w2, 16, 64);
    const int chunk = (tyi * a.tiles_x + txi) * WM + wm;   // partial row inside the image
    const int nch = a.tiles_x * a.tiles_y * WM;
    const int n = nbase + row16_fold_reg(li);
    if (li < 16 && n < a.Cout) {
      float* p = a.stat_part + (((size_t)b0 * nch + chunk) * a.Cout + n) * 2;
      p[0] = w1; p[1] = w2;
    }
  }
  DSX_STAMP(6);
}

// ===========================================================================================
// Warp-specialised, persistent variant (stride 1, sources aligned to the channel group).
//
// Why: inside one wave the activation stream (HBM, ~3 us under load) and the weight-fragment stream
// (L2, ~0.6 us) share ONE in-order vmcnt queue, so neither can be prefetched deeper than the other
// allows, and every phase of k_conv_mfma (load wait, GN+Swish VALU, MFMA, epilogue) ends up serialised.
// Here a workgroup is 8 waves:
//   waves 4-7  LOADERS : LDS-DMA (buffer_load ... lds, no VGPRs) of the raw halo patch P groups ahead
//                        into a ring; units are read back one iteration before use, then GroupNorm
//                        affine + Swish + convert and the MFMA image of the NEXT group; their vmcnt
//                        only ever counts activation DMAs.  s_setprio 1 (see below).
//   waves 0-3  COMPUTE : weight-fragment ring + MFMA on the CURRENT group's image, epilogue + fused
//                        statistics at tile ends; their vmcnt only counts weight / epilogue loads.
// Both pipes of a SIMD stay busy (one loader + one compute wave per SIMD: VALU beside MFMA), the
// workgroup is persistent over M tiles of one N tile (the weight stream stays in L2, the pipelines run
// on across tile boundaries, per-workgroup setup is paid once), and there is one raw s_barrier per
// group (never __syncthreads: its vmcnt(0) would drain the DMA ring).
// ===========================================================================================
template <typename DT, int MB, int WM, int WN, int NB, int KS, int CPG, int D, int NIT, int P, int LW>
__global__ __launch_bounds__(256 + 64 * LW, 1) void k_conv_ws(const ConvArgs a) {
  constexpr int LT = 64 * LW;                   // loader threads
  constexpr int S = 1;
  constexpr int KC = Chunk<DT>::KC;
  constexpr int CPU = Unit<DT>::N;              // channels per 16-byte unit (HBM, raw ring and LDS image alike)
  constexpr int ES = (int)sizeof(DT);
  constexpr int UPP = KC / CPU;
  constexpr int UPG = UPP * CPG;
  constexpr int UPG_LOG2 = UPG == 16 ? 4 : (UPG == 8 ? 3 : 2);
  constexpr int UB = 16;
  constexpr int PIXB = 64 * CPG + 16;
  constexpr int TAPS = KS * KS;
  constexpr int PAD = KS / 2;
  constexpr int NSTEP = CPG * TAPS * 2;
  [[maybe_unused]] constexpr bool IS_BF16 = sizeof(DT) == 2;
  constexpr int RAWB = NIT * LT * 16;           // one raw ring slot
  constexpr int NSLOT = P + 1;
  static_assert(D <= NSTEP ? NSTEP % D == 0 : (D % NSTEP == 0 && D / NSTEP <= 8), "ring depth divides the steps per group, or is a multiple of them");
  static_assert(WM * WN == 4 && P * NIT < 64, "4 compute waves; vmcnt is 6 bits");

  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];

  const int tid = threadIdx.x;
  DSX_STAMP_T(63, tid == 0);                    // kernel entry (diagnostic builds): stamp 0 - stamp 63 = the start-up ramp
  const int lane = tid & 63;
  const int wave8 = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool loader = wave8 >= 4;
  const int wave = loader ? wave8 - 4 : wave8;  // index inside the role
  const int ltid = loader ? tid - 256 : tid;
  const int c16 = lane & 15, kq = lane >> 4;     // compute waves: MFMA column (pixel) / K slice and output row group (see mfma16_step)

  const int TW = 1 << a.tw_log2, TH = 1 << a.th_log2;
  const int PW = (TW - 1) * S + KS;
  const int PH = (TH - 1) * S + KS;
  const int PPI = PH * PW;
  const int PP = PPI << a.tb_log2;
  const int RB = a.lds_row;
  const int BUFB = (PH << a.tb_log2) * RB;
  // [image 0 .. NBUF-1][GroupNorm scale/shift of three tiles' images][raw ring][FULL, FREE counters]
  constexpr int NBUF = ws_nbuf(32 * MB * WM, KS, NB, CPG);
  static_assert(NBUF >= 2 && NBUF <= 4, "two images and a barrier, or three / four and counters");
  const int C = a.C0 + a.C1;
  const int AFFB = a.has_gn ? ((2 * C * 4 + 15) & ~15) : 0;     // bytes of one tile's {scale[C], shift[C]}
  float* const aff_base = (float*)(lds + NBUF * BUFB);
  unsigned char* const raw_base = lds + NBUF * BUFB + 3 * AFFB;
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)lds;  // LDS byte address of lds[0]
  const unsigned full_addr = lds0 + NBUF * BUFB + 3 * AFFB + NSLOT * RAWB;   // FULL[4 loader waves], FREE[4 compute waves] at +16 (NBUF == 3)
  const int G = a.kchunks / CPG;                // channel groups per tile (no split-K here; host: G >= 2, TB == 1)

  // persistent work list: this workgroup owns N tile `nt` and M tiles p, p+wpn, ...
  // XCD-aware: workgroups are dealt round-robin over the 8 XCDs (blockIdx % 8 labels the L2 they
  // share), so an N tile's workgroups are pinned to as few XCDs as possible and every L2 only
  // streams its own slice of the weights (placement affects speed only, never results).
  const int wpn = a.ws_wg_per_n;
  int nt, p0;
  {
    // (no integer divisions here or below: the host passes quotients and fastdiv magics, see ConvArgs::ws_map)
    const int bid = blockIdx.x, xcd = bid & 7, k = bid >> 3;
    if (a.ws_map == 0) {                         // n_tiles in {1, 2, 4, 8} and wpn a multiple of 8 / n_tiles
      const int lg = a.ws_nt_log2, band = xcd >> lg;
      nt = xcd & ((1 << lg) - 1);
      // each XCD owns a contiguous band of M tiles (neighbouring tiles share halo rows in its L2) ...
      if (a.xcd_bands) p0 = band * (wpn >> (3 - lg)) + k;
      else p0 = (k << (3 - lg)) + band;          // ... or tiles dealt round-robin (DSX_XCD_BANDS=0)
    } else if (a.ws_map == 1) {                  // n_tiles a multiple of 8
      const int kq = (int)fastdiv((unsigned)k, a.mg_per);
      nt = xcd + 8 * (k - kq * a.ws_per);
      p0 = kq;
    } else if (a.ws_map == 3) {                  // 1 x 1 convs: XCD x owns the M tiles p = x (mod 8) for EVERY N tile -- their
      const int j = (int)fastdiv((unsigned)k, a.mg_per);   // weights are small, the input is what the L2s would otherwise
      nt = k - j * a.ws_per;                     // all fetch; consecutive workgroups of an XCD share the M tile
      p0 = (j << 3) + xcd;                       // (host: ws_per = n_tiles, wpn a multiple of 8)
    } else {
      nt = (int)fastdiv((unsigned)bid, a.mg_wpn);
      p0 = bid - nt * wpn;
    }
  }
  // (m_tiles - p0 + wpn - 1) / wpn; one tile per workgroup is the common case
  const int ntile = wpn >= a.m_tiles ? (p0 < a.m_tiles ? 1 : 0)
                                     : (a.ws_bigdiv ? (a.m_tiles - p0 + wpn - 1) / wpn
                                                                : (int)fastdiv((unsigned)(a.m_tiles - p0 + wpn - 1), a.mg_wpn));
  const int total = ntile * G;                  // (tile, group) items, in order

  if (loader) {
    // ======================================================================= LOADER WAVES
    // The loader waves are dispatched after the compute waves, and vector issue on a SIMD is arbitrated by
    // priority, then age: as the younger wave their GroupNorm/Swish VALU stream only gets the slots the MFMA
    // wave leaves over, and they become the critical path.  Static priority for the whole kernel.
    __builtin_amdgcn_s_setprio(DSX_LOADER_PRIO_EXPR);
    // Per-thread, tile-invariant description of its NIT units: LDS image offset, source pixel offset
    // relative to the tile's origin pixel, and which patch borders the unit lies on.  Per tile only the
    // origin offset and four "tile touches the image border" flags change (no divisions, no per-unit
    // bounds arithmetic): zero padding is exactly the border units of border tiles.
    struct TilePos { int tx, ty, b; };
    const int adv_x = a.ws_adv_x, adv_y = a.ws_adv_y, adv_b = a.ws_adv_b;
    auto tile_advance = [&](TilePos& t) __attribute__((always_inline)) {
      t.tx += adv_x;
      if (t.tx >= a.tiles_x) { t.tx -= a.tiles_x; t.ty += 1; }
      t.ty += adv_y;
      if (t.ty >= a.tiles_y) { t.ty -= a.tiles_y; t.b += 1; }
      t.b += adv_b;
    };
    auto tile_flags = [&](const TilePos& t) __attribute__((always_inline)) -> int {   // bit0 top, bit1 bottom, bit2 left, bit3 right
      return (PAD == 0) ? 0
                        : ((t.ty == 0 ? 1 : 0) | (t.ty == a.tiles_y - 1 ? 2 : 0) | (t.tx == 0 ? 4 : 0) |
                           (t.tx == a.tiles_x - 1 ? 8 : 0));
    };
    // source pixel index of the patch origin's *output* pixel (oy0, ox0); `rel` is added to it
    auto tile_base = [&](const TilePos& t) __attribute__((always_inline)) -> int {
      const int oy0 = t.ty << a.th_log2, ox0 = t.tx << a.tw_log2;   // even whenever a.up (TW, TH >= 2)
      return (t.b * a.Hs + (a.up ? oy0 >> 1 : oy0)) * a.Ws + (a.up ? ox0 >> 1 : ox0);
    };
    const int nunits = PP << UPG_LOG2;
    const int cvg = ltid & (UPG - 1);
    int loff[NIT];            // LDS image offset of each unit, -1: no such unit
    int rel[NIT];             // source pixel offset relative to tile_base
    int edge[NIT];            // border bits of the unit's patch pixel
    {
      constexpr int PSTEP = LT >> UPG_LOG2;
      const int pix0 = ltid >> UPG_LOG2;
      static_assert(PSTEP < 65536, "fastdiv range");
      int py = (int)fastdiv((unsigned)pix0, a.mg_pw);
      int px = pix0 - py * PW;
      const int dpy = a.ws_dpy, dpx = a.ws_dpx;   // PSTEP / PW and the remainder
#pragma unroll
      for (int it = 0; it < NIT; ++it) {
        const bool have = ltid + it * LT < nunits;
        loff[it] = have ? py * RB + px * PIXB + cvg * UB : -1;
        const int ry = py - PAD, rx = px - PAD;   // relative to the tile's first output pixel (input grid)
        // nearest x2 upsample: source = floor(input / 2); the origin is even, so floor((o + r) / 2) = o/2 + (r >> 1)
        rel[it] = a.up ? (ry >> 1) * a.Ws + (rx >> 1) : ry * a.Ws + rx;
        edge[it] = (PAD == 0) ? 0 : ((py == 0 ? 1 : 0) | (py == PH - 1 ? 2 : 0) | (px == 0 ? 4 : 0) | (px == PW - 1 ? 8 : 0));
        px += dpx; py += dpy;
        if (px >= PW) { px -= PW; py += 1; }
      }
    }

    const __amdgpu_buffer_rsrc_t rs0 = __builtin_amdgcn_make_buffer_rsrc(
        (void*)a.src0, 0, (int)min((long long)a.B * a.Hs * a.Ws * a.C0 * ES, 0x7fffffffLL), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs1 = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(a.src1 ? a.src1 : a.src0), 0,
        a.src1 ? (int)min((long long)a.B * a.Hs * a.Ws * a.C1 * ES, 0x7fffffffLL) : 0, 0x00020000);

    TilePos posI;                                                                // tile being issued
    {
      const int q1 = (int)fastdiv((unsigned)p0, a.mg_tiles_x);                  // p0 / tiles_x
      posI.b = (int)fastdiv((unsigned)p0, a.mg_per_img);
      posI.tx = p0 - q1 * a.tiles_x;
      posI.ty = q1 - posI.b * a.tiles_y;
    }
    TilePos posC = posI;                                                         // tile being consumed
    int baseI = tile_base(posI), flagsI = tile_flags(posI), flagsC = flagsI;
    int gI = 0;               // group of the next item to issue
    int tiC = 0;              // tile of the next item to consume
    int gC = 0;               // its group
    int affslot = 0;          // tiC % 3: LDS slot of that tile's scale/shift

    // DMA the raw patch of item (tiI, gI) into ring slot `slot`; every wave issues exactly NIT instructions
    auto issue = [&](int slot) __attribute__((always_inline)) {
      const int c = gI * (CPG * KC) + cvg * CPU;
      const bool first = gI * (CPG * KC) < a.C0;            // uniform: a group never straddles the sources
      const int cs = first ? a.C0 : a.C1;
      const unsigned coff = c < C ? (unsigned)((first ? c : c - a.C0) * ES) : 0x80000000u;
      unsigned char* dst = raw_base + slot * RAWB + wave * 1024;
#pragma unroll
      for (int it = 0; it < NIT; ++it) {
        const bool ok = loff[it] >= 0 && (edge[it] & flagsI) == 0;
        const unsigned vo = ok ? (unsigned)(baseI + rel[it]) * (unsigned)(cs * ES) + coff : 0x80000000u;
        auto ldst = (__attribute__((address_space(3))) void*)(dst + it * (LT * 16));
        if (first) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs0, ldst, 16, vo, 0, 0, 0);
        else __builtin_amdgcn_raw_ptr_buffer_load_lds(rs1, ldst, 16, vo, 0, 0, 0);
      }
      if (++gI == G) {
        gI = 0;
        tile_advance(posI);
        baseI = tile_base(posI); flagsI = tile_flags(posI);
      }
    };
    // Item (tiC, gC) goes raw slot -> registers (fetch) -> GroupNorm affine + Swish -> MFMA image (convert).
    // The two halves run one iteration apart: the LDS reads of item v+2 are issued before the barrier of
    // iteration v and are long complete when iteration v+1 converts them (the LDS queue is busy with the
    // compute waves' operand reads, so a read-then-wait inside one iteration costs hundreds of cycles).
    constexpr int NA = CPU / 4;          // 16-byte pieces of scale (and of shift) per unit
    f32x4_t av[2 * NA], rv[NIT];
    int cF = 0, flagsF = 0;              // channel offset / border flags of the fetched item
    bool gnF = false;
    // fetch: scale/shift (parked in LDS by the compute waves -- an ordinary global load here would make the
    // compiler wait vmcnt(0) and drain the DMA ring) and the raw units of item (tiC, gC); advances (tiC, gC)
    auto fetch = [&](int slot) __attribute__((always_inline)) {
      cF = gC * (CPG * KC) + cvg * CPU;
      flagsF = flagsC;
      gnF = a.gn_scale != nullptr && cF < C;
      if constexpr (NBUF >= 3) {
        // the scale/shift slot of tile t (t >= 3) is written by the compute waves' epilogue of tile t - 3, which is
        // over once they have consumed the first item of tile t - 2 (with the per-item barrier of NBUF == 2 the
        // three-tile lead alone guarantees this; running ahead, the loaders ask)
        if (gnF && gC == 0 && tiC >= 3) lds_wait_all_ge(full_addr + 16, (tiC - 2) * G + 1, a.handoff_timeouts);
      }
      const unsigned src = lds0 + NBUF * BUFB + 3 * AFFB + slot * RAWB + ltid * 16;
      if (gnF) {
        const unsigned af = lds0 + NBUF * BUFB + affslot * AFFB;   // slot tiC % 3
#pragma unroll
        for (int q = 0; q < NA; ++q) {
          asm volatile("ds_read_b128 %0, %1" : "=v"(av[q]) : "v"(af + (cF + 4 * q) * 4) : "memory");
          asm volatile("ds_read_b128 %0, %1" : "=v"(av[NA + q]) : "v"(af + (C + cF + 4 * q) * 4) : "memory");
        }
      }
#pragma unroll
      for (int it = 0; it < NIT; ++it)
        asm volatile("ds_read_b128 %0, %1" : "=v"(rv[it]) : "v"(src + it * (LT * 16)) : "memory");
      if (++gC == G) {
        gC = 0; ++tiC;
        if (++affslot == 3) affslot = 0;
        tile_advance(posC);
        flagsC = tile_flags(posC);
      }
    };
    auto convert = [&](int buf) __attribute__((always_inline)) {
      const unsigned dst = lds0 + buf * BUFB;
      // the wait, then empty volatile asms that every read's result passes through: volatile asms keep their
      // order, so no use of a result can be scheduled above the wait
      DSX_STAMP_T(120, tid == 256 && DSX_STAMP_CVT_COND);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int q = 0; q < 2 * NA; ++q) asm volatile("" : "+v"(av[q]));
#pragma unroll
      for (int it = 0; it < NIT; ++it) asm volatile("" : "+v"(rv[it]));
      DSX_STAMP_T(121, tid == 256 && DSX_STAMP_CVT_COND);
      // lanes without a GroupNorm in front (gnF is per lane: channels past C) multiply by 1 and add 0 -- exact -- so the
      // unit loop has one unconditional packed fma per pair instead of a select per element
      float sc[CPU], sh[CPU];
#pragma unroll
      for (int q = 0; q < NA; ++q) {
        sc[4 * q] = gnF ? av[q].x : 1.0f; sc[4 * q + 1] = gnF ? av[q].y : 1.0f;
        sc[4 * q + 2] = gnF ? av[q].z : 1.0f; sc[4 * q + 3] = gnF ? av[q].w : 1.0f;
        sh[4 * q] = gnF ? av[NA + q].x : 0.0f; sh[4 * q + 1] = gnF ? av[NA + q].y : 0.0f;
        sh[4 * q + 2] = gnF ? av[NA + q].z : 0.0f; sh[4 * q + 3] = gnF ? av[NA + q].w : 0.0f;
      }
      const bool any_gn = a.gn_scale != nullptr;   // uniform
      // one uniform branch around the whole unit loop (a branch per unit keeps the units' arithmetic from interleaving):
      // residual / attention-output 1 x 1 convs and the upsampling convs have no GroupNorm / Swish in front and copy
      // (1 x 1 only: in the 3 x 3 instantiations the extra branch cost the GroupNorm layers 3-4 %, more than the four
      // upsampling convs gained)
      const bool plain = KS == 1 && !(gnF || a.swish);
      if (plain) {
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
          if (loff[it] >= 0) {
            uint4 w = __builtin_bit_cast(uint4, rv[it]);
            const unsigned keep = ((edge[it] & flagsF) == 0 && cF < C) ? 0xffffffffu : 0u;
            w.x &= keep; w.y &= keep; w.z &= keep; w.w &= keep;
            lds_write_b128_asm(dst + loff[it], __builtin_bit_cast(f32x4_t, w));
          }
        }
      } else {
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
          if (loff[it] >= 0) {
            float v[CPU];
            Unit<DT>::unpack(__builtin_bit_cast(uint4, rv[it]), v);
            // the arithmetic runs in every lane; padding pixels (and channels past C) are forced to exactly 0
            // afterwards with one mask per packed register (padded AFTER the activation, as the reference does)
#ifndef DSX_ABL_CVT   // -DDSX_ABL_CVT: timing experiment, loaders skip the GroupNorm / Swish arithmetic (results wrong)
            if (any_gn) affine_vec(v, sc, sh);
            if (a.swish) swish_vec(v);
#endif
            if (it == NIT - 1) { asm volatile("" :: "v"(v[0]), "v"(v[CPU - 1])); DSX_STAMP_T(122, tid == 256 && DSX_STAMP_CVT_COND); }
            uint4 w = Unit<DT>::pack(v);
            const unsigned keep = ((edge[it] & flagsF) == 0 && cF < C) ? 0xffffffffu : 0u;
            w.x &= keep; w.y &= keep; w.z &= keep; w.w &= keep;
            lds_write_b128_asm(dst + loff[it], __builtin_bit_cast(f32x4_t, w));
          }
        }
      }
      DSX_STAMP_T(123, tid == 256 && DSX_STAMP_CVT_COND);
    };
    // wait until the DMAs of the item to fetch have landed; `young` = younger items that may stay in flight
    auto wait_young = [&](int young) __attribute__((always_inline)) {
      if (young >= P) wait_vmcnt<P * NIT>();
      else if (P > 1 && young == P - 1) wait_vmcnt<(P > 1 ? (P - 1) * NIT : 0)>();
      else if (P > 2 && young == P - 2) wait_vmcnt<(P > 2 ? (P - 2) * NIT : 0)>();
      else if (P > 3 && young == P - 3) wait_vmcnt<(P > 3 ? (P - 3) * NIT : 0)>();
      else wait_vmcnt<0>();
      __builtin_amdgcn_sched_barrier(0);
    };

    int issued = 0, slotI = 0, slotF = 0;   // items issued; ring slot of the next issue / next fetch
    auto issue_next = [&]() __attribute__((always_inline)) {
      issue(slotI);
      ++issued;
      if (++slotI == NSLOT) slotI = 0;
    };
    auto fetch_next = [&]() __attribute__((always_inline)) {
      fetch(slotF);
      if (++slotF == NSLOT) slotF = 0;
    };
    DSX_STAMP_T(110, tid == 256);       // tables built
    // Start-up ramp: with every CU issuing at once an LDS-DMA instruction takes ~200 cycles to issue, and the first
    // item's data has landed long before all P + 1 ring slots are requested.  Only the first K0 items are requested
    // up front; the ring fills up (at most two requests per iteration) while the first images are converted.
    // `young` of a wait is counted from `issued`: requests issued after the item waited for.
    constexpr int K0 = DSX_WS_K0_EXPR < NSLOT ? DSX_WS_K0_EXPR : NSLOT;
    while (issued < K0 && issued < total) issue_next();
    DSX_STAMP_T(111, tid == 256);       // first DMAs issued
    ws_barrier();                       // scale/shift of tiles 0 and 1 are in LDS
    wait_young(issued - 1);
    DSX_STAMP_T(112, tid == 256);       // item 0 has landed
    fetch_next();                       // item 0
    for (int k = 0; k < 2 && issued < total && issued < NSLOT; ++k) issue_next();   // (in flight during the conversion)
    convert(0);
    if constexpr (NBUF >= 3) { if (total > 0) lds_signal(full_addr, wave, 1); }   // item 0 is ready
    DSX_STAMP_T(113, tid == 256);       // item 0 converted
    if (total > 1) {
      wait_young(issued - 2);
      fetch_next();                     // item 1
    }
    if constexpr (NBUF == 2) ws_barrier();   // image of item 0 is ready
    DSX_STAMP_T(64, tid == 256);
    int bufL = 1;                       // image buffer of item v + 1
    for (int v = 0; v < total; ++v) {
      // the slots of items <= v were copied to registers an iteration or more ago: items up to v + NSLOT may be requested
      for (int k = 0; k < 2 && issued < total && issued < v + 1 + NSLOT; ++k) issue_next();
      DSX_STAMP_T(65 + 4 * v, tid == 256 && v < 9);
      if (v + 1 < total) {
        if constexpr (NBUF >= 3) {
          // buffer (v + 1) % NBUF held item v + 1 - NBUF: every compute wave must have released it (v + 2 - NBUF items consumed)
          if (v + 1 >= NBUF) lds_wait_all_ge(full_addr + 16, v + 2 - NBUF, a.handoff_timeouts);
          convert(bufL);
          lds_signal(full_addr, wave, v + 2);     // items 0 .. v + 1 are ready
        } else {
          convert((v + 1) & 1);
        }
      }
      if (++bufL == NBUF) bufL = 0;
      DSX_STAMP_T(66 + 4 * v, tid == 256 && v < 9);
      if (v + 2 < total) {
        wait_young(issued - (v + 3));
        fetch_next();                   // item v+2
      }
      DSX_STAMP_T(67 + 4 * v, tid == 256 && v < 9);
      if constexpr (NBUF == 2) ws_barrier();
      DSX_STAMP_T(68 + 4 * v, tid == 256 && v < 9);
    }
    return;
  }

  // ========================================================================= COMPUTE WAVES
  // Each wave owns MB row blocks of 32 pixels (two 16-pixel MFMA column blocks, ph = 0 / 1) x NB 32-channel N blocks.
  // A pixel fragment read from LDS feeds 2 NB MFMAs (both 16-channel halves of every N block).
  const int wm = wave / WN, wn = wave % WN;
  const int pq = ws16_pixel(c16);               // this lane's pixel inside a 16-pixel block
  int abase[MB][KS];                            // ph = 0; the ph = 1 block is `a16` bytes further (uniform)
#pragma unroll
  for (int mb = 0; mb < MB; ++mb) {
    const int m = (wm * MB + mb) * 32 + pq;
    const int tx = m & (TW - 1);
    const int ty = (m >> a.tw_log2) & (TH - 1);
    const int tb = m >> (a.tw_log2 + a.th_log2);
#pragma unroll
    for (int dy = 0; dy < KS; ++dy)
      abase[mb][dy] = (tb * PH + ty * S + dy) * RB + tx * S * PIXB + ws16_slice(kq) * 16;
  }
  // pixel m + 16 of a row block: 16 pixels to the right (tiles >= 32 wide), else 16 / TW rows down (host: TW >= 8, TB == 1)
  const int a16 = TW >= 32 ? 16 * S * PIXB : (16 >> a.tw_log2) * S * RB;
  const int blk0 = (nt * WN + wn) * NB;         // host: Cout % (32 * WN * NB) == 0, so every block exists
  const int wblock = G * (NSTEP * 1024);        // bytes of one N block's fragment stream
  const __amdgpu_buffer_rsrc_t rsw = __builtin_amdgcn_make_buffer_rsrc(
      (void*)((const unsigned char*)a.wpack + (size_t)blk0 * wblock), 0, NB * wblock, 0x00020000);
  // pack_conv's stream holds, per (chunk, tap), two 1 KiB fragments of the 32x32x16 shape; the 16 x 16 fragments ch = 0 / 1
  // of the same 32 channels x 32 K are a per-lane gather from that 2 KiB pair (every byte of it is read exactly once)
  const int wlane[2] = {ws16_woff(lane, 0), ws16_woff(lane, 1)};
  static_assert(D % 2 == 0 && NSTEP % 2 == 0, "ring slots alternate between the two fragments of a pair");
  const int qtot = G * NSTEP;                   // the stream restarts at every tile (same N blocks)
  int qn = 0;                                   // next step (fragment) to prefetch (wraps); its parity is the static `ch`
  struct WFrag { uint4 v[NB]; };                // one step's weight fragments (by value: stays in registers)
  auto load_b = [&](auto chc) __attribute__((always_inline)) -> WFrag {
    constexpr int ch = decltype(chc)::value;
    WFrag f;
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
      f.v[nb] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rsw, wlane[ch], nb * wblock + (qn >> 1) * 2048, 0));
    if (++qn == qtot) qn = 0;
    return f;
  };
  WFrag bq[D];
  static_for<D>([&](auto jc) __attribute__((always_inline)) {
    bq[decltype(jc)::value] = load_b(std::integral_constant<int, (decltype(jc)::value & 1)>{});
  });

  f32x16 acc[MB][NB];
#pragma unroll
  for (int mb = 0; mb < MB; ++mb)
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mb][nb][r] = 0.0f;

  const bool do_stats = a.stat_part != nullptr;
  const int nbase = blk0 * 32 + 8 * kq;         // this lane's 8 consecutive channels of N block 0 (+32 per block)

  // Tile walk without divisions: (tx, ty, image) of tile p0 + k*wpn, advanced by the decomposed stride.
  struct TilePos { int tx, ty, b; };
  const int per_img = a.tiles_x * a.tiles_y;
  const int adv_x = a.ws_adv_x, adv_y = a.ws_adv_y, adv_b = a.ws_adv_b;
  auto tile_advance = [&](TilePos& t) __attribute__((always_inline)) {
    t.tx += adv_x;
    if (t.tx >= a.tiles_x) { t.tx -= a.tiles_x; t.ty += 1; }
    t.ty += adv_y;
    if (t.ty >= a.tiles_y) { t.ty -= a.tiles_y; t.b += 1; }
    t.b += adv_b;
  };
  // element offsets (32-bit; host: tensors < 2^31 elements) of this lane's output rows: a tile-invariant
  // per-row part, precomputed, plus a wave-uniform per-tile part -> no per-tile vector multiplies
  int orow[MB], rrow[MB];                       // ph = 0; the pixel of ph = 1 is `p16` output pixels further (uniform)
#pragma unroll
  for (int mb = 0; mb < MB; ++mb) {
    const int m = (wm * MB + mb) * 32 + pq;
    const int pix = (m >> a.tw_log2) * a.Wo + (m & (TW - 1));
    orow[mb] = pix * a.out_ld + nbase;
    rrow[mb] = pix * a.resid_ld + nbase;
  }
  const int p16 = TW >= 32 ? 16 : (16 >> a.tw_log2) * a.Wo;
  const int o16 = p16 * a.out_ld, r16 = p16 * a.resid_ld;
  auto tile_pixel0 = [&](const TilePos& t) __attribute__((always_inline)) -> int {   // first output pixel of tile t (uniform)
    return (t.b * a.Ho + (t.ty << a.th_log2)) * a.Wo + (t.tx << a.tw_log2);
  };
  TilePos cur;             // tile ti (being multiplied)
  {
    const int q1 = (int)fastdiv((unsigned)p0, a.mg_tiles_x);
    cur.b = (int)fastdiv((unsigned)p0, a.mg_per_img);
    cur.tx = p0 - q1 * a.tiles_x;
    cur.ty = q1 - cur.b * a.tiles_y;
  }
  TilePos nxt = cur;       // tile ti + 1
  tile_advance(nxt);
  TilePos nn = nxt;        // tile ti + 2
  tile_advance(nn);

  // GroupNorm scale/shift of image b -> LDS parity slot, for the loader waves (host: C <= 1024)
  auto load_aff = [&](int b, int slot) __attribute__((always_inline)) {
    if (a.gn_scale == nullptr || tid * 4 >= C) return;
    float* dst = aff_base + (size_t)slot * (AFFB / 4);
    *(float4*)(dst + tid * 4) = *(const float4*)(a.gn_scale + (size_t)b * C + tid * 4);
    *(float4*)(dst + C + tid * 4) = *(const float4*)(a.gn_shift + (size_t)b * C + tid * 4);
  };
  // Three slots (tile % 3).  The scale/shift of tile t is written by the epilogue of tile t-3, i.e. it is visible
  // after the barrier that ends item (t-2)*G, and the loaders first read it when they fetch item t*G, after the
  // barrier that ends item t*G - 3: safe for every G >= 2.  (With two slots and a lead of two tiles the fetch of a
  // G = 2 layer raced with the epilogue that writes the slot.)
  TilePos n3 = nn;         // tile ti + 3
  tile_advance(n3);
  load_aff(cur.b, 0);
  if (ntile > 1) load_aff(nxt.b, 1);
  if (ntile > 2) load_aff(nn.b, 2);

  // Epilogue operands live in registers and are fetched one tile ahead, at the start of the previous tile's
  // epilogue (right after its own operands were consumed): the loads then have the rest of that epilogue
  // plus the ring's steps to land before an in-order vmcnt wait of the weight ring can trip over them.
  // One per-channel addend: the FiLM vector of the tile's image (the host folds the conv bias into the FiLM
  // bias, see dsx_model_finalize) or, without FiLM, the conv bias, fetched once (the workgroup never changes
  // its N tile).  Missing operands stay 0.0f: adding them is exact.
  constexpr int NR = 8 / CPU;    // 16-byte pieces of a lane's 8 residual values of one pixel
  const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
  float4 addv[NB][2], affv[2];
  constexpr bool PRE_RESID = DSX_PRE_RESID_EXPR;   // else the residual is read in the epilogue
  uint4 residv[PRE_RESID ? MB : 1][NB][2][NR];   // [row block][N block][ph][piece]; storage type; all-zero bits are 0.0 in both
  // (a uniform branch, not `cond ? *ptr : zero4`: hipcc turns that select into a flat load through a pointer select
  // between the global vector and a private copy of zero4 -- scratch, flat loads, and the addrspacecast that its spill
  // path later fails on with "Operand has incorrect register class ... $src_private_base")
#pragma unroll
  for (int nb = 0; nb < NB; ++nb)
#pragma unroll
    for (int j = 0; j < 2; ++j) addv[nb][j] = zero4;
  if (a.bias && !a.film) {
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
      for (int j = 0; j < 2; ++j) addv[nb][j] = *(const float4*)(a.bias + nbase + 32 * nb + 4 * j);
  }
#pragma unroll
  for (int mb = 0; mb < (PRE_RESID ? MB : 1); ++mb)
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
      for (int ph = 0; ph < 2; ++ph)
#pragma unroll
        for (int q = 0; q < NR; ++q) residv[mb][nb][ph][q] = make_uint4(0u, 0u, 0u, 0u);
  affv[0] = affv[1] = zero4;
  // t: the tile whose epilogue will use the operands; b_aff: image of the tile three after it
  auto prefetch_epilogue = [&](const TilePos& t, bool want_aff, int b_aff) __attribute__((always_inline)) {
    if (a.film) {
#pragma unroll
      for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          addv[nb][j] = *(const float4*)(a.film + (size_t)t.b * a.film_bs + nbase + 32 * nb + 4 * j);
    }
    if (PRE_RESID && a.resid) {
      const int r0 = tile_pixel0(t) * a.resid_ld;
#pragma unroll
      for (int mb = 0; mb < (PRE_RESID ? MB : 1); ++mb)
#pragma unroll
        for (int ph = 0; ph < 2; ++ph) {
          const DT* rp = (const DT*)a.resid + (r0 + rrow[mb] + ph * r16);
#pragma unroll
          for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int q = 0; q < NR; ++q) residv[mb][nb][ph][q] = *(const uint4*)(rp + 32 * nb + CPU * q);
        }
    }
    if (want_aff && a.gn_scale != nullptr && tid * 4 < C) {
      affv[0] = *(const float4*)(a.gn_scale + (size_t)b_aff * C + tid * 4);
      affv[1] = *(const float4*)(a.gn_shift + (size_t)b_aff * C + tid * 4);
    }
  };
  prefetch_epilogue(cur, ntile > 3, n3.b);

  // Until the loaders have staged the first image the compute waves only wait (one DMA round trip plus a conversion).
  // A residual 1 x 1 conv uses that time for the finalize of the GroupNorm between the two 3 x 3 convs of its block
  // (its statistics were complete before this launch started): one (image, group) item per compute wave, plus that
  // GroupNorm's consumer's L2 weight prefetch -- one k_gn_finalize launch less per block.
  // Every launch also uses the wait to pull the NEXT conv launch's weight slices into the L2s that will read them.
  // (not in the two-N-block 3 x 3 instantiation: it is at the register limit, and the extra live values cost its main
  // loop 10 %)
  constexpr bool PF_OK = !(NB == 2 && KS == 3);
  // (folded into one register at once: these waves wait for the first image anyway, and the main loop has no registers to spare)
  unsigned fin_pf_acc = 0u;
  if constexpr (PF_OK) fin_pf_acc = l2_prefetch_fold(l2_prefetch(a.pf, blockIdx.x, gridDim.x, tid, 256));
  if constexpr (KS == 1) {
    if (a.fin_on) {
      fin_pf_acc |= l2_prefetch_fold(l2_prefetch(a.fin.pf, blockIdx.x, gridDim.x, tid, 256));
      const int nitems = a.fin.B * a.fin.groups;
      for (int item = (int)blockIdx.x * 4 + wave; item < nitems; item += (int)gridDim.x * 4)
        gn_finalize_item<1>(a.fin, item % a.fin.B, item / a.fin.B, lane, nullptr);
    }
  }

  if constexpr (NBUF >= 3) {
    if (tid < 8) ((unsigned*)(lds + NBUF * BUFB + 3 * AFFB + NSLOT * RAWB))[tid] = 0u;   // FULL[4], FREE[4]
  }
  ws_barrier();   // scale/shift (and the zeroed counters) visible to the loaders
  if constexpr (NBUF == 2) ws_barrier();   // image of item 0 is ready
  DSX_STAMP_T(0, tid == 0);
  int g = 0, ti = 0, aslot = 0;   // aslot == ti % 3
  int ibuf = 0;                   // image buffer of the current item (v % NBUF)
  // operand fragments (one 16-pixel column block x one 64-byte chunk each, feeding 2 NB MFMAs = 32 NB cycles) are read
  // PFF fragments ahead of their MFMAs into a ring of PFF + 1
  constexpr int NPAIR = NSTEP / 2;                 // (chunk, tap) pairs of an item: two weight fragments (ch = 0, 1) each
  constexpr int NF = 2 * MB;                       // pixel fragments per pair: (mb, ph)
  constexpr int FT = NPAIR * NF;                   // pixel fragments per item
  constexpr int PFF = (DSX_PFF_EXPR) < FT ? (DSX_PFF_EXPR) : FT - 1;
  static_assert(PFF >= 1 && PFF < FT && PFF <= 15, "lgkmcnt is 4 bits");
  // The ring slot of step s of item v is (v * NSTEP + s) % D: static for D <= NSTEP; for a ring of RP groups the item
  // loop is unrolled RP times (ring phase R = v % RP static in each copy).
  constexpr int RP = D > NSTEP ? D / NSTEP : 1;
  if constexpr (RP == 1) {
    for (int v = 0; v < total; ++v) {
      constexpr int R0 = 0;
#define DSX_WS_ITEM_NEXT continue
#include "dsx_conv_ws_item.inc"
#undef DSX_WS_ITEM_NEXT
    }
  } else {
    for (int v0 = 0; v0 < total; v0 += RP)
      static_for<RP>([&](auto rc_) __attribute__((always_inline)) {
        constexpr int R0 = decltype(rc_)::value * NSTEP;
        const int v = v0 + decltype(rc_)::value;
        if (v >= total) return;
#define DSX_WS_ITEM_NEXT return
#include "dsx_conv_ws_item.inc"
#undef DSX_WS_ITEM_NEXT
      });
  }
  if constexpr (PF_OK) l2_prefetch_retire(a.pf, fin_pf_acc);   // (both sinks are always null)
}


// ===========================================================================================
// Image-resident kernel for the 8 x 8 feature maps (the bottleneck level of sr_sr3_16_128: 512 / 1024 channels).
//
// Why: with M = 64 pixels per image a tile grid cannot fill the chip, so the tiled kernels split K over
// workgroups (fp32 slabs through HBM + a reduce launch: 16.8 MB of slab traffic for a 4.7 MB weight tensor) and
// every GroupNorm costs a k_gn_finalize launch: ~32 us per layer for ~2 us of MFMA work.  Here one workgroup
// (8 waves) owns one image x one 32-channel N block and the WHOLE K extent:
//   * K is split over the 8 waves: in phase p wave w owns the 64-byte channel chunk 8p + w for all taps; its
//     slice of the (zero-bordered) halo image lives in a wave-private LDS buffer, so the main loop has no
//     workgroup barrier at all;
//   * GroupNorm is finalised here: lanes = channels, the producer's partial sums are reduced over the group with
//     wave shuffles (double, fixed order), affine + Swish are applied while the slice is written to LDS;
//   * weights stream from L2 in MFMA fragment order straight into registers (ring of D fragments);
//   * the 8 partial accumulators meet in LDS (fixed order -> bitwise reproducible), the epilogue adds bias / FiLM /
//     residual, stores, and emits the complete per-(image, channel) GroupNorm sums of the result (one partial row).
// ===========================================================================================
template <typename DT, int KS, int NPH>
__global__ __launch_bounds__(512, 1) void k_conv_img(const ConvArgs a) {
  constexpr int KC = Chunk<DT>::KC;
  constexpr int CPU = Unit<DT>::N;
  constexpr int ES = (int)sizeof(DT);
  constexpr int TAPS = KS * KS, PAD = KS / 2, PW = 8 + 2 * PAD;
  constexpr int PIXB = 80;
  constexpr int RB = KS == 3 ? 896 : 640;       // conv_lds_row(KS, 1, 3): conflict-free ds_read_b128 of 8-wide rows
  constexpr int IMGB = PW * RB;
  constexpr int NSTEP = TAPS * 2;
  constexpr int D = NSTEP;                      // weight fragments in flight: a whole phase (latency, not bandwidth, bounds the stream)
  constexpr int MAXP = NPH;                     // phases this instantiation is unrolled for (host: C <= NPH * 8 * KC)
  constexpr bool ALLW = NPH == 2;               // two phases: every weight fragment of the wave is requested up front
  static_assert(NSTEP % D == 0, "ring depth divides the steps of a phase");
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, lh = lane >> 5;
  const int nb = blockIdx.x % a.nblocks, b = blockIdx.x / a.nblocks;
  const int C = a.C0 + a.C1;
  // The 2- and 4-phase instantiations are launched for exactly that many phases (512 / 1024 channels in 16-bit storage),
  // the 8-phase one for every other count: with a run-time count the "phase exists" branches around the early loads made
  // hipcc's vmcnt bookkeeping assume the fewest loads on any path, so the GroupNorm arithmetic and the conversion waited
  // for the first 9 weight fragments instead of running under the weight stream.
  const int nphase = NPH != 8 ? NPH : C / (8 * KC);
  unsigned char* img = lds + wave * (2 * IMGB);
  float* aff = (float*)(lds + 8 * 2 * IMGB) + wave * (8 * 2 * KC);      // [phase][scale KC | shift KC]

  DSX_STAMP(0);
  // zero both buffers once: the halo border stays zero (the reference pads AFTER the activation)
  if (KS == 3) {
    // 36 border pixels x 5 sixteen-byte units x 2 buffers = 360 units: top / bottom rows, left / right columns
    for (int q = lane; q < 360; q += 64) {
      const int bufi = q >= 180, r = q - 180 * bufi, bp = r / 5, u = r - 5 * bp;
      int py, px2;
      if (bp < 10) { py = 0; px2 = bp; }
      else if (bp < 20) { py = 9; px2 = bp - 10; }
      else { py = 1 + ((bp - 20) >> 1); px2 = ((bp - 20) & 1) ? 9 : 0; }
      *(uint4*)(img + bufi * IMGB + py * RB + px2 * PIXB + u * 16) = make_uint4(0u, 0u, 0u, 0u);
    }
  }

  // ---- this wave's slice of phase p: 64 pixels x one 64-byte chunk = 4 staging units per lane
  auto load_raw = [&](int p, uint4 (&raw)[4]) __attribute__((always_inline)) {
    const int c0 = (p * 8 + wave) * KC;
    const bool first = c0 < a.C0;
    const char* src = (const char*)(first ? a.src0 : a.src1);
    const int Cs = first ? a.C0 : a.C1, cl = first ? c0 : c0 - a.C0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int q = lane + 64 * i, px = q >> 2, u = q & 3;
      raw[i] = *(const uint4*)(src + (((size_t)b * 64 + px) * Cs + cl + u * CPU) * ES);
    }
  };
  auto convert_store = [&](int p, const uint4 (&raw)[4]) __attribute__((always_inline)) {
    unsigned char* buf = img + (p & 1) * IMGB;
    const int u = lane & 3;
    float sc[CPU], sh[CPU];
    if (a.has_gn) {
#pragma unroll
      for (int j = 0; j < CPU; ++j) { sc[j] = aff[p * 2 * KC + u * CPU + j]; sh[j] = aff[p * 2 * KC + KC + u * CPU + j]; }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int px = (lane + 64 * i) >> 2;
      float v[CPU];
      Unit<DT>::unpack(raw[i], v);
      if (a.has_gn) affine_vec(v, sc, sh);
      if (a.swish) swish_vec(v);
      *(uint4*)(buf + ((px >> 3) + PAD) * RB + ((px & 7) + PAD) * PIXB + u * 16) = Unit<DT>::pack(v);
    }
  };

  // ---- GroupNorm operands first: vmcnt retires in order, so the producer's partial sums (and gamma / beta) issued
  // ahead of the activation slices and the 18-fragment weight ring arrive first, and the finalize arithmetic runs
  // under the weight stream's latency instead of behind it (it waited ~18k cycles for its operands the other way round).
  // Branch-free loads (a "load or skip" branch per phase would make hipcc wait vmcnt(0) inside every branch):
  // phases past the last re-read the last one, the element is fetched as two 8-byte halves for either partial type.
  uint2 glo[MAXP], ghi[MAXP];
  float gg[MAXP], gb[MAXP];
  if (a.has_gn) {
#pragma unroll
    for (int p = 0; p < MAXP; ++p) {
      const int pp = p < nphase ? p : nphase - 1;
      const int c0 = (pp * 8 + wave) * KC, cb = c0 & ~63, c = cb + lane;   // c < C: C0, C1 are multiples of 64
      const bool first = cb < a.C0;             // wave-uniform
      const char* part = (const char*)(first ? a.gn_part0 : a.gn_part1);
      const int nch = first ? a.gn_nchunk0 : a.gn_nchunk1, pf32 = first ? a.gn_pf32_0 : a.gn_pf32_1;
      const int Cs = first ? a.C0 : a.C1, cl = first ? c : c - a.C0;
      const char* e = part + (((size_t)b * nch) * Cs + cl) * (pf32 ? 8 : 16);
      glo[p] = *(const uint2*)e;
      ghi[p] = *(const uint2*)(e + 8);          // second half of a double2 (ignored for float partials; stays inside the workspace)
      gg[p] = a.gn_gamma[c]; gb[p] = a.gn_beta[c];
    }
  }
  uint4 raw0[4], raw1[4];
  load_raw(0, raw0);
  if (nphase > 1) load_raw(1, raw1);

  // ---- weight stream: fragments of (N block nb, chunk 8p + wave), NSTEP x 1 KiB per phase, flat over the phases
  const unsigned char* wbase = (const unsigned char*)a.wpack + (size_t)nb * a.kchunks * (NSTEP * 1024) + lane * 16;
  int pn = 0, sn = 0;                           // phase and step of the next fragment to prefetch
  auto load_w = [&]() __attribute__((always_inline)) -> uint4 {   // unconditional: the caller stops at the last phase
    const uint4 v = *(const uint4*)(wbase + ((size_t)(pn * 8 + wave) * NSTEP + sn) * 1024);
    if (++sn == NSTEP) { sn = 0; ++pn; }
    return v;
  };
  uint4 wq[D], wq1[ALLW ? D : 1];
#pragma unroll
  for (int j = 0; j < D; ++j) wq[j] = load_w();
  if constexpr (ALLW) {
    if (nphase > 1) {
#pragma unroll
      for (int j = 0; j < D; ++j) wq1[j] = load_w();
    }
  }
  DSX_STAMP(1);

  // ---- GroupNorm: scale / shift of this wave's channels of every phase (lanes = 64 consecutive channels).
  // All loads of all phases were issued at the top (one memory round trip, not one per phase).
  if (a.has_gn) {
    const int cpg = C / a.gn_groups;            // host: power of two <= 64, divides C0
    double gs[MAXP], gq[MAXP];
#pragma unroll
    for (int p = 0; p < MAXP; ++p) {
      const int pp = p < nphase ? p : nphase - 1;
      const int cb = ((pp * 8 + wave) * KC) & ~63, c = cb + lane;
      const bool first = cb < a.C0;
      const int nch = first ? a.gn_nchunk0 : a.gn_nchunk1, pf32 = first ? a.gn_pf32_0 : a.gn_pf32_1;
      const double d0 = __builtin_bit_cast(double, ((unsigned long long)glo[p].y << 32) | glo[p].x);
      const double d1 = __builtin_bit_cast(double, ((unsigned long long)ghi[p].y << 32) | ghi[p].x);
      gs[p] = pf32 ? (double)__builtin_bit_cast(float, glo[p].x) : d0;
      gq[p] = pf32 ? (double)__builtin_bit_cast(float, glo[p].y) : d1;
      if (nch > 1 && p < nphase) {              // producers with several partial rows (split-K reduce, k_chan_stats): rare
        const char* part = (const char*)(first ? a.gn_part0 : a.gn_part1);
        const int Cs = first ? a.C0 : a.C1, cl = first ? c : c - a.C0;
        for (int k = 1; k < nch; ++k) {
          const size_t idx = (((size_t)b * nch + k) * Cs + cl) * 2;
          if (pf32) { const float2 v = *(const float2*)((const float*)part + idx); gs[p] += v.x; gq[p] += v.y; }
          else { const double2 v = *(const double2*)((const double*)part + idx); gs[p] += v.x; gq[p] += v.y; }
        }
      }
    }
    // Group sums over <= 64 lanes in fp32 (a channel's sums over the 64 pixels already are fp32 values of the producer's
    // epilogue; the xor butterfly gives every lane of a group the same bits), the cancellation-prone E[x^2] - mean^2 in
    // double, the reciprocal square root by v_rsq_f32 (1 ulp): a dozen instructions per phase instead of a double-
    // precision butterfly, division and square root (~3 k cycles of this latency-bound kernel's prologue with 8 waves/CU).
    const float inv_n = 1.0f / (64.0f * (float)cpg);     // H * W * channels per group: a power of two, exact
#pragma unroll
    for (int p = 0; p < MAXP; ++p) {
      if (p < nphase) {
        const int c0 = (p * 8 + wave) * KC, c = (c0 & ~63) + lane;
        float s = (float)gs[p], q = (float)gq[p];
        for (int o = 1; o < cpg; o <<= 1) { s += __shfl_xor(s, o, 64); q += __shfl_xor(q, o, 64); }
        const double mean = (double)s * (double)inv_n;
        double var = (double)q * (double)inv_n - mean * mean;
        if (var < 0) var = 0;
        const float rstd = __builtin_amdgcn_rsqf((float)var + a.gn_eps);
        const float meanf = (float)mean;
        if (c >= c0 && c < c0 + KC) {
          const float sc = rstd * gg[p];
          aff[p * 2 * KC + (c - c0)] = sc;
          aff[p * 2 * KC + KC + (c - c0)] = gb[p] - meanf * sc;
        }
      }
    }
  }
  // the next image-resident conv's weight slices -> this XCD's L2 (its workgroups with the same label need them): issued
  // behind this kernel's own first loads, a whole kernel duration ahead of their use
  const PfAcc pf_acc = l2_prefetch(a.pf, blockIdx.x, gridDim.x, tid, 512);
  DSX_STAMP(2);
  __builtin_amdgcn_s_waitcnt(0xC07F);           // lgkmcnt(0): the zero fill and the aff table are in LDS
  convert_store(0, raw0);
  if (nphase > 1) convert_store(1, raw1);
  DSX_STAMP(3);

  int abase[2][KS];
#pragma unroll
  for (int mb = 0; mb < 2; ++mb) {
    const int m = mb * 32 + li;
#pragma unroll
    for (int dy = 0; dy < KS; ++dy) abase[mb][dy] = ((m >> 3) + dy) * RB + (m & 7) * PIXB + lh * 16;
  }
  f32x16 acc[2];
#pragma unroll
  for (int mb = 0; mb < 2; ++mb)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[mb][r] = 0.f;

  for (int p = 0; p < nphase; ++p) {
    const unsigned char* buf = img + (p & 1) * IMGB;
    if (p + 2 < nphase) load_raw(p + 2, raw0);   // in flight during this phase's MFMAs
    const bool more = p + 1 < nphase;
    static_for<NSTEP>([&](auto sc_) __attribute__((always_inline)) {
      constexpr int s = decltype(sc_)::value;
      constexpr int tap = s >> 1, fs = s & 1, dy = tap / KS, dx = tap % KS;
      uint4 wcur = wq[s % D];
      if constexpr (ALLW) {
        if (p == 1) wcur = wq1[s % D];           // uniform
      } else {
        if (more) wq[s % D] = load_w();          // uniform: the next phase's fragment into the slot just consumed
      }
#pragma unroll
      for (int mb = 0; mb < 2; ++mb) {
        const f32x4_t px = *(const f32x4_t*)(buf + abase[mb][dy] + dx * PIXB + fs * 32);
        acc[mb] = mfma_step<DT>(wcur, px, acc[mb]);
      }
    });
    if (p + 2 < nphase) convert_store(p + 2, raw0);   // refill this buffer with the slice of phase p + 2
    DSX_STAMP(4 + p);
  }

  // thread -> pixel tid >> 3, channels 4 * (tid & 7) .. + 3 of the block.  Its epilogue operands are requested here, in
  // front of the barrier that waits for the slowest wave's MFMAs (the weight registers are dead): a memory round trip
  // that used to follow the reduction.
  const int px = tid >> 3, cg = (tid & 7) * 4;
  const int n0 = nb * 32 + cg;
  const size_t opix = (size_t)b * 64 + px;
  float4 eb = make_float4(0.f, 0.f, 0.f, 0.f), ef = eb;
  float er[4] = {0.f, 0.f, 0.f, 0.f};
  if (a.bias) eb = *(const float4*)(a.bias + n0);
  if (a.film) ef = *(const float4*)(a.film + (size_t)b * a.film_bs + n0);
  if (a.resid) {
#pragma unroll
    for (int j = 0; j < 4; ++j) er[j] = act_load<DT>(a.resid, opix * a.resid_ld + n0 + j);
  }
  // ---- the 8 partial sums meet in LDS: part[w][mb][r][lane]
  __syncthreads();
  DSX_STAMP(12);
  float* part = (float*)lds;
#pragma unroll
  for (int mb = 0; mb < 2; ++mb)   // [w][mb][lane][16 registers], 80-byte lane stride: conflict-free 16-byte accesses
#pragma unroll
    for (int j = 0; j < 4; ++j)
      *(float4*)(part + ((wave * 2 + mb) * 64 + lane) * 20 + 4 * j) =
          make_float4(acc[mb][4 * j], acc[mb][4 * j + 1], acc[mb][4 * j + 2], acc[mb][4 * j + 3]);
  __syncthreads();
  DSX_STAMP(13);
  // thread -> pixel tid >> 3, channels 4 * (tid & 7) .. + 3 of the block.  Accumulator register r of lane
  // (li, lh) of block mb is pixel 32 mb + li, channel 16 lh + r.
  const int mbo = px >> 5, lo = (px & 31) + 32 * (cg >> 4), r0 = cg & 15;
  float x[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int w = 0; w < 8; ++w) {
    const float4 t = *(const float4*)(part + ((w * 2 + mbo) * 64 + lo) * 20 + r0);
    x[0] += t.x; x[1] += t.y; x[2] += t.z; x[3] += t.w;
  }
  // (same order of additions as before the hoist: bias, FiLM, residual; absent operands are exact zeros)
  x[0] += eb.x; x[1] += eb.y; x[2] += eb.z; x[3] += eb.w;
  x[0] += ef.x; x[1] += ef.y; x[2] += ef.z; x[3] += ef.w;
#pragma unroll
  for (int j = 0; j < 4; ++j) x[j] += er[j];
  const int okind = a.out_bf16;
  if (okind == 0) {
    *(float4*)((float*)a.out + opix * a.out_ld + n0) = make_float4(x[0], x[1], x[2], x[3]);
  } else {
    uint2 w2;
    if (okind == 1) { w2.x = pack_bf16x2(x[0], x[1]); w2.y = pack_bf16x2(x[2], x[3]); }
    else { w2.x = pack_f16x2(x[0], x[1]); w2.y = pack_f16x2(x[2], x[3]); }
    *(uint2*)((unsigned short*)a.out + opix * a.out_ld + n0) = w2;
  }
  DSX_STAMP(14);
  // ---- GroupNorm sums of the result over the image's 64 pixels (complete: one partial row per image)
  if (a.stat_part != nullptr) {
    float s1[4], s2[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) { s1[j] = x[j]; s2[j] = x[j] * x[j]; }
#pragma unroll
    for (int o = 8; o < 64; o <<= 1)
#pragma unroll
      for (int j = 0; j < 4; ++j) { s1[j] += __shfl_xor(s1[j], o, 64); s2[j] += __shfl_xor(s2[j], o, 64); }
    __syncthreads();                             // `part` is read; reuse its first bytes
    float* red = (float*)lds;                    // [8 waves][32 channels][2]
    if (lane < 8) {
#pragma unroll
      for (int j = 0; j < 4; ++j) { red[(wave * 32 + cg + j) * 2] = s1[j]; red[(wave * 32 + cg + j) * 2 + 1] = s2[j]; }
    }
    __syncthreads();
    if (tid < 64) {
      const int ch = tid >> 1, k = tid & 1;
      float t = 0.f;
#pragma unroll
      for (int w = 0; w < 8; ++w) t += red[(w * 32 + ch) * 2 + k];
      a.stat_part[((size_t)b * a.Cout + nb * 32 + ch) * 2 + k] = t;
    }
  }
  l2_prefetch_retire(a.pf, pf_acc);
  DSX_STAMP(15);
}

static size_t conv_img_lds(int dtype, int ks) {
  const int KC = dtype != 0 ? 32 : 16;
  const int pw = ks == 3 ? 10 : 8, rb = ks == 3 ? 896 : 640;
  return (size_t)8 * 2 * pw * rb + (size_t)8 * 8 * 2 * KC * sizeof(float);
}
bool conv_img_applicable(int dtype, int ks, int stride, const ConvArgs& a, bool gn, int gn_groups) {
  static const int on = getenv("DSX_IMG") ? atoi(getenv("DSX_IMG")) : 1;
  if (!on || stride != 1 || a.up || !(ks == 1 || ks == 3)) return false;
  if (a.Hs != 8 || a.Ws != 8 || a.Ho != 8 || a.Wo != 8) return false;
  const int KC = dtype != 0 ? 32 : 16, C = a.C0 + a.C1;
  if (a.C0 % (8 * KC) || a.C1 % (8 * KC) || C / (8 * KC) > 8 || C < 8 * KC) return false;
  if (a.Cout % 32 || (a.out_ld & 3) || (a.resid && (a.resid_ld & 3))) return false;
  if (gn) {
    if (gn_groups < 1 || C % gn_groups) return false;
    const int cpg = C / gn_groups;
    if (cpg > 64 || (cpg & (cpg - 1)) || a.C0 % cpg) return false;
  }
  return true;
}
template <typename DT, int KS, int NPH> static hipError_t launch_img_one(const ConvArgs* ap, size_t lds, hipStream_t st) {
  auto kern = k_conv_img<DT, KS, NPH>;
  if (!ap) return hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipLaunchKernelGGL(kern, dim3((unsigned)(ap->B * ap->nblocks)), dim3(512), lds, st, *ap);
  return hipGetLastError();
}
template <typename DT, int KS> static hipError_t launch_img_ks(const ConvArgs* a, size_t lds, hipStream_t st) {
  if (!a) {   // one-time attributes of every instantiation
    hipError_t e = launch_img_one<DT, KS, 2>(a, lds, st);
    if (e == hipSuccess) e = launch_img_one<DT, KS, 4>(a, lds, st);
    if (e == hipSuccess) e = launch_img_one<DT, KS, 8>(a, lds, st);
    return e;
  }
  const int nphase = (a->C0 + a->C1) / (8 * Chunk<DT>::KC);   // 2 / 4: the instantiation unrolled for exactly that count
  return nphase == 2 ? launch_img_one<DT, KS, 2>(a, lds, st)
       : nphase == 4 ? launch_img_one<DT, KS, 4>(a, lds, st) : launch_img_one<DT, KS, 8>(a, lds, st);
}
template <typename DT> static hipError_t launch_img_dt(int ks, const ConvArgs* a, size_t lds, hipStream_t st) {
  return ks == 3 ? launch_img_ks<DT, 3>(a, lds, st) : launch_img_ks<DT, 1>(a, lds, st);
}
hipError_t launch_conv_img(int dtype, int ks, const ConvArgs& a, hipStream_t st) {
  const size_t lds = conv_img_lds(dtype, ks);
  return dtype == 1 ? launch_img_dt<__bf16>(ks, &a, lds, st)
       : dtype == 2 ? launch_img_dt<_Float16>(ks, &a, lds, st) : launch_img_dt<float>(ks, &a, lds, st);
}


// ===========================================================================================
// First conv of the UNet (downs.0: 3 x 3, 1..7 input channels -> 16..64 output channels, no GroupNorm in front).
//
// With so few input channels the per-tap channel chunk of the implicit-GEMM kernels is almost all padding
// (6 of 32 channels x 9 taps) and their staging falls back to per-element loads.  Here the whole receptive
// field is ONE K dimension, k = tap * Cin + c (K = 9 Cin <= 63, padded to 64): a 16 x 16 pixel tile's halo
// patch sits in LDS as [y][x][Cin], a lane gathers its 8 consecutive k's of a step with 2-byte LDS reads through
// a per-lane table of k -> patch offsets that is the same for every pixel, weights (packed [N block][k step] in
// fragment order) live in registers.  The layer is HBM-bound: it writes B * H * W * Cout activations and reads
// almost nothing.  Epilogue + fused GroupNorm statistics as in the other kernels (one partial row per (tile, wave)).
// ===========================================================================================
template <typename DT>
__global__ __launch_bounds__(256) void k_conv_first(const ConvArgs a) {
  constexpr bool F32 = sizeof(DT) == 4;
  constexpr int ES = (int)sizeof(DT);
  constexpr int KSTEP = F32 ? 2 : 16;            // k per MFMA
  constexpr int NS16 = 4;                        // 16-bit: K padded to 64
  __shared__ __attribute__((aligned(16))) unsigned char sm[18 * 18 * 7 * 4 + 64 * 4];
  DT* patch = (DT*)sm;
  int* kofft = (int*)(sm + 18 * 18 * 7 * 4);     // fp32 path: k -> element offset inside the patch

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, lh = lane >> 5;
  const int Cin = a.C0 + a.C1, K = 9 * Cin;
  const int tiles_x = a.Wo >> 4, tiles_y = a.Ho >> 4;
  const int t = blockIdx.x % (tiles_x * tiles_y), b = blockIdx.x / (tiles_x * tiles_y);
  const int ty = t / tiles_x, tx = t - ty * tiles_x;
  const int oy0 = ty * 16, ox0 = tx * 16;
  const int NBLK = (a.Cout + 31) >> 5;           // 1 or 2

  // ---- halo patch -> LDS [18][18][Cin] (zero outside the image)
  for (int e = tid; e < 18 * 18 * Cin; e += 256) {
    const int c = e % Cin, pp = e / Cin, px = pp % 18, py = pp / 18;
    const int iy = oy0 + py - 1, ix = ox0 + px - 1;
    DT v = (DT)0.f;
    if (iy >= 0 && iy < a.Hs && ix >= 0 && ix < a.Ws) {
      const size_t pix = ((size_t)b * a.Hs + iy) * a.Ws + ix;
      v = c < a.C0 ? ((const DT*)a.src0)[pix * a.C0 + c] : ((const DT*)a.src1)[pix * a.C1 + (c - a.C0)];
    }
    patch[e] = v;
  }
  if (F32 && tid < 64) {
    const int k = tid < K ? tid : 0;             // padded k: weight 0, any finite element
    const int tap = k / Cin, c = k - tap * Cin;
    kofft[tid] = ((tap / 3) * 18 + (tap % 3)) * Cin + c;
  }
  // 16-bit path: this lane's k's (the same for every pixel): k = 16 s + 8 lh + j
  int koff[F32 ? 1 : NS16 * 8];
  if constexpr (!F32) {
#pragma unroll
    for (int s = 0; s < NS16; ++s)
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        int k = 16 * s + 8 * lh + j;
        if (k >= K) k = 0;
        const int tap = k / Cin, c = k - tap * Cin;
        koff[s * 8 + j] = (((tap / 3) * 18 + (tap % 3)) * Cin + c) * 2;
      }
  }
  // weights: [N block][k step][lane][16 B] (16-bit) or [N block][k step][lane] floats (fp32), rows permuted as in pack_conv
  uint4 wf[F32 ? 1 : 2 * NS16];
  if constexpr (!F32) {
#pragma unroll
    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
      for (int s = 0; s < NS16; ++s)
        wf[nb * NS16 + s] = nb < NBLK ? ((const uint4*)a.wpack)[(nb * NS16 + s) * 64 + lane] : make_uint4(0u, 0u, 0u, 0u);
  }
  __syncthreads();

  f32x16 acc[2][2];
#pragma unroll
  for (int mb = 0; mb < 2; ++mb)
#pragma unroll
    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mb][nb][r] = 0.f;

#pragma unroll
  for (int mb = 0; mb < 2; ++mb) {
    const int m = (wave * 2 + mb) * 32 + li;     // pixel of the 16 x 16 tile
    const int base = ((m >> 4) * 18 + (m & 15)) * Cin;
    if constexpr (F32) {
      const int nsteps = (K + 1) >> 1;
      const float* wp = (const float*)a.wpack;
      for (int s = 0; s < nsteps; ++s) {
        const float pv = ((const float*)patch)[base + kofft[2 * s + lh]];
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) {
          if (nb < NBLK) acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(wp[(nb * 32 + s) * 64 + lane], pv, acc[mb][nb], 0, 0, 0);
        }
      }
    } else {
      const unsigned char* pb = (const unsigned char*)patch + base * 2;
#pragma unroll
      for (int s = 0; s < NS16; ++s) {
        unsigned w4[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const unsigned lo = *(const unsigned short*)(pb + koff[s * 8 + 2 * q]);
          const unsigned hi = *(const unsigned short*)(pb + koff[s * 8 + 2 * q + 1]);
          w4[q] = lo | (hi << 16);
        }
        const f32x4_t px = __builtin_bit_cast(f32x4_t, make_uint4(w4[0], w4[1], w4[2], w4[3]));
#pragma unroll
        for (int nb = 0; nb < 2; ++nb)
          if (nb < NBLK) acc[mb][nb] = mfma_step<DT>(wf[nb * NS16 + s], px, acc[mb][nb]);
      }
    }
  }
  (void)KSTEP; (void)ES;

  // ---- epilogue: + bias, store (16 consecutive channels per lane), GroupNorm partial sums per (tile, wave)
  const int okind = a.out_bf16;
#pragma unroll
  for (int nb = 0; nb < 2; ++nb) {
    const int nbase = nb * 32 + 16 * lh;
    if (nb >= NBLK) continue;
    float s1[16], s2[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) { s1[r] = 0.f; s2[r] = 0.f; }
    const bool live = nbase < a.Cout;            // Cout is a multiple of 16
#pragma unroll
    for (int mb = 0; mb < 2; ++mb) {
      const int m = (wave * 2 + mb) * 32 + li;
      const size_t opix = ((size_t)b * a.Ho + (oy0 + (m >> 4))) * a.Wo + (ox0 + (m & 15));
      float x[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) x[r] = acc[mb][nb][r];
      if (live) {
        if (a.bias) {
#pragma unroll
          for (int j = 0; j < 4; ++j) { const float4 tt = *(const float4*)(a.bias + nbase + 4 * j); x[4 * j] += tt.x; x[4 * j + 1] += tt.y; x[4 * j + 2] += tt.z; x[4 * j + 3] += tt.w; }
        }
        store16<true>(a.out, opix * a.out_ld + nbase, x, okind, 16);
#pragma unroll
        for (int r = 0; r < 16; ++r) { s1[r] += x[r]; s2[r] += x[r] * x[r]; }
      }
    }
    if (a.stat_part != nullptr) {
      float w1 = row16_fold(s1, lane), w2 = row16_fold(s2, lane);
      w1 += __shfl_xor(w1, 16, 64);
      w2 += __shfl_xor(w2, 16, 64);
      const int nch = tiles_x * tiles_y * 4, chunk = t * 4 + wave;
      const int n = nbase + row16_fold_reg(li);
      if (li < 16 && n < a.Cout) {
        float* pp = a.stat_part + (((size_t)b * nch + chunk) * a.Cout + n) * 2;
        pp[0] = w1; pp[1] = w2;
      }
    }
  }
}

bool conv_first_applicable(int ks, int stride, const ConvArgs& a, bool gn) {
  static const int on = getenv("DSX_FIRST") ? atoi(getenv("DSX_FIRST")) : 1;
  const int C = a.C0 + a.C1;
  return on && ks == 3 && stride == 1 && !a.up && !gn && !a.swish && C >= 1 && C <= 7 && a.Cout >= 16 && a.Cout <= 64 &&
         (a.Cout & 15) == 0 && (a.Ho & 15) == 0 && (a.Wo & 15) == 0 && a.Ho == a.Hs && a.Wo == a.Ws && a.out_ld == a.Cout;
}
hipError_t launch_conv_first(int dtype, const ConvArgs& a, hipStream_t st) {
  const unsigned grid = (unsigned)(a.B * (a.Ho >> 4) * (a.Wo >> 4));
  if (dtype == 1) hipLaunchKernelGGL(k_conv_first<__bf16>, dim3(grid), dim3(256), 0, st, a);
  else if (dtype == 2) hipLaunchKernelGGL(k_conv_first<_Float16>, dim3(grid), dim3(256), 0, st, a);
  else hipLaunchKernelGGL(k_conv_first<float>, dim3(grid), dim3(256), 0, st, a);
  return hipGetLastError();
}

// ------------------------------------------------------------------ dispatch
struct TileCfg { int MB, WM, WN; };
static constexpr TileCfg kTiles[TILE_COUNT] = {
    {8, 1, 4},  // 256 x 128
    {4, 1, 4},  // 128 x 128
    {2, 1, 4},  // 64 x 128
    {4, 2, 2},  // 256 x 64
    {2, 2, 2},  // 128 x 64
    {1, 2, 2},  // 64 x 64
    {1, 4, 1},  // 128 x 32: layers with <= 32 output channels (the final conv, 16-channel Hagen levels): all four
                // waves along M, no wave multiplies padding columns
    {2, 4, 1},  // 256 x 32: the same with a 16 x 16 pixel tile (two-chunk variant only, see conv_g2_lds_bytes)
};

ConvTileInfo conv_tile_info(int tile) {
  const TileCfg& t = kTiles[tile];
  return ConvTileInfo{32 * t.MB * t.WM, 32 * t.WN};
}

int conv_tile_wm(int tile) { return kTiles[tile].WM; }
bool conv_tile_fuses_stats(int tile) { return kTiles[tile].MB <= 2; }
bool conv_ws_fuses_stats(int tile) {
  return tile == TILE_128x128 || tile == TILE_64x128 || tile == TILE_128x64 || tile == TILE_64x64 || tile == TILE_256x64;
}

static constexpr int conv_cpg(int ks) { return ks == 1 ? 2 : 1; }

// patch-pixel budget of a tile's staging registers
static constexpr int max_px(int tile, int ks, int stride) {
  const int bm = 32 * kTiles[tile].MB * kTiles[tile].WM;
  if (ks == 1) return bm;  // no halo
  return stride == 2 ? 400 : (bm == 256 ? 400 : (bm == 128 ? 220 : 144));
}
static constexpr int max_it(int dtype, int tile, int ks, int stride) {
  const int upg = 4 * conv_cpg(ks);   // 16-byte units per pixel per group, either storage type
  (void)dtype;
  return (max_px(tile, ks, stride) * upg + 255) / 256;
}

static int patch_pixels(int ks, int stride, const ConvArgs& a) {
  const int TW = 1 << a.tw_log2, TH = 1 << a.th_log2;
  return (((TH - 1) * stride + ks) * ((TW - 1) * stride + ks)) << a.tb_log2;
}

int conv_chunk_multiple(int ks) { return conv_cpg(ks); }

// LDS bytes per patch row.  The A fragment of a 32-row block is read with ds_read_b128, whose 16-lane
// groups cover rows {0-3,12-15,20-27} / {4-11,16-19,28-31}: with 16-wide tiles the pitch must be a multiple
// of 256 B, with 8-wide tiles an odd multiple of 128 B, for the 16 reads to fall on 16 distinct 16-B slots.
int conv_lds_row(int ks, int stride, int tw_log2) {
  const int pixb = 64 * conv_cpg(ks) + 16;
  const int pw = ((1 << tw_log2) - 1) * stride + ks;
  int rb = (pw * pixb + 15) & ~15;
  if (ks == 1 || stride != 1) return rb;   // no halo: consecutive pixels already conflict-free
  if (tw_log2 == 4) rb = (rb + 255) & ~255;
  else if (tw_log2 == 3) { rb = (rb + 127) & ~127; if (((rb >> 7) & 1) == 0) rb += 128; }
  return rb;
}

// ---- two-chunk-per-group 3x3 variant (ConvArgs::cpg == 2): 64 input channels per staged group.  For the 64-channel
// layers of the 128^2 level the whole K extent is one group: load + GroupNorm/Swish + one barrier + 36 MFMA steps +
// epilogue, three workgroups per CU overlapping each other's phases, instead of the persistent kernel's per-group
// hand-offs (which dominate when a tile has only two groups).
int conv_lds_row_g2(int tw_log2) {
  const int pixb = 64 * 2 + 16;
  const int pw = ((1 << tw_log2) - 1) + 3;
  int rb = (pw * pixb + 15) & ~15;
  if (tw_log2 == 4) rb = (rb + 255) & ~255;
  else if (tw_log2 == 3) { rb = (rb + 127) & ~127; if (((rb >> 7) & 1) == 0) rb += 128; }
  return rb;
}
// patch pixels the two-chunk variant's staging registers are sized for, per tile
static constexpr int g2_max_px(int tile) { return tile == TILE_256x32 ? 324 : 220; }
size_t conv_g2_lds_bytes(int tile, const ConvArgs& a) {
  if (tile != TILE_128x64 && tile != TILE_256x32 && tile != TILE_128x32) return 0;
  const ConvTileInfo ti = conv_tile_info(tile);
  if ((1 << (a.tw_log2 + a.th_log2 + a.tb_log2)) != ti.BM || a.tb_log2 != 0) return 0;
  if (patch_pixels(3, 1, a) > g2_max_px(tile) || a.lds_row != conv_lds_row_g2(a.tw_log2)) return 0;
  if (a.kchunks % 2 || a.stage_mode != 0) return 0;
  const int ph = ((1 << a.th_log2) - 1) + 3;
  const size_t bufb = (size_t)ph * a.lds_row;
  const size_t need = (a.kchunks / 2 > 1 ? 2 : 1) * bufb;     // a single group never touches the second buffer
  return need <= 160 * 1024 ? need : 0;
}
template <typename DT, int TILE> static hipError_t launch_g2_tile(const ConvArgs* ap, size_t lds, hipStream_t st) {
  constexpr TileCfg t = kTiles[TILE];
  auto kern = k_conv_mfma<DT, t.MB, t.WM, t.WN, 3, 1, 2, 6, (g2_max_px(TILE) * 8 + 255) / 256>;
  if (!ap) return hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  dim3 grid((unsigned)(ap->m_tiles * ap->n_tiles * ap->ksplit));
  hipLaunchKernelGGL(kern, grid, dim3(256), lds, st, *ap);
  return hipGetLastError();
}
template <typename DT> static hipError_t launch_g2(int tile, const ConvArgs* ap, size_t lds, hipStream_t st) {
  if (!ap) {   // one-time attributes of every instantiation
    hipError_t e = launch_g2_tile<DT, TILE_128x64>(ap, lds, st);
    if (e == hipSuccess) e = launch_g2_tile<DT, TILE_256x32>(ap, lds, st);
    if (e == hipSuccess) e = launch_g2_tile<DT, TILE_128x32>(ap, lds, st);
    return e;
  }
  return tile == TILE_256x32 ? launch_g2_tile<DT, TILE_256x32>(ap, lds, st)
       : tile == TILE_128x32 ? launch_g2_tile<DT, TILE_128x32>(ap, lds, st) : launch_g2_tile<DT, TILE_128x64>(ap, lds, st);
}

size_t conv_lds_bytes(int dtype, int tile, int ks, int stride, const ConvArgs& a) {
  if (a.cpg == 2 && ks == 3 && stride == 1) return conv_g2_lds_bytes(tile, a);
  if (tile < 0 || tile >= TILE_COUNT || tile == TILE_256x32) return 0;   // (256 x 32 exists as the two-chunk variant only)
  if (!(ks == 1 || ks == 3) || !(stride == 1 || (stride == 2 && ks == 3 && tile == TILE_64x64))) return 0;
  const ConvTileInfo ti = conv_tile_info(tile);
  if ((1 << (a.tw_log2 + a.th_log2 + a.tb_log2)) != ti.BM) return 0;
  const int pp = patch_pixels(ks, stride, a);
  if (pp > max_px(tile, ks, stride)) return 0;
  if (a.lds_row != conv_lds_row(ks, stride, a.tw_log2)) return 0;
  const int ph = ((1 << a.th_log2) - 1) * stride + ks;
  const size_t bufb = (size_t)(ph << a.tb_log2) * a.lds_row;
  if (2 * bufb > 64 * 1024) return 0;
  return 2 * bufb;
}

// ap == nullptr: only set the kernel's dynamic-LDS attribute (conv_init)
template <typename DT, int TILE, int KS, int S>
static hipError_t launch_one(const ConvArgs* ap, size_t lds, hipStream_t st) {
  constexpr TileCfg t = kTiles[TILE];
  constexpr int CPG = conv_cpg(KS);
  constexpr int D = KS == 1 ? 4 : DSX_RING_DEPTH;
  constexpr int MI = max_it(Kind<DT>::value, TILE, KS, S);
  auto kern = k_conv_mfma<DT, t.MB, t.WM, t.WN, KS, S, CPG, D, MI>;
  if (!ap)
    return hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
  const ConvArgs& a = *ap;
  dim3 grid((unsigned)(a.m_tiles * a.n_tiles * a.ksplit));
  hipLaunchKernelGGL(kern, grid, dim3(256), lds, st, a);
  return hipGetLastError();
}

template <typename DT>
static hipError_t launch_dt(int tile, int ks, int stride, const ConvArgs* a, size_t lds, hipStream_t st) {
  if (stride == 2) return launch_one<DT, TILE_64x64, 3, 2>(a, lds, st);
#define DSX_TILE_CASE(T) \
  case T: return ks == 3 ? launch_one<DT, T, 3, 1>(a, lds, st) : launch_one<DT, T, 1, 1>(a, lds, st);
  switch (tile) {
    DSX_TILE_CASE(TILE_256x128)
    DSX_TILE_CASE(TILE_128x128)
    DSX_TILE_CASE(TILE_64x128)
    DSX_TILE_CASE(TILE_256x64)
    DSX_TILE_CASE(TILE_128x64)
    DSX_TILE_CASE(TILE_128x32)
    default: return ks == 3 ? launch_one<DT, TILE_64x64, 3, 1>(a, lds, st) : launch_one<DT, TILE_64x64, 1, 1>(a, lds, st);
  }
#undef DSX_TILE_CASE
}

hipError_t launch_conv(int dtype, int tile, int ks, int stride, const ConvArgs& a, hipStream_t st) {
  const size_t lds = conv_lds_bytes(dtype, tile, ks, stride, a);
  if (lds == 0 || a.ksplit < 1 || a.n_tiles < 1) return hipErrorInvalidValue;
  if (a.cpg == 2)
    return dtype == 1 ? launch_g2<__bf16>(tile, &a, lds, st) : dtype == 2 ? launch_g2<_Float16>(tile, &a, lds, st) : launch_g2<float>(tile, &a, lds, st);
  return dtype == 1 ? launch_dt<__bf16>(tile, ks, stride, &a, lds, st)
       : dtype == 2 ? launch_dt<_Float16>(tile, ks, stride, &a, lds, st)
                    : launch_dt<float>(tile, ks, stride, &a, lds, st);
}

// ---- warp-specialised variant: per (dtype, tile, ks) constants.  Its waves are laid out differently from
// k_conv_mfma's: MB row blocks x NB N blocks per wave (the 128 x 128 tile is 2 x 2 waves of 64 x 64).
struct WsTileCfg { int MB, WM, WN, NB; };
static constexpr WsTileCfg ws_tile(int tile) {
  return tile == TILE_128x128 ? WsTileCfg{2, 2, 2, 2}
       : tile == TILE_64x128  ? WsTileCfg{2, 1, 4, 1}
       : tile == TILE_128x64  ? WsTileCfg{2, 2, 2, 1}
       : tile == TILE_256x64  ? WsTileCfg{4, 2, 2, 1}    // 16 x 16 pixels: half the tiles (and epilogues) of 128 x 64
                              : WsTileCfg{1, 2, 2, 1};   // TILE_64x64
}
static constexpr bool ws_tile_ok(int tile) {
  return tile == TILE_128x128 || tile == TILE_64x128 || tile == TILE_128x64 || tile == TILE_64x64 || tile == TILE_256x64;
}
int conv_ws_tile_wm(int tile) { return ws_tile(tile).WM; }
static constexpr int ws_depth(int tile, int ks, int cpg) {   // P: groups of raw activations in flight beyond the current one
  const int bm = 32 * kTiles[tile].MB * kTiles[tile].WM;
#ifndef DSX_WS_DEPTH_C2_64
#define DSX_WS_DEPTH_C2_64 3
#endif
  if (ks == 3 && cpg == 2) return bm == 64 ? DSX_WS_DEPTH_C2_64 : 2;   // two-chunk groups are twice as long (and twice the ring bytes)
  if (ks == 3 && cpg == 4) return 1;
  if (ks == 1 && cpg == 4) return bm == 64 ? 2 : 1;                     // four-chunk groups of a 1 x 1 conv
  return DSX_WS_DEPTH_EXPR;
}
// chunks per group of k_conv_ws: the family default, or two for a 3 x 3 conv that asks for it (ConvArgs::ws_cpg)
static constexpr int ws_cpg_of(int ks, int ws_cpg) {
  return (ks == 3 && (ws_cpg == 2 || ws_cpg == 4)) ? ws_cpg : ((ks == 1 && ws_cpg == 4) ? 4 : conv_cpg(ks));
}
// LDS row pitch of a 3 x 3 image with `cpg` chunks per pixel (the rule of conv_lds_row for 64 cpg + 16 byte pixels)
int conv_lds_row_3x3_c(int tw_log2, int cpg) {
  const int pixb = 64 * cpg + 16;
  const int pw = ((1 << tw_log2) - 1) + 3;
  int rb = (pw * pixb + 15) & ~15;
  if (tw_log2 == 4) rb = (rb + 255) & ~255;
  else if (tw_log2 == 3) { rb = (rb + 127) & ~127; if (((rb >> 7) & 1) == 0) rb += 128; }
  return rb;
}
// LDS row pitch of a 1 x 1 conv's image with four chunks per pixel (272-byte pixels: no halo, consecutive pixels)
int conv_lds_row_1x1_c4(int tw_log2) { return (((1 << tw_log2) * (64 * 4 + 16)) + 15) & ~15; }
static constexpr int kWsLoaderWaves = 4;
// patch pixels the WS loaders are sized for: one image per tile, square-ish tiles (16x8 / 8x16 -> 18x10,
// 8x8 -> 10x10, 16x4 -> 18x6); other shapes fall back to k_conv_mfma
static constexpr int ws_max_px(int tile, int ks) {
  const int bm = 32 * kTiles[tile].MB * kTiles[tile].WM;
  return ks == 1 ? bm : (bm == 256 ? 324 : (bm == 128 ? 180 : 108));
}
static constexpr int ws_nit(int tile, int ks, int cpg) {   // staging units per loader thread per group
  const int upg = 4 * cpg;
  return (ws_max_px(tile, ks) * upg + kWsLoaderWaves * 64 - 1) / (kWsLoaderWaves * 64);
}
size_t conv_ws_lds_bytes(int dtype, int tile, int ks, const ConvArgs& a) {
  if (!ws_tile_ok(tile) || !(ks == 1 || ks == 3)) return 0;
  if (tile == TILE_256x64 && ks != 3) return 0;
  if (ws_tile(tile).NB == 2 && dtype == 0) return 0;
  const int cpg = ws_cpg_of(ks, a.ws_cpg);
  if (ks == 3 && cpg == 4) {
    const int KC = dtype != 0 ? 32 : 16;
    if (tile != TILE_64x128 || a.cpg == 2 || (a.kchunks & 3) || a.C0 % (4 * KC) || a.C1 % (4 * KC)) return 0;
    if ((1 << (a.tw_log2 + a.th_log2 + a.tb_log2)) != conv_tile_info(tile).BM) return 0;
    if (a.lds_row != conv_lds_row_3x3_c(a.tw_log2, 4)) return 0;
  } else if (ks == 3 && cpg == 2) {
    // two-chunk 3 x 3 groups: instantiated for the 64-pixel tile (the 16 x 16 maps, where the loaders' per-item costs
    // bound the item); 144-byte pixels, the row pitch of the other two-chunk kernel
    const int KC = dtype != 0 ? 32 : 16;
    if (!(tile == TILE_64x128 || tile == TILE_128x128) || a.cpg == 2 || (a.kchunks & 1) || a.C0 % (2 * KC) || a.C1 % (2 * KC)) return 0;
    if ((1 << (a.tw_log2 + a.th_log2 + a.tb_log2)) != conv_tile_info(tile).BM) return 0;
    if (a.lds_row != conv_lds_row_g2(a.tw_log2)) return 0;
  } else if (ks == 1 && cpg == 4) {
    // four-chunk 1 x 1 groups (128 input channels per item): the 64- and 128-pixel tiles with 128 output channels
    const int KC = dtype != 0 ? 32 : 16;
    if (!(tile == TILE_64x128 || tile == TILE_128x128) || (a.kchunks & 3) || a.C0 % (4 * KC) || a.C1 % (4 * KC)) return 0;
    if ((1 << (a.tw_log2 + a.th_log2 + a.tb_log2)) != conv_tile_info(tile).BM) return 0;
    if (a.lds_row != conv_lds_row_1x1_c4(a.tw_log2)) return 0;
  } else if (conv_lds_bytes(dtype, tile, ks, 1, a) == 0) return 0;
  if (patch_pixels(ks, 1, a) > ws_max_px(tile, ks)) return 0;
  if (a.up && (a.tw_log2 == 0 || a.th_log2 == 0)) return 0;   // the loaders assume an even tile origin when upsampling
  const int ph = ((1 << a.th_log2) - 1) + ks;
  const size_t bufb = (size_t)(ph << a.tb_log2) * a.lds_row;
  const size_t rawb = (size_t)ws_nit(tile, ks, cpg) * (kWsLoaderWaves * 64 * 16);
  const size_t affb = a.has_gn ? (((size_t)2 * (a.C0 + a.C1) * 4 + 15) & ~(size_t)15) : 0;  // never keyed on a pointer
  const WsTileCfg wt = ws_tile(tile);
  const size_t nbuf = (size_t)ws_nbuf(32 * wt.MB * wt.WM, ks, wt.NB, cpg);
  const size_t total = nbuf * bufb + 3 * affb + (size_t)(ws_depth(tile, ks, cpg) + 1) * rawb + (nbuf >= 3 ? 32 : 0);
  if (a.tb_log2 != 0 || a.kchunks / cpg < 2) return 0;   // one image per tile, >= 2 channel groups
  // whole 32-channel blocks, float4 epilogue, scale/shift staged by 256 threads x float4
  if ((long long)a.B * a.Ho * a.Wo * std::max(a.out_ld, a.resid_ld) >= (1LL << 31)) return 0;   // 32-bit element offsets
  const int al = dtype != 0 ? 7 : 3;   // 16-byte rows in elements of the storage type
  if (a.Cout % (32 * ws_tile(tile).WN * ws_tile(tile).NB) != 0 || (a.out_ld & al) != 0 || (a.resid_ld & al) != 0 ||
      a.C0 + a.C1 > 1024)
    return 0;
  if (dtype != 0 && !(a.act_bf16 && a.out_bf16)) return 0;   // this kernel reads and writes the storage type only
  return total <= 160 * 1024 ? total : 0;
}

template <typename DT, int TILE, int KS, int CPG = conv_cpg(KS)>
static hipError_t launch_ws_one(const ConvArgs* ap, size_t lds, hipStream_t st) {
  if constexpr (!ws_tile_ok(TILE) || (ws_tile(TILE).NB == 2 && sizeof(DT) == 4)) {
    return ap ? hipErrorInvalidValue : hipSuccess;   // (the fp32 build has no two-N-block variant: registers)
  } else {
    constexpr WsTileCfg t = ws_tile(TILE);
    // weight ring, in steps: a full group for one N block per wave, half of it (same bytes, same time) for two
    // (1 x 1 with two N blocks per wave or MB 4: a deeper ring spills, and hipcc's spill path fails on this kernel)
    constexpr int D = KS == 1 ? ((t.NB == 2 || t.MB == 4) ? DSX_RING_1X1_NB2 : DSX_RING_1X1_NB1) : ((t.NB == 2 || t.MB == 4) ? 6 : 18);   // (MB 4: four MFMAs per step, and the registers are needed)
    constexpr int NIT = ws_nit(TILE, KS, CPG);
    constexpr int P = ws_depth(TILE, KS, CPG);
    auto kern = k_conv_ws<DT, t.MB, t.WM, t.WN, t.NB, KS, CPG, D, NIT, P, kWsLoaderWaves>;
    if (!ap)
      return hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    const ConvArgs& a = *ap;
    dim3 grid((unsigned)(a.n_tiles * a.ws_wg_per_n));
    hipLaunchKernelGGL(kern, grid, dim3(256 + 64 * kWsLoaderWaves), lds, st, a);
    return hipGetLastError();
  }
}
template <typename DT>
static hipError_t launch_ws_dt(int tile, int ks, const ConvArgs* a, size_t lds, hipStream_t st) {
#define DSX_WS_CASE(T) \
  case T: return ks == 3 ? launch_ws_one<DT, T, 3>(a, lds, st) : launch_ws_one<DT, T, 1>(a, lds, st);
  if ((tile == TILE_64x128 || tile == TILE_128x128) && ks == 1) {   // + the four-chunk 1 x 1 form of these tiles (ConvArgs::ws_cpg == 4)
    if (!a) {
      hipError_t e = launch_ws_one<DT, TILE_64x128, 1, 4>(a, lds, st);
      if (e == hipSuccess) e = launch_ws_one<DT, TILE_128x128, 1, 4>(a, lds, st);
      if (e != hipSuccess) return e;
    } else if (a->ws_cpg == 4) {
      return tile == TILE_64x128 ? launch_ws_one<DT, TILE_64x128, 1, 4>(a, lds, st) : launch_ws_one<DT, TILE_128x128, 1, 4>(a, lds, st);
    }
  }
  if ((tile == TILE_64x128 || tile == TILE_128x128) && ks == 3) {   // + the two-chunk form of these tiles (ConvArgs::ws_cpg == 2)
    if (!a) {
      hipError_t e = launch_ws_one<DT, TILE_64x128, 3, 2>(a, lds, st);
      if (e == hipSuccess) e = launch_ws_one<DT, TILE_128x128, 3, 2>(a, lds, st);
      if (e == hipSuccess) e = launch_ws_one<DT, TILE_64x128, 3, 4>(a, lds, st);
      if (e != hipSuccess) return e;
    } else if (a->ws_cpg == 2) {
      return tile == TILE_64x128 ? launch_ws_one<DT, TILE_64x128, 3, 2>(a, lds, st) : launch_ws_one<DT, TILE_128x128, 3, 2>(a, lds, st);
    } else if (a->ws_cpg == 4 && tile == TILE_64x128) {
      return launch_ws_one<DT, TILE_64x128, 3, 4>(a, lds, st);
    }
  }
  switch (tile) {
    DSX_WS_CASE(TILE_128x128)
    DSX_WS_CASE(TILE_64x128)
    DSX_WS_CASE(TILE_128x64)
    DSX_WS_CASE(TILE_256x64)
    DSX_WS_CASE(TILE_64x64)
    default: return hipErrorInvalidValue;
  }
#undef DSX_WS_CASE
}
hipError_t launch_conv_ws(int dtype, int tile, int ks, const ConvArgs& a, hipStream_t st) {
  const size_t lds = conv_ws_lds_bytes(dtype, tile, ks, a);
  if (lds == 0 || a.ksplit != 1 || a.ws_wg_per_n < 1 || a.stage_mode != 0) return hipErrorInvalidValue;
  if (a.bias && a.film) return hipErrorInvalidValue;   // the planner folds the conv bias into the FiLM bias
  return dtype == 1 ? launch_ws_dt<__bf16>(tile, ks, &a, lds, st)
       : dtype == 2 ? launch_ws_dt<_Float16>(tile, ks, &a, lds, st) : launch_ws_dt<float>(tile, ks, &a, lds, st);
}

hipError_t conv_init() {
  static bool done = false;
  if (done) return hipSuccess;
  for (int dtype = 0; dtype < 3; ++dtype)
    for (int ks = 1; ks <= 3; ks += 2)
      for (int tile = 0; tile < TILE_COUNT; ++tile)
        for (int stride = 1; stride <= 2; ++stride) {
          if (stride == 2 && !(ks == 3 && tile == TILE_64x64)) continue;
          hipError_t e = dtype == 1 ? launch_dt<__bf16>(tile, ks, stride, nullptr, 0, nullptr)
                       : dtype == 2 ? launch_dt<_Float16>(tile, ks, stride, nullptr, 0, nullptr)
                                    : launch_dt<float>(tile, ks, stride, nullptr, 0, nullptr);
          if (e != hipSuccess) return e;
          if (stride == 1 && ws_tile_ok(tile)) {
            e = dtype == 1 ? launch_ws_dt<__bf16>(tile, ks, nullptr, 0, nullptr)
              : dtype == 2 ? launch_ws_dt<_Float16>(tile, ks, nullptr, 0, nullptr)
                           : launch_ws_dt<float>(tile, ks, nullptr, 0, nullptr);
            if (e != hipSuccess) return e;
          }
        }
  {
    hipError_t e = launch_g2<float>(0, nullptr, 0, nullptr);
    if (e == hipSuccess) e = launch_g2<__bf16>(0, nullptr, 0, nullptr);
    if (e == hipSuccess) e = launch_g2<_Float16>(0, nullptr, 0, nullptr);
    if (e != hipSuccess) return e;
  }
  for (int ks = 1; ks <= 3; ks += 2) {
    hipError_t e = launch_img_dt<float>(ks, nullptr, 0, nullptr);
    if (e == hipSuccess) e = launch_img_dt<__bf16>(ks, nullptr, 0, nullptr);
    if (e == hipSuccess) e = launch_img_dt<_Float16>(ks, nullptr, 0, nullptr);
    if (e != hipSuccess) return e;
  }
  done = true;
  return hipSuccess;
}

// ------------------------------------------------------------ naive direct conv
__global__ void k_conv_naive(const NaiveConvArgs na) {
  const ConvArgs& a = na.c;
  const long long total = (long long)a.B * a.Ho * a.Wo * a.Cout;
  const int C = a.C0 + a.C1;
  const int pad = na.ks / 2;
  const int Hi = a.up ? a.Hs * 2 : a.Hs, Wi = a.up ? a.Ws * 2 : a.Ws;
  for (long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x; idx < total;
       idx += (long long)gridDim.x * blockDim.x) {
    const int n = (int)(idx % a.Cout);
    long long p = idx / a.Cout;
    const int ox = (int)(p % a.Wo); p /= a.Wo;
    const int oy = (int)(p % a.Ho);
    const int b = (int)(p / a.Ho);
    float acc = 0.f;
    for (int dy = 0; dy < na.ks; ++dy) {
      const int iy = oy * na.stride + dy - pad;
      if (iy < 0 || iy >= Hi) continue;
      for (int dx = 0; dx < na.ks; ++dx) {
        const int ix = ox * na.stride + dx - pad;
        if (ix < 0 || ix >= Wi) continue;
        const int sy = a.up ? iy >> 1 : iy, sx = a.up ? ix >> 1 : ix;
        const size_t so = ((size_t)b * a.Hs + sy) * a.Ws + sx;
        const float* w = na.w + (((size_t)n * na.ks + dy) * na.ks + dx) * C;
        for (int c = 0; c < C; ++c) {
          float v = c < a.C0 ? act_load_kind(a.src0, so * a.C0 + c, a.act_bf16)
                             : act_load_kind(a.src1, so * a.C1 + (c - a.C0), a.act_bf16);
          if (a.gn_scale) v = v * a.gn_scale[(size_t)b * C + c] + a.gn_shift[(size_t)b * C + c];
          if (a.swish) v = swish_f(v);
          acc = fmaf(v, w[c], acc);
        }
      }
    }
    const size_t opix = ((size_t)b * a.Ho + oy) * a.Wo + ox;
    float v = acc + (a.bias ? a.bias[n] : 0.f);
    if (a.film) v += a.film[(size_t)b * a.film_bs + n];
    if (a.resid) v += act_load_kind(a.resid, opix * a.resid_ld + n, a.act_bf16);
    if (na.sigmoid_out) v = 1.0f / (1.0f + __expf(-v));
    act_store(a.out, opix * a.out_ld + n, v, a.out_bf16);
  }
}

hipError_t launch_conv_naive(const NaiveConvArgs& a, hipStream_t st) {
  const long long total = (long long)a.c.B * a.c.Ho * a.c.Wo * a.c.Cout;
  long long blocks = (total + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(k_conv_naive, dim3((unsigned)blocks), dim3(256), 0, st, a);
  return hipGetLastError();
}

}  // namespace dsx
