// dsx_attn.hip — fused single-head self-attention for the UNet's SelfAttention block
// (model/sr3_modules/unet.py:113-142, model/ddpm_modules/unet.py:99-129):
//
//     attn = softmax_{keys}( q . k / sqrt(C) ) ;  out = attn . v          (n_head = 1, head_dim = C)
//
// The reference materialises the (B, 1, H, W, H, W) fp32 score tensor (64 MB per image at the 64x64
// bottleneck of a 512^2 tile); here no L x L tensor ever reaches HBM: one workgroup owns 32 query rows of one
// image and walks the keys in tiles of 128 with an online softmax (running max / running sum per query).
//
// Work split inside the workgroup (4 wave64):
//   S^T = K . Q^T : wave w multiplies ITS 32 keys of the tile by the 32 queries over the whole head dimension
//                   (accumulated over head-dimension chunks staged through LDS): D[key][query], query on the lane.
//   softmax      : row max / row sum per query = 16 registers + one lane^32 exchange + a 4-entry LDS exchange
//                   between the waves; P^T goes to LDS as [query][key] in the operand type.
//   O^T = V^T . P^T: wave w owns the output channels 32w..32w+31 of every 128-channel chunk (the accumulator is
//                   again [channel][query] with the query on the lane, so the online rescale is lane-local).
//                   V is stored token-major in HBM but the MFMA wants 8 consecutive keys per lane: the 16-bit
//                   path transposes V on its way into LDS (key pairs packed into 32-bit words, 16-byte slots
//                   XOR-swizzled so that both the transposed writes and the ds_read_b128 operand reads are
//                   conflict-free); the fp32 path (v_mfma_f32_32x32x2_f32: one element per lane) reads V as stored.
//   MFMA         : bf16 / fp16  v_mfma_f32_32x32x16_{bf16,f16};  fp32  v_mfma_f32_32x32x2_f32 (exact fp32 FMA chain:
//                   the <= 1e-3 parity path).
// Head dimensions are padded to a multiple of 128 (columns past C read as zero), L is arbitrary (keys past L
// are masked to -inf, query rows past L are not stored).
#include "dsx_kernels.h"
#include <type_traits>
#include <utility>

namespace dsx {

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_a;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8_a;

namespace {

template <class F, int... I>
__device__ __forceinline__ __attribute__((always_inline)) void attn_static_for_impl(F&& f, std::integer_sequence<int, I...>) {
  (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F> __device__ __forceinline__ void attn_static_for(F&& f) {
  attn_static_for_impl(f, std::make_integer_sequence<int, N>{});
}

constexpr int BQ = 32;      // query rows per workgroup
constexpr int BK = 128;     // keys per tile (32 per wave)

template <int ST> struct AttnT;       // ST: storage / operand kind (0 fp32, 1 bf16, 2 fp16)
template <> struct AttnT<0> {
  static constexpr int ES = 4;
  static constexpr int DC = 64;       // head-dimension columns per staged K/Q unit (32 KiB of K)
  static constexpr int VK = 64;       // keys per staged V unit (x 128 channels = 32 KiB)
};
template <> struct AttnT<1> { static constexpr int ES = 2; static constexpr int DC = 128; static constexpr int VK = 128; };
template <> struct AttnT<2> { static constexpr int ES = 2; static constexpr int DC = 128; static constexpr int VK = 128; };

__device__ __forceinline__ unsigned pack16(float lo, float hi, std::integral_constant<int, 1>) {
  const __bf16 l = (__bf16)lo, h = (__bf16)hi;
  return (unsigned)__builtin_bit_cast(unsigned short, l) | ((unsigned)__builtin_bit_cast(unsigned short, h) << 16);
}
__device__ __forceinline__ unsigned pack16(float lo, float hi, std::integral_constant<int, 2>) {
  const _Float16 l = (_Float16)lo, h = (_Float16)hi;
  return (unsigned)__builtin_bit_cast(unsigned short, l) | ((unsigned)__builtin_bit_cast(unsigned short, h) << 16);
}

template <int ST>
__device__ __forceinline__ f32x16 mfma16(const uint4 a, const uint4 b, f32x16 c) {
  if constexpr (ST == 1)
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_a, a), __builtin_bit_cast(bf16x8_a, b), c, 0, 0, 0);
  else
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_a, a), __builtin_bit_cast(f16x8_a, b), c, 0, 0, 0);
}

// 16-byte slot swizzle of the transposed V image (rows = channels, 256 B = 128 keys per row):
// conflict-free for the transposed 4-byte writes (rows 8*seg + j) and for the ds_read_b128 operand reads
// (16 consecutive rows of a lane group)
__device__ __forceinline__ int vt_slot(int row, int slot) { return slot ^ (((row >> 3) ^ row) & 15); }

}  // namespace

// NDB: 128-channel chunks of the (padded) head dimension.  CS: output-channel split -- CS workgroups share an (image,
// query tile): each computes the scores over the whole head dimension (the K stream and the cheap S = K.Q^T work are
// repeated) but only NDB / CS of the 128-channel chunks of O = P.V, i.e. it stages 1/CS of V.  For the 16 x 16 maps of
// the SR3 UNet (L = 256: 8 query tiles x 16 images = 128 workgroups on 256 CUs, 16 barrier-separated staged units each)
// CS = 2 fills the chip and shortens every workgroup to 12 units.
template <int ST, int NDB, int CS>
__global__ __launch_bounds__(256, (NDB <= 1 ? 2 : 1)) void k_attn(const AttnArgs a) {
  static_assert(NDB % CS == 0, "whole chunks per workgroup");
  constexpr int NDO = NDB / CS;                 // output chunks of this workgroup
  using T = AttnT<ST>;
  constexpr int ES = T::ES, DC = T::DC, VK = T::VK;
  constexpr bool F32 = ST == 0;
  constexpr int KUNITS = NDB * (128 / DC);      // staged K/Q units per key tile
  constexpr int VPC = BK / VK;                  // staged V units per 128-channel chunk
  constexpr int VUNITS = NDO * VPC;
  // LDS rows (bytes): +16 B (16-bit: conflict-free ds_read_b128 of 32-row fragments) / +4 B (fp32: ds_read_b32)
  constexpr int KROW = F32 ? (DC + 1) * 4 : DC * 2 + 16;
  constexpr int PROW = F32 ? (BK + 1) * 4 : BK * 2 + 16;
  constexpr int VROW = F32 ? 128 * 4 : BK * 2;  // fp32: [key][128 channels]; 16-bit: [channel][128 keys], swizzled
  constexpr int KV_BYTES = (BK * KROW > (F32 ? VK : 128) * VROW) ? BK * KROW : (F32 ? VK : 128) * VROW;
  constexpr int Q_BYTES = BQ * KROW;
  constexpr int P_BYTES = BQ * PROW;

  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* kv_lds = smem;                       // K unit or V unit (time-shared)
  unsigned char* q_lds = smem + KV_BYTES;
  unsigned char* p_lds = q_lds + Q_BYTES;
  float* red = (float*)(p_lds + P_BYTES);             // [2][4][32]: per-wave row max / row sum

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, lh = lane >> 5;

  // workgroup -> (image, query tile); query tiles of one image share an XCD (they read the same K / V)
  const int QT = (a.L + BQ - 1) / BQ;
  const int total = a.B * QT * CS;
  int j = blockIdx.x;
  if ((total & 7) == 0) j = (blockIdx.x & 7) * (total >> 3) + (blockIdx.x >> 3);
  const int half = CS == 1 ? 0 : j % CS;        // which NDO chunks of the output (the CS workgroups of a tile are neighbours: same XCD)
  const int jq = CS == 1 ? j : j / CS;
  const int b = jq / QT, q0 = (jq - b * QT) * BQ;
  const int chunk0 = half * NDO;

  const size_t row0 = (size_t)b * a.L;                // first token row of the image in the qkv tensor
  const size_t ldb = (size_t)a.ld * ES;               // bytes per token row

  // ---- staging plans.  Global reads: 16 lanes x 16 B = one 256-byte row segment per 16 lanes.
  constexpr int SEGS = (DC * ES) / 16;                // 16-byte segments per K/Q unit row (16)
  static_assert(SEGS == 16, "a staged K/Q row is 256 bytes");
  const int seg = tid & 15, rsub = tid >> 4;          // rsub: 0..15
#ifndef DSX_ATTN_PD
#define DSX_ATTN_PD 2
#endif
  constexpr int PD = (KUNITS + VUNITS) < DSX_ATTN_PD ? (KUNITS + VUNITS) : DSX_ATTN_PD;   // staged units in flight (global -> registers)
  uint4 pre[PD][10];                                  // 8 K (or V) pieces + 2 Q pieces each

  // Every load goes through one buffer descriptor over this image's L token rows (q, k and v are column ranges of
  // the same rows): rows past L fall outside the descriptor and read as zero, columns past C get the forced
  // out-of-range offset -> no predicate, no branch, no wait around any load (a per-load "load or zero" branch makes
  // hipcc serialise the loads behind vmcnt(0) waits).
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
      (void*)((const char*)a.q + row0 * ldb), 0, (int)((size_t)a.L * ldb), 0x00020000);
  const unsigned koff = (unsigned)((const char*)a.k - (const char*)a.q), voff = (unsigned)((const char*)a.v - (const char*)a.q);
  auto ld16 = [&](unsigned tensor_off, int row, int col_elem) __attribute__((always_inline)) -> uint4 {
    // 16 bytes = 16/ES elements at column col_elem of token `row` of the image (host: C is a multiple of 16/ES)
    const unsigned vo = col_elem < a.C ? (unsigned)row * (unsigned)ldb + tensor_off + (unsigned)col_elem * ES : 0x80000000u;
    return __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rs, vo, 0, 0));
  };

  // unit ordinal u within a key tile: [0, KUNITS) K/Q units, [KUNITS, KUNITS + VUNITS) V units
  auto issue_unit = [&](int key0, int u, uint4 (&pre)[10]) __attribute__((always_inline)) {
    if (u < KUNITS) {
      const int c0 = u * DC + seg * (16 / ES);
#pragma unroll
      for (int it = 0; it < 8; ++it) {
        const int key = key0 + rsub + 16 * it;
        pre[it] = ld16(koff, key, c0);
      }
#pragma unroll
      for (int it = 0; it < 2; ++it) {
        const int q = q0 + rsub + 16 * it;
        pre[8 + it] = ld16(0u, q, c0);
      }
    } else {
      const int v = u - KUNITS, chunk = v / VPC, part = v - chunk * VPC;
      if constexpr (F32) {
        // [VK keys][128 channels] as stored: thread -> (key = tid/32 + 8*it, 16-byte segment tid%32)
        const int s32 = tid & 31, r8 = tid >> 5;
        const int c0 = (chunk0 + chunk) * 128 + s32 * 4;
#pragma unroll
        for (int it = 0; it < 8; ++it) {
          const int key = key0 + part * VK + r8 + 8 * it;
          pre[it] = ld16(voff, key, c0);
        }
      } else {
        // key pairs (2i, 2i+1) x 16-byte channel segment: thread -> (segment tid%16, pair tid/16 + 16*it)
        const int c0 = (chunk0 + chunk) * 128 + seg * 8;
#pragma unroll
        for (int it = 0; it < 4; ++it) {
          const int key = key0 + 2 * (rsub + 16 * it);
          pre[2 * it] = ld16(voff, key, c0);
          pre[2 * it + 1] = ld16(voff, key + 1, c0);
        }
      }
    }
  };
  auto store_unit = [&](int u, const uint4 (&pre)[10]) __attribute__((always_inline)) {
    if (u < KUNITS) {
      auto put = [&](unsigned char* dst, const uint4 v) __attribute__((always_inline)) {
        if constexpr (F32) {   // rows are 260 bytes apart (conflict-free one-float-per-lane reads): 4-byte stores
          unsigned* d = (unsigned*)dst;
          d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
        } else {
          *(uint4*)dst = v;
        }
      };
#pragma unroll
      for (int it = 0; it < 8; ++it) put(kv_lds + (rsub + 16 * it) * KROW + seg * 16, pre[it]);
#pragma unroll
      for (int it = 0; it < 2; ++it) put(q_lds + (rsub + 16 * it) * KROW + seg * 16, pre[8 + it]);
    } else {
      if constexpr (F32) {
        const int s32 = tid & 31, r8 = tid >> 5;
#pragma unroll
        for (int it = 0; it < 8; ++it) *(uint4*)(kv_lds + (r8 + 8 * it) * VROW + s32 * 16) = pre[it];
      } else {
#pragma unroll
        for (int it = 0; it < 4; ++it) {
          const int pair = rsub + 16 * it;            // keys 2*pair, 2*pair+1 -> 32-bit word `pair` of a channel row
          const unsigned ea[4] = {pre[2 * it].x, pre[2 * it].y, pre[2 * it].z, pre[2 * it].w};
          const unsigned eb[4] = {pre[2 * it + 1].x, pre[2 * it + 1].y, pre[2 * it + 1].z, pre[2 * it + 1].w};
#pragma unroll
          for (int jj = 0; jj < 8; ++jj) {
            const unsigned lo = (jj & 1) ? (ea[jj >> 1] >> 16) : (ea[jj >> 1] & 0xffffu);
            const unsigned hi = (jj & 1) ? (eb[jj >> 1] & 0xffff0000u) : (eb[jj >> 1] << 16);
            const int row = seg * 8 + jj;             // channel inside the 128-channel chunk
            *(unsigned*)(kv_lds + row * VROW + vt_slot(row, pair >> 2) * 16 + (pair & 3) * 4) = lo | hi;
          }
        }
      }
    }
  };

  f32x16 oacc[NDO];
#pragma unroll
  for (int c = 0; c < NDO; ++c)
#pragma unroll
    for (int r = 0; r < 16; ++r) oacc[c][r] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;              // per query (lane & 31), replicated in every wave
  f32x16 sacc;

  const int ntiles = (a.L + BK - 1) / BK;
  constexpr int UPT = KUNITS + VUNITS;
  static_assert(UPT % PD == 0 && UPT >= PD, "the register slot of a unit is a compile-time constant");
  attn_static_for<PD>([&](auto pc) __attribute__((always_inline)) {
    constexpr int u0 = decltype(pc)::value;
    issue_unit(0, u0, pre[u0]);
  });
  for (int t = 0; t < ntiles; ++t) {
    const int key0 = t * BK;
#pragma unroll
    for (int r = 0; r < 16; ++r) sacc[r] = 0.f;
    attn_static_for<UPT>([&](auto uc) __attribute__((always_inline)) {
      constexpr int u = decltype(uc)::value;
      __syncthreads();                                // every wave is done with the previous unit's LDS image
      store_unit(u, pre[u % PD]);
      __syncthreads();
      {                                               // the loads of unit u + PD fly during the MFMAs of u .. u + PD - 1
        constexpr int nu = (u + PD) % UPT;
        const int nk = (u + PD >= UPT) ? key0 + BK : key0;
        if (nk < a.L) issue_unit(nk, nu, pre[u % PD]);
      }
      if constexpr (u < KUNITS) {
        // ---- S^T[key][query] += K[key][d] . Q[query][d] over this unit's DC columns
        if constexpr (F32) {
          const float* kr = (const float*)(kv_lds + (wave * 32 + li) * KROW);
          const float* qr = (const float*)(q_lds + li * KROW);
#pragma unroll 8
          for (int s = 0; s < DC / 2; ++s)
            sacc = __builtin_amdgcn_mfma_f32_32x32x2f32(kr[2 * s + lh], qr[2 * s + lh], sacc, 0, 0, 0);
        } else {
          const unsigned char* kr = kv_lds + (wave * 32 + li) * KROW + lh * 16;
          const unsigned char* qr = q_lds + li * KROW + lh * 16;
#pragma unroll
          for (int s = 0; s < DC / 16; ++s)
            sacc = mfma16<ST>(*(const uint4*)(kr + s * 32), *(const uint4*)(qr + s * 32), sacc);
        }
        if constexpr (u == KUNITS - 1) {
          // ---- online softmax over this tile's 128 keys.  Register r of lane (query li, half lh) is key
          //      32*wave + (r & 3) + 8*(r >> 2) + 4*lh of the tile.
          float s[16];
          float mx = -INFINITY;
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int key = key0 + wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            float v = F32 ? sacc[r] / a.div : sacc[r] * a.inv_div;     // the reference divides by sqrt(C) (unet.py:134)
            if (key >= a.L) v = -INFINITY;
            s[r] = v;
            mx = fmaxf(mx, v);
          }
          mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
          if (lh == 0) red[wave * 32 + li] = mx;
          __syncthreads();
          const float tmax = fmaxf(fmaxf(red[li], red[32 + li]), fmaxf(red[64 + li], red[96 + li]));
          const float m_new = fmaxf(m_run, tmax);     // finite: every tile holds at least one valid key
          const float alpha = F32 ? expf(m_run - m_new) : __expf(m_run - m_new);  // first tile: exp(-inf) = 0
          float psum = 0.f;
#pragma unroll
          for (int r = 0; r < 16; ++r) { s[r] = F32 ? expf(s[r] - m_new) : __expf(s[r] - m_new); psum += s[r]; }
          psum += __shfl_xor(psum, 32, 64);
          if (lh == 0) red[128 + wave * 32 + li] = psum;
          // P^T -> LDS as [query][key] in the operand type
          if constexpr (F32) {
            float* pr = (float*)(p_lds + li * PROW);
#pragma unroll
            for (int r = 0; r < 16; ++r) pr[wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh] = s[r];
          } else {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
              uint2 w;
              w.x = pack16(s[4 * g], s[4 * g + 1], std::integral_constant<int, ST>{});
              w.y = pack16(s[4 * g + 2], s[4 * g + 3], std::integral_constant<int, ST>{});
              *(uint2*)(p_lds + li * PROW + (wave * 32 + 8 * g + 4 * lh) * 2) = w;
            }
          }
          __syncthreads();
          const float tsum = (red[128 + li] + red[160 + li]) + (red[192 + li] + red[224 + li]);
          l_run = l_run * alpha + tsum;
          m_run = m_new;
#pragma unroll
          for (int c = 0; c < NDO; ++c)
#pragma unroll
            for (int r = 0; r < 16; ++r) oacc[c][r] *= alpha;
        }
      } else {
        // ---- O^T[channel][query] += V^T[channel][key] . P^T[key][query] for this unit's keys
        constexpr int v = u - KUNITS, chunk = v / VPC, part = v - chunk * VPC;
        f32x16 acc = oacc[chunk];
        if constexpr (F32) {
          const float* vr = (const float*)kv_lds + wave * 32 + li;            // V[key][channel], 128 floats per key
          const float* pr = (const float*)(p_lds + li * PROW) + part * VK;
#pragma unroll 8
          for (int s = 0; s < VK / 2; ++s)
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(vr[(2 * s + lh) * 128], pr[2 * s + lh], acc, 0, 0, 0);
        } else {
          const int row = wave * 32 + li;
          const unsigned char* vr = kv_lds + row * VROW;
          const unsigned char* pr = p_lds + li * PROW + lh * 16;
#pragma unroll
          for (int s = 0; s < BK / 16; ++s)
            acc = mfma16<ST>(*(const uint4*)(vr + vt_slot(row, 2 * s + lh) * 16), *(const uint4*)(pr + s * 32), acc);
        }
        oacc[chunk] = acc;
      }
    });
  }

  // ---- out[query][channel] = O^T / l.  Register r of lane (query li, half lh): channel 128c + 32*wave + (r&3) + 8*(r>>2) + 4*lh
  const int q = q0 + li;
  if (q >= a.L) return;
  const float inv_l = 1.0f / l_run;
  char* orow = (char*)a.out + ((size_t)b * a.L + q) * (size_t)a.ldo * ES;
#pragma unroll
  for (int c = 0; c < NDO; ++c) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int ch = (chunk0 + c) * 128 + wave * 32 + 8 * g + 4 * lh;
      if (ch >= a.C) continue;                        // C is a multiple of 4 (host check): whole groups of 4
      const float x0 = oacc[c][4 * g] * inv_l, x1 = oacc[c][4 * g + 1] * inv_l;
      const float x2 = oacc[c][4 * g + 2] * inv_l, x3 = oacc[c][4 * g + 3] * inv_l;
      if constexpr (F32) {
        *(float4*)(orow + (size_t)ch * 4) = make_float4(x0, x1, x2, x3);
      } else {
        uint2 w;
        w.x = pack16(x0, x1, std::integral_constant<int, ST>{});
        w.y = pack16(x2, x3, std::integral_constant<int, ST>{});
        *(uint2*)(orow + (size_t)ch * 2) = w;
      }
    }
  }
}

namespace {
template <int ST> constexpr size_t attn_lds_bytes() {
  using T = AttnT<ST>;
  constexpr bool F32 = ST == 0;
  constexpr int KROW = F32 ? (T::DC + 1) * 4 : T::DC * 2 + 16;
  constexpr int PROW = F32 ? (BK + 1) * 4 : BK * 2 + 16;
  constexpr int VROW = F32 ? 128 * 4 : BK * 2;
  constexpr int KV = (BK * KROW > (F32 ? T::VK : 128) * VROW) ? BK * KROW : (F32 ? T::VK : 128) * VROW;
  return (size_t)KV + (size_t)BQ * KROW + (size_t)BQ * PROW + 2 * 4 * 32 * sizeof(float);
}
template <int ST, int NDB, int CS = 1> hipError_t launch_attn_one(const AttnArgs& a, hipStream_t st) {
  const int QT = (a.L + BQ - 1) / BQ;
  hipLaunchKernelGGL((k_attn<ST, NDB, CS>), dim3((unsigned)(a.B * QT * CS)), dim3(256), attn_lds_bytes<ST>(), st, a);
  return hipGetLastError();
}
template <int ST> hipError_t launch_attn_st(const AttnArgs& a, hipStream_t st) {
  const int ndb = (a.C + 127) / 128;
  switch (ndb) {
    case 1: return launch_attn_one<ST, 1>(a, st);
    case 2: return launch_attn_one<ST, 2>(a, st);
    case 3: case 4: {
      // few query tiles (the 16 x 16 maps at batch 16: 128 workgroups): two workgroups per tile, half of the output
      // channels each (measured: 17.9 -> see DESIGN.md); DSX_ATTN_CS=1 keeps one
      static const int cs_on = getenv("DSX_ATTN_CS") ? atoi(getenv("DSX_ATTN_CS")) : 2;
      const int QT = (a.L + BQ - 1) / BQ;
      if (cs_on == 2 && a.B * QT <= 160) return launch_attn_one<ST, 4, 2>(a, st);
      return launch_attn_one<ST, 4>(a, st);
    }
    case 5: case 6: case 7: case 8: return launch_attn_one<ST, 8>(a, st);
    default: return hipErrorInvalidValue;             // head dimension > 1024 (no reference config)
  }
}
}  // namespace

bool attn_supported(int C, int L) { return C >= 8 && C <= 1024 && (C & 7) == 0 && L >= 1 && (long long)L * 3 * C * 4 < (1LL << 31); }

hipError_t launch_attn(const AttnArgs& a, hipStream_t st) {
  if (!attn_supported(a.C, a.L) || a.B < 1) return hipErrorInvalidValue;
  const int epu = a.storage == 0 ? 4 : 8;             // elements per 16-byte unit
  if ((a.ld % epu) || (a.ldo % 4) || ((uintptr_t)a.q & 15) || ((uintptr_t)a.k & 15) || ((uintptr_t)a.v & 15) ||
      ((uintptr_t)a.out & 15))
    return hipErrorInvalidValue;                      // 16-byte loads / 8- or 16-byte stores
  switch (a.storage) {
    case 0: return launch_attn_st<0>(a, st);
    case 1: return launch_attn_st<1>(a, st);
    case 2: return launch_attn_st<2>(a, st);
    default: return hipErrorInvalidValue;
  }
}

}  // namespace dsx
