// dsx_conv.hip — fused GroupNorm-apply + Swish + KxK convolution (+bias +FiLM
// +residual) as an implicit GEMM on gfx950 MFMA, NHWC fp32 activations in HBM.
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
//                  loaded once (coalesced float4 along C), normalised + activated
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
//   epilogue     : accumulator rows are pixels, columns are channels -> each
//                  store instruction writes 2 x 128 B contiguous NHWC segments.
#include "dsx_kernels.h"

namespace dsx {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;


__device__ __forceinline__ float swish_f(float v) {
  return __fdividef(v, 1.0f + __expf(-v));
}

__device__ __forceinline__ unsigned pack_bf16x2(float lo, float hi) {
  __bf16 l = (__bf16)lo, h = (__bf16)hi;  // RNE (v_cvt_pk_bf16_f32)
  unsigned short ls = __builtin_bit_cast(unsigned short, l);
  unsigned short hs = __builtin_bit_cast(unsigned short, h);
  return (unsigned)ls | ((unsigned)hs << 16);
}

template <typename DT> struct Chunk;
template <> struct Chunk<float> { static constexpr int KC = 16; };
template <> struct Chunk<__bf16> { static constexpr int KC = 32; };

// MB   : 32-row M blocks per wave;  WM x WN waves (WM*WN == 4); every wave owns ONE
//        32-channel N block, so with WM == 1 no weight fragment is loaded twice.
// CPG  : channel chunks staged per barrier ("group"); 1 for 3x3, 2 for 1x1 (few steps per chunk)
// D    : depth of the weight-fragment prefetch ring (global -> VGPR), steps ahead
template <typename DT, int MB, int WM, int WN, int KS, int S, int CPG, int D, int MAX_IT>
__global__ __launch_bounds__(256) void k_conv_mfma(const ConvArgs a) {
  constexpr int KC = Chunk<DT>::KC;
  constexpr int UPP = KC / 4;                 // 4-channel staging units per pixel per chunk
  constexpr int UPG = UPP * CPG;              // ... per group
  constexpr int UPG_LOG2 = UPG == 16 ? 4 : (UPG == 8 ? 3 : 2);
  constexpr int UB = 4 * (int)sizeof(DT);     // LDS bytes per staging unit
  constexpr int PIXB = 64 * CPG + 16;         // LDS bytes per patch pixel (odd multiple of 16: conflict-free b128 reads)
  constexpr int TAPS = KS * KS;
  constexpr int PAD = KS / 2;
  constexpr int NSTEP = CPG * TAPS * 2;       // MFMA steps per group: (chunk, tap, 32-B half)
  constexpr bool IS_BF16 = sizeof(DT) == 2;
  static_assert(NSTEP % D == 0, "ring depth must divide the steps per group");
  static_assert(WM * WN == 4, "4 waves");

  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int li = lane & 31, lh = lane >> 5;

  const int TW = 1 << a.tw_log2, TH = 1 << a.th_log2;
  const int PW = (TW - 1) * S + KS;
  const int PH = (TH - 1) * S + KS;
  const int PPI = PH * PW;                       // patch pixels per image
  const int PP = PPI << a.tb_log2;               // patch pixels per tile
  const int BUFB = (PP * PIXB + 15) & ~15;
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

  // ---- staging plan: which source pixel feeds each of this thread's units
  const int nunits = PP << UPG_LOG2;
  int soff[MAX_IT];   // source pixel index, or -1 (zero padding / outside batch)
  int simg[MAX_IT];   // image index (GroupNorm scale/shift lookup when a tile spans images)
#pragma unroll
  for (int it = 0; it < MAX_IT; ++it) {
    const int u = tid + it * 256;
    const int pix = u >> UPG_LOG2;
    int so = -1, b = b0;
    if (u < nunits) {
      const int tb = pix / PPI;
      const int rem = pix - tb * PPI;
      const int py = rem / PW;
      const int px = rem - py * PW;
      const int iy = iy0 + py, ix = ix0 + px;
      b = b0 + tb;
      if (b < a.B && iy >= 0 && iy < Hi && ix >= 0 && ix < Wi) {
        const int sy = a.up ? (iy >> 1) : iy;
        const int sx = a.up ? (ix >> 1) : ix;
        so = (b * a.Hs + sy) * a.Ws + sx;
      }
    }
    soff[it] = so;
    simg[it] = b;
  }
  const int cvg = tid & (UPG - 1);  // this thread's 4-channel unit inside a group (same for all its units)

  // ---- A-fragment LDS base offsets for this wave's MB row blocks
  int abase[MB];
#pragma unroll
  for (int mb = 0; mb < MB; ++mb) {
    const int m = (wm * MB + mb) * 32 + li;
    const int tx = m & (TW - 1);
    const int ty = (m >> a.tw_log2) & (TH - 1);
    const int tb = m >> (a.tw_log2 + a.th_log2);
    abase[mb] = ((tb * PH + ty * S) * PW + tx * S) * PIXB + lh * 16;
  }

  // ---- B fragments: fragment-packed weights, one coalesced 16-B load per lane per step,
  //      kept D steps ahead in a register ring (the step index runs on across groups)
  int blk = nt * WN + wn;
  if (blk >= a.nblocks) blk = a.nblocks - 1;  // results of a clamped block are never stored
  const uint4* wpb = (const uint4*)a.wpack + (size_t)blk * kgroups * (NSTEP * 64) + lane;
  const int q_end = g1 * NSTEP;               // one past this workgroup's last step
  uint4 bq[D];
#pragma unroll
  for (int j = 0; j < D; ++j) bq[j] = wpb[(size_t)min(g0 * NSTEP + j, q_end - 1) * 64];

  f32x16 acc[MB];
#pragma unroll
  for (int mb = 0; mb < MB; ++mb)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[mb][r] = 0.0f;

  float4 stg[MAX_IT];

  // issue the global loads of group g into registers
  auto stage_load = [&](int g) {
    const int c = g * (CPG * KC) + cvg * 4;
#pragma unroll
    for (int it = 0; it < MAX_IT; ++it) {
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      const int so = soff[it];
      if (so >= 0) {
        if (!a.scalar_stage) {
          if (c < a.C0) v = *(const float4*)(a.src0 + (size_t)so * a.C0 + c);
          else if (c < C) v = *(const float4*)(a.src1 + (size_t)so * a.C1 + (c - a.C0));
        } else {
          float e[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int cc = c + j;
            e[j] = 0.f;
            if (cc < a.C0) e[j] = a.src0[(size_t)so * a.C0 + cc];
            else if (cc < C) e[j] = a.src1[(size_t)so * a.C1 + (cc - a.C0)];
          }
          v = make_float4(e[0], e[1], e[2], e[3]);
        }
      }
      stg[it] = v;
    }
  };

  // GroupNorm affine + Swish in registers, convert, park in LDS buffer `buf`
  auto stage_store = [&](int g, int buf) {
    const int c = g * (CPG * KC) + cvg * 4;
    unsigned char* dst = lds + buf * BUFB;
    const bool has_gn = a.gn_scale != nullptr && c < C;
    float sc[4] = {1.f, 1.f, 1.f, 1.f}, sh[4] = {0.f, 0.f, 0.f, 0.f};
    auto load_affine = [&](int b) {
      const size_t gi = (size_t)b * C + c;
      if (!a.scalar_stage) {
        const float4 s4 = *(const float4*)(a.gn_scale + gi);
        const float4 h4 = *(const float4*)(a.gn_shift + gi);
        sc[0] = s4.x; sc[1] = s4.y; sc[2] = s4.z; sc[3] = s4.w;
        sh[0] = h4.x; sh[1] = h4.y; sh[2] = h4.z; sh[3] = h4.w;
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          sc[j] = (c + j < C) ? a.gn_scale[gi + j] : 0.f;
          sh[j] = (c + j < C) ? a.gn_shift[gi + j] : 0.f;
        }
      }
    };
    if (has_gn && !multi_img) load_affine(b0);  // one image per tile: same (b, c) for every unit
#pragma unroll
    for (int it = 0; it < MAX_IT; ++it) {
      const int u = tid + it * 256;
      if (u < nunits) {
        float4 v = stg[it];
        if (soff[it] >= 0 && c < C) {
          if (has_gn) {
            if (multi_img) load_affine(simg[it]);
            v.x = v.x * sc[0] + sh[0];
            v.y = v.y * sc[1] + sh[1];
            v.z = v.z * sc[2] + sh[2];
            v.w = v.w * sc[3] + sh[3];
          }
          if (a.swish) {
            v.x = swish_f(v.x); v.y = swish_f(v.y); v.z = swish_f(v.z); v.w = swish_f(v.w);
          }
          if (a.scalar_stage) {  // channels past C inside the last 4-group must stay 0
            if (c + 1 >= C) v.y = 0.f;
            if (c + 2 >= C) v.z = 0.f;
            if (c + 3 >= C) v.w = 0.f;
          }
        }
        const int pix = u >> UPG_LOG2;
        unsigned char* p = dst + pix * PIXB + cvg * UB;
        if constexpr (IS_BF16) {
          uint2 w;
          w.x = pack_bf16x2(v.x, v.y);
          w.y = pack_bf16x2(v.z, v.w);
          *(uint2*)p = w;
        } else {
          *(float4*)p = v;
        }
      }
    }
  };

  // ---- main loop over channel groups (double-buffered LDS, one barrier per group)
  stage_load(g0);
  stage_store(g0, 0);
  __syncthreads();

  for (int g = g0; g < g1; ++g) {
    const bool more = (g + 1) < g1;
    const unsigned char* abuf = lds + ((g - g0) & 1) * BUFB;
    const int qbase = g * NSTEP;

    if (more) stage_load(g + 1);

#pragma unroll
    for (int s = 0; s < NSTEP; ++s) {
      const uint4 bcur = bq[s % D];
      bq[s % D] = wpb[(size_t)min(qbase + s + D, q_end - 1) * 64];
      const int cg = s / (TAPS * 2), tap = (s >> 1) % TAPS, fs = s & 1;
      const int dy = tap / KS, dx = tap % KS;
      const int aoff = (dy * PW + dx) * PIXB + cg * 64 + fs * 32;
#pragma unroll
      for (int mb = 0; mb < MB; ++mb) {
        const uint4 av = *(const uint4*)(abuf + abase[mb] + aoff);
        if constexpr (IS_BF16) {
          acc[mb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, av),
                                                            __builtin_bit_cast(bf16x8, bcur), acc[mb], 0, 0, 0);
        } else {
          const float4 af = __builtin_bit_cast(float4, av);
          const float4 bf = __builtin_bit_cast(float4, bcur);
          acc[mb] = __builtin_amdgcn_mfma_f32_32x32x2f32(af.x, bf.x, acc[mb], 0, 0, 0);
          acc[mb] = __builtin_amdgcn_mfma_f32_32x32x2f32(af.y, bf.y, acc[mb], 0, 0, 0);
          acc[mb] = __builtin_amdgcn_mfma_f32_32x32x2f32(af.z, bf.z, acc[mb], 0, 0, 0);
          acc[mb] = __builtin_amdgcn_mfma_f32_32x32x2f32(af.w, bf.w, acc[mb], 0, 0, 0);
        }
      }
    }

    if (more) stage_store(g + 1, (g + 1 - g0) & 1);
    __syncthreads();
  }

  // ---- epilogue: NHWC stores; split-K slices write raw partial sums to their slab
  const int n = (nt * WN + wn) * 32 + li;
  if (n >= a.Cout) return;
  const bool partial = a.ksplit > 1;
  float* outp = a.out + (partial ? (size_t)split * a.slab_stride : 0);
  const float bias = (!partial && a.bias) ? a.bias[n] : 0.f;
#pragma unroll
  for (int mb = 0; mb < MB; ++mb) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = (r & 3) + 8 * (r >> 2) + 4 * lh;
      const int m = (wm * MB + mb) * 32 + row;
      const int tx = m & (TW - 1);
      const int ty = (m >> a.tw_log2) & (TH - 1);
      const int b = b0 + (m >> (a.tw_log2 + a.th_log2));
      if (b < a.B) {
        const size_t opix = ((size_t)b * a.Ho + (oy0 + ty)) * a.Wo + (ox0 + tx);
        float v = acc[mb][r] + bias;
        if (!partial) {
          if (a.film) v += a.film[(size_t)b * a.film_bs + n];
          if (a.resid) v += a.resid[opix * a.resid_ld + n];
        }
        outp[opix * a.out_ld + n] = v;
      }
    }
  }
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
};

ConvTileInfo conv_tile_info(int tile) {
  const TileCfg& t = kTiles[tile];
  return ConvTileInfo{32 * t.MB * t.WM, 32 * t.WN};
}

static constexpr int conv_cpg(int ks) { return ks == 1 ? 2 : 1; }

// patch-pixel budget of a tile's staging registers
static constexpr int max_px(int tile, int ks, int stride) {
  const int bm = 32 * kTiles[tile].MB * kTiles[tile].WM;
  if (ks == 1) return bm;  // no halo
  return stride == 2 ? 400 : (bm == 256 ? 400 : (bm == 128 ? 220 : 144));
}
static constexpr int max_it(int dtype, int tile, int ks, int stride) {
  const int upg = (dtype == 1 ? 8 : 4) * conv_cpg(ks);
  return (max_px(tile, ks, stride) * upg + 255) / 256;
}

static int patch_pixels(int ks, int stride, const ConvArgs& a) {
  const int TW = 1 << a.tw_log2, TH = 1 << a.th_log2;
  return (((TH - 1) * stride + ks) * ((TW - 1) * stride + ks)) << a.tb_log2;
}

int conv_chunk_multiple(int ks) { return conv_cpg(ks); }

size_t conv_lds_bytes(int dtype, int tile, int ks, int stride, const ConvArgs& a) {
  if (tile < 0 || tile >= TILE_COUNT) return 0;
  if (!(ks == 1 || ks == 3) || !(stride == 1 || (stride == 2 && ks == 3 && tile == TILE_64x64))) return 0;
  const ConvTileInfo ti = conv_tile_info(tile);
  if ((1 << (a.tw_log2 + a.th_log2 + a.tb_log2)) != ti.BM) return 0;
  const int pp = patch_pixels(ks, stride, a);
  if (pp > max_px(tile, ks, stride)) return 0;
  const size_t pixb = 64 * conv_cpg(ks) + 16;
  const size_t bufb = ((size_t)pp * pixb + 15) & ~(size_t)15;
  if (2 * bufb > 64 * 1024) return 0;
  return 2 * bufb;
}

// ap == nullptr: only set the kernel's dynamic-LDS attribute (conv_init)
template <typename DT, int TILE, int KS, int S>
static hipError_t launch_one(const ConvArgs* ap, size_t lds, hipStream_t st) {
  constexpr TileCfg t = kTiles[TILE];
  constexpr int CPG = conv_cpg(KS);
  constexpr int D = KS == 1 ? 4 : 6;
  constexpr int MI = max_it(sizeof(DT) == 2 ? 1 : 0, TILE, KS, S);
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
    default: return ks == 3 ? launch_one<DT, TILE_64x64, 3, 1>(a, lds, st) : launch_one<DT, TILE_64x64, 1, 1>(a, lds, st);
  }
#undef DSX_TILE_CASE
}

hipError_t launch_conv(int dtype, int tile, int ks, int stride, const ConvArgs& a, hipStream_t st) {
  const size_t lds = conv_lds_bytes(dtype, tile, ks, stride, a);
  if (lds == 0 || a.ksplit < 1 || a.n_tiles < 1) return hipErrorInvalidValue;
  return dtype == 1 ? launch_dt<__bf16>(tile, ks, stride, &a, lds, st)
                    : launch_dt<float>(tile, ks, stride, &a, lds, st);
}

hipError_t conv_init() {
  static bool done = false;
  if (done) return hipSuccess;
  for (int dtype = 0; dtype < 2; ++dtype)
    for (int ks = 1; ks <= 3; ks += 2)
      for (int tile = 0; tile < TILE_COUNT; ++tile)
        for (int stride = 1; stride <= 2; ++stride) {
          if (stride == 2 && !(ks == 3 && tile == TILE_64x64)) continue;
          hipError_t e = dtype == 1 ? launch_dt<__bf16>(tile, ks, stride, nullptr, 0, nullptr)
                                    : launch_dt<float>(tile, ks, stride, nullptr, 0, nullptr);
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
          float v = c < a.C0 ? a.src0[so * a.C0 + c] : a.src1[so * a.C1 + (c - a.C0)];
          if (a.gn_scale) v = v * a.gn_scale[(size_t)b * C + c] + a.gn_shift[(size_t)b * C + c];
          if (a.swish) v = swish_f(v);
          acc = fmaf(v, w[c], acc);
        }
      }
    }
    const size_t opix = ((size_t)b * a.Ho + oy) * a.Wo + ox;
    float v = acc + (a.bias ? a.bias[n] : 0.f);
    if (a.film) v += a.film[(size_t)b * a.film_bs + n];
    if (a.resid) v += a.resid[opix * a.resid_ld + n];
    if (na.sigmoid_out) v = 1.0f / (1.0f + __expf(-v));
    a.out[opix * a.out_ld + n] = v;
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
