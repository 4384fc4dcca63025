// dsx_kernels.h — launch interface between the host runtime (dsx_runtime.cpp)
// and the gfx950 kernels (dsx_kernels.hip).  Internal; the public ABI is
// include/dsx.h.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace dsx {

// ---------------------------------------------------------------------------
// L2 weight prefetch for the NEXT conv launch.  A conv's weight stream is read once per step and comes from HBM
// (196 MB of weights per step do not stay in the 32 MB of L2); a wave can keep ~18 KiB of it in flight, so at HBM
// latency the stream, not the MFMAs, paces the small-map layers.  The launch in front of a conv (its k_gn_finalize, or
// the previous image-resident conv) therefore touches the 128-byte lines of the weight slices that the conv's
// workgroups on the same XCD will read, so that they wait in that XCD's L2.  Workgroups are dealt round-robin over the 8 XCDs: blocks with equal
// (linear id % 8) share an L2, and the consumer kernels key their N slices on that same label.  Placement only
// affects speed: a different dispatch order makes the prefetch useless, never wrong.
//   slices : `nslices` consecutive ranges of `slice_bytes` at `base`
//   label x needs slice x % nslices when nslices < 8 (8 % nslices == 0), else every slice s with s % 8 == x
// The loads are ordinary (compiler-counted) loads whose values are OR-ed into a word that is stored through `sink`
// at the end of the kernel; `sink` is always nullptr, so nothing is ever stored, but the loads cannot be dropped
// and no register is overwritten behind the compiler's back.
// ---------------------------------------------------------------------------
struct PrefetchArgs {
  const void* base;       // nullptr: nothing to prefetch
  unsigned slice_bytes;
  int nslices;
  unsigned* sink;         // always nullptr
};
// n / d for n, d < 65536 as a multiply-high: magic = d == 1 ? 0 : floor(2^32 / d) + 1 (exact: n * (magic * d - 2^32) < 2^32)
inline unsigned fastdiv_magic(unsigned d) { return d <= 1 ? 0u : (unsigned)((1ull << 32) / d) + 1u; }
#if defined(__HIPCC__)
__device__ __forceinline__ unsigned fastdiv(unsigned n, unsigned magic) { return magic ? __umulhi(n, magic) : n; }
// The loaded words stay in four registers of their own until l2_prefetch_retire: the first line a thread requests of
// each of its first four slices is a plain assignment, so no s_waitcnt stands between the prefetch and the kernel's own
// work (an `acc |= load` there made the issuing waves of k_conv_img wait vmcnt(0), i.e. for their whole weight stream,
// before they converted their slice of the image: they reached the workgroup barrier ~5 k cycles after the others).
// Further lines (a share longer than the thread count, more than four slices per label: not in the plans built here)
// are OR-ed in and do wait.
struct PfAcc { unsigned v[4]; };
__device__ __forceinline__ PfAcc l2_prefetch(const PrefetchArgs& pf, unsigned lin_block, unsigned nblocks_total, int tid, int nthreads) {
  PfAcc acc = {{0u, 0u, 0u, 0u}};
  if (pf.base == nullptr || pf.nslices <= 0) return acc;
  const unsigned label = lin_block & 7u, j = lin_block >> 3, n8 = (nblocks_total + 7u) >> 3;
  const unsigned lines = pf.slice_bytes >> 7;            // whole 128-byte lines (slices are multiples of 1 KiB)
  const unsigned per = (lines + n8 - 1u) / n8, l0 = j * per;
  const unsigned l1 = l0 + per < lines ? l0 + per : lines;
  const int s0 = pf.nslices < 8 ? (int)(label % (unsigned)pf.nslices) : (int)label;
  const int sstep = pf.nslices < 8 ? pf.nslices : 8;     // nslices < 8: exactly one slice
  int sl = s0;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    if (sl < pf.nslices) {
      const char* b = (const char*)pf.base + (size_t)sl * pf.slice_bytes;
      unsigned l = l0 + (unsigned)tid;
      if (l < l1) acc.v[k] = *(const unsigned*)(b + (size_t)l * 128);
      for (l += (unsigned)nthreads; l < l1; l += (unsigned)nthreads) acc.v[k] |= *(const unsigned*)(b + (size_t)l * 128);
      sl += sstep;
    }
  }
  for (; sl < pf.nslices; sl += sstep) {
    const char* b = (const char*)pf.base + (size_t)sl * pf.slice_bytes;
    for (unsigned l = l0 + (unsigned)tid; l < l1; l += (unsigned)nthreads) acc.v[3] |= *(const unsigned*)(b + (size_t)l * 128);
  }
  return acc;
}
__device__ __forceinline__ unsigned l2_prefetch_fold(const PfAcc& acc) { return acc.v[0] | acc.v[1] | acc.v[2] | acc.v[3]; }   // waits for the loads
__device__ __forceinline__ void l2_prefetch_retire(const PrefetchArgs& pf, unsigned acc) {
  if (pf.sink != nullptr) *pf.sink = acc;
}
__device__ __forceinline__ void l2_prefetch_retire(const PrefetchArgs& pf, const PfAcc& acc) {
  if (pf.sink != nullptr) *pf.sink = acc.v[0] | acc.v[1] | acc.v[2] | acc.v[3];   // never taken: keeps the prefetch loads alive
}
#endif

// GroupNorm of the concatenation of (t0, t1) -> scale/shift[b][C0+C1]
struct GnFinArgs {
  const void* part0; int C0, nchunk0, f32_0;   // partials: double (k_chan_stats) or float (conv epilogue)
  const void* part1; int C1, nchunk1, f32_1;
  int B, groups;
  double count;          // elements per channel (H*W)
  const float* gamma;    // [C0+C1]
  const float* beta;
  float eps;
  float* scale;          // [B][C0+C1]
  float* shift;
  PrefetchArgs pf;       // weight slices of the conv this GroupNorm feeds (see l2_prefetch)
};
#if defined(__HIPCC__)
// One (image, group) item of the GroupNorm finalize, by one wave (or NW waves): threads stride over (channel-in-group,
// chunk) partials, fixed assignment + fixed butterfly order -> bitwise reproducible.  Used by k_gn_finalize (one wave per
// workgroup) and by the loader waves of a residual 1 x 1 conv that hosts the finalize of the block's second GroupNorm.
// NW waves share the item (k_gn_finalize_wide: NW = 4 for the 128^2 / 64^2 maps with hundreds of partial rows); their
// wave sums meet in LDS (`red`, 2 * NW doubles) and are added in wave order.
template <int NW>
__device__ __forceinline__ void gn_finalize_item(const GnFinArgs& a, const int b, const int g, const int tid, double* red) {
  constexpr int NT = 64 * NW;
  const int C = a.C0 + a.C1;
  const int cpg = C / a.groups;
  const int c_lo = g * cpg;
  double s = 0, q = 0;
  // channels of this group that live in source 0 / source 1; the loads of a lane are independent, so
  // they are issued 8 at a time (the kernel is pure load latency otherwise)
  const int n0 = max(0, min(a.C0, c_lo + cpg) - c_lo);      // first n0 channels from source 0
  const int n1 = cpg - n0;
  auto accumulate = [&](const void* part, int is_f32, int nsrc, int nchunk, int Csrc, int cbase) __attribute__((always_inline)) {
    const int items = nsrc * nchunk;
    for (int i0 = tid; i0 < items; i0 += NT * 8) {
      double ps[8], pq[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int i = i0 + u * NT;
        ps[u] = 0; pq[u] = 0;
        if (i < items) {
          const int ch = i / nsrc, c = cbase + (i - ch * nsrc);
          const size_t idx = (((size_t)b * nchunk + ch) * Csrc + c) * 2;
          if (is_f32) { const float2 v = *(const float2*)((const float*)part + idx); ps[u] = v.x; pq[u] = v.y; }
          else { const double2 v = *(const double2*)((const double*)part + idx); ps[u] = v.x; pq[u] = v.y; }
        }
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) { s += ps[u]; q += pq[u]; }
    }
  };
  // gamma / beta of this thread's channel: issued with the partial sums (one memory round trip, not two)
  const int c_own = c_lo + tid;
  const bool own = tid < cpg;
  const float g_own = own ? a.gamma[c_own] : 0.f, b_own = own ? a.beta[c_own] : 0.f;
  if (n0 > 0) accumulate(a.part0, a.f32_0, n0, a.nchunk0, a.C0, c_lo);
  if (n1 > 0) accumulate(a.part1, a.f32_1, n1, a.nchunk1, a.C1, c_lo + n0 - a.C0);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { s += __shfl_xor(s, o, 64); q += __shfl_xor(q, o, 64); }
  if constexpr (NW > 1) {
    if ((tid & 63) == 0) { red[2 * (tid >> 6)] = s; red[2 * (tid >> 6) + 1] = q; }
    __syncthreads();
    s = 0; q = 0;
#pragma unroll
    for (int w = 0; w < NW; ++w) { s += red[2 * w]; q += red[2 * w + 1]; }
  }
  const double n = a.count * cpg;
  const double mean = s / n;
  double var = q / n - mean * mean;
  if (var < 0) var = 0;
  const float rstd = (float)(1.0 / sqrt(var + (double)a.eps));
  const float meanf = (float)mean;
  if (own) {
    const float sc = rstd * g_own;
    a.scale[(size_t)b * C + c_own] = sc;
    a.shift[(size_t)b * C + c_own] = b_own - meanf * sc;
  }
  for (int c = c_lo + tid + NT; c < c_lo + cpg; c += NT) {   // groups wider than the workgroup (not in the reference configs)
    const float sc = rstd * a.gamma[c];
    a.scale[(size_t)b * C + c] = sc;
    a.shift[(size_t)b * C + c] = a.beta[c] - meanf * sc;
  }
}
#endif

// ---------------------------------------------------------------------------
// Fused conv:  out = conv_{KSxKS, stride S}( act( gn(x) ) ) + bias + film + resid
//   x is the channel-concatenation of up to two NHWC fp32 tensors (skip
//   connections are never materialised), optionally nearest-upsampled x2.
//   Implicit GEMM on MFMA: M = output pixels, N = Cout, K = taps x Cin.
// ---------------------------------------------------------------------------
struct ConvArgs {
  const void* src0;       // NHWC activations in the storage type: fp32, or bf16 when act_bf16
  const void* src1;
  int act_bf16;           // storage kind of src0 / src1 / resid: 0 fp32, 1 bf16, 2 fp16 (the MFMA operand type)
  int out_bf16;           // storage kind of out (0 for split-K slabs and the network's final output)
  int C0, C1;             // channels per source (C1 == 0: single source)
  int B, Hs, Ws;          // source spatial size (before the optional upsample)
  int up;                 // 1: nearest x2 upsample fused into the patch load
  int Ho, Wo;             // output spatial size
  const float* gn_scale;  // [B][C0+C1] or nullptr:  v = x*scale + shift
  const float* gn_shift;
  int has_gn;             // GroupNorm affine fused (== gn_scale != nullptr once the workspace exists)
  int swish;              // v = v*sigmoid(v) after the affine
  int stage_mode;         // 0: sources aligned to the channel group (buffer loads); 1: 16-byte loads with
                          // per-lane source select; 2: channel counts not multiples of the 16-byte unit
                          // (4 fp32 / 8 bf16 channels): per-element loads
  const void* wpack;      // MFMA-fragment-ordered weights (fp32 or bf16)
  const float* bias;      // [Cout] or nullptr
  const float* film;      // film[b*film_bs + n] or nullptr (FiLM / time-embedding add)
  int film_bs;
  const void* resid;      // [B][Ho][Wo][resid_ld] (storage type) or nullptr
  int resid_ld;
  void* out;              // [B][Ho][Wo][out_ld]
  int out_ld;
  int Cout;
  int nblocks;            // ceil(Cout/32)
  int kchunks;            // ceil((C0+C1)/KC), padded to conv_chunk_multiple(ks)
  int tw_log2, th_log2, tb_log2;  // output tile = TB images x TH x TW pixels
  int tiles_x, tiles_y, m_tiles, n_tiles;
  int lds_row;            // LDS bytes per patch row (conv_lds_row)
  int cpg;                // 0: the kernel family's default chunks per staged group (1 for 3x3, 2 for 1x1);
                          // 2 with a 3x3: the two-chunk variant of k_conv_mfma (64 input channels per barrier: a
                          // 64-channel layer is ONE group, no K loop)
  int ws_wg_per_n;        // warp-specialised kernel: persistent workgroups per N tile (0: k_conv_mfma)
  int ws_cpg;             // k_conv_ws: chunks per (tile, group) item other than the family default -- 2 for a 3 x 3 conv (lds_row =
                          // conv_lds_row_g2), 4 for a 1 x 1 conv (conv_lds_row_1x1_c4); 64- and 128-pixel tiles: the loaders'
                          // per-item costs are paid once per 64 / 128 input channels
  int xcd_bands;          // k_conv_ws: an XCD's workgroups take a contiguous band of M tiles (else round-robin)
  // k_conv_ws start-up without integer divisions (a dozen of them cost ~2000 cycles before the first DMA could be issued):
  // the host passes the quotients it can compute and multiply-high magics (fastdiv) for the per-workgroup ones.
  int ws_map;             // blockIdx -> (N tile, first M tile): 0 N tiles dealt over XCDs (n_tiles in {1,2,4,8}), 1 n_tiles % 8 == 0, 2 plain,
                          // 3 M tiles dealt over XCDs, every N tile on each (1 x 1 convs; ws_per = n_tiles)
  int ws_nt_log2;         // ws_map 0: log2(n_tiles)
  int ws_per;             // ws_map 1: n_tiles / 8
  int ws_adv_x, ws_adv_y, ws_adv_b;   // tile walk stride wpn decomposed: wpn % tiles_x, (wpn / tiles_x) % tiles_y, wpn / (tiles_x * tiles_y)
  int ws_dpy, ws_dpx;     // loader tables: (loader threads / units per pixel) / PW and the remainder
  unsigned mg_tiles_x, mg_per_img, mg_pw, mg_wpn, mg_per;   // fastdiv magics (dividends < 65536, see fastdiv)
  int ws_bigdiv;          // m_tiles + wpn >= 65536: the tiles-per-workgroup quotient needs a real division
  float* stat_part;       // fused GroupNorm partials [B][tiles_x*tiles_y*WM][Cout][2] (fp32) or nullptr
  unsigned* handoff_timeouts;  // k_conv_ws with image counters: incremented when a bounded FULL / FREE spin gives up (never in
                               // a correct run; dsx_exec_handoff_timeouts reads it) -- a lost hand-off must not pass silently
  unsigned long long* stamp;  // diagnostic s_memtime stamps of workgroup `stamp_block` (or nullptr)
  int stamp_block;
  int ablate;             // -DDSX_DIAG builds only: DSX_ABLATE timing experiments (results are wrong when non-zero)
  // k_conv_img only (GroupNorm finalised inside the consumer): the partial sums of the sources as their producers
  // left them ([B][nchunk][C][2], double from k_chan_stats or float from a fused epilogue) and the affine parameters
  const void* gn_part0; int gn_nchunk0, gn_pf32_0;
  const void* gn_part1; int gn_nchunk1, gn_pf32_1;
  const float* gn_gamma; const float* gn_beta;
  int gn_groups; float gn_eps;
  PrefetchArgs pf;        // weight slices of the next conv launch (see l2_prefetch)
  int fin_on;             // k_conv_ws: the compute waves, idle until the first image is staged, finalize another GroupNorm
  GnFinArgs fin;          //   (the one between the two convs behind this residual 1 x 1 conv; fin.pf = its consumer's weights)
  int ksplit;             // split-K slices (1 = none); slice s writes raw sums to out + s*slab_stride
  int groups_per_split;   // channel groups per slice
  long long slab_stride;  // elements between slabs
};

// tile configurations compiled for the MFMA conv kernel
enum ConvTile { TILE_256x128 = 0, TILE_128x128, TILE_64x128, TILE_256x64, TILE_128x64, TILE_64x64, TILE_128x32, TILE_256x32, TILE_COUNT };
struct ConvTileInfo { int BM, BN; };
ConvTileInfo conv_tile_info(int tile);
int conv_tile_wm(int tile);   // waves along M (one statistics row per (tile, wm))
bool conv_tile_fuses_stats(int tile);   // k_conv_mfma
bool conv_ws_fuses_stats(int tile);     // k_conv_ws
int conv_ws_tile_wm(int tile);          // k_conv_ws lays its waves out differently
// input-channel chunks are staged in groups of this many (weights are packed/padded to it)
int conv_chunk_multiple(int ks);
int conv_lds_row(int ks, int stride, int tw_log2);
// two-chunk-per-group 3x3 variant of k_conv_mfma (ConvArgs::cpg == 2): TILE_128x64 (experiment) and the narrow-output
// tiles TILE_256x32 / TILE_128x32 (convs with <= 32 output channels, e.g. the UNet's final conv)
int conv_lds_row_g2(int tw_log2);
int conv_lds_row_3x3_c(int tw_log2, int cpg);   // 3 x 3 conv of k_conv_ws with cpg chunks per item (ConvArgs::ws_cpg == 4)
int conv_lds_row_1x1_c4(int tw_log2);   // 1 x 1 conv of k_conv_ws with four chunks per item (ConvArgs::ws_cpg == 4)
size_t conv_g2_lds_bytes(int tile, const ConvArgs& a);
// LDS bytes needed by a launch; 0 if the geometry is not supported by `tile`
size_t conv_lds_bytes(int dtype, int tile, int ks, int stride, const ConvArgs& a);
hipError_t launch_conv(int dtype, int tile, int ks, int stride, const ConvArgs& a, hipStream_t st);
// warp-specialised persistent variant (stride 1, stage_mode 0, no split-K); 0 bytes = not applicable
size_t conv_ws_lds_bytes(int dtype, int tile, int ks, const ConvArgs& a);
hipError_t launch_conv_ws(int dtype, int tile, int ks, const ConvArgs& a, hipStream_t st);
// image-resident kernel for 8 x 8 feature maps (one workgroup = one image x 32 output channels, the whole K inside
// the workgroup, GroupNorm finalised in the prologue, statistics of the result in the epilogue): no split-K slabs,
// no reduce launch, no k_gn_finalize launch.  `gn` says whether a GroupNorm precedes the conv.
bool conv_img_applicable(int dtype, int ks, int stride, const ConvArgs& a, bool gn, int gn_groups);
hipError_t launch_conv_img(int dtype, int ks, const ConvArgs& a, hipStream_t st);
// first conv of the UNet (1..7 input channels): the 9 taps x Cin receptive field as ONE K dimension on the MFMA;
// a.wpack = the im2col-ordered pack (pack_first in dsx_runtime.cpp).  One GroupNorm partial row per (16x16 tile, wave).
bool conv_first_applicable(int ks, int stride, const ConvArgs& a, bool gn);
hipError_t launch_conv_first(int dtype, const ConvArgs& a, hipStream_t st);
// one-time function attributes (dynamic LDS limit); call outside any stream capture
hipError_t conv_init();
hipError_t ops_init();

// Plain direct convolution (one thread per output element), any odd KS, stride 1,
// weights [Cout][KS][KS][Cin] fp32.  Used for the 7x7 ForegroundMask conv and as
// an on-device cross-check of the MFMA path (DSX_CONV_IMPL=naive).
struct NaiveConvArgs {
  ConvArgs c;
  const float* w;  // [Cout][KS][KS][C]
  int ks, stride;
  int sigmoid_out;
};
hipError_t launch_conv_naive(const NaiveConvArgs& a, hipStream_t st);

// ---------------------------------------------------------------------------
// GroupNorm statistics
// ---------------------------------------------------------------------------
// per-(image, pixel-chunk, channel) partial {sum, sumsq} in double:
//   part[((b*nchunk + ch)*C + c)*2 + {0,1}]
hipError_t launch_chan_stats(const void* x, int bf16, int B, int HW, int C, int nchunk, double* part,
                             hipStream_t st);
hipError_t launch_gn_finalize(const GnFinArgs& a, hipStream_t st);

// ---------------------------------------------------------------------------
// time embedding + all FiLM vectors of one forward in one launch
// ---------------------------------------------------------------------------
struct TembArgs {
  int flavour;            // 0 sr3 (PositionalEncoding), 1 ddpm (TimeEmbedding)
  int B;                  // rows to produce
  int n_time;             // B or 1 (broadcast)
  const float* time;      // direct time values, or nullptr -> table[*step_ctr]
  const float* table;     // per-step tcond table (device); per_sample: [step][B]
  const int* step_ctr;
  int per_sample;
  int inner;              // C0
  const float* freq;      // [inner/2]
  const float* w1; const float* b1;   // [4*inner][inner], [4*inner]
  const float* w2; const float* b2;   // [inner][4*inner], [inner]
  const float* wf; const float* bf;   // all FiLM linears stacked: [F][inner], [F]
  int F;
  float* film;            // [B][F]
};
hipError_t launch_temb(const TembArgs& a, hipStream_t st);

// ---------------------------------------------------------------------------
// attention (single head, d = C):  S = QK^T/sqrt(C); P = softmax(S); O = PV
// ---------------------------------------------------------------------------
// out[m][n] = sum_s slab[s][m][n] + bias[n] + film[b][n] + resid[m][n]   (split-K epilogue)
struct SplitKReduceArgs {
  const float* slab; int nsplit; long long slab_stride;
  long long M; int N; int HW;            // b = m / HW
  const float* bias; const float* film; int film_bs;
  const void* resid; int resid_ld;
  void* out;
  int act_bf16;                          // storage kind of resid and out (0 fp32, 1 bf16, 2 fp16)
  float* stat_part;                      // nullptr, or GroupNorm partials [B][HW/16][N][2] of `out` (N % 64 == 0, HW % 16 == 0)
};
hipError_t launch_splitk_reduce(const SplitKReduceArgs& a, hipStream_t st);

// fused single-head attention (dsx_attn.hip): out[b][i][:] = softmax_j(q_i . k_j / div) . v_j ; no L x L tensor in HBM.
// q / k / v: token-major rows of `ld` elements (the three thirds of the qkv conv's output), out: rows of `ldo`.
struct AttnArgs {
  const void* q; const void* k; const void* v; int ld;
  void* out; int ldo;
  int storage;            // element type of q, k, v and out: 0 fp32, 1 bf16, 2 fp16
  int B, L, C;            // images, tokens per image, head dimension (= channels)
  float div, inv_div;     // sqrt(C) and its reciprocal
};
bool attn_supported(int C, int L);
hipError_t launch_attn(const AttnArgs& a, hipStream_t st);

// ---------------------------------------------------------------------------
// layout, sampler update, RNG, tiling
// ---------------------------------------------------------------------------
// dst[b][hw][c] = src[(b*ctot + coff + c)*HW + hw]   (channel slice of an NCHW tensor -> NHWC)
hipError_t launch_nchw_slice_to_nhwc(const float* src, void* dst, int dst_bf16, int B, int C, int ctot, int coff,
                                     int HW, hipStream_t st);
hipError_t launch_nhwc_to_nchw(const float* src, float* dst, int B, int C, int H, int W, hipStream_t st);

struct UpdateArgs {
  float* x;               // state, NHWC [B][H][W][C], always fp32
  void* x_act;            // the first conv's copy of the state in the activation storage type (16-bit builds),
                          // or nullptr when the conv reads `x` itself
  int x_act_kind;         // its storage kind (1 bf16, 2 fp16)
  const float* net;       // UNet output, NHWC
  int use_noise;          // 0 -> Philox normals keyed by (seed, step); 1 -> injected draws
  // per-call values live in device memory so that one captured graph serves every call:
  // loop_params[0] = Philox seed, loop_params[1] = address of the injected noise [steps][B][C][H][W] (NCHW,
  // reference draw order)
  const unsigned long long* loop_params;
  const float* tab;       // device table [6][n_steps]: tcond,a,b,c1,c2,sigma
  int n_steps;            // column stride of `tab` (its capacity)
  const int* step_ctr;
  int predict_eps, clip;
  int per_sample;         // the table columns hold [step][B] values instead of one per step
  int B, C, H, W;
};
hipError_t launch_update(const UpdateArgs& a, hipStream_t st);
hipError_t launch_advance(int* step_ctr, hipStream_t st);
hipError_t launch_randn(float* out, long long n, unsigned long long seed, unsigned long long subseq,
                        hipStream_t st);

// the tiles of one launch: ids first, first + stride, ... (count of them); the tables a kernel indexes with an id
// (`starts` [..][3], `regions` [..][8], `off` [..]) live on the device -- the plan's own (dsx_tileplan), or a per-call
// table with first = 0, stride = 1
struct TileSeq { long long first, stride, count; };
hipError_t launch_tiles_gather(const float* frames, int H, int W, int ph, int pw, const int* starts /*dev*/, TileSeq seq,
                               float* tiles, hipStream_t st);
// crop + dataset normalisation fused (SplitDataset.__getitem__): norm = {mean_inp, std_inp, mean_t0, std_t0, mean_t1, std_t1}
hipError_t launch_tiles_gather_norm(const float* f0, const float* f1, int H, int W, int ph, int pw, const int* starts,
                                    TileSeq seq, float w0, float w1, const double norm[6], int from_norm_target,
                                    float* tin, float* ttar, hipStream_t st);
// source of a paste: whole predicted tiles (count, C, ph, pw) of the sequence, or the gathered packed exchange buffer
// [world][rank_stride] (valid regions [C][h][w] of rank q's tiles q, q + world, ... back to back; `off` = pixel offset
// of each tile id inside its rank's run)
struct StitchSrc {
  const float* base; int packed; int ph, pw;
  const long long* off; long long rank_stride; int world;
};
// paste (gt == nullptr) or paste + per-(tile, workgroup, channel) partial sums for RangeInvariantPsnr:
// part[count][gx][C][8] doubles {sum p, sum p^2, sum g, sum g^2, sum g p, min g, max g, 0}
hipError_t launch_stitch(const StitchSrc& s, int C, const int* regions /*dev*/, TileSeq seq, float* canvas, int H, int W,
                         const float* gt, double* part, int gx, hipStream_t st);
// valid regions of the sequence's tiles -> this rank's flat run (the crop of tile_stitcher.py:38-56 before the collective)
hipError_t launch_tiles_pack(const float* tiles, int C, int ph, int pw, const int* regions, const long long* off,
                             TileSeq seq, float* flat, hipStream_t st);

// relu(u) * sigmoid-mask reduction of the TimePredictor head
hipError_t launch_masked_mean(const float* u, const float* mask, int B, long long n, float* out,
                              hipStream_t st);

}  // namespace dsx
