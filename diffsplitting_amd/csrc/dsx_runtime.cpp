// dsx_runtime.cpp — host side of libdsx.so: UNet topology + parameter table in
// the reference's state_dict order, weight repacking into MFMA fragment order,
// the per-(B,H,W) launch plan, the graph-captured sampling loop and the tile
// planner.  C ABI in include/dsx.h.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <memory>
#include <string>
#include <vector>

#include "../../include/dsx.h"
#include "dsx_kernels.h"

using namespace dsx;

// ------------------------------------------------------------------ errors
static thread_local std::string g_err;
static int fail(int code, const char* fmt, ...) {
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  g_err = buf;
  return code;
}
#define HIP_TRY(expr)                                                                     \
  do {                                                                                    \
    hipError_t e__ = (expr);                                                              \
    if (e__ != hipSuccess)                                                                \
      return fail(DSX_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e__),    \
                  __FILE__, __LINE__);                                                    \
  } while (0)

extern "C" const char* dsx_last_error(void) { return g_err.c_str(); }
extern "C" int dsx_abi_version(void) { return DSX_ABI_VERSION; }
extern "C" int dsx_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); return 0; }
  return n;
}

// ------------------------------------------------------------------ model
namespace {

struct Param {
  std::string name;
  std::vector<int64_t> shape;
  std::vector<float> host;
  bool set = false;
  int64_t numel() const {
    int64_t n = 1;
    for (auto s : shape) n *= s;
    return n;
  }
};

struct ConvW {
  int pw = -1, pb = -1;  // param indices (weight, bias)
  int cin = 0, cout = 0, ks = 1;
  int kchunks = 0, nblocks = 0;
  void* pack = nullptr;     // device, fragment order
  void* pack_first = nullptr;   // device, im2col order of k_conv_first (few-input-channel 3x3 convs only)
  float* bias = nullptr;    // device
  float* naive = nullptr;   // device [Cout][ks][ks][Cin] (debug / 7x7 only)
};
struct GnW {
  int pg = -1, pb = -1;
  int C = 0;
  float* gamma = nullptr;
  float* beta = nullptr;
};
struct LinW {
  int pw = -1, pb = -1;
  int in = 0, out = 0;
};

struct Module {
  int kind;  // 0 conv_in, 1 res, 2 down, 3 up, 4 final
  int section;  // 0 downs, 1 mid, 2 ups, 3 final
  int cin = 0, cout = 0, skip = 0;
  bool attn = false;
  ConvW conv;             // conv_in / down / up / final conv
  GnW gn1, gn2, gna;      // res: block1/2 norms, attention norm; final: gn1
  ConvW conv1, conv2, res, qkv, out;
  bool has_res = false;
  LinW film;
  int film_off = -1;
};

}  // namespace

struct dsx_model {
  dsx_unet_cfg cfg;
  std::vector<Param> params;
  std::vector<Module> mods;
  // time embedding
  int p_invfreq = -1;
  LinW t1, t2;
  std::vector<float> freq;  // inner/2
  bool freq_set = false;
  int F = 0;                // stacked FiLM outputs
  // device
  bool finalized = false;
  int dtype = 0;
  char* arena = nullptr;
  size_t arena_bytes = 0;
  float *d_freq = nullptr, *d_w1 = nullptr, *d_b1 = nullptr, *d_w2 = nullptr, *d_b2 = nullptr;
  float *d_wf = nullptr, *d_bf = nullptr;
  bool want_naive = false;
};

static int add_param(dsx_model* m, const std::string& name, std::initializer_list<int64_t> shape) {
  Param p;
  p.name = name;
  p.shape.assign(shape.begin(), shape.end());
  m->params.push_back(std::move(p));
  return (int)m->params.size() - 1;
}
static ConvW add_conv(dsx_model* m, const std::string& pfx, int cin, int cout, int ks, bool bias) {
  ConvW c;
  c.cin = cin; c.cout = cout; c.ks = ks;
  c.pw = add_param(m, pfx + ".weight", {cout, cin, ks, ks});
  if (bias) c.pb = add_param(m, pfx + ".bias", {cout});
  return c;
}
static GnW add_gn(dsx_model* m, const std::string& pfx, int C) {
  GnW g;
  g.C = C;
  g.pg = add_param(m, pfx + ".weight", {C});
  g.pb = add_param(m, pfx + ".bias", {C});
  return g;
}
static LinW add_lin(dsx_model* m, const std::string& pfx, int in, int out) {
  LinW l;
  l.in = in; l.out = out;
  l.pw = add_param(m, pfx + ".weight", {out, in});
  l.pb = add_param(m, pfx + ".bias", {out});
  return l;
}

// ResnetBlocWithAttn in state_dict order (sr3 unet.py:94-158, ddpm unet.py:78-146)
static void add_res(dsx_model* m, const std::string& pfx, Module& md) {
  const dsx_unet_cfg& c = m->cfg;
  const std::string rb = pfx + ".res_block";
  if (c.with_time_emb) {
    md.film = add_lin(m, c.flavour == DSX_FLAVOUR_SR3 ? rb + ".noise_func.noise_func.0" : rb + ".mlp.1",
                      c.inner_channel, md.cout);
    md.film_off = m->F;
    m->F += md.cout;
  }
  md.gn1 = add_gn(m, rb + ".block1.block.0", md.cin);
  md.conv1 = add_conv(m, rb + ".block1.block.3", md.cin, md.cout, 3, true);
  md.gn2 = add_gn(m, rb + ".block2.block.0", md.cout);
  md.conv2 = add_conv(m, rb + ".block2.block.3", md.cout, md.cout, 3, true);
  md.has_res = md.cin != md.cout;
  if (md.has_res) md.res = add_conv(m, rb + ".res_conv", md.cin, md.cout, 1, true);
  if (md.attn) {
    md.gna = add_gn(m, pfx + ".attn.norm", md.cout);
    md.qkv = add_conv(m, pfx + ".attn.qkv", md.cout, 3 * md.cout, 1, false);
    md.out = add_conv(m, pfx + ".attn.out", md.cout, md.cout, 1, true);
  }
}

extern "C" int dsx_model_create(const dsx_unet_cfg* cfg, dsx_model** out) {
  if (!cfg || !out) return fail(DSX_ERR_INVALID, "null argument");
  if (cfg->n_mults < 1 || cfg->n_mults > 8 || cfg->n_attn_res < 0 || cfg->n_attn_res > 8)
    return fail(DSX_ERR_INVALID, "bad n_mults/n_attn_res");
  if (cfg->inner_channel < 4 || cfg->inner_channel % 4 || cfg->norm_groups < 1)
    return fail(DSX_ERR_INVALID, "inner_channel must be a positive multiple of 4");
  for (int i = 0; i < cfg->n_mults; ++i)
    if ((cfg->inner_channel * cfg->channel_mults[i]) % cfg->norm_groups)
      return fail(DSX_ERR_INVALID, "norm_groups must divide every level's channel count");
  dsx_model* m = new dsx_model();
  m->cfg = *cfg;
  const int inner = cfg->inner_channel;
  auto in_attn = [&](int res) {
    for (int i = 0; i < cfg->n_attn_res; ++i)
      if (cfg->attn_res[i] == res) return true;
    return false;
  };
  // time embedding MLP (sr3 unet.py:177-187 / ddpm unet.py:163-173)
  if (cfg->with_time_emb) {
    if (cfg->flavour == DSX_FLAVOUR_SR3) {
      m->t1 = add_lin(m, "noise_level_mlp.1", inner, 4 * inner);
      m->t2 = add_lin(m, "noise_level_mlp.3", 4 * inner, inner);
    } else {
      m->p_invfreq = add_param(m, "time_mlp.0.inv_freq", {inner / 2});
      m->t1 = add_lin(m, "time_mlp.1", inner, 4 * inner);
      m->t2 = add_lin(m, "time_mlp.3", 4 * inner, inner);
    }
  }
  // downs
  int pre = inner, now_res = cfg->image_size, idx = 0;
  std::vector<int> feat{pre};
  {
    Module md{};
    md.kind = 0; md.section = 0; md.cin = cfg->in_channel; md.cout = inner;
    md.conv = add_conv(m, "downs.0", cfg->in_channel, inner, 3, true);
    m->mods.push_back(md);
    idx = 1;
  }
  for (int ind = 0; ind < cfg->n_mults; ++ind) {
    const bool last = ind == cfg->n_mults - 1;
    const bool use_attn = in_attn(now_res);
    const int ch = inner * cfg->channel_mults[ind];
    for (int r = 0; r < cfg->res_blocks; ++r) {
      Module md{};
      md.kind = 1; md.section = 0; md.cin = pre; md.cout = ch; md.attn = use_attn;
      add_res(m, "downs." + std::to_string(idx++), md);
      m->mods.push_back(md);
      feat.push_back(ch);
      pre = ch;
    }
    if (!last) {
      Module md{};
      md.kind = 2; md.section = 0; md.cin = pre; md.cout = pre;
      md.conv = add_conv(m, "downs." + std::to_string(idx++) + ".conv", pre, pre, 3, true);
      m->mods.push_back(md);
      feat.push_back(pre);
      now_res /= 2;
    }
  }
  for (int k = 0; k < 2; ++k) {
    Module md{};
    md.kind = 1; md.section = 1; md.cin = pre; md.cout = pre; md.attn = (k == 0);
    add_res(m, "mid." + std::to_string(k), md);
    m->mods.push_back(md);
  }
  idx = 0;
  for (int ind = cfg->n_mults - 1; ind >= 0; --ind) {
    const bool last = ind < 1;
    const bool use_attn = in_attn(now_res);
    const int ch = inner * cfg->channel_mults[ind];
    for (int r = 0; r < cfg->res_blocks + 1; ++r) {
      Module md{};
      md.kind = 1; md.section = 2; md.skip = feat.back(); feat.pop_back();
      md.cin = pre + md.skip; md.cout = ch; md.attn = use_attn;
      if (md.cin % cfg->norm_groups) {
        delete m;
        return fail(DSX_ERR_INVALID, "norm_groups must divide concatenated channel counts");
      }
      add_res(m, "ups." + std::to_string(idx++), md);
      m->mods.push_back(md);
      pre = ch;
    }
    if (!last) {
      Module md{};
      md.kind = 3; md.section = 2; md.cin = pre; md.cout = pre;
      md.conv = add_conv(m, "ups." + std::to_string(idx++) + ".conv", pre, pre, 3, true);
      m->mods.push_back(md);
      now_res *= 2;
    }
  }
  {
    Module md{};
    md.kind = 4; md.section = 3; md.cin = pre;
    md.cout = cfg->out_channel > 0 ? cfg->out_channel : cfg->in_channel;
    md.gn1 = add_gn(m, "final_conv.block.0", pre);
    md.conv = add_conv(m, "final_conv.block.3", pre, md.cout, 3, true);
    m->mods.push_back(md);
  }
  const char* env = getenv("DSX_CONV_IMPL");
  m->want_naive = env && !strcmp(env, "naive");
  *out = m;
  return DSX_OK;
}

extern "C" void dsx_model_destroy(dsx_model* m) {
  if (!m) return;
  if (m->arena) (void)hipFree(m->arena);
  delete m;
}
extern "C" int dsx_model_num_params(const dsx_model* m) { return m ? (int)m->params.size() : 0; }
extern "C" int dsx_model_param_info(const dsx_model* m, int i, char* name, int cap, int* ndim,
                                    int64_t shape[4]) {
  if (!m || i < 0 || i >= (int)m->params.size()) return fail(DSX_ERR_INVALID, "bad param index");
  const Param& p = m->params[i];
  if (name && cap > 0) snprintf(name, cap, "%s", p.name.c_str());
  if (ndim) *ndim = (int)p.shape.size();
  if (shape)
    for (size_t k = 0; k < 4; ++k) shape[k] = k < p.shape.size() ? p.shape[k] : 1;
  return DSX_OK;
}
extern "C" int dsx_model_set_param(dsx_model* m, int i, const float* data, int64_t numel) {
  if (!m || !data || i < 0 || i >= (int)m->params.size()) return fail(DSX_ERR_INVALID, "bad argument");
  Param& p = m->params[i];
  if (numel != p.numel())
    return fail(DSX_ERR_INVALID, "param %s: got %lld elements, expected %lld", p.name.c_str(),
                (long long)numel, (long long)p.numel());
  p.host.assign(data, data + numel);
  p.set = true;
  m->finalized = false;
  return DSX_OK;
}
extern "C" int dsx_model_set_posenc_freq(dsx_model* m, const float* f, int count) {
  if (!m || !f || count != m->cfg.inner_channel / 2) return fail(DSX_ERR_INVALID, "bad freq table");
  m->freq.assign(f, f + count);
  m->freq_set = true;
  m->finalized = false;
  return DSX_OK;
}

// fp32 -> bf16 round-to-nearest-even (finite inputs)
static inline uint16_t f2bf(float f) {
  uint32_t u;
  memcpy(&u, &f, 4);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);  // NaN stays NaN
  u += 0x7fffu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}

// packed geometry of one conv: 64-byte input-channel chunks (padded to the staging group) and 32-channel N blocks
static void conv_geometry(int cout, int cin, int ks, int dtype, int& kchunks, int& nblocks) {
  const int KC = dtype != DSX_DTYPE_F32 ? 32 : 16;
  const int mult = conv_chunk_multiple(ks);
  kchunks = ((cin + KC - 1) / KC + mult - 1) / mult * mult;
  nblocks = (cout + 31) / 32;
}

// fp32 -> fp16 round-to-nearest-even
static inline uint16_t f2h(float f) {
  const _Float16 h = (_Float16)f;
  uint16_t u;
  memcpy(&u, &h, 2);
  return u;
}

// OIHW fp32 -> [nblk][kchunk][tap][half][lane][16 B] (see dsx_conv.hip header)
static void pack_conv(const float* w, int cout, int cin, int ks, int dtype, std::vector<char>& dst,
                      int& kchunks, int& nblocks) {
  const int KC = dtype != DSX_DTYPE_F32 ? 32 : 16, EPL = dtype != DSX_DTYPE_F32 ? 8 : 4, taps = ks * ks;
  conv_geometry(cout, cin, ks, dtype, kchunks, nblocks);
  dst.assign((size_t)nblocks * kchunks * taps * 2 * 64 * 16, 0);
  for (int nb = 0; nb < nblocks; ++nb)
    for (int kc = 0; kc < kchunks; ++kc)
      for (int tap = 0; tap < taps; ++tap)
        for (int fs = 0; fs < 2; ++fs)
          for (int lane = 0; lane < 64; ++lane) {
            // MFMA A row i = q + 8*j + 4*hh carries output channel 16*hh + 4*j + q of the block, so that a
            // lane's 16 accumulator registers are 16 consecutive channels (dsx_conv.hip, store16)
            const int i = lane & 31, h = lane >> 5;
            const int n = nb * 32 + 16 * ((i >> 2) & 1) + 4 * (i >> 3) + (i & 3);
            char* p = dst.data() + ((((size_t)(nb * kchunks + kc) * taps + tap) * 2 + fs) * 64 + lane) * 16;
            for (int j = 0; j < EPL; ++j) {
              const int c = kc * KC + (KC / 2) * fs + EPL * h + j;
              float v = 0.f;
              if (n < cout && c < cin) v = w[((size_t)n * cin + c) * taps + tap];
              if (dtype == DSX_DTYPE_BF16) { uint16_t b = f2bf(v); memcpy(p + 2 * j, &b, 2); }
              else if (dtype == DSX_DTYPE_F16) { uint16_t b = f2h(v); memcpy(p + 2 * j, &b, 2); }
              else memcpy(p + 4 * j, &v, 4);
            }
          }
}

// OIHW fp32 3x3 weights with cin <= 7 -> k_conv_first's operand: K = 9 cin as one dimension, k = tap * cin + c.
// 16-bit: [nblk][4 k-steps][lane][8 elements], element j of lane (i, h) = W[n(i)][16 s + 8 h + j];
// fp32: [nblk][32 k-steps][lane] floats, lane (i, h) = W[n(i)][2 s + h].  Rows permuted like pack_conv.
static void pack_first(const float* w, int cout, int cin, int dtype, std::vector<char>& dst) {
  const int K = 9 * cin, nblocks = (cout + 31) / 32;
  auto wk = [&](int n, int k) -> float {
    if (n >= cout || k >= K) return 0.f;
    const int tap = k / cin, c = k % cin;
    return w[((size_t)n * cin + c) * 9 + tap];
  };
  auto row = [](int nb, int i) { return nb * 32 + 16 * ((i >> 2) & 1) + 4 * (i >> 3) + (i & 3); };
  if (dtype == DSX_DTYPE_F32) {
    dst.assign((size_t)nblocks * 32 * 64 * 4, 0);
    for (int nb = 0; nb < nblocks; ++nb)
      for (int s = 0; s < 32; ++s)
        for (int lane = 0; lane < 64; ++lane) {
          const float v = wk(row(nb, lane & 31), 2 * s + (lane >> 5));
          memcpy(dst.data() + (((size_t)nb * 32 + s) * 64 + lane) * 4, &v, 4);
        }
  } else {
    dst.assign((size_t)nblocks * 4 * 64 * 16, 0);
    for (int nb = 0; nb < nblocks; ++nb)
      for (int s = 0; s < 4; ++s)
        for (int lane = 0; lane < 64; ++lane)
          for (int j = 0; j < 8; ++j) {
            const float v = wk(row(nb, lane & 31), 16 * s + 8 * (lane >> 5) + j);
            const uint16_t hbits = dtype == DSX_DTYPE_BF16 ? f2bf(v) : f2h(v);
            memcpy(dst.data() + ((((size_t)nb * 4 + s) * 64 + lane) * 8 + j) * 2, &hbits, 2);
          }
  }
}

namespace {
// One host image of everything the model keeps on the device (fragment-ordered conv weights, biases, GroupNorm
// affine parameters, time-embedding MLP, stacked FiLM linears).  With `write` false only the layout is computed
// (offsets and the total size): dsx_model_finalize_packed uploads a cached image into exactly this layout.
struct DevImage {
  bool write;
  std::vector<char> buf;
  size_t size = 0;
  struct Fix { void** dst; size_t off; };
  std::vector<Fix> fix;
  size_t reserve(size_t bytes) {
    const size_t off = (size + 255) & ~(size_t)255;
    size = off + bytes;
    if (write) buf.resize(size);
    return off;
  }
  void put(const void* src, size_t bytes, void** dst) {
    const size_t off = reserve(bytes);
    if (write) memcpy(buf.data() + off, src, bytes);
    fix.push_back({dst, off});
  }
};
}  // namespace

// layout (and, with img.write, contents) of the device image for `dtype`; the conv geometry is stored in the model
static int build_image(dsx_model* m, int dtype, DevImage& img) {
  const int inner = m->cfg.inner_channel;
  if (img.write) {
    for (int i = 0; i < (int)m->params.size(); ++i) {
      if (m->params[i].set || i == m->p_invfreq) continue;  // inv_freq is derived below if absent
      return fail(DSX_ERR_MISSING, "parameter %s was never set", m->params[i].name.c_str());
    }
    if (m->cfg.with_time_emb) {
      if (m->cfg.flavour == DSX_FLAVOUR_DDPM) {
        Param& p = m->params[m->p_invfreq];
        if (p.set) m->freq = p.host;
        else {  // ddpm unet.py:22-26
          m->freq.resize(inner / 2);
          for (int k = 0; k < inner / 2; ++k) m->freq[k] = expf((float)(2 * k) * (float)(-log(10000.0) / inner));
        }
      } else if (!m->freq_set) {  // sr3 unet.py:24-28
        m->freq.resize(inner / 2);
        for (int k = 0; k < inner / 2; ++k)
          m->freq[k] = expf((float)(-log(1e4)) * ((float)k / (float)(inner / 2)));
      }
    }
  }
  auto put_param = [&](int pi, float** dst) {   // one fp32 parameter as it is
    img.put(img.write ? m->params[pi].host.data() : nullptr, (size_t)m->params[pi].numel() * 4, (void**)dst);
  };
  auto put_conv = [&](ConvW& c) {
    if (c.pw < 0) return;
    conv_geometry(c.cout, c.cin, c.ks, dtype, c.kchunks, c.nblocks);
    std::vector<char> pk;
    size_t bytes = (size_t)c.nblocks * c.kchunks * c.ks * c.ks * 2 * 64 * 16;
    if (img.write) { pack_conv(m->params[c.pw].host.data(), c.cout, c.cin, c.ks, dtype, pk, c.kchunks, c.nblocks); bytes = pk.size(); }
    img.put(pk.data(), bytes, &c.pack);
    if (c.ks == 3 && c.cin <= 7) {
      std::vector<char> pf;
      size_t fb = (size_t)((c.cout + 31) / 32) * (dtype == DSX_DTYPE_F32 ? 32 * 64 * 4 : 4 * 64 * 16);
      if (img.write) { pack_first(m->params[c.pw].host.data(), c.cout, c.cin, dtype, pf); fb = pf.size(); }
      img.put(pf.data(), fb, &c.pack_first);
    }
    if (c.pb >= 0) put_param(c.pb, &c.bias);
    if (m->want_naive) {
      std::vector<float> nv;
      if (img.write) {
        nv.resize((size_t)c.cout * c.ks * c.ks * c.cin);
        const float* w = m->params[c.pw].host.data();
        for (int n = 0; n < c.cout; ++n)
          for (int ci = 0; ci < c.cin; ++ci)
            for (int t = 0; t < c.ks * c.ks; ++t)
              nv[((size_t)n * c.ks * c.ks + t) * c.cin + ci] = w[((size_t)n * c.cin + ci) * c.ks * c.ks + t];
      }
      img.put(nv.data(), (size_t)c.cout * c.ks * c.ks * c.cin * 4, (void**)&c.naive);
    }
  };
  auto put_gn = [&](GnW& g) {
    if (g.pg < 0) return;
    put_param(g.pg, &g.gamma);
    put_param(g.pb, &g.beta);
  };
  std::vector<float> wf, bf;
  if (img.write) { wf.resize((size_t)m->F * inner); bf.resize(m->F); }
  for (auto& md : m->mods) {
    put_conv(md.conv);
    put_gn(md.gn1); put_gn(md.gn2); put_gn(md.gna);
    put_conv(md.conv1); put_conv(md.conv2);
    if (md.has_res) put_conv(md.res);
    if (md.attn) { put_conv(md.qkv); put_conv(md.out); }
    if (md.film_off >= 0 && img.write) {
      memcpy(wf.data() + (size_t)md.film_off * inner, m->params[md.film.pw].host.data(),
             (size_t)md.cout * inner * 4);
      memcpy(bf.data() + md.film_off, m->params[md.film.pb].host.data(), (size_t)md.cout * 4);
      // the FiLM vector is only ever added to conv1's output: carry conv1's bias in it (one per-channel addend
      // in the conv epilogue instead of two; plan_res passes no bias for that conv)
      if (md.conv1.pb >= 0)
        for (int n = 0; n < md.cout; ++n) bf[md.film_off + n] += m->params[md.conv1.pb].host[n];
    }
  }
  if (m->cfg.with_time_emb) {
    img.put(m->freq.data(), (size_t)(inner / 2) * 4, (void**)&m->d_freq);
    put_param(m->t1.pw, &m->d_w1);
    put_param(m->t1.pb, &m->d_b1);
    put_param(m->t2.pw, &m->d_w2);
    put_param(m->t2.pb, &m->d_b2);
    img.put(wf.data(), (size_t)m->F * inner * 4, (void**)&m->d_wf);
    img.put(bf.data(), (size_t)m->F * 4, (void**)&m->d_bf);
  }
  return DSX_OK;
}

static int upload_image(dsx_model* m, int dtype, const DevImage& img, const void* bytes) {
  if (m->arena) { (void)hipFree(m->arena); m->arena = nullptr; }
  HIP_TRY(hipMalloc((void**)&m->arena, img.size));
  HIP_TRY(hipMemcpy(m->arena, bytes, img.size, hipMemcpyHostToDevice));
  m->arena_bytes = img.size;
  for (auto& f : img.fix) *f.dst = m->arena + f.off;
  m->dtype = dtype;
  m->finalized = true;
  return DSX_OK;
}

static bool dtype_ok(int dtype) { return dtype == DSX_DTYPE_F32 || dtype == DSX_DTYPE_BF16 || dtype == DSX_DTYPE_F16; }

extern "C" int dsx_model_finalize(dsx_model* m, int dtype) {
  if (!m) return fail(DSX_ERR_INVALID, "null model");
  if (!dtype_ok(dtype)) return fail(DSX_ERR_INVALID, "bad dtype");
  DevImage img;
  img.write = true;
  int rc = build_image(m, dtype, img);
  if (rc) return rc;
  return upload_image(m, dtype, img, img.buf.data());
}

// ---- packed-weight cache (the one-time repack of a *_gen.pth, model/model.py:153-166): export the device image of a
// finalized model, and finalize a fresh model straight from such an image (no parameters set, no repacking)
extern "C" int dsx_model_packed_bytes(dsx_model* m, int dtype, size_t* bytes) {
  if (!m || !bytes || !dtype_ok(dtype)) return fail(DSX_ERR_INVALID, "bad argument");
  DevImage img;
  img.write = false;
  int rc = build_image(m, dtype, img);
  if (rc) return rc;
  *bytes = img.size;
  return DSX_OK;
}
extern "C" int dsx_model_export_packed(const dsx_model* m, void* host_buf, size_t capacity) {
  if (!m || !host_buf) return fail(DSX_ERR_INVALID, "null argument");
  if (!m->finalized) return fail(DSX_ERR_STATE, "finalize the model before exporting its packed image");
  if (capacity < m->arena_bytes) return fail(DSX_ERR_INVALID, "buffer of %zu bytes < %zu", capacity, m->arena_bytes);
  HIP_TRY(hipMemcpy(host_buf, m->arena, m->arena_bytes, hipMemcpyDeviceToHost));
  return DSX_OK;
}
extern "C" int dsx_model_finalize_packed(dsx_model* m, int dtype, const void* host_img, size_t bytes) {
  if (!m || !host_img || !dtype_ok(dtype)) return fail(DSX_ERR_INVALID, "bad argument");
  DevImage img;
  img.write = false;
  int rc = build_image(m, dtype, img);
  if (rc) return rc;
  if (img.size != bytes)
    return fail(DSX_ERR_INVALID, "packed image of %zu bytes does not fit this model / dtype (%zu expected)", bytes, img.size);
  return upload_image(m, dtype, img, host_img);
}

extern "C" double dsx_model_flops(const dsx_model* m, int H, int W) {
  if (!m) return 0;
  double fl = 0;
  int h = H, w = W;
  const int inner = m->cfg.inner_channel;
  auto conv = [&](const ConvW& c, int hh, int ww) {
    if (c.pw >= 0) fl += 2.0 * hh * ww * (double)c.cout * c.cin * c.ks * c.ks;
  };
  if (m->cfg.with_time_emb) fl += 2.0 * (inner * 4.0 * inner) * 2;
  for (auto& md : m->mods) {
    if (md.kind == 0) conv(md.conv, h, w);
    else if (md.kind == 2) { h /= 2; w /= 2; conv(md.conv, h, w); }
    else if (md.kind == 3) { h *= 2; w *= 2; conv(md.conv, h, w); }
    else if (md.kind == 4) conv(md.conv, h, w);
    else {
      conv(md.conv1, h, w); conv(md.conv2, h, w);
      if (md.has_res) conv(md.res, h, w);
      if (md.film_off >= 0) fl += 2.0 * inner * md.cout;
      if (md.attn) {
        conv(md.qkv, h, w); conv(md.out, h, w);
        const double L = (double)h * w;
        fl += 2.0 * 2.0 * L * L * md.cout;
      }
    }
  }
  return fl;
}

// ------------------------------------------------------------------ executor
namespace {

struct Tensor {
  void* p = nullptr;       // NHWC in the storage type `st`
  int C = 0, H = 0, W = 0;
  int id = -1;             // index into dsx_exec::stats (copies of a Tensor share it)
  int st = 0;              // storage kind: 0 fp32, 1 bf16, 2 fp16 (DSX_DTYPE_*)
  int esz() const { return st ? 2 : 4; }
  char* at(size_t elem) const { return (char*)p + elem * esz(); }
};
struct StatInfo {          // GroupNorm partial sums of one tensor, produced at most once
  void* part = nullptr;    // double [B][nchunk][C][2] (k_chan_stats) or float (fused into the conv epilogue)
  int nchunk = 0;
  bool planned = false;
  bool f32 = false;
};

struct OpInfo {            // what one launch of the plan computes (for profiling / roofline)
  int kind;                // DSX_OP_*
  std::string desc;
  double flops;            // algorithmic 2*MAC
  double bytes;            // algorithmic HBM bytes: inputs + outputs + weights, each once
};

}  // namespace

struct dsx_exec {
  dsx_model* m = nullptr;
  int B = 0, H = 0, W = 0, cond_c = 0, x_c = 0;
  char* ws = nullptr;
  size_t ws_bytes = 0, ws_used = 0;
  bool sizing = true;
  std::vector<std::function<hipError_t(hipStream_t)>> ops;  // the UNet forward
  std::vector<OpInfo> op_info;                               // parallel to ops
  int conv_ordinal = 0;
  bool overflow = false;
  // L2 weight prefetch (l2_prefetch in dsx_kernels.h): the k_gn_finalize launch planned for the conv being planned,
  // and the previous image-resident conv, receive the weight slices of that conv (filled in once its kernel and
  // tiling are known)
  std::shared_ptr<PrefetchArgs> pending_gn_pf;
  std::shared_ptr<PrefetchArgs> prev_img_pf;
  std::shared_ptr<PrefetchArgs> prev_ws_pf;    // slot of the previous k_conv_ws launch: the next conv's weight slices
  // GroupNorm finalize hosted by the residual 1 x 1 conv in front of it (k_conv_ws loader waves): armed by plan_conv for
  // the conv plan_res marks, consumed by the plan_gn that follows it immediately.  Shapes only: both planner passes agree.
  struct HostedFin { bool on = false; GnFinArgs a{}; std::shared_ptr<PrefetchArgs> pf; };
  std::shared_ptr<HostedFin> fin_host;
  bool fin_host_armed = false;
  unsigned long long* stamp_buf = nullptr;
  std::vector<StatInfo> stats;
  // fixed buffers
  Tensor in_cond, in_x, out;   // NHWC
  float* x_state = nullptr;    // sampler state, NHWC fp32 (== in_x.p unless activations are stored in bf16)
  float* film = nullptr;       // [B][F]
  float* time_buf = nullptr;   // [B] direct time values
  int* step_ctr = nullptr;
  float* table = nullptr;      // [6][cap]
  int table_cap = 0;
  bool temb_from_table = false;
  unsigned* handoff_timeouts = nullptr;        // device counter: bounded FULL / FREE spins of k_conv_ws that gave up (0 in a correct run)
  unsigned long long* loop_params = nullptr;   // device {seed, noise address}: per-call values the captured step reads
  // per-call host data (step table, seed, noise address) is staged in pinned memory, one slot per call in flight: a
  // slot is reused only after the event recorded behind its copies has completed, so a second dsx_sample_loop on the
  // same executor never overwrites bytes an earlier call's asynchronous copy has yet to read
  struct Staging { float* host = nullptr; size_t floats = 0; hipEvent_t ev = nullptr; bool busy = false; };
  Staging staging[4];
  int staging_next = 0;
  hipStream_t last_stream = nullptr;           // stream of the most recent graph launches
  dsx_step_table cur_tab{};
  // graph
  hipGraphExec_t graph_exec = nullptr;
  hipGraph_t graph = nullptr;
  std::vector<float> graph_sig;
  // time predictor head
  float* tp_w = nullptr; float* tp_b = nullptr; float* tp_mask = nullptr;
  int launches = 0;
};

static void add_op(dsx_exec* ex, int kind, const std::string& desc, double flops, double bytes,
                   std::function<hipError_t(hipStream_t)> fn) {
  ex->ops.push_back(std::move(fn));
  ex->op_info.push_back(OpInfo{kind, desc, flops, bytes});
}
static std::string fmt(const char* f, ...) {
  char buf[256];
  va_list ap;
  va_start(ap, f);
  vsnprintf(buf, sizeof buf, f, ap);
  va_end(ap);
  return buf;
}

static char* ws_alloc(dsx_exec* ex, size_t bytes) {
  size_t off = (ex->ws_used + 255) & ~(size_t)255;
  ex->ws_used = off + bytes;
  if (ex->sizing) return nullptr;
  if (ex->ws_used > ex->ws_bytes) ex->overflow = true;   // sizing and planning passes diverged: reported by build_plan
  return ex->ws + off;
}
// activations are stored in the MFMA operand type (bf16 build: bf16); `f32` forces fp32 (network output)
static Tensor new_tensor(dsx_exec* ex, int C, int H, int W, bool f32 = false) {
  Tensor t;
  t.C = C; t.H = H; t.W = W;
  t.st = f32 ? 0 : ex->m->dtype;
  t.p = ws_alloc(ex, (size_t)ex->B * H * W * C * t.esz());
  t.id = (int)ex->stats.size();
  ex->stats.push_back(StatInfo());
  return t;
}

static int ilog2(int v) { int l = 0; while ((1 << (l + 1)) <= v) ++l; return l; }
static int pow2_divisor(int v, int cap) {  // largest power of two dividing v, <= cap
  int p = 1;
  while (p * 2 <= cap && v % (p * 2) == 0) p *= 2;
  return p;
}

// tile preference lists, overridable for tuning: DSX_TILES_WIDE / DSX_TILES_NARROW = "0,2,3,4"
static std::vector<int> tile_order(const char* env, std::initializer_list<int> dflt) {
  std::vector<int> v(dflt);
  const char* e = getenv(env);
  if (e && *e) {
    v.clear();
    for (const char* p = e; *p;) {
      v.push_back(atoi(p));
      while (*p && *p != ',') ++p;
      if (*p == ',') ++p;
    }
  }
  return v;
}

// geometry of `tile` for this conv (unsplit); false if the tile cannot be used
static bool tile_geometry(int dtype, int tile, int ks, int stride, const ConvArgs& a, ConvArgs& c) {
  if (tile < 0 || tile >= TILE_COUNT) return false;
  if (stride == 2 && tile != TILE_64x64) return false;
  const ConvTileInfo ti = conv_tile_info(tile);
  c = a;
  const int TW = pow2_divisor(a.Wo, 16);
  const int TH = pow2_divisor(a.Ho, std::max(1, ti.BM / TW));
  const int TB = ti.BM / (TW * TH);
  if (TB < 1 || TW * TH * TB != ti.BM) return false;
  c.tw_log2 = ilog2(TW); c.th_log2 = ilog2(TH); c.tb_log2 = ilog2(TB);
  c.tiles_x = a.Wo / TW; c.tiles_y = a.Ho / TH;
  c.m_tiles = c.tiles_x * c.tiles_y * ((a.B + TB - 1) / TB);
  c.n_tiles = (c.nblocks * 32 + ti.BN - 1) / ti.BN;
  c.ksplit = 1; c.groups_per_split = a.kchunks / conv_chunk_multiple(ks); c.slab_stride = 0;
  c.lds_row = conv_lds_row(ks, stride, c.tw_log2);
#ifdef DSX_DIAG
  c.ablate = getenv("DSX_ABLATE") ? atoi(getenv("DSX_ABLATE")) : 0;   // diagnostic build only
#else
  c.ablate = 0;
#endif
  return conv_lds_bytes(dtype, tile, ks, stride, c) != 0;
}

// choose tile + geometry (+ split-K) for one conv; returns false if no MFMA config fits.
// Pass 1: the first tile of the preference list whose plain grid fills the chip.
// Pass 2: small-M layers — the first tile of the split list, K split across workgroups
//         (slabs + a reduce launch) until the grid fills the chip.
static bool pick_conv(int dtype, int ks, int stride, ConvArgs& a, int& tile_out) {
  static const std::vector<int> wide = tile_order("DSX_TILES_WIDE", {TILE_128x128, TILE_64x128, TILE_64x64});
  static const std::vector<int> narrow = tile_order("DSX_TILES_NARROW", {TILE_64x64, TILE_128x64});
  static const std::vector<int> slim = tile_order("DSX_TILES_SLIM", {TILE_128x32, TILE_64x64, TILE_128x64});   // Cout <= 32
  static const std::vector<int> wide2 = tile_order("DSX_TILES_WIDE_SPLIT", {TILE_64x128, TILE_128x128, TILE_64x64});
  static const std::vector<int> narrow2 = tile_order("DSX_TILES_NARROW_SPLIT", {TILE_64x64, TILE_128x64});
  static const int min_grid = getenv("DSX_MIN_GRID") ? atoi(getenv("DSX_MIN_GRID")) : 512;
  static const int splitk_on = getenv("DSX_SPLITK") ? atoi(getenv("DSX_SPLITK")) : 1;
  const bool is_wide = a.Cout > 64;
  const int kgroups = a.kchunks / conv_chunk_multiple(ks);
  ConvArgs c;
  // Pass 0: the warp-specialised persistent kernel only needs ~one workgroup per CU: take the widest
  // tile (least re-staging of the activations per output channel) that still gives >= ws_min work items.
  static const int ws_on = getenv("DSX_WS") ? atoi(getenv("DSX_WS")) : 1;
  // (read at every plan, not cached: the parity tests lower it to force the persistent kernel onto small grids)
  const int ws_min = getenv("DSX_WS_MIN_GRID") ? atoi(getenv("DSX_WS_MIN_GRID")) : 224;
  // the 128 x 128 tile (2 x 2 waves of 64 pixels x 64 channels: each LDS pixel fragment feeds two MFMAs and each
  // converted group twice the MFMA work of the 64 x 128 tile) wherever it still fills the chip
  static const std::vector<int> ws_wide3 = tile_order("DSX_TILES_WS_WIDE", {TILE_128x128, TILE_64x128});
  static const std::vector<int> ws_wide1 = tile_order("DSX_TILES_WS_WIDE_1X1", {TILE_128x128, TILE_64x128});
  // 1 x 1 without GroupNorm / Swish in front (residual and attention-output convs): the loaders only copy, the kernel is
  // bound by the weight stream, and only the one-N-block tiles have the deep weight ring (measured: 32^2 layers -2 us each)
  static const std::vector<int> ws_wide1raw = tile_order("DSX_TILES_WS_WIDE_1X1_RAW", {TILE_64x128, TILE_128x128});
  const std::vector<int>& ws_wide = ks == 1 ? ((a.has_gn || a.swish) ? ws_wide1 : ws_wide1raw) : ws_wide3;
  static const std::vector<int> ws_narrow = tile_order("DSX_TILES_WS_NARROW", {TILE_256x64, TILE_128x64, TILE_64x64});
  // Pass -1 (experiment, off: DSX_CPG2=1 enables it): few input channels at a large map (the 64-channel layers of the
  // 128^2 level) on the two-chunk variant of k_conv_mfma: every 64 input channels are one staged group, three
  // workgroups per CU overlap their load / MFMA / epilogue phases.  Measured slower than the persistent kernel
  // (64->64 @128^2: 60 us vs 42 us), see DESIGN.md.
  static const int g2_on = getenv("DSX_CPG2") ? atoi(getenv("DSX_CPG2")) : 0;
  static const int g2_max_c = getenv("DSX_CPG2_MAXC") ? atoi(getenv("DSX_CPG2_MAXC")) : 64;
  if (g2_on && ks == 3 && stride == 1 && a.Cout == 64 && a.stage_mode == 0 && a.kchunks % 2 == 0 &&
      a.C0 % 64 == 0 && a.C1 % 64 == 0 && a.C0 + a.C1 <= g2_max_c) {
    ConvArgs g = a;
    const int tile = TILE_128x64;
    const ConvTileInfo ti = conv_tile_info(tile);
    const int TW = pow2_divisor(a.Wo, 16), TH = pow2_divisor(a.Ho, std::max(1, ti.BM / TW));
    if (TW * TH == ti.BM) {
      g.cpg = 2;
      g.tw_log2 = ilog2(TW); g.th_log2 = ilog2(TH); g.tb_log2 = 0;
      g.tiles_x = a.Wo / TW; g.tiles_y = a.Ho / TH;
      g.m_tiles = g.tiles_x * g.tiles_y * a.B;
      g.n_tiles = 1;
      g.ksplit = 1; g.groups_per_split = a.kchunks / 2; g.slab_stride = 0;
      g.lds_row = conv_lds_row_g2(g.tw_log2);
      g.ablate = 0;
      if (conv_g2_lds_bytes(tile, g) != 0 && g.m_tiles >= 512) { a = g; tile_out = tile; return true; }
    }
  }
  // Narrow outputs (<= 32 channels: the UNet's final conv, 64 -> 3 at full resolution): HBM-bound layers that the
  // generic 128 x 32 tile walked as two 32-channel groups with a barrier pair each and a 16 x 8 pixel halo.  The
  // two-chunk variant stages ALL input channels of a 16 x 16 (or 16 x 8) pixel patch once -- one load phase, one
  // conversion, one barrier, 36 MFMA steps per row block -- with three workgroups per CU overlapping their phases.
  static const int narrow_on = getenv("DSX_NARROW_G2") ? atoi(getenv("DSX_NARROW_G2")) : 1;
  {
    const int gw2 = 2 * (dtype != DSX_DTYPE_F32 ? 32 : 16);          // channels per two-chunk group
    if (narrow_on && ks == 3 && stride == 1 && a.Cout <= 32 && a.stage_mode == 0 && a.kchunks % 2 == 0 &&
        a.C0 % gw2 == 0 && a.C1 % gw2 == 0 && !a.up) {
      static const std::vector<int> narrow_tiles = tile_order("DSX_TILES_NARROW_G2", {TILE_256x32, TILE_128x32});
      for (int tile : narrow_tiles) {
        ConvArgs g = a;
        const ConvTileInfo ti = conv_tile_info(tile);
        const int TW = pow2_divisor(a.Wo, 16), TH = pow2_divisor(a.Ho, std::max(1, ti.BM / TW));
        if (TW * TH != ti.BM) continue;
        g.cpg = 2;
        g.tw_log2 = ilog2(TW); g.th_log2 = ilog2(TH); g.tb_log2 = 0;
        g.tiles_x = a.Wo / TW; g.tiles_y = a.Ho / TH;
        g.m_tiles = g.tiles_x * g.tiles_y * a.B;
        g.n_tiles = 1;
        g.ksplit = 1; g.groups_per_split = a.kchunks / 2; g.slab_stride = 0;
        g.lds_row = conv_lds_row_g2(g.tw_log2);
        g.ablate = 0;
        if (conv_g2_lds_bytes(tile, g) != 0) { a = g; tile_out = tile; return true; }
      }
    }
  }
  static const int ws_1x1 = getenv("DSX_WS_1X1") ? atoi(getenv("DSX_WS_1X1")) : 1;
  if (ws_on && (ks != 1 || ws_1x1) && stride == 1 && a.stage_mode == 0) {
    for (int tile : (is_wide ? ws_wide : ws_narrow)) {
      if (!tile_geometry(dtype, tile, ks, stride, a, c)) continue;
      if (conv_ws_lds_bytes(dtype, tile, ks, c) == 0) continue;
      if ((long long)c.m_tiles * c.n_tiles >= ws_min) { a = c; tile_out = tile; return true; }
    }
  }
  for (int tile : (is_wide ? wide : (a.Cout <= 32 ? slim : narrow))) {
    if (!tile_geometry(dtype, tile, ks, stride, a, c)) continue;
    if ((long long)c.m_tiles * c.n_tiles >= min_grid) { a = c; tile_out = tile; return true; }
  }
  int best = -1;
  long long best_eff = -1;
  ConvArgs best_a = a;
  for (int tile : (is_wide ? wide2 : (a.Cout <= 32 ? slim : narrow2))) {
    if (!tile_geometry(dtype, tile, ks, stride, a, c)) continue;
    const long long grid = (long long)c.m_tiles * c.n_tiles;
    long long eff = grid;
    if (splitk_on && grid < min_grid && kgroups >= 4 && (a.Cout % 16) == 0 && a.out_ld == a.Cout) {
      const int want = (int)((min_grid + grid - 1) / grid);
      int S = std::min(want, kgroups / 2);  // at least two channel groups per slice
      const int gps = (kgroups + S - 1) / S;
      S = (kgroups + gps - 1) / gps;
      if (S > 1) {
        c.ksplit = S; c.groups_per_split = gps;
        c.slab_stride = (long long)a.B * a.Ho * a.Wo * a.Cout;
        eff = grid * S;
      }
    }
    if (eff > best_eff) { best = tile; best_a = c; best_eff = eff; }
    if (eff >= min_grid) break;
  }
  if (best < 0) return false;
  a = best_a;
  tile_out = best;
  return true;
}

struct ConvSpec {
  const ConvW* w;
  Tensor x0, x1;       // x1.p == nullptr / C == 0: single source
  bool up = false;
  int stride = 1;
  const GnW* gn = nullptr;   // GroupNorm over cat(x0, x1) in front of the conv (finalised by k_gn_finalize, or inside
                             // the consumer by k_conv_img)
  bool has_resid = false;
  bool host_fin = false;     // plan_res: this residual 1 x 1 conv may host the finalize of the block's second GroupNorm
  bool swish = false;
  const float* film = nullptr; int film_bs = 0;
  const void* resid = nullptr; int resid_ld = 0;
  Tensor out;
  bool want_stats = false;   // a GroupNorm will read `out`: produce its statistics in the epilogue
  bool bias_in_film = false; // the conv bias is already part of the FiLM vector (dsx_model_finalize)
};

static void plan_stats(dsx_exec* ex, const Tensor& t);
static void plan_gn(dsx_exec* ex, const GnW& g, const Tensor& t0, const Tensor* t1, float** scale, float** shift);

// workgroups per N tile of a k_conv_ws launch (whole XCD groups per N tile, see the kernel)
static int ws_wg_per_n(const ConvArgs& a) {
  int wpn = std::min(a.m_tiles, std::max(1, 256 / std::max(1, a.n_tiles)));
  if (a.n_tiles <= 8 && 8 % a.n_tiles == 0) {
    const int unit = 8 / a.n_tiles;
    wpn = std::max(unit, wpn / unit * unit);
    if (wpn > a.m_tiles) wpn = (a.m_tiles + unit - 1) / unit * unit;
  }
  return wpn;
}

static int plan_conv(dsx_exec* ex, const ConvSpec& s) {
  ex->pending_gn_pf.reset();
  ConvArgs a{};
  a.src0 = s.x0.p; a.C0 = s.x0.C;
  a.src1 = s.x1.C ? s.x1.p : nullptr; a.C1 = s.x1.C;
  a.B = ex->B; a.Hs = s.x0.H; a.Ws = s.x0.W; a.up = s.up ? 1 : 0;
  a.Ho = s.out.H; a.Wo = s.out.W;
  a.swish = s.swish ? 1 : 0;
  a.has_gn = s.gn ? 1 : 0;
  a.act_bf16 = ex->m->dtype;   // storage kind of the sources / residual
  a.out_bf16 = s.out.st;
  {
    const int gw = conv_chunk_multiple(s.w->ks) * (ex->m->dtype != 0 ? 32 : 16);  // channels per staged group
    const int um = ex->m->dtype != 0 ? 7 : 3;                                       // channels per 16-byte unit - 1
    a.stage_mode = ((a.C0 & um) || (a.C1 & um)) ? 2 : ((a.C1 == 0 || a.C0 % gw == 0) ? 0 : 1);
  }
  a.wpack = s.w->pack; a.bias = s.bias_in_film ? nullptr : s.w->bias;
  a.film = s.film; a.film_bs = s.film_bs;
  a.resid = s.resid; a.resid_ld = s.resid_ld;
  a.out = s.out.p; a.out_ld = s.out.C; a.Cout = s.w->cout;
  a.nblocks = s.w->nblocks; a.kchunks = s.w->kchunks;
  if (a.C0 + a.C1 != s.w->cin) return fail(DSX_ERR_INVALID, "conv channel mismatch");
  {
    // the conv kernels address their sources with 32-bit byte offsets (0x80000000 = forced out of bounds, the
    // zero padding): a source tensor of 2 GiB or more would silently read as zeros
    const long long esz_src = ex->m->dtype != DSX_DTYPE_F32 ? 2 : 4;
    const long long src_bytes = (long long)a.B * a.Hs * a.Ws * std::max(a.C0, a.C1) * esz_src;
    if (src_bytes >= (1LL << 31))
      return fail(DSX_ERR_INVALID,
                  "conv source of %lld bytes (B=%d, %dx%d, %d channels) exceeds the 2 GiB the kernels address; "
                  "use a smaller batch per executor", src_bytes, a.B, a.Hs, a.Ws, std::max(a.C0, a.C1));
  }
  const int dtype = ex->m->dtype, ks = s.w->ks, stride = s.stride;
  const double npix = (double)a.B * a.Ho * a.Wo;
  const double cin = a.C0 + a.C1;
  const double flops = 2.0 * npix * a.Cout * cin * ks * ks;
  const double wbytes = (double)a.Cout * cin * ks * ks * (dtype != 0 ? 2 : 4);
  const double esz = dtype != 0 ? 2.0 : 4.0;   // activation element size in HBM
  const double bytes = esz * ((double)a.B * a.Hs * a.Ws * cin + npix * a.Cout * (s.has_resid || s.resid ? 1 : 0)) +
                       (a.out_bf16 ? 2.0 : 4.0) * npix * a.Cout + wbytes;
  {  // diagnostics: DSX_STAMP_OP=<conv ordinal>[,<block>] -> in-kernel phase stamps of that launch
    static const char* se = getenv("DSX_STAMP_OP");
    if (se) {
      const int want = atoi(se);
      const char* comma = strchr(se, ',');
      if (ex->conv_ordinal == want) {
        a.stamp = (unsigned long long*)ws_alloc(ex, 128 * 8);
        a.stamp_block = comma ? atoi(comma + 1) : 0;
        ex->stamp_buf = a.stamp;
      }
    }
    ex->conv_ordinal++;
  }
  // ---- the UNet's first conv (few input channels): im2col-in-K kernel
  if (!ex->m->want_naive && !s.film && !s.resid && !s.has_resid && conv_first_applicable(ks, stride, a, s.gn != nullptr)) {
    ex->launches++;
    a.wpack = s.w->pack_first;
    if (s.want_stats) {
      StatInfo& si = ex->stats[s.out.id];
      si.nchunk = (a.Ho >> 4) * (a.Wo >> 4) * 4;
      si.part = ws_alloc(ex, (size_t)a.B * si.nchunk * a.Cout * 2 * sizeof(float));
      si.planned = true;
      si.f32 = true;
      a.stat_part = (float*)si.part;
    }
    if (ex->sizing) return DSX_OK;
    add_op(ex, DSX_OP_CONV_MFMA, fmt("conv3x3 %d->%d @%dx%d first", (int)cin, a.Cout, a.Ho, a.Wo), flops, bytes,
           [=](hipStream_t st) { return launch_conv_first(dtype, a, st); });
    return DSX_OK;
  }
  // ---- 8 x 8 maps: the image-resident kernel (GroupNorm finalised in its prologue, statistics in its epilogue)
  if (!ex->m->want_naive && conv_img_applicable(dtype, ks, stride, a, s.gn != nullptr, ex->m->cfg.norm_groups)) {
    ex->launches++;
    if (s.gn) {
      plan_stats(ex, s.x0);
      if (s.x1.C) plan_stats(ex, s.x1);
      const StatInfo& s0 = ex->stats[s.x0.id];
      a.gn_part0 = s0.part; a.gn_nchunk0 = s0.nchunk; a.gn_pf32_0 = s0.f32 ? 1 : 0;
      if (s.x1.C) {
        const StatInfo& s1 = ex->stats[s.x1.id];
        a.gn_part1 = s1.part; a.gn_nchunk1 = s1.nchunk; a.gn_pf32_1 = s1.f32 ? 1 : 0;
      }
      a.gn_gamma = s.gn->gamma; a.gn_beta = s.gn->beta; a.gn_groups = ex->m->cfg.norm_groups; a.gn_eps = 1e-5f;
    }
    if (s.want_stats) {
      StatInfo& si = ex->stats[s.out.id];
      si.nchunk = 1;
      si.part = ws_alloc(ex, (size_t)a.B * a.Cout * 2 * sizeof(float));
      si.planned = true;
      si.f32 = true;
      a.stat_part = (float*)si.part;
    }
    if (ex->sizing) return DSX_OK;
    static const int pf_on = getenv("DSX_PREFETCH") ? atoi(getenv("DSX_PREFETCH")) : 1;
    const PrefetchArgs mine{a.wpack, (unsigned)((size_t)a.kchunks * ks * ks * 2 * 1024), a.nblocks, nullptr};   // one slice per N block
    if (pf_on && ex->prev_img_pf) *ex->prev_img_pf = mine;      // the previous image-resident conv warms the L2s for this one
    auto pf = std::make_shared<PrefetchArgs>(PrefetchArgs{nullptr, 0u, 0, nullptr});
    ex->prev_img_pf = pf;
    add_op(ex, DSX_OP_CONV_MFMA, fmt("conv%dx%d %d->%d @%dx%d img", ks, ks, (int)cin, a.Cout, a.Ho, a.Wo), flops, bytes,
           [=](hipStream_t st) { ConvArgs b = a; b.pf = *pf; return launch_conv_img(dtype, ks, b, st); });
    return DSX_OK;
  }
  static const int fuse_stats = getenv("DSX_FUSE_STATS") ? atoi(getenv("DSX_FUSE_STATS")) : 1;
  static const int ws_enabled = getenv("DSX_WS") ? atoi(getenv("DSX_WS")) : 1;
  static const int ws_1x1_enabled = getenv("DSX_WS_1X1") ? atoi(getenv("DSX_WS_1X1")) : 1;
  int tile = -1;
  const bool mfma_ok = !ex->m->want_naive && pick_conv(dtype, ks, stride, a, tile);   // (keyed on a.has_gn, never on pointers)
  const bool use_ws = mfma_ok && a.cpg != 2 && ws_enabled && (ks != 1 || ws_1x1_enabled) && stride == 1 && a.stage_mode == 0 &&
                      a.ksplit == 1 && conv_ws_lds_bytes(dtype, tile, ks, a) != 0;
  if (use_ws && ks == 3) {
    // (these knobs are read at every plan: the tests flip them)
    // Two 64-byte chunks per (tile, group) item where the geometry allows it: under the 16 x 16 MFMA shape the loaders,
    // not the MFMAs, bound an item (their VALU stream gets 8 of every 16 issue cycles), and their per-item costs (DMA
    // issue, wait, fetch, barrier: ~1.3 k of 3.2 k cycles per 32 channels on the 64-pixel tile) are paid once per 64
    // channels this way.  Measured (A/B in one gpurun call): the 512-channel 16 x 16 layers -8 .. -11 %, a 256 -> 512 layer
    // with only 4 two-chunk groups +7 % (hence the 12-chunk minimum there); the 128 x 128 tile -2.6 % over its 22 launches.
    const int ws_g2 = getenv("DSX_WS_G2") ? atoi(getenv("DSX_WS_G2")) : 1;
    const int g2_min64 = getenv("DSX_WS_G2_MIN64") ? atoi(getenv("DSX_WS_G2_MIN64")) : 12;     // chunks: fewer -> one-chunk groups
    const int g2_min128 = getenv("DSX_WS_G2_MIN128") ? atoi(getenv("DSX_WS_G2_MIN128")) : 4;
    ConvArgs t2 = a;
    t2.ws_cpg = 2; t2.lds_row = conv_lds_row_g2(a.tw_log2);
    const int min_chunks = conv_tile_info(tile).BM == 64 ? g2_min64 : g2_min128;
    if (ws_g2 && a.kchunks >= min_chunks && conv_ws_lds_bytes(dtype, tile, ks, t2) != 0) { a.ws_cpg = 2; a.lds_row = t2.lds_row; }
    const int g4_min64 = getenv("DSX_WS_G4_MIN64") ? atoi(getenv("DSX_WS_G4_MIN64")) : 16;   // four chunks (64-pixel tile)
    ConvArgs t4 = a;
    t4.ws_cpg = 4; t4.lds_row = conv_lds_row_3x3_c(a.tw_log2, 4);
    if (ws_g2 && a.kchunks >= g4_min64 && conv_ws_lds_bytes(dtype, tile, ks, t4) != 0) { a.ws_cpg = 4; a.lds_row = t4.lds_row; }
  }
  if (use_ws && ks == 1) {   // the same for 1 x 1 convs: four chunks (128 input channels) per item
    const int ws_c4 = getenv("DSX_WS_C4") ? atoi(getenv("DSX_WS_C4")) : 1;
    const int c4_min = getenv("DSX_WS_C4_MIN") ? atoi(getenv("DSX_WS_C4_MIN")) : 8;   // chunks: at least two groups
    ConvArgs t4 = a;
    t4.ws_cpg = 4; t4.lds_row = conv_lds_row_1x1_c4(a.tw_log2);
    if (ws_c4 && a.kchunks >= c4_min && conv_ws_lds_bytes(dtype, tile, ks, t4) != 0) { a.ws_cpg = 4; a.lds_row = t4.lds_row; }
  }
  // (Tried and rejected in round 3, measured: the GroupNorm finalised by the consuming conv's own compute waves during
  // their start-up wait -- 14 to 22 k_gn_finalize launches fewer, but every such conv started 3-7 us later, the same
  // or more than the launch it replaced cost inside the captured graph: step +0.4 .. +1.1 %.  DESIGN.md section 4.)
  if (s.gn) {   // every other kernel takes the per-channel scale / shift a k_gn_finalize launch prepares
    float *sc = nullptr, *sh = nullptr;
    plan_gn(ex, *s.gn, s.x0, s.x1.C ? &s.x1 : nullptr, &sc, &sh);
    a.gn_scale = sc; a.gn_shift = sh;
  }
  ex->launches++;
  if (fuse_stats && s.want_stats && mfma_ok && (use_ws ? conv_ws_fuses_stats(tile) : conv_tile_fuses_stats(tile)) &&
      a.ksplit == 1 && a.tb_log2 == 0 &&
      (a.Cout & 15) == 0 &&
      a.out_ld == a.Cout && (a.resid_ld & 7) == 0) {
    StatInfo& si = ex->stats[s.out.id];
    si.nchunk = a.tiles_x * a.tiles_y * (use_ws ? conv_ws_tile_wm(tile) : conv_tile_wm(tile));
    si.part = ws_alloc(ex, (size_t)a.B * si.nchunk * a.Cout * 2 * sizeof(float));
    si.planned = true;
    si.f32 = true;
    a.stat_part = (float*)si.part;
  }
  float* slab = nullptr;
  float* reduce_stats = nullptr;
  if (mfma_ok && a.ksplit > 1) {
    slab = (float*)ws_alloc(ex, (size_t)a.ksplit * a.slab_stride * sizeof(float));
    ex->launches++;
    if (fuse_stats && s.want_stats && a.Cout % 64 == 0 && a.out_ld == a.Cout && (a.Ho * a.Wo) % 16 == 0) {
      StatInfo& si = ex->stats[s.out.id];   // statistics in the reduce launch
      si.nchunk = a.Ho * a.Wo / 16;
      si.part = ws_alloc(ex, (size_t)a.B * si.nchunk * a.Cout * 2 * sizeof(float));
      si.planned = true;
      si.f32 = true;
      reduce_stats = (float*)si.part;
    }
  }
  const int host_on = getenv("DSX_HOST_FIN") ? atoi(getenv("DSX_HOST_FIN")) : 1;   // (read at every plan: the tests flip it)
  const bool host_fin = s.host_fin && host_on && use_ws;   // (shapes only: the sizing pass arms it too, plan_gn counts launches)
  if (host_fin) ex->fin_host_armed = true;
  if (ex->sizing) return DSX_OK;
  if (mfma_ok) {
    const ConvTileInfo ti = conv_tile_info(tile);
    const std::string d = fmt("conv%dx%d%s%s %d->%d @%dx%d tile%dx%d", ks, ks, stride == 2 ? "s2" : "",
                              a.up ? "up" : "", (int)cin, a.Cout, a.Ho, a.Wo, ti.BM, ti.BN);
    if (a.ksplit > 1) {
      ConvArgs p = a;  // slices write raw sums into fp32 slabs; a reduce launch applies the epilogue
      p.out = slab; p.out_bf16 = 0;
      SplitKReduceArgs ra{};
      ra.slab = slab; ra.nsplit = a.ksplit; ra.slab_stride = a.slab_stride;
      ra.M = (long long)a.B * a.Ho * a.Wo; ra.N = a.Cout; ra.HW = a.Ho * a.Wo;
      ra.bias = a.bias; ra.film = a.film; ra.film_bs = a.film_bs;
      ra.resid = a.resid; ra.resid_ld = a.resid_ld; ra.out = a.out; ra.act_bf16 = a.act_bf16;
      ra.stat_part = reduce_stats;
      add_op(ex, DSX_OP_CONV_MFMA, d + fmt(" splitK%d", a.ksplit), flops, bytes,
             [=](hipStream_t st) { return launch_conv(dtype, tile, ks, stride, p, st); });
      add_op(ex, DSX_OP_SPLITK_REDUCE, fmt("splitk_reduce x%d %d ch @%dx%d", a.ksplit, a.Cout, a.Ho, a.Wo), 0.0,
             (4.0 * a.ksplit + esz) * (double)ra.M * ra.N,
             [=](hipStream_t st) { return launch_splitk_reduce(ra, st); });
    } else {
      ConvArgs w = a;
      w.handoff_timeouts = ex->handoff_timeouts;
      w.xcd_bands = getenv("DSX_XCD_BANDS") ? atoi(getenv("DSX_XCD_BANDS")) : 1;
      w.ws_wg_per_n = ws_wg_per_n(a);
      // 1 x 1 convs with several N tiles: an XCD takes every N tile of its M tiles (ws_map 3).  With the N tiles dealt over
      // the XCDs (the 3 x 3 choice: there the weights are the larger operand) every L2 fetched most of the input: 65.7 MB
      // per launch of the 512 -> 1536 qkv conv against 18.4 MB algorithmic (PMC, profiles/r03_pmc_summary.txt).
      const int map3_on = getenv("DSX_WS_MAP3") ? atoi(getenv("DSX_WS_MAP3")) : 1;
      bool map3 = false;
      if (map3_on && use_ws && ks == 1 && a.n_tiles >= 2 && a.n_tiles <= 32) {
        int wpn3 = std::max(8, (256 / a.n_tiles) / 8 * 8);
        if (wpn3 > a.m_tiles) wpn3 = (a.m_tiles + 7) / 8 * 8;
        w.ws_wg_per_n = wpn3;
        map3 = true;
      }
      {   // division-free start-up of k_conv_ws: quotients and fastdiv magics (see ConvArgs::ws_map)
        const int NT = a.n_tiles, wpn = w.ws_wg_per_n, per_img = a.tiles_x * a.tiles_y;
        w.ws_map = map3 ? 3 : ((NT <= 8 && 8 % NT == 0 && wpn % (8 / NT) == 0) ? 0 : ((NT % 8) == 0 ? 1 : 2));
        w.ws_nt_log2 = NT <= 8 ? ilog2(NT) : 0;
        w.ws_per = map3 ? NT : (NT >> 3);
        w.ws_adv_x = wpn % a.tiles_x; w.ws_adv_y = (wpn / a.tiles_x) % a.tiles_y; w.ws_adv_b = wpn / per_img;
        const int PW = ((1 << a.tw_log2) - 1) + ks;                       // stride 1
        const int upg = 4 * (a.ws_cpg ? a.ws_cpg : conv_chunk_multiple(ks));   // 16-byte units per pixel and group
        const int pstep = 256 / upg;                                      // loader threads / units per pixel
        w.ws_dpy = pstep / PW; w.ws_dpx = pstep - w.ws_dpy * PW;
        w.mg_tiles_x = fastdiv_magic((unsigned)a.tiles_x); w.mg_per_img = fastdiv_magic((unsigned)per_img);
        w.mg_pw = fastdiv_magic((unsigned)PW); w.mg_wpn = fastdiv_magic((unsigned)wpn);
        w.mg_per = fastdiv_magic((unsigned)std::max(1, w.ws_per));
        w.ws_bigdiv = ((long long)a.m_tiles + wpn >= 65536 || a.tiles_x >= 65536 || per_img >= 65536) ? 1 : 0;
        // the magics are exact for dividends below 65536; check the ones this launch can produce (a few thousand
        // multiplications per conv at plan time) rather than trust the bound
        auto exact = [](unsigned d, unsigned magic, unsigned nmax) {
          for (unsigned n = 0; n <= nmax; ++n) {
            const unsigned q = magic ? (unsigned)(((unsigned long long)n * magic) >> 32) : n;
            if (q != n / d) return false;
          }
          return true;
        };
        const unsigned grid = (unsigned)(NT * wpn);
        const unsigned nt_max = w.ws_bigdiv ? 0u : (unsigned)(a.m_tiles + wpn);
        if (!exact((unsigned)a.tiles_x, w.mg_tiles_x, grid) || !exact((unsigned)per_img, w.mg_per_img, grid) ||
            !exact((unsigned)PW, w.mg_pw, 255u) || !exact((unsigned)wpn, w.mg_wpn, std::max(grid, nt_max)) ||
            !exact((unsigned)std::max(1, w.ws_per), w.mg_per, grid >> 3))
          return fail(DSX_ERR_INVALID, "planner: fastdiv magic not exact for conv %dx%d @%dx%d", ks, ks, a.Ho, a.Wo);
      }
      if (use_ws) {
        // this conv's k_gn_finalize launch pulls the weight slices into the L2 of the XCD group that will read them
        // (k_conv_ws keys its N tile on blockIdx % 8 in exactly these two cases)
        static const int pf_on = getenv("DSX_PREFETCH") ? atoi(getenv("DSX_PREFETCH")) : 1;
        const int NT = a.n_tiles;
        const bool keyed = !map3 && ((NT <= 8 && 8 % NT == 0 && w.ws_wg_per_n % (8 / NT) == 0) || (NT % 8) == 0);
        const size_t wblock = (size_t)a.kchunks * ks * ks * 2 * 1024;     // bytes of one 32-channel N block's fragments
        const PrefetchArgs mine{a.wpack, (unsigned)(wblock * (ti.BN / 32)), NT, nullptr};
        static const int pf_ws = getenv("DSX_PREFETCH_WS") ? atoi(getenv("DSX_PREFETCH_WS")) : 1;
        if (pf_on && ex->pending_gn_pf && keyed) *ex->pending_gn_pf = mine;
        // no finalize launch in front (1 x 1 without GroupNorm, upsampling conv): the previous k_conv_ws launch carries it
        else if (pf_on && pf_ws && ex->prev_ws_pf && keyed) *ex->prev_ws_pf = mine;
        auto npf = std::make_shared<PrefetchArgs>(PrefetchArgs{nullptr, 0u, 0, nullptr});
        ex->prev_ws_pf = npf;
        w.pf = PrefetchArgs{nullptr, 0u, 0, nullptr};
        if (host_fin) {
          auto hf = std::make_shared<dsx_exec::HostedFin>();
          ex->fin_host = hf;
          add_op(ex, DSX_OP_CONV_MFMA, d + (a.ws_cpg == 4 ? " ws c4 +gn" : " ws +gn"), flops, bytes, [=](hipStream_t st) {
            ConvArgs b = w;
            b.pf = *npf;
            if (hf->on) { b.fin_on = 1; b.fin = hf->a; if (hf->pf) b.fin.pf = *hf->pf; }
            return launch_conv_ws(dtype, tile, ks, b, st);
          });
        } else {
          add_op(ex, DSX_OP_CONV_MFMA, d + (a.ws_cpg == 2 ? " ws c2" : (a.ws_cpg == 4 ? " ws c4" : " ws")), flops, bytes,
                 [=](hipStream_t st) { ConvArgs b = w; b.pf = *npf; return launch_conv_ws(dtype, tile, ks, b, st); });
        }
      } else {
        add_op(ex, DSX_OP_CONV_MFMA, a.cpg == 2 ? d + " g2" : d, flops, bytes,
               [=](hipStream_t st) { return launch_conv(dtype, tile, ks, stride, a, st); });
      }
    }
  } else {
    if (!s.w->naive)
      return fail(DSX_ERR_INVALID,
                  "no MFMA tile fits conv %dx%d (%dx%d out, B=%d); set DSX_CONV_IMPL=naive", ks, ks,
                  a.Ho, a.Wo, a.B);
    NaiveConvArgs na{};
    na.c = a; na.w = s.w->naive; na.ks = ks; na.stride = stride; na.sigmoid_out = 0;
    add_op(ex, DSX_OP_CONV_NAIVE, fmt("conv%dx%d-naive %d->%d @%dx%d", ks, ks, (int)cin, a.Cout, a.Ho, a.Wo),
           flops, bytes, [=](hipStream_t st) { return launch_conv_naive(na, st); });
  }
  return DSX_OK;
}

static void plan_stats(dsx_exec* ex, const Tensor& t) {
  StatInfo& si = ex->stats[t.id];
  if (si.planned) return;
  const int HW = t.H * t.W;
  int nchunk = std::max(1, 512 / ex->B);
  nchunk = std::min(nchunk, std::max(1, HW / 16));
  nchunk = std::min(nchunk, 64);
  si.nchunk = nchunk;
  si.part = ws_alloc(ex, (size_t)ex->B * nchunk * t.C * 2 * sizeof(double));
  si.planned = true;
  si.f32 = false;
  ex->launches++;
  if (ex->sizing) return;
  const void* x = t.p; double* part = (double*)si.part;
  const int B = ex->B, C = t.C, xbf = t.st;
  add_op(ex, DSX_OP_GN_STATS, fmt("gn_stats C=%d @%dx%d", C, t.H, t.W), 0.0, (xbf ? 2.0 : 4.0) * B * HW * C,
         [=](hipStream_t st) { return launch_chan_stats(x, xbf, B, HW, C, nchunk, part, st); });
}

// GroupNorm over cat(t0, t1) -> device scale/shift [B][C]
static void plan_gn(dsx_exec* ex, const GnW& g, const Tensor& t0, const Tensor* t1, float** scale, float** shift) {
  plan_stats(ex, t0);
  if (t1) plan_stats(ex, *t1);
  const int C = t0.C + (t1 ? t1->C : 0);
  *scale = (float*)ws_alloc(ex, (size_t)ex->B * C * sizeof(float));
  *shift = (float*)ws_alloc(ex, (size_t)ex->B * C * sizeof(float));
  const bool hosted = ex->fin_host_armed;      // the launch in front is a residual 1 x 1 conv whose loader waves do it
  ex->fin_host_armed = false;
  if (!hosted) ex->launches++;
  if (ex->sizing) return;
  GnFinArgs a{};
  a.part0 = ex->stats[t0.id].part; a.C0 = t0.C; a.nchunk0 = ex->stats[t0.id].nchunk;
  a.f32_0 = ex->stats[t0.id].f32 ? 1 : 0;
  a.part1 = t1 ? ex->stats[t1->id].part : nullptr; a.C1 = t1 ? t1->C : 0;
  a.nchunk1 = t1 ? ex->stats[t1->id].nchunk : 0;
  a.f32_1 = (t1 && ex->stats[t1->id].f32) ? 1 : 0;
  a.B = ex->B; a.groups = ex->m->cfg.norm_groups; a.count = (double)t0.H * t0.W;
  a.gamma = g.gamma; a.beta = g.beta; a.eps = 1e-5f;
  a.scale = *scale; a.shift = *shift;
  auto pf = std::make_shared<PrefetchArgs>(PrefetchArgs{nullptr, 0u, 0, nullptr});
  ex->pending_gn_pf = pf;    // plan_conv fills it in once it has chosen the consumer's kernel and tiling
  if (hosted && ex->fin_host) {
    ex->fin_host->a = a; ex->fin_host->pf = pf; ex->fin_host->on = true;
    ex->fin_host.reset();
    return;
  }
  add_op(ex, DSX_OP_GN_FINALIZE, fmt("gn_finalize C=%d", C), 0.0, 0.0,
         [=](hipStream_t st) { GnFinArgs b = a; b.pf = *pf; return launch_gn_finalize(b, st); });
}

static int plan_res(dsx_exec* ex, const Module& md, const Tensor& x0, const Tensor* x1, Tensor& y) {
  int rc;
  const int H = x0.H, W = x0.W;
  Tensor h = new_tensor(ex, md.cout, H, W);
  ConvSpec c1{};
  c1.w = &md.conv1; c1.x0 = x0; if (x1) c1.x1 = *x1;
  c1.gn = &md.gn1; c1.swish = true;
  if (md.film_off >= 0 && !ex->sizing) { c1.film = ex->film + md.film_off; c1.film_bs = ex->m->F; }
  c1.bias_in_film = md.film_off >= 0 && md.conv1.pb >= 0;
  c1.out = h; c1.want_stats = true;
  if ((rc = plan_conv(ex, c1))) return rc;
  Tensor r;
  if (md.has_res) {
    r = new_tensor(ex, md.cout, H, W);
    ConvSpec cr{};
    cr.w = &md.res; cr.x0 = x0; if (x1) cr.x1 = *x1; cr.out = r;
    // the hosted finalize reads h's GroupNorm partial sums: only when conv1's epilogue produced them (fused statistics;
    // both planner passes agree, `planned` is set from shapes).  Otherwise c2's plan_gn adds a k_chan_stats launch
    // AFTER this conv and the finalize must stay behind it as a launch of its own.
    cr.host_fin = ex->stats[h.id].planned;
    if ((rc = plan_conv(ex, cr))) return rc;
  } else {
    r = x0;
  }
  Tensor o = new_tensor(ex, md.cout, H, W);
  ConvSpec c2{};
  c2.w = &md.conv2; c2.x0 = h; c2.gn = &md.gn2; c2.swish = true;
  c2.resid = r.p; c2.resid_ld = md.cout; c2.has_resid = true; c2.out = o; c2.want_stats = true;
  if ((rc = plan_conv(ex, c2))) return rc;
  ex->fin_host_armed = false; ex->fin_host.reset();
  if (!md.attn) { y = o; return DSX_OK; }
  // SelfAttention (unet.py:113-142)
  const int C = md.cout, L = H * W, B = ex->B;
  Tensor qkv = new_tensor(ex, 3 * C, H, W);
  ConvSpec cq{};
  cq.w = &md.qkv; cq.x0 = o; cq.gn = &md.gna; cq.out = qkv;
  if ((rc = plan_conv(ex, cq))) return rc;
  Tensor av = new_tensor(ex, C, H, W);
  ex->launches += 1;
  if (!attn_supported(C, L)) return fail(DSX_ERR_INVALID, "attention with head dimension %d is not supported (8..1024, multiple of 8)", C);
  if (!ex->sizing) {
    AttnArgs g{};
    g.q = qkv.p; g.k = qkv.at(C); g.v = qkv.at(2 * (size_t)C); g.ld = 3 * C;
    g.out = av.p; g.ldo = C; g.storage = qkv.st;
    g.B = B; g.L = L; g.C = C; g.div = sqrtf((float)C); g.inv_div = 1.0f / g.div;
    const double esz = qkv.st ? 2.0 : 4.0;
    add_op(ex, DSX_OP_ATTN_GEMM, fmt("attn fused L=%d d=%d", L, C), 4.0 * B * L * (double)L * C,
           B * esz * 4.0 * L * C, [=](hipStream_t st) { return launch_attn(g, st); });
  }
  Tensor o2 = new_tensor(ex, C, H, W);
  ConvSpec co{};
  co.w = &md.out; co.x0 = av; co.resid = o.p; co.resid_ld = C; co.has_resid = true; co.out = o2; co.want_stats = true;
  if ((rc = plan_conv(ex, co))) return rc;
  y = o2;
  return DSX_OK;
}

static int build_plan(dsx_exec* ex) {
  dsx_model* m = ex->m;
  ex->ws_used = 0;
  ex->ops.clear();
  ex->conv_ordinal = 0;
  ex->prev_img_pf.reset();
  ex->prev_ws_pf.reset();
  ex->pending_gn_pf.reset();
  ex->op_info.clear();
  ex->stats.clear();
  ex->launches = 0;
  const int B = ex->B;
  ex->step_ctr = (int*)ws_alloc(ex, 256);
  ex->loop_params = (unsigned long long*)(ex->sizing ? nullptr : (char*)ex->step_ctr + 64);
  ex->handoff_timeouts = (unsigned*)(ex->sizing ? nullptr : (char*)ex->step_ctr + 128);   // (zeroed with the block at create)
  ex->time_buf = (float*)ws_alloc(ex, (size_t)B * sizeof(float));
  ex->film = m->F ? (float*)ws_alloc(ex, (size_t)B * m->F * sizeof(float)) : nullptr;
  ex->in_cond = Tensor();
  if (ex->cond_c) ex->in_cond = new_tensor(ex, ex->cond_c, ex->H, ex->W);
  ex->in_x = new_tensor(ex, ex->x_c, ex->H, ex->W);
  ex->x_state = ex->in_x.st ? (float*)ws_alloc(ex, (size_t)B * ex->H * ex->W * ex->x_c * sizeof(float))
                              : (float*)ex->in_x.p;
  std::vector<Tensor> feats;
  Tensor x;
  int rc;
  for (auto& md : m->mods) {
    if (md.kind == 0) {
      Tensor o = new_tensor(ex, md.cout, ex->H, ex->W);
      ConvSpec c{};
      c.w = &md.conv;
      if (ex->cond_c) { c.x0 = ex->in_cond; c.x1 = ex->in_x; } else c.x0 = ex->in_x;
      c.out = o; c.want_stats = true;
      if ((rc = plan_conv(ex, c))) return rc;
      x = o; feats.push_back(x);
    } else if (md.kind == 2) {
      if ((x.H & 1) || (x.W & 1)) return fail(DSX_ERR_INVALID, "H and W must be divisible by 2^(levels-1)");
      Tensor o = new_tensor(ex, md.cout, x.H / 2, x.W / 2);
      ConvSpec c{};
      c.w = &md.conv; c.x0 = x; c.stride = 2; c.out = o; c.want_stats = true;
      if ((rc = plan_conv(ex, c))) return rc;
      x = o; feats.push_back(x);
    } else if (md.kind == 3) {
      Tensor o = new_tensor(ex, md.cout, x.H * 2, x.W * 2);
      ConvSpec c{};
      c.w = &md.conv; c.x0 = x; c.up = true; c.out = o; c.want_stats = true;
      if ((rc = plan_conv(ex, c))) return rc;
      x = o;
    } else if (md.kind == 1) {
      Tensor y;
      if (md.section == 2) {
        Tensor skip = feats.back();
        feats.pop_back();
        if (skip.H != x.H || skip.W != x.W || skip.C != md.skip) return fail(DSX_ERR_INVALID, "skip mismatch");
        if ((rc = plan_res(ex, md, x, &skip, y))) return rc;
      } else {
        if ((rc = plan_res(ex, md, x, nullptr, y))) return rc;
      }
      x = y;
      if (md.section == 0) feats.push_back(x);
    } else {
      Tensor o = new_tensor(ex, md.cout, x.H, x.W, /*f32=*/true);   // the network's output feeds the fp32 sampler update
      ConvSpec c{};
      c.w = &md.conv; c.x0 = x; c.gn = &md.gn1; c.swish = true; c.out = o;
      if ((rc = plan_conv(ex, c))) return rc;
      x = o;
    }
  }
  ex->out = x;
  if (ex->out.st) return fail(DSX_ERR_STATE, "internal error: the network output must be an fp32 tensor");
  if (ex->overflow)
    return fail(DSX_ERR_STATE, "internal error: the planning pass needs more workspace than the sizing pass reserved");
  return DSX_OK;
}

extern "C" int dsx_exec_create(dsx_model* m, int B, int H, int W, int cond_channels, dsx_exec** out) {
  if (!m || !out || B < 1 || H < 1 || W < 1) return fail(DSX_ERR_INVALID, "bad argument");
  if (!m->finalized) return fail(DSX_ERR_STATE, "dsx_model_finalize must precede dsx_exec_create");
  if (cond_channels < 0 || cond_channels >= m->cfg.in_channel) return fail(DSX_ERR_INVALID, "bad cond_channels");
  HIP_TRY(conv_init());
  HIP_TRY(ops_init());
  dsx_exec* ex = new dsx_exec();
  ex->m = m; ex->B = B; ex->H = H; ex->W = W;
  ex->cond_c = cond_channels; ex->x_c = m->cfg.in_channel - cond_channels;
  ex->sizing = true;
  int rc = build_plan(ex);
  if (rc) { delete ex; return rc; }
  ex->ws_bytes = ex->ws_used + 4096;
  hipError_t e = hipMalloc((void**)&ex->ws, ex->ws_bytes);
  if (e != hipSuccess) {
    delete ex;
    return fail(DSX_ERR_HIP, "hipMalloc(%zu) for the activation workspace failed: %s", ex->ws_bytes,
                hipGetErrorString(e));
  }
  ex->sizing = false;
  const size_t sized = ex->ws_used;
  rc = build_plan(ex);
  if (rc == DSX_OK && ex->ws_used != sized)
    rc = fail(DSX_ERR_STATE, "internal error: sizing pass reserved %zu bytes, planning pass used %zu", sized, ex->ws_used);
  if (rc) { (void)hipFree(ex->ws); delete ex; return rc; }
  e = hipMemset(ex->step_ctr, 0, 256);
  if (e != hipSuccess) { (void)hipFree(ex->ws); delete ex; return fail(DSX_ERR_HIP, "hipMemset failed"); }
  *out = ex;
  return DSX_OK;
}

// Host-only: both planner passes for (cfg, dtype, B, H, W) without a device (the workspace base is a fake
// address that is never dereferenced).  Tests use it to pin that sizing and planning agree under every tile
// preference setting.
extern "C" int dsx_plan_dry_run(const dsx_unet_cfg* cfg, int dtype, int B, int H, int W, int cond_channels,
                                size_t* sizing_bytes, size_t* planning_bytes, int* launches) {
  if (!cfg || B < 1 || H < 1 || W < 1) return fail(DSX_ERR_INVALID, "bad argument");
  if (dtype != DSX_DTYPE_F32 && dtype != DSX_DTYPE_BF16 && dtype != DSX_DTYPE_F16) return fail(DSX_ERR_INVALID, "bad dtype");
  dsx_model* m = nullptr;
  int rc = dsx_model_create(cfg, &m);
  if (rc) return rc;
  if (cond_channels < 0 || cond_channels >= m->cfg.in_channel) { dsx_model_destroy(m); return fail(DSX_ERR_INVALID, "bad cond_channels"); }
  m->dtype = dtype;
  for (auto& md : m->mods)
    for (ConvW* c : {&md.conv, &md.conv1, &md.conv2, &md.res, &md.qkv, &md.out})
      if (c->pw >= 0) conv_geometry(c->cout, c->cin, c->ks, dtype, c->kchunks, c->nblocks);
  dsx_exec* ex = new dsx_exec();
  ex->m = m; ex->B = B; ex->H = H; ex->W = W;
  ex->cond_c = cond_channels; ex->x_c = m->cfg.in_channel - cond_channels;
  ex->sizing = true;
  rc = build_plan(ex);
  const size_t sized = ex->ws_used;
  size_t planned = 0;
  if (rc == DSX_OK) {
    ex->sizing = false;
    ex->ws = (char*)(uintptr_t)0x100000000ull;   // never dereferenced: no launch happens
    ex->ws_bytes = ~(size_t)0 >> 1;
    rc = build_plan(ex);
    planned = ex->ws_used;
  }
  if (sizing_bytes) *sizing_bytes = sized;
  if (planning_bytes) *planning_bytes = planned;
  if (launches) *launches = ex->launches;
  ex->ws = nullptr;
  delete ex;
  dsx_model_destroy(m);
  return rc;
}

extern "C" void dsx_exec_destroy(dsx_exec* ex) {
  if (!ex) return;
  if (ex->graph_exec) (void)hipGraphExecDestroy(ex->graph_exec);
  if (ex->graph) (void)hipGraphDestroy(ex->graph);
  if (ex->table) (void)hipFree(ex->table);
  if (ex->tp_w) (void)hipFree(ex->tp_w);
  if (ex->ws) (void)hipFree(ex->ws);
  for (auto& sl : ex->staging) {
    if (sl.ev) { if (sl.busy) (void)hipEventSynchronize(sl.ev); (void)hipEventDestroy(sl.ev); }
    if (sl.host) (void)hipHostFree(sl.host);
  }
  delete ex;
}
extern "C" size_t dsx_exec_workspace_bytes(const dsx_exec* ex) { return ex ? ex->ws_bytes : 0; }
// Bounded spins of the conv kernel's loader -> compute hand-off (three-image tiles) that gave up since the executor was
// created: 0 in every correct run; anything else means wrong pixels were produced.  Synchronises the device.
extern "C" int dsx_exec_handoff_timeouts(dsx_exec* ex, unsigned* count) {
  if (!ex || !count) return fail(DSX_ERR_INVALID, "null argument");
  HIP_TRY(hipDeviceSynchronize());
  HIP_TRY(hipMemcpy(count, ex->handoff_timeouts, sizeof(unsigned), hipMemcpyDeviceToHost));
  return DSX_OK;
}
extern "C" int dsx_exec_num_launches(const dsx_exec* ex) { return ex ? ex->launches : 0; }

extern "C" int dsx_exec_num_ops(const dsx_exec* ex) { return ex ? (int)ex->ops.size() : 0; }
extern "C" int dsx_exec_op_info(const dsx_exec* ex, int i, char* desc, int cap, int* kind, double* flops,
                                double* bytes) {
  if (!ex || i < 0 || i >= (int)ex->op_info.size()) return fail(DSX_ERR_INVALID, "bad op index");
  const OpInfo& o = ex->op_info[i];
  if (desc && cap > 0) snprintf(desc, cap, "%s", o.desc.c_str());
  if (kind) *kind = o.kind;
  if (flops) *flops = o.flops;
  if (bytes) *bytes = o.bytes;
  return DSX_OK;
}
// diagnostics: copy the 128 in-kernel stamps of the launch chosen with DSX_STAMP_OP (zeros if none)
extern "C" int dsx_exec_time_kind(dsx_exec* ex, int kind, int iters, float* ms_per_replay, int* launches,
                                  void* stream) {
  if (!ex || !ms_per_replay || iters < 1) return fail(DSX_ERR_INVALID, "bad argument");
  hipStream_t st = (hipStream_t)stream;
  if (!st) return fail(DSX_ERR_INVALID, "dsx_exec_time_kind needs a non-default stream (stream capture)");
  // kind >= 0: the launches of that kind; -1: every launch of the forward; <= -2: every launch except kind (-kind - 2)
  auto selected = [&](int k) { return kind >= 0 ? k == kind : (kind == -1 ? true : k != -kind - 2); };
  int n = 0;
  for (auto& oi : ex->op_info) n += selected(oi.kind) ? 1 : 0;
  if (launches) *launches = n;
  if (n == 0) { *ms_per_replay = 0.f; return DSX_OK; }
  hipGraph_t graph = nullptr;
  hipGraphExec_t gexec = nullptr;
  HIP_TRY(hipStreamSynchronize(st));
  HIP_TRY(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
  hipError_t err = hipSuccess;
  for (size_t i = 0; i < ex->ops.size() && err == hipSuccess; ++i)
    if (selected(ex->op_info[i].kind)) err = ex->ops[i](st);
  hipError_t e2 = hipStreamEndCapture(st, &graph);
  if (err != hipSuccess || e2 != hipSuccess || !graph) {
    if (graph) (void)hipGraphDestroy(graph);
    return fail(DSX_ERR_HIP, "capture of the kernel family failed: %s", hipGetErrorString(err != hipSuccess ? err : e2));
  }
  HIP_TRY(hipGraphInstantiate(&gexec, graph, nullptr, nullptr, 0));
  hipEvent_t e0, e1;
  HIP_TRY(hipEventCreate(&e0));
  HIP_TRY(hipEventCreate(&e1));
  HIP_TRY(hipGraphLaunch(gexec, st));   // warm-up replay
  HIP_TRY(hipEventRecord(e0, st));
  for (int it = 0; it < iters; ++it) HIP_TRY(hipGraphLaunch(gexec, st));
  HIP_TRY(hipEventRecord(e1, st));
  HIP_TRY(hipStreamSynchronize(st));
  float ms = 0.f;
  HIP_TRY(hipEventElapsedTime(&ms, e0, e1));
  *ms_per_replay = ms / iters;
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
  (void)hipGraphExecDestroy(gexec); (void)hipGraphDestroy(graph);
  return DSX_OK;
}

extern "C" int dsx_exec_read_stamps(dsx_exec* ex, unsigned long long* out128) {
  if (!ex || !out128) return fail(DSX_ERR_INVALID, "null argument");
  memset(out128, 0, 128 * 8);
  if (!ex->stamp_buf) return DSX_OK;
  HIP_TRY(hipDeviceSynchronize());
  HIP_TRY(hipMemcpy(out128, ex->stamp_buf, 128 * 8, hipMemcpyDeviceToHost));
  return DSX_OK;
}

// Eager, event-timed replay of the UNet plan on `stream` (inputs: whatever the
// buffers hold).  ms_per_op[i] = mean over `iters` of the hipEvent time around launch i.
extern "C" int dsx_exec_profile(dsx_exec* ex, int iters, float* ms_per_op, void* stream) {
  if (!ex || !ms_per_op || iters < 1) return fail(DSX_ERR_INVALID, "bad argument");
  hipStream_t st = (hipStream_t)stream;
  const size_t n = ex->ops.size();
  std::vector<hipEvent_t> ev(2 * n);
  for (auto& e : ev) HIP_TRY(hipEventCreate(&e));
  std::vector<double> acc(n, 0.0);
  for (int it = 0; it < iters; ++it) {
    static const bool trace = getenv("DSX_TRACE") != nullptr;   // debugging: name each launch, sync after it
    for (size_t i = 0; i < n; ++i) {
      if (trace) { fprintf(stderr, "[dsx] op %zu: %s\n", i, ex->op_info[i].desc.c_str()); fflush(stderr); }
      HIP_TRY(hipEventRecord(ev[2 * i], st));
      HIP_TRY(ex->ops[i](st));
      HIP_TRY(hipEventRecord(ev[2 * i + 1], st));
      if (trace) HIP_TRY(hipStreamSynchronize(st));
    }
    HIP_TRY(hipStreamSynchronize(st));
    for (size_t i = 0; i < n; ++i) {
      float ms = 0.f;
      HIP_TRY(hipEventElapsedTime(&ms, ev[2 * i], ev[2 * i + 1]));
      acc[i] += ms;
    }
  }
  for (size_t i = 0; i < n; ++i) ms_per_op[i] = (float)(acc[i] / iters);
  for (auto& e : ev) (void)hipEventDestroy(e);
  return DSX_OK;
}

// time embedding + UNet body on `st`; inputs already in ex->in_cond / ex->in_x
static int run_unet(dsx_exec* ex, bool from_table, int n_time, hipStream_t st, int per_sample = 0) {
  dsx_model* m = ex->m;
  if (m->cfg.with_time_emb) {
    TembArgs t{};
    t.flavour = m->cfg.flavour; t.B = ex->B; t.n_time = n_time;
    t.time = from_table ? nullptr : ex->time_buf;
    t.table = ex->table; t.step_ctr = ex->step_ctr; t.per_sample = per_sample;
    t.inner = m->cfg.inner_channel; t.freq = m->d_freq;
    t.w1 = m->d_w1; t.b1 = m->d_b1; t.w2 = m->d_w2; t.b2 = m->d_b2;
    t.wf = m->d_wf; t.bf = m->d_bf; t.F = m->F; t.film = ex->film;
    HIP_TRY(launch_temb(t, st));
  }
  for (auto& op : ex->ops) HIP_TRY(op(st));
  return DSX_OK;
}

static int load_inputs(dsx_exec* ex, const float* cond_nchw, const float* x_nchw, int x_total_c,
                       int x_c_off, hipStream_t st);

extern "C" int dsx_unet_forward(dsx_exec* ex, const float* x, const float* time, int n_time, float* y,
                                void* stream) {
  if (!ex || !x || !y) return fail(DSX_ERR_INVALID, "null argument");
  hipStream_t st = (hipStream_t)stream;
  dsx_model* m = ex->m;
  if (m->cfg.with_time_emb) {
    if (!time || !(n_time == 1 || n_time == ex->B)) return fail(DSX_ERR_INVALID, "n_time must be 1 or B");
    HIP_TRY(hipMemcpyAsync(ex->time_buf, time, (size_t)n_time * 4, hipMemcpyDeviceToDevice, st));
  }
  // x is (B, in_channel, H, W): channels [0,cond_c) feed the cond tensor, the rest the state tensor
  int rc = load_inputs(ex, ex->cond_c ? x : nullptr, x, m->cfg.in_channel, ex->cond_c, st);
  if (rc) return rc;
  if ((rc = run_unet(ex, false, n_time, st))) return rc;
  HIP_TRY(launch_nhwc_to_nchw((const float*)ex->out.p, y, ex->B, ex->out.C, ex->H, ex->W, st));
  return DSX_OK;
}

static int load_inputs(dsx_exec* ex, const float* cond_nchw, const float* x_nchw, int x_total_c,
                       int x_c_off, hipStream_t st) {
  const int HW = ex->H * ex->W;
  if (ex->cond_c) {
    if (!cond_nchw) return fail(DSX_ERR_INVALID, "this executor was created with cond_channels > 0");
    const int ctot = (cond_nchw == x_nchw) ? x_total_c : ex->cond_c;
    HIP_TRY(dsx::launch_nchw_slice_to_nhwc(cond_nchw, ex->in_cond.p, ex->in_cond.st, ex->B, ex->cond_c, ctot,
                                           0, HW, st));
  }
  HIP_TRY(dsx::launch_nchw_slice_to_nhwc(x_nchw, ex->in_x.p, ex->in_x.st, ex->B, ex->x_c, x_total_c, x_c_off,
                                         HW, st));
  if (ex->in_x.st)   // the sampler state itself stays fp32
    HIP_TRY(dsx::launch_nchw_slice_to_nhwc(x_nchw, ex->x_state, 0, ex->B, ex->x_c, x_total_c, x_c_off, HW, st));
  return DSX_OK;
}

// ------------------------------------------------------------------ sampler
static int ensure_table(dsx_exec* ex, const dsx_step_table* tab) {
  const int T = tab->n_steps * (tab->per_sample > 0 ? tab->per_sample : 1);   // values per column
  if (T > ex->table_cap) {
    // the table's column stride (= capacity) is baked into captured graphs: drop them
    if (ex->graph_exec) {
      if (ex->last_stream) HIP_TRY(hipStreamSynchronize(ex->last_stream));
      (void)hipGraphExecDestroy(ex->graph_exec); ex->graph_exec = nullptr;
    }
    if (ex->graph) { (void)hipGraphDestroy(ex->graph); ex->graph = nullptr; }
    if (ex->table) (void)hipFree(ex->table);
    ex->table = nullptr;
    const int cap = std::max(T, 2048);
    HIP_TRY(hipMalloc((void**)&ex->table, (size_t)6 * cap * sizeof(float)));
    ex->table_cap = cap;
  }
  return DSX_OK;
}

// a pinned staging slot holding this call's table [6][cap] followed by {seed, noise address}; *out = its host pointer
static int stage_call(dsx_exec* ex, const dsx_step_table* tab, uint64_t seed, const float* noise, dsx_exec::Staging** out) {
  const int cap = ex->table_cap;
  const int T = tab->n_steps * (tab->per_sample > 0 ? tab->per_sample : 1);
  const size_t need = (size_t)6 * cap + 4;   // + 16 bytes of loop parameters
  dsx_exec::Staging& sl = ex->staging[ex->staging_next];
  ex->staging_next = (ex->staging_next + 1) % 4;
  if (sl.busy) { HIP_TRY(hipEventSynchronize(sl.ev)); sl.busy = false; }
  if (sl.floats < need) {
    if (sl.host) (void)hipHostFree(sl.host);
    sl.host = nullptr; sl.floats = 0;
    HIP_TRY(hipHostMalloc((void**)&sl.host, need * sizeof(float), hipHostMallocDefault));
    sl.floats = need;
  }
  if (!sl.ev) HIP_TRY(hipEventCreateWithFlags(&sl.ev, hipEventDisableTiming));
  memset(sl.host, 0, (size_t)6 * cap * sizeof(float));
  const float* cols[6] = {tab->tcond, tab->a, tab->b, tab->c1, tab->c2, tab->sigma};
  for (int k = 0; k < 6; ++k)
    if (cols[k]) memcpy(sl.host + (size_t)k * cap, cols[k], (size_t)T * 4);
  unsigned long long lp[2] = {seed, (unsigned long long)(uintptr_t)noise};
  memcpy(sl.host + (size_t)6 * cap, lp, 16);
  *out = &sl;
  return DSX_OK;
}

static int enqueue_step(dsx_exec* ex, const dsx_step_table* tab, bool use_noise, hipStream_t st) {
  int rc = run_unet(ex, true, tab->per_sample > 0 ? ex->B : 1, st, tab->per_sample > 0 ? 1 : 0);
  if (rc) return rc;
  UpdateArgs u{};
  u.x = ex->x_state; u.x_act = ex->in_x.st ? ex->in_x.p : nullptr; u.x_act_kind = ex->in_x.st;
  u.net = (const float*)ex->out.p; u.use_noise = use_noise ? 1 : 0; u.loop_params = ex->loop_params;
  u.tab = ex->table; u.n_steps = ex->table_cap; u.step_ctr = ex->step_ctr;
  u.predict_eps = tab->predict_eps; u.clip = tab->clip; u.per_sample = tab->per_sample > 0 ? 1 : 0;
  u.B = ex->B; u.C = ex->x_c; u.H = ex->H; u.W = ex->W;
  HIP_TRY(launch_update(u, st));
  HIP_TRY(launch_advance(ex->step_ctr, st));
  return DSX_OK;
}

extern "C" int dsx_sample_loop(dsx_exec* ex, const dsx_step_table* tab, const float* cond, float* x,
                               const float* noise, uint64_t seed, const int32_t* snap_steps, int n_snap,
                               float* snaps, int use_graph, void* stream) {
  if (!ex || !tab || !x || tab->n_steps < 1) return fail(DSX_ERR_INVALID, "bad argument");
  if (!tab->tcond || !tab->c1 || !tab->c2 || !tab->sigma || (tab->predict_eps && (!tab->a || !tab->b)))
    return fail(DSX_ERR_INVALID, "step table columns missing");
  if (ex->out.C != ex->x_c)
    return fail(DSX_ERR_INVALID, "UNet out_channel (%d) must equal the state channels (%d)", ex->out.C, ex->x_c);
  if (n_snap > 0 && (!snap_steps || !snaps)) return fail(DSX_ERR_INVALID, "snapshot arrays missing");
  if (tab->per_sample != 0 && tab->per_sample != ex->B)
    return fail(DSX_ERR_INVALID, "per_sample step table for %d samples, executor batch %d", tab->per_sample, ex->B);
  hipStream_t st = (hipStream_t)stream;
  const int T = tab->n_steps;
  int rc = ensure_table(ex, tab);
  if (rc) return rc;
  // per-call values (step table; seed, noise address: the captured step does not bake them in) go to device memory
  // from a pinned staging slot of this call's own
  dsx_exec::Staging* sl = nullptr;
  if ((rc = stage_call(ex, tab, seed, noise, &sl))) return rc;
  HIP_TRY(hipMemcpyAsync(ex->table, sl->host, (size_t)6 * ex->table_cap * 4, hipMemcpyHostToDevice, st));
  HIP_TRY(hipMemcpyAsync(ex->loop_params, sl->host + (size_t)6 * ex->table_cap, 16, hipMemcpyHostToDevice, st));
  HIP_TRY(hipEventRecord(sl->ev, st));
  sl->busy = true;
  HIP_TRY(hipMemsetAsync(ex->step_ctr, 0, 4, st));
  if ((rc = load_inputs(ex, cond, x, ex->x_c, 0, st))) return rc;

  const size_t snap_elems = (size_t)ex->B * ex->x_c * ex->H * ex->W;
  bool graph_ok = false;
  if (use_graph) {
    // the captured step bakes in the mode flags only (not the step count, the seed or the noise address)
    std::vector<float> sig = {(float)tab->predict_eps, (float)tab->clip, noise ? 1.f : 0.f, tab->per_sample > 0 ? 1.f : 0.f};
    if (!ex->graph_exec || sig != ex->graph_sig) {
      if (ex->graph_exec) {
        // a replaced executable graph may still have launches queued: wait for them before destroying it
        if (ex->last_stream) HIP_TRY(hipStreamSynchronize(ex->last_stream));
        (void)hipGraphExecDestroy(ex->graph_exec); ex->graph_exec = nullptr;
      }
      if (ex->graph) { (void)hipGraphDestroy(ex->graph); ex->graph = nullptr; }
      hipStream_t cs;
      HIP_TRY(hipStreamCreateWithFlags(&cs, hipStreamNonBlocking));
      hipError_t e = hipStreamBeginCapture(cs, hipStreamCaptureModeThreadLocal);
      if (e == hipSuccess) {
        rc = enqueue_step(ex, tab, noise != nullptr, cs);
        hipError_t e2 = hipStreamEndCapture(cs, &ex->graph);
        if (rc == DSX_OK && e2 == hipSuccess) e2 = hipGraphInstantiate(&ex->graph_exec, ex->graph, nullptr, nullptr, 0);
        if (rc != DSX_OK || e2 != hipSuccess) {
          (void)hipGetLastError();
          if (ex->graph) { (void)hipGraphDestroy(ex->graph); ex->graph = nullptr; }
          ex->graph_exec = nullptr;
        }
      } else {
        (void)hipGetLastError();
      }
      (void)hipStreamDestroy(cs);
      if (ex->graph_exec) ex->graph_sig = sig;
      else if (rc != DSX_OK) return rc;
    }
    graph_ok = ex->graph_exec != nullptr;
    if (!graph_ok) return fail(DSX_ERR_HIP, "hipGraph capture of the sampling step failed");
    ex->last_stream = st;
  }
  int snap_i = 0;
  for (int s = 0; s < T; ++s) {
    if (graph_ok) HIP_TRY(hipGraphLaunch(ex->graph_exec, st));
    else if ((rc = enqueue_step(ex, tab, noise != nullptr, st))) return rc;
    while (snap_i < n_snap && snap_steps[snap_i] == s) {
      HIP_TRY(launch_nhwc_to_nchw(ex->x_state, snaps + (size_t)snap_i * snap_elems, ex->B, ex->x_c, ex->H,
                                  ex->W, st));
      ++snap_i;
    }
  }
  HIP_TRY(launch_nhwc_to_nchw(ex->x_state, x, ex->B, ex->x_c, ex->H, ex->W, st));
  return DSX_OK;
}

// ---- single reverse steps: one-row step tables through dsx_sample_loop, no graph (nothing to replay)
extern "C" int dsx_sr3_step(dsx_exec* ex, float noise_level, float sqrt_recip_ac, float sqrt_recipm1_ac, float coef1,
                            float coef2, float sigma, int clip_denoised, const float* cond, float* x, const float* noise,
                            uint64_t seed, void* stream) {
  dsx_step_table t{};
  t.n_steps = 1; t.tcond = &noise_level; t.a = &sqrt_recip_ac; t.b = &sqrt_recipm1_ac; t.c1 = &coef1; t.c2 = &coef2;
  t.sigma = &sigma; t.predict_eps = 1; t.clip = clip_denoised ? 1 : 0; t.per_sample = 0;
  return dsx_sample_loop(ex, &t, cond, x, noise, seed, nullptr, 0, nullptr, 0, stream);
}
extern "C" int dsx_indi_step(dsx_exec* ex, float t_cur, float c_x0, float c_xt, float noise_scale, float* x,
                             const float* noise, uint64_t seed, void* stream) {
  dsx_step_table t{};
  t.n_steps = 1; t.tcond = &t_cur; t.c1 = &c_x0; t.c2 = &c_xt; t.sigma = &noise_scale;
  t.predict_eps = 0; t.clip = 0; t.per_sample = 0;
  return dsx_sample_loop(ex, &t, nullptr, x, noise, seed, nullptr, 0, nullptr, 0, stream);
}

extern "C" int dsx_randn(float* out, int64_t n, uint64_t seed, uint64_t subseq, void* stream) {
  if (!out || n < 0) return fail(DSX_ERR_INVALID, "bad argument");
  if (n == 0) return DSX_OK;
  HIP_TRY(launch_randn(out, n, seed, subseq, (hipStream_t)stream));
  return DSX_OK;
}

// ------------------------------------------------------------------ time predictor head
extern "C" int dsx_time_predictor_set_mask(dsx_exec* ex, const float* w, const float* b) {
  if (!ex || !w || !b) return fail(DSX_ERR_INVALID, "null argument");
  const int cin = ex->m->cfg.in_channel;
  if (ex->out.C != 1) return fail(DSX_ERR_INVALID, "TimePredictor head expects out_channel == 1");
  const size_t nw = (size_t)49 * cin;
  std::vector<float> hw(nw);
  for (int ci = 0; ci < cin; ++ci)
    for (int t = 0; t < 49; ++t) hw[(size_t)t * cin + ci] = w[(size_t)ci * 49 + t];  // (1,in,7,7) -> [49][in]
  if (!ex->tp_w) HIP_TRY(hipMalloc((void**)&ex->tp_w, (nw + 64) * 4 + (size_t)ex->B * ex->H * ex->W * 4));
  ex->tp_b = ex->tp_w + nw;
  ex->tp_mask = ex->tp_w + nw + 64;
  HIP_TRY(hipMemcpy(ex->tp_w, hw.data(), nw * 4, hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(ex->tp_b, b, 4, hipMemcpyHostToDevice));
  return DSX_OK;
}

extern "C" int dsx_time_predictor_forward(dsx_exec* ex, const float* x, float* t_out, void* stream) {
  if (!ex || !x || !t_out) return fail(DSX_ERR_INVALID, "null argument");
  if (!ex->tp_w) return fail(DSX_ERR_STATE, "dsx_time_predictor_set_mask first");
  hipStream_t st = (hipStream_t)stream;
  int rc = load_inputs(ex, nullptr, x, ex->m->cfg.in_channel, 0, st);
  if (rc) return rc;
  if ((rc = run_unet(ex, false, 1, st))) return rc;
  NaiveConvArgs na{};
  na.c.src0 = ex->in_x.p; na.c.C0 = ex->x_c; na.c.C1 = 0;
  na.c.act_bf16 = ex->in_x.st; na.c.out_bf16 = 0;
  na.c.B = ex->B; na.c.Hs = ex->H; na.c.Ws = ex->W; na.c.Ho = ex->H; na.c.Wo = ex->W;
  na.c.bias = ex->tp_b; na.c.out = ex->tp_mask; na.c.out_ld = 1; na.c.Cout = 1;
  na.w = ex->tp_w; na.ks = 7; na.stride = 1; na.sigmoid_out = 1;
  HIP_TRY(launch_conv_naive(na, st));
  HIP_TRY(launch_masked_mean((const float*)ex->out.p, ex->tp_mask, ex->B, (long long)ex->H * ex->W, t_out, st));
  return DSX_OK;
}

// ------------------------------------------------------------------ tiling (host integer math)
namespace {
struct TilePlanner {
  int64_t D[3], g[3], p[3];
  int mode;
  int64_t dim_count(int d) const {  // tiling_manager.py:34-50
    if (g[d] == 1 && p[d] == 1) return D[d];
    const int64_t ex = p[d] - g[d];
    if (mode == DSX_TILING_PAD) return (D[d] + g[d] - 1) / g[d];
    const int64_t num = D[d] - ex;
    if (mode == DSX_TILING_SHIFT) return num <= 0 ? 0 : (num + g[d] - 1) / g[d];
    return num < 0 ? 0 : num / g[d];
  }
  int64_t grid_count(int d) const {  // :58-68
    int64_t n = 1;
    for (int k = d + 1; k < 3; ++k) n *= dim_count(k);
    return n;
  }
  int64_t total() const { return grid_count(0) * dim_count(0); }
  int64_t grid_start(int d, int64_t k) const {  // :121-143
    const int64_t ex = (p[d] - g[d]) / 2;
    if (g[d] == 1 && p[d] == 1) return k;
    if (mode == DSX_TILING_PAD) return k * g[d];
    if (mode == DSX_TILING_TRIM) return k * g[d] + ex;
    if (k < dim_count(d) - 1) return k * g[d] + ex;
    return D[d] - g[d] - ex;
  }
  void location(int64_t idx, int64_t loc[3]) const {  // :145-154
    for (int d = 0; d < 3; ++d) {
      const int64_t gc = grid_count(d);
      loc[d] = grid_start(d, idx / gc);
      idx %= gc;
    }
  }
};
static int make_planner(const int64_t* ds, const int64_t* gs, const int64_t* ps, int mode, TilePlanner& t) {
  if (!ds || !gs || !ps) return fail(DSX_ERR_INVALID, "null shape");
  if (mode < 0 || mode > 2) return fail(DSX_ERR_INVALID, "bad tiling mode");
  for (int d = 0; d < 3; ++d) {
    t.D[d] = ds[d]; t.g[d] = gs[d]; t.p[d] = ps[d];
    if (ds[d] < 1 || gs[d] < 1 || ps[d] < gs[d] || ((ps[d] - gs[d]) & 1))  // tiling_manager.py:21-29
      return fail(DSX_ERR_INVALID, "patch must be >= grid with even padding in dim %d", d);
  }
  t.mode = mode;
  return DSX_OK;
}
}  // namespace

extern "C" int64_t dsx_tile_plan(const int64_t data_shape[3], const int64_t grid_shape[3],
                                 const int64_t patch_shape[3], int mode, int64_t* grid_start,
                                 int64_t* patch_start, int64_t capacity) {
  TilePlanner t;
  int rc = make_planner(data_shape, grid_shape, patch_shape, mode, t);
  if (rc) return rc;
  const int64_t n = t.total();
  if (grid_start || patch_start) {
    if (capacity < n) return fail(DSX_ERR_INVALID, "capacity %lld < %lld tiles", (long long)capacity, (long long)n);
    for (int64_t i = 0; i < n; ++i) {
      int64_t loc[3];
      t.location(i, loc);
      for (int d = 0; d < 3; ++d) {
        if (grid_start) grid_start[i * 3 + d] = loc[d];
        if (patch_start) patch_start[i * 3 + d] = loc[d] - (t.p[d] - t.g[d]) / 2;
      }
    }
  }
  return n;
}

extern "C" int dsx_tile_regions(const int64_t data_shape[3], const int64_t grid_shape[3],
                                const int64_t patch_shape[3], int mode, int32_t* regions, int64_t capacity) {
  TilePlanner t;
  int rc = make_planner(data_shape, grid_shape, patch_shape, mode, t);
  if (rc) return rc;
  if (!regions) return fail(DSX_ERR_INVALID, "null regions");
  const int64_t n = t.total();
  if (capacity < n) return fail(DSX_ERR_INVALID, "capacity too small");
  for (int64_t i = 0; i < n; ++i) {
    int64_t gs[3], vgs[3], vge[3], ps[3];
    t.location(i, gs);
    for (int d = 0; d < 3; ++d) {  // tile_stitcher.py:26-56
      const int64_t ge = gs[d] + t.g[d];
      ps[d] = gs[d] - (t.p[d] - t.g[d]) / 2;
      const int64_t pe = ps[d] + t.p[d];
      vgs[d] = gs[d]; vge[d] = ge;
      if (mode == DSX_TILING_SHIFT) {
        if (ps[d] == 0) vgs[d] = 0;
        if (pe == t.D[d]) vge[d] = t.D[d];
      }
    }
    int32_t* r = regions + i * 8;
    r[0] = (int32_t)vgs[0]; r[1] = (int32_t)vgs[1]; r[2] = (int32_t)vgs[2];
    r[3] = (int32_t)(vge[1] - vgs[1]); r[4] = (int32_t)(vge[2] - vgs[2]);
    r[5] = (int32_t)(vgs[1] - ps[1]); r[6] = (int32_t)(vgs[2] - ps[2]); r[7] = 0;
  }
  return DSX_OK;
}

// ---- legacy per-call forms: the caller passes host tables, which are uploaded for the call (one small allocation,
// one synchronous copy, freed on every path).  The stall-free forms are the dsx_tileplan_* entry points below.
namespace {
struct DevTemp {           // a device buffer for the duration of one call
  void* p = nullptr;
  ~DevTemp() { if (p) (void)hipFree(p); }
  hipError_t upload(const void* host, size_t bytes) {
    hipError_t e = hipMalloc(&p, bytes);
    if (e != hipSuccess) { p = nullptr; return e; }
    return hipMemcpy(p, host, bytes, hipMemcpyHostToDevice);
  }
};
static int check_starts(const int64_t* patch_start, const int64_t* tile_ids, int64_t count, const int64_t data_shape[3],
                        const int64_t patch_shape[3], std::vector<int>& starts) {
  starts.resize((size_t)count * 3);
  for (int64_t i = 0; i < count; ++i) {
    const int64_t id = tile_ids ? tile_ids[i] : i;
    for (int d = 0; d < 3; ++d) starts[i * 3 + d] = (int)patch_start[id * 3 + d];
    if (starts[i * 3] < 0 || starts[i * 3] >= data_shape[0] || starts[i * 3 + 1] < 0 ||
        starts[i * 3 + 1] + patch_shape[1] > data_shape[1] || starts[i * 3 + 2] < 0 ||
        starts[i * 3 + 2] + patch_shape[2] > data_shape[2])
      return fail(DSX_ERR_INVALID, "tile %lld lies outside the frames", (long long)id);
  }
  return DSX_OK;
}
static int check_regions(const int32_t* regions, int64_t count, const int64_t data_shape[3], int ph, int pw) {
  for (int64_t i = 0; i < count; ++i) {
    const int32_t* r = regions + i * 8;
    if (r[0] < 0 || r[0] >= data_shape[0] || r[1] < 0 || r[1] + r[3] > data_shape[1] || r[2] < 0 ||
        r[2] + r[4] > data_shape[2] || r[5] < 0 || r[5] + r[3] > ph || r[6] < 0 || r[6] + r[4] > pw)
      return fail(DSX_ERR_INVALID, "region %lld out of bounds", (long long)i);
  }
  return DSX_OK;
}
}  // namespace

extern "C" int dsx_tiles_gather(const float* frames, const int64_t data_shape[3], const int64_t patch_shape[3],
                                const int64_t* patch_start, const int64_t* tile_ids, int64_t count,
                                float* tiles, void* stream) {
  if (!frames || !data_shape || !patch_shape || !patch_start || !tiles || count < 0)
    return fail(DSX_ERR_INVALID, "bad argument");
  if (count == 0) return DSX_OK;
  std::vector<int> starts;
  int rc = check_starts(patch_start, tile_ids, count, data_shape, patch_shape, starts);
  if (rc) return rc;
  DevTemp d;
  HIP_TRY(d.upload(starts.data(), starts.size() * 4));
  HIP_TRY(launch_tiles_gather(frames, (int)data_shape[1], (int)data_shape[2], (int)patch_shape[1],
                              (int)patch_shape[2], (const int*)d.p, TileSeq{0, 1, count}, tiles, (hipStream_t)stream));
  return DSX_OK;   // ~DevTemp: hipFree waits for the launch
}

extern "C" int dsx_stitch(const float* tiles, int64_t count, int C, int ph, int pw, const int32_t* regions,
                          float* canvas, const int64_t data_shape[3], void* stream) {
  if (!tiles || !regions || !canvas || !data_shape || count < 0 || C < 1)
    return fail(DSX_ERR_INVALID, "bad argument");
  if (count == 0) return DSX_OK;
  int rc = check_regions(regions, count, data_shape, ph, pw);
  if (rc) return rc;
  DevTemp d;
  HIP_TRY(d.upload(regions, (size_t)count * 32));
  const StitchSrc src{tiles, 0, ph, pw, nullptr, 0, 1};
  HIP_TRY(launch_stitch(src, C, (const int*)d.p, TileSeq{0, 1, count}, canvas, (int)data_shape[1], (int)data_shape[2],
                        nullptr, nullptr, 0, (hipStream_t)stream));
  return DSX_OK;
}

// stitch + RangeInvariantPsnr partial sums in one pass over the tiles (no second pass over the canvas)
extern "C" int dsx_stitch_psnr_blocks(int ph, int pw) {
  int gx = (ph * pw + 255) / 256;
  return gx > 16 ? 16 : (gx < 1 ? 1 : gx);
}
extern "C" int dsx_stitch_psnr(const float* tiles, int64_t count, int C, int ph, int pw, const int32_t* regions,
                               float* canvas, const int64_t data_shape[3], const float* gt_canvas, double* partials_dev,
                               void* stream) {
  if (!tiles || !regions || !canvas || !data_shape || !gt_canvas || !partials_dev || count < 0 || C < 1 || C > 4)
    return fail(DSX_ERR_INVALID, "bad argument (1 <= C <= 4)");
  if (count == 0) return DSX_OK;
  int rc = check_regions(regions, count, data_shape, ph, pw);
  if (rc) return rc;
  DevTemp d;
  HIP_TRY(d.upload(regions, (size_t)count * 32));
  const StitchSrc src{tiles, 0, ph, pw, nullptr, 0, 1};
  HIP_TRY(launch_stitch(src, C, (const int*)d.p, TileSeq{0, 1, count}, canvas, (int)data_shape[1], (int)data_shape[2],
                        gt_canvas, partials_dev, dsx_stitch_psnr_blocks(ph, pw), (hipStream_t)stream));
  return DSX_OK;
}

// tiles of both channels cut out of device-resident frames AND normalised in the same pass: the batch source of tiled
// prediction without the per-tile host crop + host->device copy of the reference's DataLoader(batch_size = 1)
extern "C" int dsx_tiles_gather_norm(const float* frames0, const float* frames1, const int64_t data_shape[3],
                                     const int64_t patch_shape[3], const int64_t* patch_start, const int64_t* tile_ids,
                                     int64_t count, float w0, float w1, const double norm[6], int from_norm_target,
                                     float* tiles_in, float* tiles_target, void* stream) {
  if (!frames0 || !frames1 || !data_shape || !patch_shape || !patch_start || !norm || !tiles_in || !tiles_target || count < 0)
    return fail(DSX_ERR_INVALID, "bad argument");
  if (norm[1] == 0.0 || norm[3] == 0.0 || norm[5] == 0.0) return fail(DSX_ERR_INVALID, "zero standard deviation");
  if (count == 0) return DSX_OK;
  std::vector<int> starts;
  int rc = check_starts(patch_start, tile_ids, count, data_shape, patch_shape, starts);
  if (rc) return rc;
  DevTemp d;
  HIP_TRY(d.upload(starts.data(), starts.size() * 4));
  HIP_TRY(launch_tiles_gather_norm(frames0, frames1, (int)data_shape[1], (int)data_shape[2], (int)patch_shape[1],
                                   (int)patch_shape[2], (const int*)d.p, TileSeq{0, 1, count}, w0, w1, norm,
                                   from_norm_target, tiles_in, tiles_target, (hipStream_t)stream));
  return DSX_OK;
}

// ------------------------------------------------------------------ tile plan with device-resident tables
// One handle per (data, grid, patch, mode): patch starts and valid regions of every tile are uploaded ONCE; every call
// names its tiles as the arithmetic sequence first, first + stride, ... (a rank's shard r, r + W, ... or a batch of
// it) and the kernels index the plan's tables by tile id.  No allocation, copy or synchronisation per call.
struct dsx_tileplan {
  TilePlanner t;
  int64_t total = 0;
  std::vector<int32_t> starts, regions;          // [total][3], [total][8]
  int* d_starts = nullptr;                       // one device allocation: starts, then regions
  int* d_regions = nullptr;
  struct Offsets {                               // pack layout for `world` ranks (dsx_tileplan_pack_layout)
    int world = 0;
    std::vector<int64_t> off, rank_pixels;
    long long* d_off = nullptr;
  };
  std::vector<Offsets> offs;
};

static void plan_layout(const dsx_tileplan* p, int world, dsx_tileplan::Offsets& o) {
  o.world = world;
  o.off.assign((size_t)p->total, 0);
  o.rank_pixels.assign((size_t)world, 0);
  for (int64_t id = 0; id < p->total; ++id) {    // rank q's run: its tiles q, q + world, ... back to back
    const int q = (int)(id % world);
    o.off[id] = o.rank_pixels[q];
    o.rank_pixels[q] += (int64_t)p->regions[id * 8 + 3] * p->regions[id * 8 + 4];
  }
}

extern "C" int dsx_tileplan_create(const int64_t data_shape[3], const int64_t grid_shape[3], const int64_t patch_shape[3],
                                   int mode, dsx_tileplan** out) {
  if (!out) return fail(DSX_ERR_INVALID, "null argument");
  auto p = std::make_unique<dsx_tileplan>();
  int rc = make_planner(data_shape, grid_shape, patch_shape, mode, p->t);
  if (rc) return rc;
  for (int d = 0; d < 3; ++d)
    if (data_shape[d] >= (1LL << 31)) return fail(DSX_ERR_INVALID, "data extent exceeds 32 bits");
  p->total = p->t.total();
  p->starts.resize((size_t)p->total * 3);
  p->regions.resize((size_t)p->total * 8);
  std::vector<int64_t> ps((size_t)p->total * 3);
  if (p->total) {
    if (dsx_tile_plan(data_shape, grid_shape, patch_shape, mode, nullptr, ps.data(), p->total) < 0) return DSX_ERR_INVALID;
    rc = dsx_tile_regions(data_shape, grid_shape, patch_shape, mode, p->regions.data(), p->total);
    if (rc) return rc;
    for (int64_t i = 0; i < p->total; ++i) {
      for (int d = 0; d < 3; ++d) p->starts[i * 3 + d] = (int32_t)ps[i * 3 + d];
      if (ps[i * 3] < 0 || ps[i * 3] >= data_shape[0] || ps[i * 3 + 1] < 0 || ps[i * 3 + 1] + patch_shape[1] > data_shape[1] ||
          ps[i * 3 + 2] < 0 || ps[i * 3 + 2] + patch_shape[2] > data_shape[2])
        return fail(DSX_ERR_INVALID, "tile %lld lies outside the frames (this tiling mode needs padded frames)", (long long)i);
    }
    rc = check_regions(p->regions.data(), p->total, data_shape, (int)patch_shape[1], (int)patch_shape[2]);
    if (rc) return rc;
    // stitch_predictions pastes tile after tile (tile_stitcher.py:68-80): where two valid regions overlap -- the
    // shifted last tile of a ragged extent re-covers a strip of the tile before it -- the LATER tile's pixels stay.
    // The device pastes all tiles at once, so the earlier tile's region is clipped to what survives: every canvas
    // pixel is then written exactly once, by the tile the sequential loop leaves there (and is packed only once).
    const int64_t cy = p->t.dim_count(1), cx = p->t.dim_count(2);
    for (int64_t i = 0; i < p->total; ++i) {
      int32_t* r = p->regions.data() + i * 8;
      const int64_t iy = (i / cx) % cy, ix = i % cx;
      if (cy >= 2 && iy == cy - 2) {
        const int32_t* last = p->regions.data() + (i + cx) * 8;          // same frame and column, last row of tiles
        if (last[1] < r[1] + r[3]) r[3] = std::max(0, last[1] - r[1]);
      }
      if (cx >= 2 && ix == cx - 2) {
        const int32_t* last = p->regions.data() + (i + 1) * 8;
        if (last[2] < r[2] + r[4]) r[4] = std::max(0, last[2] - r[2]);
      }
    }
  }
  *out = p.release();
  return DSX_OK;
}
extern "C" void dsx_tileplan_destroy(dsx_tileplan* p) {
  if (!p) return;
  if (p->d_starts) (void)hipFree(p->d_starts);
  for (auto& o : p->offs) if (o.d_off) (void)hipFree(o.d_off);
  delete p;
}
extern "C" int64_t dsx_tileplan_total(const dsx_tileplan* p) { return p ? p->total : 0; }
// the paste regions the plan's device kernels use (dsx_tile_regions clipped where a later tile overwrites)
extern "C" int dsx_tileplan_regions(const dsx_tileplan* p, int32_t* regions, int64_t capacity) {
  if (!p || !regions) return fail(DSX_ERR_INVALID, "null argument");
  if (capacity < p->total) return fail(DSX_ERR_INVALID, "capacity too small");
  memcpy(regions, p->regions.data(), (size_t)p->total * 32);
  return DSX_OK;
}

// host only: pixel offset of every tile inside its rank's packed run and the pixels of every rank's run
extern "C" int dsx_tileplan_pack_layout(const dsx_tileplan* p, int world, int64_t* off, int64_t* rank_pixels) {
  if (!p || world < 1) return fail(DSX_ERR_INVALID, "bad argument");
  dsx_tileplan::Offsets o;
  plan_layout(p, world, o);
  if (off) memcpy(off, o.off.data(), o.off.size() * 8);
  if (rank_pixels) memcpy(rank_pixels, o.rank_pixels.data(), o.rank_pixels.size() * 8);
  return DSX_OK;
}

static int plan_device(dsx_tileplan* p) {       // first device use: upload the tables (once)
  if (p->d_starts || p->total == 0) return DSX_OK;
  const size_t nb = (size_t)p->total * (3 + 8) * 4;
  HIP_TRY(hipMalloc((void**)&p->d_starts, nb));
  p->d_regions = p->d_starts + p->total * 3;
  hipError_t e = hipMemcpy(p->d_starts, p->starts.data(), (size_t)p->total * 12, hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(p->d_regions, p->regions.data(), (size_t)p->total * 32, hipMemcpyHostToDevice);
  if (e != hipSuccess) {
    (void)hipFree(p->d_starts); p->d_starts = nullptr; p->d_regions = nullptr;
    return fail(DSX_ERR_HIP, "upload of the tile tables failed: %s", hipGetErrorString(e));
  }
  return DSX_OK;
}
static int plan_offsets(dsx_tileplan* p, int world, const dsx_tileplan::Offsets** out) {
  for (auto& o : p->offs) if (o.world == world) { *out = &o; return DSX_OK; }
  dsx_tileplan::Offsets o;
  plan_layout(p, world, o);
  if (p->total) {
    HIP_TRY(hipMalloc((void**)&o.d_off, (size_t)p->total * 8));
    hipError_t e = hipMemcpy(o.d_off, o.off.data(), (size_t)p->total * 8, hipMemcpyHostToDevice);
    if (e != hipSuccess) { (void)hipFree(o.d_off); return fail(DSX_ERR_HIP, "upload of the pack offsets failed"); }
  }
  p->offs.push_back(std::move(o));
  *out = &p->offs.back();
  return DSX_OK;
}
static int seq_ok(const dsx_tileplan* p, int64_t first, int64_t stride, int64_t count) {
  if (first < 0 || stride < 1 || count < 0) return fail(DSX_ERR_INVALID, "bad tile sequence");
  if (count > 0 && first + (count - 1) * stride >= p->total) return fail(DSX_ERR_INVALID, "tile sequence leaves the plan (%lld tiles)", (long long)p->total);
  if (count > 65535) return fail(DSX_ERR_INVALID, "at most 65535 tiles per call");
  return DSX_OK;
}

extern "C" int dsx_tileplan_gather(dsx_tileplan* p, const float* frames, int64_t first, int64_t stride, int64_t count,
                                   float* tiles, void* stream) {
  if (!p || !frames || !tiles) return fail(DSX_ERR_INVALID, "null argument");
  int rc = seq_ok(p, first, stride, count);
  if (rc || count == 0) return rc;
  if ((rc = plan_device(p))) return rc;
  HIP_TRY(launch_tiles_gather(frames, (int)p->t.D[1], (int)p->t.D[2], (int)p->t.p[1], (int)p->t.p[2], p->d_starts,
                              TileSeq{first, stride, count}, tiles, (hipStream_t)stream));
  return DSX_OK;
}
extern "C" int dsx_tileplan_gather_norm(dsx_tileplan* p, const float* frames0, const float* frames1, int64_t first,
                                        int64_t stride, int64_t count, float w0, float w1, const double norm[6],
                                        int from_norm_target, float* tiles_in, float* tiles_target, void* stream) {
  if (!p || !frames0 || !frames1 || !norm || !tiles_in || !tiles_target) return fail(DSX_ERR_INVALID, "null argument");
  if (norm[1] == 0.0 || norm[3] == 0.0 || norm[5] == 0.0) return fail(DSX_ERR_INVALID, "zero standard deviation");
  int rc = seq_ok(p, first, stride, count);
  if (rc || count == 0) return rc;
  if ((rc = plan_device(p))) return rc;
  HIP_TRY(launch_tiles_gather_norm(frames0, frames1, (int)p->t.D[1], (int)p->t.D[2], (int)p->t.p[1], (int)p->t.p[2],
                                   p->d_starts, TileSeq{first, stride, count}, w0, w1, norm, from_norm_target, tiles_in,
                                   tiles_target, (hipStream_t)stream));
  return DSX_OK;
}
// paste whole predicted tiles (count, C, ph, pw) of the sequence; gt_canvas != NULL: also the PSNR partial sums
// (count * dsx_stitch_psnr_blocks * C * 8 doubles, as dsx_stitch_psnr)
extern "C" int dsx_tileplan_stitch(dsx_tileplan* p, const float* tiles, int C, int64_t first, int64_t stride, int64_t count,
                                   float* canvas, const float* gt_canvas, double* partials_dev, void* stream) {
  if (!p || !tiles || !canvas || C < 1) return fail(DSX_ERR_INVALID, "bad argument");
  if (gt_canvas && (!partials_dev || C > 4)) return fail(DSX_ERR_INVALID, "PSNR sums need a partials buffer and C <= 4");
  int rc = seq_ok(p, first, stride, count);
  if (rc || count == 0) return rc;
  if ((rc = plan_device(p))) return rc;
  const int ph = (int)p->t.p[1], pw = (int)p->t.p[2];
  const StitchSrc src{tiles, 0, ph, pw, nullptr, 0, 1};
  HIP_TRY(launch_stitch(src, C, p->d_regions, TileSeq{first, stride, count}, canvas, (int)p->t.D[1], (int)p->t.D[2],
                        gt_canvas, partials_dev, dsx_stitch_psnr_blocks(ph, pw), (hipStream_t)stream));
  return DSX_OK;
}
// valid regions of the sequence's predicted tiles -> this rank's packed run (`flat_rank`: the start of the run of rank
// first % world; tiles land at their final offsets, so batches of a shard pack into one buffer independently)
extern "C" int dsx_tileplan_pack(dsx_tileplan* p, const float* tiles, int C, int world, int64_t first, int64_t count,
                                 float* flat_rank, void* stream) {
  if (!p || !tiles || !flat_rank || C < 1 || world < 1) return fail(DSX_ERR_INVALID, "bad argument");
  int rc = seq_ok(p, first, world, count);
  if (rc || count == 0) return rc;
  if ((rc = plan_device(p))) return rc;
  const dsx_tileplan::Offsets* o = nullptr;
  if ((rc = plan_offsets(p, world, &o))) return rc;
  HIP_TRY(launch_tiles_pack(tiles, C, (int)p->t.p[1], (int)p->t.p[2], p->d_regions, o->d_off, TileSeq{first, world, count},
                            flat_rank, (hipStream_t)stream));
  return DSX_OK;
}
// paste ALL tiles from the gathered exchange buffer [world][rank_stride_elems] (rank q's run at q * rank_stride_elems)
extern "C" int dsx_tileplan_paste_packed(dsx_tileplan* p, const float* flat_all, int C, int world, int64_t rank_stride_elems,
                                         float* canvas, const float* gt_canvas, double* partials_dev, void* stream) {
  if (!p || !flat_all || !canvas || C < 1 || world < 1) return fail(DSX_ERR_INVALID, "bad argument");
  if (gt_canvas && (!partials_dev || C > 4)) return fail(DSX_ERR_INVALID, "PSNR sums need a partials buffer and C <= 4");
  if (p->total == 0) return DSX_OK;
  int rc = seq_ok(p, 0, 1, p->total);
  if (rc) return rc;
  if ((rc = plan_device(p))) return rc;
  const dsx_tileplan::Offsets* o = nullptr;
  if ((rc = plan_offsets(p, world, &o))) return rc;
  for (int q = 0; q < world; ++q)
    if (o->rank_pixels[q] * C > rank_stride_elems) return fail(DSX_ERR_INVALID, "rank stride smaller than rank %d's run", q);
  const int ph = (int)p->t.p[1], pw = (int)p->t.p[2];
  const StitchSrc src{flat_all, 1, ph, pw, o->d_off, rank_stride_elems, world};
  HIP_TRY(launch_stitch(src, C, p->d_regions, TileSeq{0, 1, p->total}, canvas, (int)p->t.D[1], (int)p->t.D[2], gt_canvas,
                        partials_dev, dsx_stitch_psnr_blocks(ph, pw), (hipStream_t)stream));
  return DSX_OK;
}
