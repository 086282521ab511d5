// engine.hip — host side of libasyrp_hip.so: parameter store, workspace pool, UNet schedule, the
// DDIM loops and the C ABI declared in include/asyrp.h.  No torch, no CPU compute fallback: every
// arithmetic op of the hot path is a launch of a kernel from kernels.hip.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstring>
#include <map>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/asyrp.h"
#include "kernels.h"

using namespace asyrp;

namespace {

thread_local std::string g_err;

int fail(int code, const std::string& msg) {
  g_err = msg;
  return code;
}

}  // namespace
namespace asyrp {
// the thread's last-error string, for the translation units that report through the C ABI's asyrp_last_error (bench_hooks.hip)
int set_last_error(int code, const char* msg) { return fail(code, msg ? msg : ""); }
}  // namespace asyrp
namespace {

#define HIPCHK(x)                                                                                   \
  do {                                                                                              \
    hipError_t _e = (x);                                                                            \
    if (_e != hipSuccess) return fail(ASYRP_EHIP, std::string(#x) + ": " + hipGetErrorString(_e)); \
  } while (0)
#define TRY(x)            \
  do {                    \
    int _r = (x);         \
    if (_r != 0) return _r; \
  } while (0)

struct ParamSpec {
  std::string key;
  std::vector<int64_t> shape;
  int64_t numel() const {
    int64_t n = 1;
    for (auto d : shape) n *= d;
    return n;
  }
};

// ---- device workspace pool: size-keyed free lists.  Immediate reuse of a returned buffer is safe because every launch of
// an engine goes to ONE in-order stream: the engine binds to the stream of its first compute call and, when a later call
// arrives on another stream, makes that stream wait (event) for everything enqueued on the previous one before any
// buffer is handed out again (bind_stream below).  `live_` tracks what an ABI call has taken and not yet returned, so
// an early error return cannot leak workspace: every ABI entry point ends with reclaim(). ----
struct Pool {
  std::multimap<size_t, float*> free_;
  std::unordered_map<float*, size_t> size_;
  std::unordered_map<float*, char> live_;
  size_t total_bytes = 0;
  int get(size_t nfloats, float** out) {
    if (nfloats == 0) nfloats = 1;
    nfloats = (nfloats + 63) & ~(size_t)63;
    auto it = free_.find(nfloats);
    if (it != free_.end()) {
      *out = it->second;
      free_.erase(it);
      live_[*out] = 1;
      return 0;
    }
    float* p = nullptr;
    hipError_t e = hipMalloc(&p, nfloats * sizeof(float));
    if (e != hipSuccess) return fail(ASYRP_EHIP, std::string("hipMalloc(workspace): ") + hipGetErrorString(e));
    size_[p] = nfloats;
    total_bytes += nfloats * sizeof(float);
    live_[p] = 1;
    *out = p;
    return 0;
  }
  // training forward: buffers the inference schedule would recycle are HELD instead (the backward pass reads them);
  // release_held() returns them after the backward (or when the tape is discarded)
  bool defer = false;
  std::vector<float*> held_;
  void put(float* p) {
    if (!p) return;
    auto it = size_.find(p);
    if (it == size_.end() || !live_.erase(p)) return;
    if (defer) held_.push_back(p);
    else free_.emplace(it->second, p);
  }
  void release_held() {
    for (float* p : held_) free_.emplace(size_[p], p);
    held_.clear();
  }
  // end of an ABI call: whatever an error path left outstanding goes back to the free lists
  void reclaim() {
    for (auto& kv : live_) free_.emplace(size_[kv.first], kv.first);
    live_.clear();
  }
  void destroy() {
    for (auto& kv : size_) (void)hipFree(kv.first);
    size_.clear();
    free_.clear();
    live_.clear();
    held_.clear();
    total_bytes = 0;
  }
};

struct Act {
  float* p = nullptr;
  int C = 0, H = 0, W = 0;
  // GroupNorm partial statistics written by the producing conv's epilogue ([B][st_nblk][C][2] doubles), or null
  double* st = nullptr;
  int st_nblk = 0;
  long long per_image() const { return (long long)H * W * C; }
};

struct ProfRec {
  hipEvent_t a, b;
  int variant;
  double flops, bytes;
};

// ---- what the training forward keeps for the backward pass through decoder #2 and the DeltaBlock (DDPM family) ----
struct TapeRes { std::string p; Act x0, x1; bool has_x1 = false; Act h1; float *sc1 = nullptr, *sh1 = nullptr, *mr1 = nullptr,
                 *sc2 = nullptr, *sh2 = nullptr, *mr2 = nullptr; int Cout = 0; bool shortcut = false;
                 // sub-module names (DDPM: norm1/conv1/norm2/conv2/nin_shortcut; iDDPM: in_layers.0/.2, out_layers.0/.3, skip_connection)
                 const char *n1 = ".norm1", *c1 = ".conv1", *n2 = ".norm2", *c2 = ".conv2", *sk = ".nin_shortcut";
                 int mode = 0;                                   // 2: iDDPM ResBlock(up=True): nearest x2 on both branches
                 const float* film = nullptr; int ld_film = 0; };   // iDDPM: GN(h)*(1+scale)+shift, scale = film[n][c]
struct TapeAttn { std::string p; Act x, qkv; float *sc = nullptr, *sh = nullptr, *mr = nullptr, *P = nullptr; int heads = 1; };
struct TapeUp { std::string p; int C = 0, H = 0, W = 0; };
struct TapeEntry { int type; int idx; };   // 0 ResnetBlock, 1 AttnBlock, 2 Upsample (forward order)
struct Tape {
  bool valid = false;
  int B = 0;
  int64_t id = 0;                       // generation: asyrp_train_forward stamps it, asyrp_train_backward must present it
  std::vector<TapeEntry> order;
  std::vector<TapeRes> res;
  std::vector<TapeAttn> attn;
  std::vector<TapeUp> up;
  Act out_h; float *out_sc = nullptr, *out_sh = nullptr, *out_mr = nullptr;      // norm_out input and its GroupNorm terms
  Act d_h, d_d1; float *d_sc = nullptr, *d_sh = nullptr, *d_mr = nullptr;        // DeltaBlock: bottleneck h, conv1 output
  float *d_sc0 = nullptr, *d_sh0 = nullptr, *d_mr0 = nullptr;                    // iDDPM DeltaBlock: its first GroupNorm (on h)
  bool d_temb = true;
  float c1 = 1.f;
  float* temb_act = nullptr;            // swish(temb) [B][temb_ch]
  void clear() { valid = false; order.clear(); res.clear(); attn.clear(); up.clear(); }
};

}  // namespace

struct asyrp_engine {
  asyrp_config cfg;
  int max_batch = 0, device = 0;
  std::vector<ParamSpec> specs;
  std::unordered_map<std::string, int> spec_idx;
  std::vector<std::vector<float>> host;
  std::vector<char> loaded;
  std::vector<char> dirty;    // loaded since the last finalize: only these tensors (and what is fused from them) are re-packed
  bool finalized = false;
  hipStream_t bound_stream = nullptr;   // stream of the previous compute call (workspace reuse is ordered on it)
  bool has_bound = false;
  hipEvent_t bind_ev = nullptr;
  Tape tape;                            // asyrp_train_forward -> asyrp_train_backward
  int64_t tape_gen = 0;                 // last generation handed out
  bool bwd_weights = false;             // transposed decoder weight images built (first training call)

  std::unordered_map<std::string, float*> dev;   // packed parameter -> device pointer
  struct XW { void* p = nullptr; float wscale = 1.f; int cout_pad = 0; size_t halfs = 0; size_t phase_halfs = 0; /* polyphase images */ };
  std::unordered_map<std::string, XW> xw;        // conv weight name -> f16x3 image (conv_f16x3.hip)
  int math = MATH_F16X3;                         // cfg.conv_math (kernel family)
  int np = 3;                                    // f16 family: matrix products per term; 1 for conv_math = ASYRP_MATH_F16
  size_t param_bytes = 0;
  std::unordered_map<std::string, int> tproj_off;   // ResnetBlock / DeltaBlock prefix -> column in tproj
  int tproj_total = 0;
  int temb_ch = 0, bott_ch = 0, bott_res = 0;
  std::vector<float> alphas;
  float* d_freqs = nullptr;
  int n_freqs = 0;
  float* d_t = nullptr;   // [max_batch] timesteps for the fused step / loops
  Pool pool;

  // iDDPM / ADM family: the module list of UNetModel.__init__ (models/improved_ddpm/unet.py:527-658)
  struct Layer { int type = 0; /* 0 conv3x3, 1 ResBlock, 2 AttentionBlock */ int cin = 0, cout = 0, mode = 0; /* 1 down, 2 up */
                 std::string p; };
  std::vector<std::vector<Layer>> in_blocks, out_blocks;
  std::vector<Layer> mid_block;
  int final_ch = 0;

  bool prof_on = false;
  std::vector<ProfRec> prof;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> ev_free;
};

namespace {

// ---------------------------------------------------------------------------------------------------
// parameter inventory of the reference DDPM (models/ddpm/diffusion.py:327-444), state_dict names
// ---------------------------------------------------------------------------------------------------
struct SpecBuilder {
  std::vector<ParamSpec>& v;
  void conv(const std::string& p, int cin, int cout, int k) {
    v.push_back({p + ".weight", {cout, cin, k, k}});
    v.push_back({p + ".bias", {cout}});
  }
  void lin(const std::string& p, int cin, int cout) {
    v.push_back({p + ".weight", {cout, cin}});
    v.push_back({p + ".bias", {cout}});
  }
  void norm(const std::string& p, int c) {
    v.push_back({p + ".weight", {c}});
    v.push_back({p + ".bias", {c}});
  }
  void res(const std::string& p, int cin, int cout, int temb) {
    norm(p + ".norm1", cin);
    conv(p + ".conv1", cin, cout, 3);
    lin(p + ".temb_proj", temb, cout);
    norm(p + ".norm2", cout);
    conv(p + ".conv2", cout, cout, 3);
    if (cin != cout) conv(p + ".nin_shortcut", cin, cout, 1);
  }
  void attn(const std::string& p, int c) {
    norm(p + ".norm", c);
    conv(p + ".q", c, c, 1);
    conv(p + ".k", c, c, 1);
    conv(p + ".v", c, c, 1);
    conv(p + ".proj_out", c, c, 1);
  }
};

bool has_attn(const asyrp_config& c, int res) {
  for (int i = 0; i < c.n_attn; ++i)
    if (c.attn_resolutions[i] == res) return true;
  return false;
}

std::string S(const char* fmt, int a = 0, int b = 0) {
  char buf[128];
  snprintf(buf, sizeof buf, fmt, a, b);
  return buf;
}

void build_specs_ddpm(asyrp_engine* e) {
  const asyrp_config& c = e->cfg;
  SpecBuilder sb{e->specs};
  const int ch = c.ch, temb = c.ch * 4, L = c.n_levels;
  auto in_mult = [&](int i) { return i == 0 ? 1 : c.ch_mult[i - 1]; };
  sb.lin("temb.dense.0", ch, temb);
  sb.lin("temb.dense.1", temb, temb);
  sb.conv("conv_in", c.in_channels, ch, 3);
  int res = c.resolution, block_in = ch;
  for (int i = 0; i < L; ++i) {
    block_in = ch * in_mult(i);
    const int block_out = ch * c.ch_mult[i];
    for (int j = 0; j < c.num_res_blocks; ++j) {
      sb.res(S("down.%d.block.%d", i, j), block_in, block_out, temb);
      block_in = block_out;
      if (has_attn(c, res)) sb.attn(S("down.%d.attn.%d", i, j), block_in);
    }
    if (i != L - 1) {
      sb.conv(S("down.%d.downsample.conv", i), block_in, block_in, 3);
      res /= 2;
    }
  }
  sb.res("mid.block_1", block_in, block_in, temb);
  sb.attn("mid.attn_1", block_in);
  sb.res("mid.block_2", block_in, block_in, temb);
  e->bott_ch = block_in;
  e->bott_res = res;
  e->temb_ch = temb;
  // decoder is constructed deepest level first but named up.0 .. up.L-1 (diffusion.py:399-423)
  std::vector<std::vector<ParamSpec>> ups(L);
  for (int i = L - 1; i >= 0; --i) {
    SpecBuilder ub{ups[i]};
    const int block_out = ch * c.ch_mult[i];
    int skip_in = ch * c.ch_mult[i];
    for (int j = 0; j < c.num_res_blocks + 1; ++j) {
      if (j == c.num_res_blocks) skip_in = ch * in_mult(i);
      ub.res(S("up.%d.block.%d", i, j), block_in + skip_in, block_out, temb);
      block_in = block_out;
      if (has_attn(c, res)) ub.attn(S("up.%d.attn.%d", i, j), block_in);
    }
    if (i != 0) {
      ub.conv(S("up.%d.upsample.conv", i), block_in, block_in, 3);
      res *= 2;
    }
  }
  for (int i = 0; i < L; ++i)
    for (auto& s : ups[i]) e->specs.push_back(s);
  sb.norm("norm_out", block_in);
  sb.conv("conv_out", block_in, c.out_channels, 3);
  for (int d = 0; d < c.n_delta; ++d) {   // DeltaBlock (diffusion.py:228-263)
    const std::string p = S("layer_%d", d);
    sb.conv(p + ".conv1", e->bott_ch, e->bott_ch, 1);
    sb.lin(p + ".temb_proj", temb, e->bott_ch);
    sb.norm(p + ".norm2", e->bott_ch);
    sb.conv(p + ".conv2", e->bott_ch, e->bott_ch, 1);
  }
}

// parameter inventory of the reference UNetModel (models/improved_ddpm/unet.py:469-658 == models/guided_diffusion/unet.py)
// with resblock_updown=True, use_scale_shift_norm=True (every arch dict of the reference), + DeltaBlocks (:776-853)
void build_specs_iddpm(asyrp_engine* e) {
  const asyrp_config& c = e->cfg;
  auto& v = e->specs;
  const int mc = c.ch, emb = mc * 4, L = c.n_levels;
  auto conv = [&](const std::string& p, int cin, int cout, int k) {
    v.push_back({p + ".weight", {cout, cin, k, k}});
    v.push_back({p + ".bias", {cout}});
  };
  auto conv1d = [&](const std::string& p, int cin, int cout) {
    v.push_back({p + ".weight", {cout, cin, 1}});
    v.push_back({p + ".bias", {cout}});
  };
  auto norm = [&](const std::string& p, int ch) {
    v.push_back({p + ".weight", {ch}});
    v.push_back({p + ".bias", {ch}});
  };
  auto lin = [&](const std::string& p, int cin, int cout) {
    v.push_back({p + ".weight", {cout, cin}});
    v.push_back({p + ".bias", {cout}});
  };
  auto emit = [&](const asyrp_engine::Layer& l) {
    if (l.type == 0) {
      conv(l.p, l.cin, l.cout, 3);
    } else if (l.type == 1) {
      norm(l.p + ".in_layers.0", l.cin);
      conv(l.p + ".in_layers.2", l.cin, l.cout, 3);
      lin(l.p + ".emb_layers.1", emb, 2 * l.cout);
      norm(l.p + ".out_layers.0", l.cout);
      conv(l.p + ".out_layers.3", l.cout, l.cout, 3);
      if (l.cin != l.cout) conv(l.p + ".skip_connection", l.cin, l.cout, 1);
    } else {
      norm(l.p + ".norm", l.cin);
      conv1d(l.p + ".qkv", l.cin, 3 * l.cin);
      conv1d(l.p + ".proj_out", l.cin, l.cin);
    }
  };
  auto mk = [&](int type, int cin, int cout, int mode, const std::string& p) {
    asyrp_engine::Layer l;
    l.type = type; l.cin = cin; l.cout = cout; l.mode = mode; l.p = p;
    return l;
  };
  lin("time_embed.0", mc, emb);
  lin("time_embed.2", emb, emb);
  if (c.num_classes > 0) v.push_back({"label_emb.weight", {c.num_classes, emb}});   // built by the reference, never used (:519-520,676-688)
  int ch = mc * c.ch_mult[0], res = c.resolution;
  std::vector<int> chans;
  e->in_blocks.clear(); e->out_blocks.clear(); e->mid_block.clear();
  e->in_blocks.push_back({mk(0, c.in_channels, ch, 0, "input_blocks.0.0")});
  chans.push_back(ch);
  for (int level = 0; level < L; ++level) {
    for (int j = 0; j < c.num_res_blocks; ++j) {
      const int n = (int)e->in_blocks.size();
      std::vector<asyrp_engine::Layer> ls;
      ls.push_back(mk(1, ch, mc * c.ch_mult[level], 0, S("input_blocks.%d.0", n)));
      ch = mc * c.ch_mult[level];
      if (has_attn(c, res)) ls.push_back(mk(2, ch, ch, 0, S("input_blocks.%d.1", n)));
      e->in_blocks.push_back(ls);
      chans.push_back(ch);
    }
    if (level != L - 1) {
      const int n = (int)e->in_blocks.size();
      e->in_blocks.push_back({mk(1, ch, ch, 1, S("input_blocks.%d.0", n))});
      chans.push_back(ch);
      res /= 2;
    }
  }
  e->mid_block = {mk(1, ch, ch, 0, "middle_block.0"), mk(2, ch, ch, 0, "middle_block.1"), mk(1, ch, ch, 0, "middle_block.2")};
  e->bott_ch = ch;
  e->bott_res = res;
  e->temb_ch = emb;
  for (int level = L - 1; level >= 0; --level) {
    for (int i = 0; i < c.num_res_blocks + 1; ++i) {
      const int ich = chans.back();
      chans.pop_back();
      const int n = (int)e->out_blocks.size();
      std::vector<asyrp_engine::Layer> ls;
      ls.push_back(mk(1, ch + ich, mc * c.ch_mult[level], 0, S("output_blocks.%d.0", n)));
      ch = mc * c.ch_mult[level];
      if (has_attn(c, res)) ls.push_back(mk(2, ch, ch, 0, S("output_blocks.%d.%d", n, (int)ls.size())));
      if (level && i == c.num_res_blocks) {
        ls.push_back(mk(1, ch, ch, 2, S("output_blocks.%d.%d", n, (int)ls.size())));
        res *= 2;
      }
      e->out_blocks.push_back(ls);
    }
  }
  e->final_ch = ch;
  for (auto& b : e->in_blocks) for (auto& l : b) emit(l);
  for (auto& l : e->mid_block) emit(l);
  for (auto& b : e->out_blocks) for (auto& l : b) emit(l);
  norm("out.0", ch);
  conv("out.2", ch, c.out_channels, 3);
  for (int d = 0; d < c.n_delta; ++d) {
    const std::string p = S("layer_%d", d);
    norm(p + ".in_layers.0", e->bott_ch);
    conv(p + ".in_layers.2", e->bott_ch, e->bott_ch, 1);
    lin(p + ".emb_layers.1", emb, e->bott_ch);
    norm(p + ".out_layers.0", e->bott_ch);
    conv(p + ".out_layers.3", e->bott_ch, e->bott_ch, 1);
  }
}

bool ends_with(const std::string& s, const std::string& suf) {
  return s.size() >= suf.size() && s.compare(s.size() - suf.size(), suf.size(), suf) == 0;
}

int upload(asyrp_engine* e, const std::string& name, const std::vector<float>& v) {
  float* d = nullptr;
  auto it = e->dev.find(name);
  if (it != e->dev.end()) {
    d = it->second;
  } else {
    HIPCHK(hipMalloc(&d, std::max<size_t>(v.size(), 1) * sizeof(float)));
    e->dev[name] = d;
    e->param_bytes += v.size() * sizeof(float);
  }
  HIPCHK(hipMemcpy(d, v.data(), v.size() * sizeof(float), hipMemcpyHostToDevice));
  return 0;
}

const std::vector<float>& hostp(asyrp_engine* e, const std::string& key) { return e->host[e->spec_idx.at(key)]; }

// conv weight in PyTorch layout [Cout][Cin][k][k] -> f16 hi/lo image of conv_f16x3.hip, scaled by a power of two so
// that max|w| lands in [1024, 2048) (keeps w_lo a normal f16 for weights down to 1e-4 of the largest one)
int pack_x3(asyrp_engine* e, const std::string& name, const std::vector<float>& w, int cout, int cin, int k) {
  float mx = 0.f;
  for (float v : w) mx = std::max(mx, std::fabs(v));
  float wscale = 1.f;
  if (mx > 0.f && std::isfinite(mx)) wscale = std::ldexp(1.0f, 10 - (int)std::floor(std::log2(mx)));
  asyrp_engine::XW x;
  auto it = e->xw.find(name);
  const size_t halfs = f16x3_packed_halfs(cout, cin, k);
  if (it != e->xw.end() && it->second.halfs == halfs) {
    x = it->second;
  } else {
    if (it != e->xw.end()) { (void)hipFree(it->second.p); e->param_bytes -= it->second.halfs * 2; }
    HIPCHK(hipMalloc(&x.p, halfs * 2));
    x.halfs = halfs;
    e->param_bytes += halfs * 2;
  }
  x.wscale = wscale;
  x.cout_pad = ((cout + 127) / 128) * 128;
  float* tmp = nullptr;
  HIPCHK(hipMalloc(&tmp, std::max<size_t>(w.size(), 1) * sizeof(float)));
  HIPCHK(hipMemcpy(tmp, w.data(), w.size() * sizeof(float), hipMemcpyHostToDevice));
  hipError_t le = launch_pack_f16x3(tmp, x.p, cout, cin, k, wscale, nullptr);
  hipError_t se = hipDeviceSynchronize();
  (void)hipFree(tmp);
  if (le != hipSuccess || se != hipSuccess) return fail(ASYRP_EHIP, "f16x3 weight packing failed for " + name);
  e->xw[name] = x;
  return 0;
}

// 1x1 conv weight [Cout][Cin] -> the fragment-major image of gemm1x1.hip under `name` + "#g1" (same power-of-two scale rule)
int pack_g1(asyrp_engine* e, const std::string& name, const std::vector<float>& w, int cout, int cin) {
  float mx = 0.f;
  for (float v : w) mx = std::max(mx, std::fabs(v));
  float wscale = 1.f;
  if (mx > 0.f && std::isfinite(mx)) wscale = std::ldexp(1.0f, 10 - (int)std::floor(std::log2(mx)));
  const std::string key = name + "#g1";
  asyrp_engine::XW x;
  auto it = e->xw.find(key);
  const size_t halfs = gemm1x1_packed_halfs(cout, cin);
  if (it != e->xw.end() && it->second.halfs == halfs) {
    x = it->second;
  } else {
    if (it != e->xw.end()) { (void)hipFree(it->second.p); e->param_bytes -= it->second.halfs * 2; }
    HIPCHK(hipMalloc(&x.p, halfs * 2));
    x.halfs = halfs;
    e->param_bytes += halfs * 2;
  }
  x.wscale = wscale;
  x.cout_pad = ((cout + 127) / 128) * 128;
  float* tmp = nullptr;
  HIPCHK(hipMalloc(&tmp, std::max<size_t>(w.size(), 1) * sizeof(float)));
  HIPCHK(hipMemcpy(tmp, w.data(), w.size() * sizeof(float), hipMemcpyHostToDevice));
  hipError_t le = launch_gemm1x1_pack(tmp, x.p, cout, cin, wscale, nullptr);
  hipError_t se = hipDeviceSynchronize();
  (void)hipFree(tmp);
  if (le != hipSuccess || se != hipSuccess) return fail(ASYRP_EHIP, "1x1 weight packing failed for " + name);
  e->xw[key] = x;
  return 0;
}

// 3x3 conv weight + the block's 1x1 shortcut weight -> ONE f16x3 image: the shortcut's slices follow the conv's K-steps
// (conv_f16x3.hip, fused shortcut); one common power-of-two scale.  Stored under `name` + "#sc".
int pack_x3_fused(asyrp_engine* e, const std::string& name, const std::vector<float>& w3, int cout, int cin3,
                  const std::vector<float>& w1, int cin1) {
  float mx = 0.f;
  for (float v : w3) mx = std::max(mx, std::fabs(v));
  for (float v : w1) mx = std::max(mx, std::fabs(v));
  float wscale = 1.f;
  if (mx > 0.f && std::isfinite(mx)) wscale = std::ldexp(1.0f, 10 - (int)std::floor(std::log2(mx)));
  const size_t h3 = f16x3_packed_halfs(cout, cin3, 3), h1 = f16x3_packed_halfs(cout, cin1, 1);
  const std::string key = name + "#sc";
  asyrp_engine::XW x;
  auto it = e->xw.find(key);
  if (it != e->xw.end() && it->second.halfs == h3 + h1) {
    x = it->second;
  } else {
    if (it != e->xw.end()) { (void)hipFree(it->second.p); e->param_bytes -= it->second.halfs * 2; }
    HIPCHK(hipMalloc(&x.p, (h3 + h1) * 2));
    x.halfs = h3 + h1;
    e->param_bytes += (h3 + h1) * 2;
  }
  x.wscale = wscale;
  x.cout_pad = ((cout + 127) / 128) * 128;
  float *t3 = nullptr, *t1 = nullptr;
  HIPCHK(hipMalloc(&t3, std::max<size_t>(w3.size(), 1) * sizeof(float)));
  HIPCHK(hipMalloc(&t1, std::max<size_t>(w1.size(), 1) * sizeof(float)));
  HIPCHK(hipMemcpy(t3, w3.data(), w3.size() * sizeof(float), hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(t1, w1.data(), w1.size() * sizeof(float), hipMemcpyHostToDevice));
  hipError_t le = launch_pack_f16x3(t3, x.p, cout, cin3, 3, wscale, nullptr);
  if (le == hipSuccess) le = launch_pack_f16x3(t1, reinterpret_cast<char*>(x.p) + h3 * 2, cout, cin1, 1, wscale, nullptr);
  hipError_t se = hipDeviceSynchronize();
  (void)hipFree(t3);
  (void)hipFree(t1);
  if (le != hipSuccess || se != hipSuccess) return fail(ASYRP_EHIP, "f16x3 fused weight packing failed for " + name);
  e->xw[key] = x;
  return 0;
}

// "nearest x2, then 3x3" collapses, per output phase (py, px), into a 2x2 convolution on the source grid (conv_f16x3.hip,
// K32Cfg<8, 2, 16, 1, 2>): output row 2i+py reads source rows (2i+py+ky-1)>>1 for ky = 0..2, i.e. rows {i-1: ky=0 | i: ky=1,2} for
// py = 0 and {i: ky=0,1 | i+1: ky=2} for py = 1 (same along x).  -> [4 phases][Cout][Cin][2][2], the sums of the taps that meet
// on one source pixel (added in fp32: the rounding of the sum is the one difference from evaluating the nine taps apart).
std::vector<float> polyphase_weights(const float* w, int cout, int cin) {
  std::vector<float> o((size_t)4 * cout * cin * 4, 0.f);
  for (int py = 0; py < 2; ++py)
    for (int px = 0; px < 2; ++px)
      for (size_t oc = 0; oc < (size_t)cout * cin; ++oc) {
        const float* k = w + oc * 9;
        float* d = o.data() + (((size_t)(py * 2 + px) * cout * cin) + oc) * 4;
        for (int ky = 0; ky < 3; ++ky)
          for (int kx = 0; kx < 3; ++kx) {
            const int a = ((py + ky - 1 + 2) >> 1) - 1 + (1 - py), b = ((px + kx - 1 + 2) >> 1) - 1 + (1 - px);   // tap slot 0/1
            d[a * 2 + b] += k[ky * 3 + kx];
          }
      }
  return o;
}

// the four phase images of one up-sampled 3x3 conv under `name` + "#up": one allocation, phases phase_halfs apart, one scale
int pack_x3_up(asyrp_engine* e, const std::string& name, const std::vector<float>& w, int cout, int cin) {
  const std::vector<float> wp = polyphase_weights(w.data(), cout, cin);
  float mx = 0.f;
  for (float v : wp) mx = std::max(mx, std::fabs(v));
  float wscale = 1.f;
  if (mx > 0.f && std::isfinite(mx)) wscale = std::ldexp(1.0f, 10 - (int)std::floor(std::log2(mx)));
  const size_t ph = f16x3_packed_halfs(cout, cin, 2);
  const std::string key = name + "#up";
  asyrp_engine::XW x;
  auto it = e->xw.find(key);
  if (it != e->xw.end() && it->second.halfs == 4 * ph) {
    x = it->second;
  } else {
    if (it != e->xw.end()) { (void)hipFree(it->second.p); e->param_bytes -= it->second.halfs * 2; }
    HIPCHK(hipMalloc(&x.p, 4 * ph * 2));
    x.halfs = 4 * ph;
    e->param_bytes += 4 * ph * 2;
  }
  x.phase_halfs = ph;
  x.wscale = wscale;
  x.cout_pad = ((cout + 127) / 128) * 128;
  float* tmp = nullptr;
  HIPCHK(hipMalloc(&tmp, wp.size() * sizeof(float)));
  HIPCHK(hipMemcpy(tmp, wp.data(), wp.size() * sizeof(float), hipMemcpyHostToDevice));
  hipError_t le = hipSuccess;
  for (int q = 0; q < 4 && le == hipSuccess; ++q)
    le = launch_pack_f16x3(tmp + (size_t)q * cout * cin * 4, reinterpret_cast<char*>(x.p) + (size_t)q * ph * 2, cout, cin, 2, wscale, nullptr);
  hipError_t se = hipDeviceSynchronize();
  (void)hipFree(tmp);
  if (le != hipSuccess || se != hipSuccess) return fail(ASYRP_EHIP, "polyphase weight packing failed for " + name);
  e->xw[key] = x;
  return 0;
}

// conv weight [Cout][Cin][k][k] -> GEMM B operand [k*k][Cin][Cout]
std::vector<float> pack_conv(const std::vector<float>& w, int cout, int cin, int k) {
  std::vector<float> o((size_t)k * k * cin * cout);
  for (int co = 0; co < cout; ++co)
    for (int ci = 0; ci < cin; ++ci)
      for (int t = 0; t < k * k; ++t) o[((size_t)t * cin + ci) * cout + co] = w[((size_t)co * cin + ci) * k * k + t];
  return o;
}

struct Ctx {
  asyrp_engine* e;
  hipStream_t s;
  int B;
  float* tproj = nullptr;   // [B][tproj_total], or ONE row shared by the batch (tproj_ld == 0)
  int tproj_ld = 0;         // row pitch of tproj per image: tproj_total, or 0 when every image reads the same row
  // asyrp_run_edit knows every timestep of the edit in advance and all images of a batch share it: the timestep embedding and
  // every block's Linear(swish(temb)) are computed for ALL steps in two launches before the loops (round 6); a step then points
  // here at its row and unet_core skips its own two launches.  Per row the arithmetic is the per-step launches' (one workgroup
  // per row in temb_mlp_kernel; a fixed tile and a K-only accumulation order in the projection GEMM), so the bits are the same.
  const float* tproj_pre = nullptr;
  const float* dh_in = nullptr;   // injected delta-h tensor (NHWC) -> slerp mix instead of the DeltaBlocks
  int coeff_per_image = 0;        // hs_coeff is [B][index + 2]: one tuple per image (batched strength sweeps, asyrp_run_edit)
  int use_mask = 0;
  Tape* tape = nullptr;           // non-null while the training forward runs the DeltaBlock and decoder #2
  // dual-decoder steps (DDPM family): the skip-connection half of every decoder ResnetBlock's conv1 is the same in both
  // decoder passes (conv1_shared); decoder #1 leaves it here, decoder #2 consumes it
  bool skip_share = false;
  std::unordered_map<std::string, Act> skip_part;
};

float* P(Ctx& c, const std::string& name) {
  auto it = c.e->dev.find(name);
  return it == c.e->dev.end() ? nullptr : it->second;
}

int new_act(Ctx& c, int C, int H, int W, Act* a) {
  a->C = C;
  a->H = H;
  a->W = W;
  return c.e->pool.get((size_t)c.B * H * W * C, &a->p);
}
void drop(Ctx& c, Act& a) {
  c.e->pool.put(a.p);
  if (a.st) c.e->pool.put(reinterpret_cast<float*>(a.st));
  a.p = nullptr;
  a.st = nullptr;
}

int variant_of(const GemmArgs& g) {
  if (g.math == MATH_F16X3 && g.wpk) return 100000 + gemm_resolve_tile_x(g) * 1000 + g.ks * 100 + g.stride * 10;
  return gemm_resolve_tile(g) * 1000 + g.ks * 100 + g.stride * 10 + (g.bT ? 1 : 0);
}

// Bracket one launch with HIP events on the launch stream when profiling is on (bench.py's roofline objects).
// variant ids: GEMM families as documented at asyrp_profile_read; 200000 + T = fused attention over T tokens.
template <class F>
int run_timed(Ctx& c, int variant, double flops, double bytes, F&& launch) {
  asyrp_engine* e = c.e;
  if (!e->prof_on) {
    HIPCHK(launch());
    return 0;
  }
  std::pair<hipEvent_t, hipEvent_t> ev;
  if (!e->ev_free.empty()) {
    ev = e->ev_free.back();
    e->ev_free.pop_back();
  } else {
    HIPCHK(hipEventCreate(&ev.first));
    HIPCHK(hipEventCreate(&ev.second));
  }
  ProfRec r;
  r.a = ev.first;
  r.b = ev.second;
  r.variant = variant;
  r.flops = flops;
  r.bytes = bytes;
  HIPCHK(hipEventRecord(r.a, c.s));
  HIPCHK(launch());
  HIPCHK(hipEventRecord(r.b, c.s));
  e->prof.push_back(r);
  return 0;
}

int run_gemm(Ctx& c, const GemmArgs& g) {
  double fl = 0, by = 0;
  if (c.e->prof_on) gemm_work(g, &fl, &by);
  return run_timed(c, c.e->prof_on ? variant_of(g) : 0, fl, by, [&]() { return launch_gemm(g, c.s); });
}

// A/B switch of the polyphase form of the up-sampled 3x3 convolutions (ASYRP_POLYPHASE=0 keeps the 3x3 form over the virtual
// up-sampling); like the other switches its value is recorded in bench.py's line
static bool polyphase_enabled() {
  static const bool on = [] { const char* e = ab_env("ASYRP_POLYPHASE"); return !(e && e[0] == '0'); }();
  return on;
}
// 1x1 convolutions of the 16 x 16 feature maps (the attention blocks of every 256-pixel config) on the barrier-free kernel of gemm1x1.hip;
// at 32 x 32 and above the 256 x 128 implicit-GEMM tile is faster, at 8 x 8 the 64-pixel tiles (profiles/rd3p_*) (ASYRP_GEMM1X1=0: the implicit-GEMM
// tiles, for A/B; =256: its 8-wave form); a function of the layer shape only.  Recorded in bench.py's line like the other switches.
static int gemm1x1_tile() {
  static const int t = [] {
    const char* e = ab_env("ASYRP_GEMM1X1");
    if (e && e[0] == '0') return 0;
    return (e && e[0] == '2') ? (int)XT_G1_256 : (int)XT_G1_128;
  }();
  return t;
}
static void try_gemm1x1(Ctx& c, const std::string& wname, GemmArgs& g) {
  // 16 x 16 maps, and (round 5) 8 x 8 maps on the pair form of the 4-wave kernel: two images per workgroup instead of the 64 x 64
  // implicit-GEMM tile that ran these layers (mid attention q|k|v / proj_out, the DeltaBlock's 1x1 convolutions) at 35 TFLOP/s
  const int hw = g.Hout * g.Wout;
  const bool pair8 = hw == 64 && gemm1x1_tile() == XT_G1_128 && (!g.pscale || g.Cin <= 1024);
  if (!gemm1x1_tile() || g.ks != 1 || !(hw == 256 || pair8) || g.s0 || g.sk > 1 || !gemm1x1_ok(g)) return;
  auto it = c.e->xw.find(wname + "#g1");
  if (it == c.e->xw.end()) return;
  g.tile = gemm1x1_tile();
  g.wpk = it->second.p;
  g.cout_pad = it->second.cout_pad;
  g.alpha = 1.0f / (it->second.wscale * f16x3_act_scale());
}
// A/B switch of the dedicated first-convolution kernel (ASYRP_CONV_IN=0: the implicit-GEMM tile with scalar-gather staging)
static bool conv_in_enabled() {
  static const bool on = [] { const char* e = ab_env("ASYRP_CONV_IN"); return !(e && e[0] == '0'); }();
  return on;
}
// y = conv(act(x0|x1)) + bias (+chan_add) (+resid);  act = optional per-(image,channel) affine (+SiLU)
int conv(Ctx& c, const Act& x0, const Act* x1, const std::string& wname, const std::string& bname, int Cout, int ks,
         int stride, int ups, const float* pscale, const float* pshift, int silu, const float* chan_add,
         const Act* resid, Act* out, bool want_stats = false, int rups = 0, const Act* sc0 = nullptr,
         const Act* sc1 = nullptr, bool* fused = nullptr, int ldb_full = 0) {
  const int Hin = x0.H, Win = x0.W;
  int Ho = Hin, Wo = Win;
  if (ups) { Ho *= 2; Wo *= 2; }
  if (stride == 2) { Ho /= 2; Wo /= 2; }
  TRY(new_act(c, Cout, Ho, Wo, out));
  GemmArgs g;
  memset(&g, 0, sizeof g);
  g.nz = c.e->cfg.nominal_batch;
  g.a0 = x0.p; g.c0 = x0.C; g.lda0 = x0.C; g.a0_zo = x0.per_image();
  if (x1) { g.a1 = x1->p; g.c1 = x1->C; g.lda1 = x1->C; g.a1_zo = x1->per_image(); }
  g.Hin = Hin; g.Win = Win; g.Hout = Ho; g.Wout = Wo;
  g.Cin = x0.C + (x1 ? x1->C : 0);
  g.Cout = Cout;
  g.ks = ks; g.stride = stride; g.ups = ups;
  g.pad = (ks == 3 && stride == 1) ? 1 : 0;   // stride 2: pad right/bottom only (diffusion.py:105)
  g.pscale = pscale; g.pshift = pshift; g.silu = silu;
  g.w = P(c, wname);
  if (!g.w) return fail(ASYRP_EKEY, "missing packed weight " + wname);
  g.ldb = ldb_full ? ldb_full : Cout;   // (a launch may use the leading Cout columns of a wider packed weight)
  g.bias = bname.empty() ? nullptr : P(c, bname);
  g.chan_add = chan_add; g.ld_chan_add = c.tproj_ld;
  if (resid) { g.resid = resid->p; g.ldr = resid->C; g.r_zo = resid->per_image(); g.rups = rups; }
  g.alpha = 1.0f;
  g.out = out->p; g.ldo = Cout; g.o_zo = out->per_image();
  g.ZI = 1; g.Z = c.B;
  g.math = MATH_F32;
  // the UNet's first convolution (3 input channels): an HBM-write-bound fp32 stencil (conv_in.hip) instead of a K = 32 tile of zeros
  if (c.e->math == MATH_F16X3 && x0.C == 3 && !x1 && !sc0 && !ldb_full && conv_in_enabled() && conv_in_supported(g)) {
    if (want_stats) {
      out->st_nblk = conv_in_stat_blocks(g);
      float* sp = nullptr;
      TRY(c.e->pool.get((size_t)c.B * out->st_nblk * Cout * 4, &sp));
      out->st = reinterpret_cast<double*>(sp);
      g.stats = out->st;
    }
    double fl = 0, by = 0;
    if (c.e->prof_on) gemm_work(g, &fl, &by);
    if (fused) *fused = false;
    auto xit = c.e->xw.find(wname);     // the scale pack_x3 derived from max|w| of this tensor
    g.cin_wmul = (xit != c.e->xw.end()) ? xit->second.wscale : 0.f;
    return run_timed(c, 400000 + Cout, fl, by, [&]() { return launch_conv_in(g, c.s); });
  }
  if (sc0 && c.e->math != MATH_F16X3) {   // fused shortcut exists only in the f16x3 family
    if (fused) *fused = false;
    drop(c, *out);
    return 0;
  }
  if (c.e->math == MATH_F16X3) {
    auto it = c.e->xw.find(wname);
    if (it == c.e->xw.end()) return fail(ASYRP_EKEY, "missing f16x3 weight image " + wname);
    g.math = MATH_F16X3;
    g.np = c.e->np;
    g.wpk = it->second.p;
    g.cout_pad = it->second.cout_pad;
    g.alpha = 1.0f / (it->second.wscale * f16x3_act_scale());   // both powers of two: exact
    if (fused) *fused = false;
    if (sc0) {   // try the fused 1x1 shortcut: needs the fused weight image and the main tile
      auto fit = c.e->xw.find(wname + "#sc");
      const float* fb = P(c, bname + "#sc");
      // 8 x 8 layers: the quad form (four images per workgroup, split-K) without the fusion beats the fused 64-pixel form --
      // 46 us + a 1x1 launch against 150 us -- so these blocks take the two-launch form (the shortcut enters the reduce as the residual)
      if (fit != c.e->xw.end() && fb && !resid && !splitk_unfused(g)) {
        GemmArgs t = g;
        t.s0 = sc0->p; t.sc0 = sc0->C; t.lds0 = sc0->C; t.s0_zo = sc0->per_image();
        if (sc1) { t.s1 = sc1->p; t.sc1 = sc1->C; t.lds1 = sc1->C; t.s1_zo = sc1->per_image(); }
        t.Cin2 = sc0->C + (sc1 ? sc1->C : 0);
        t.wpk = fit->second.p;
        t.cout_pad = fit->second.cout_pad;
        t.alpha = 1.0f / (fit->second.wscale * f16x3_act_scale());
        t.bias = fb;
        if (gemm_can_fuse_shortcut(t)) {
          g = t;
          *fused = true;
        }
      }
      if (!(fused && *fused)) {   // not fusable here: nothing is launched, the caller runs the two-launch form
        drop(c, *out);
        return 0;
      }
    }
    // the UNet's last conv: taps folded into N (conv_out.hip) when its image exists and the launch has that shape
    {
      auto cit = c.e->xw.find(wname + "#co");
      if (cit != c.e->xw.end() && !sc0 && !want_stats) {
        GemmArgs t = g;
        t.wpk = cit->second.p;
        t.cout_pad = cit->second.cout_pad;
        t.alpha = 1.0f / (cit->second.wscale * f16x3_act_scale());
        if (conv_out_supported(t)) {
          double fl = 0, by = 0;
          if (c.e->prof_on) gemm_work(g, &fl, &by);
          return run_timed(c, 300000 + Cout, fl, by, [&]() { return launch_conv_out(t, c.s); });
        }
      }
    }
    // nearest x2 + 3x3: the polyphase form (four 2x2-tap phases on the source grid, 4/9 of the products) when the layer has its
    // phase images and runs on the main tile; the training forward keeps the 3x3 form its backward pass is written against
    if (ups && ks == 3 && stride == 1 && !c.tape && !resid && !sc0 && polyphase_enabled()) {
      auto uit = c.e->xw.find(wname + "#up");
      if (uit != c.e->xw.end() && (g.Cin & 31) == 0 && (long long)Hin * Win >= 1024) {
        g.poly = 1;
        g.ups = 0;
        g.Hout = Hin; g.Wout = Win;                       // the M space is the source grid; `out` stays 2H x 2W
        g.wpk = uit->second.p;
        g.w_phase = (long long)uit->second.phase_halfs * 2;
        g.cout_pad = uit->second.cout_pad;
        g.alpha = 1.0f / (uit->second.wscale * f16x3_act_scale());
      }
    }
    // 8x8 layers: M x N has fewer tiles than the chip has CUs and K is thousands deep -> split K over 8 workgroups per
    // tile and reduce in a second, tiny launch.  The decision depends on the layer shape only (never on the batch), so
    // an image's result does not depend on what it is batched with.
    if (ks == 1 && !sc0) try_gemm1x1(c, wname, g);
    const int sk = splitk_factor(g);
    if (sk > 1) {
      g.sk = sk;
      g.tile = splitk_tile(g);
      TRY(c.e->pool.get((size_t)sk * c.B * Ho * Wo * Cout, &g.part));
    }
    if (want_stats) {   // the output will be group-normalised: its statistics come out of this launch's epilogue
      out->st_nblk = (sk > 1) ? splitk_stat_blocks(Ho * Wo) : gemm_mblocks(g);
      float* sp = nullptr;
      TRY(c.e->pool.get((size_t)c.B * out->st_nblk * Cout * 4, &sp));
      out->st = reinterpret_cast<double*>(sp);
      g.stats = out->st;
    }
    if (sk > 1) {
      TRY(run_gemm(c, g));
      HIPCHK(launch_splitk_reduce(g, c.s));
      c.e->pool.put(g.part);
      return 0;
    }
  }
  return run_gemm(c, g);
}

// GroupNorm(32, eps) of the virtual concat (x0|x1) -> scale/shift [B][C] (pool buffers returned to the caller).
// Per-channel partial statistics come from the producing conv's epilogue when it wrote them (Act::st); a tensor without
// them (h-space mix output, fp32-MFMA mode) gets one standalone reduction pass.
int gn(Ctx& c, const Act& x0, const Act* x1, const std::string& prefix, float eps, float** scale, float** shift,
       const float* film_scale = nullptr, const float* film_shift = nullptr, int ld_film = 0, float** mr = nullptr) {
  const int C = x0.C + (x1 ? x1->C : 0);
  const int HW = x0.H * x0.W;
  TRY(c.e->pool.get((size_t)c.B * C, scale));
  TRY(c.e->pool.get((size_t)c.B * C, shift));
  float* tmp[2] = {nullptr, nullptr};
  GnFin2Args a;
  memset(&a, 0, sizeof a);
  const Act* src[2] = {&x0, x1};
  for (int k = 0; k < 2; ++k) {
    if (!src[k]) continue;
    const double* st = src[k]->st;
    int nblk = src[k]->st_nblk;
    if (!st) {
      nblk = gn_nblk_of(HW);
      TRY(c.e->pool.get((size_t)c.B * nblk * src[k]->C * 4, &tmp[k]));
      HIPCHK(launch_gn_partial(src[k]->p, src[k]->C, src[k]->per_image(), HW, c.B, src[k]->C,
                               reinterpret_cast<double*>(tmp[k]), c.s));
      st = reinterpret_cast<double*>(tmp[k]);
    }
    if (k == 0) { a.p0 = st; a.nblk0 = nblk; a.C0 = src[k]->C; }
    else { a.p1 = st; a.nblk1 = nblk; a.C1 = src[k]->C; }
  }
  a.N = c.B; a.HW = HW;
  a.gamma = P(c, prefix + ".weight");
  a.beta = P(c, prefix + ".bias");
  if (!a.gamma || !a.beta) return fail(ASYRP_EKEY, "missing norm params " + prefix);
  a.eps = eps;
  a.film_scale = film_scale; a.film_shift = film_shift; a.ld_film = ld_film;
  a.scale = *scale; a.shift = *shift;
  if (mr) {
    TRY(c.e->pool.get((size_t)c.B * 64, mr));
    a.mr = *mr;
  }
  HIPCHK(launch_gn_finalize2(a, c.s));
  for (int k = 0; k < 2; ++k)
    if (tmp[k]) c.e->pool.put(tmp[k]);
  return 0;
}

// every block's Linear(swish(temb)) at once: [B x temb_ch] x [temb_ch x O_total] + bias on the fp32-MFMA GEMM (exact fp32 fma
// chain per output), one launch per UNet evaluation
int tproj_gemm(Ctx& c, const float* temb_act, int rows = 0, float* out = nullptr) {
  asyrp_engine* e = c.e;
  if (rows <= 0) rows = c.B;
  GemmArgs g;
  memset(&g, 0, sizeof g);
  g.a0 = temb_act; g.c0 = e->temb_ch; g.lda0 = e->temb_ch;
  g.Hin = rows; g.Win = 1; g.Hout = rows; g.Wout = 1; g.Cin = e->temb_ch; g.Cout = e->tproj_total; g.ks = 1; g.stride = 1;
  g.w = P(c, "__tproj.weight"); g.ldb = e->temb_ch; g.bT = 1;      // W is [O_total][temb_ch]
  g.bias = P(c, "__tproj.bias");
  g.alpha = 1.0f; g.out = out ? out : c.tproj; g.ldo = e->tproj_total; g.ZI = 1; g.Z = 1; g.math = MATH_F32;
  g.tile = TILE_64x64;          // fixed: the rows are the batch, so the tile must not depend on M
  if (!g.w || !g.bias) return fail(ASYRP_EKEY, "missing timestep projection pack");
  return run_gemm(c, g);
}

// timestep embedding + MLP of `rows` timesteps (t_dev[rows]) and every block's Linear(swish(temb)) for them: the two launches at the
// head of a UNet evaluation (rows = the batch), or of a whole edit (rows = its steps, asyrp_run_edit).
// DDPM: get_timestep_embedding [sin|cos] -> temb.dense.0/1 (models/ddpm/diffusion.py:42-60, 477-480); iDDPM: [cos|sin] -> time_embed.0/2
// (improved_ddpm/nn.py:103-121, unet.py:513-517)
int timestep_rows(Ctx& c, const float* t_dev, int rows, float* temb, float* temb_act, float* tproj_out) {
  asyrp_engine* e = c.e;
  const bool iddpm = e->cfg.family == ASYRP_FAMILY_IDDPM;
  const char* n0 = iddpm ? "time_embed.0" : "temb.dense.0";
  const char* n1 = iddpm ? "time_embed.2" : "temb.dense.1";
  HIPCHK(launch_temb_mlp(t_dev, e->d_freqs, e->n_freqs, iddpm ? 0 : 1, P(c, std::string(n0) + ".weight"), P(c, std::string(n0) + ".bias"),
                         P(c, std::string(n1) + ".weight"), P(c, std::string(n1) + ".bias"), e->cfg.ch, e->temb_ch, temb, temb_act, rows,
                         c.s));
  return tproj_gemm(c, temb_act, rows, tproj_out);
}

// Dual-decoder steps run the decoder twice over the SAME skip tensors (models/ddpm/diffusion.py:541-577: h + delta_h and h).
// conv1 of a decoder ResnetBlock sees swish(GroupNorm(cat(h, skip))): a skip channel's normalised value depends on h only
// through the statistics of its own group, so every skip channel outside the one group that may straddle the two sources is
// identical in both passes, and so is its contribution  sum_{c in clean skip} W[:, c] * act(x[c]).  That partial is computed
// once (a conv over the clean skip channels with the matching slice of the packed weight image and of the scale/shift rows)
// and enters both passes' conv1 -- now over (h | straddling skip channels) only -- as the residual operand.  25-33 % of
// conv1's products disappear from every second pass.  Same products, one more fp32 rounding where the partial joins
// (the single-decoder steps keep the one-launch form).  ASYRP_SKIP_SHARE=0 disables it.
struct SkipPlan { int dirty, nclean; };
static bool skip_share_enabled() {
  static const bool on = [] { const char* e = ab_env("ASYRP_SKIP_SHARE"); return !(e && e[0] == '0'); }();
  return on;
}
static bool skip_plan(const Act& x0, const Act& x1, SkipPlan* pl) {
  const int Cin = x0.C + x1.C;
  if ((Cin % 32) || (x0.C % 32)) return false;
  const int gs = Cin / 32;                                  // channels per GroupNorm(32) group
  int dirty = (x0.C % gs) ? (x0.C / gs + 1) * gs - x0.C : 0;   // skip channels of the group that straddles h | skip
  dirty = (dirty + 31) / 32 * 32;                            // whole K = 32 steps
  pl->dirty = dirty;
  pl->nclean = x1.C - dirty;
  return pl->nclean >= 32 && (pl->nclean % 32) == 0;
}
// conv1 of a decoder ResnetBlock in a dual-decoder step; *done = false (nothing launched) when the layer does not qualify
// (DDPM: conv1 + the block's timestep projection as chan_add; iDDPM/ADM: in_layers.2, no per-channel vector)
int conv1_shared(Ctx& c, const std::string& p, const std::string& wname, const std::string& bname, const Act& x0, const Act& x1,
                 const float* sc, const float* sh, int Cout, const float* chan_add, Act* h1, bool* done) {
  asyrp_engine* e = c.e;
  *done = false;
  SkipPlan pl;
  if (!skip_plan(x0, x1, &pl)) return 0;
  auto it = e->xw.find(wname);
  if (it == e->xw.end()) return 0;
  const int Cin = x0.C + x1.C, c_clean = x0.C + pl.dirty, H = x0.H, W = x0.W;
  GemmArgs b;
  memset(&b, 0, sizeof b);
  b.nz = e->cfg.nominal_batch;
  b.Hin = H; b.Win = W; b.Hout = H; b.Wout = W; b.Cout = Cout;
  b.ks = 3; b.stride = 1; b.pad = 1; b.silu = 1; b.ld_ps = Cin;
  b.w = P(c, wname); b.ldb = Cout;
  b.math = MATH_F16X3; b.np = e->np; b.cout_pad = it->second.cout_pad;
  b.alpha = 1.0f / (it->second.wscale * f16x3_act_scale());
  b.ldo = Cout; b.o_zo = (long long)H * W * Cout; b.ZI = 1; b.Z = c.B;
  // this pass: (h | straddling skip channels), + bias + timestep projection + the shared partial
  GemmArgs g = b;
  g.a0 = x0.p; g.c0 = x0.C; g.lda0 = x0.C; g.a0_zo = x0.per_image();
  if (pl.dirty) { g.a1 = x1.p; g.c1 = pl.dirty; g.lda1 = x1.C; g.a1_zo = x1.per_image(); }
  g.Cin = c_clean;
  g.pscale = sc; g.pshift = sh;
  g.wpk = it->second.p;
  // the shared partial: clean skip channels [dirty, x1.C) = concat channels [c_clean, Cin)
  GemmArgs s = b;
  s.a0 = x1.p + pl.dirty; s.c0 = pl.nclean; s.lda0 = x1.C; s.a0_zo = x1.per_image();
  s.Cin = pl.nclean;
  s.pscale = sc + c_clean; s.pshift = sh + c_clean;
  s.wpk = reinterpret_cast<const char*>(it->second.p) + (size_t)(c_clean / 16) * 9 * 4 * it->second.cout_pad * 16;
  if (splitk_factor_shared(g) > 1 || splitk_factor_shared(s) > 1) return 0;
  Act part;
  auto f = c.skip_part.find(p);
  const bool second = (f != c.skip_part.end());
  if (!second) {
    TRY(new_act(c, Cout, H, W, &part));
    s.out = part.p;
    TRY(run_gemm(c, s));
    c.skip_part[p] = part;
  } else {
    part = f->second;
  }
  TRY(new_act(c, Cout, H, W, h1));
  g.bias = P(c, bname);
  g.chan_add = chan_add; g.ld_chan_add = c.tproj_ld;
  g.resid = part.p; g.ldr = Cout; g.r_zo = part.per_image();
  g.out = h1->p;
  h1->st_nblk = gemm_mblocks(g);
  float* sp = nullptr;
  TRY(e->pool.get((size_t)c.B * h1->st_nblk * Cout * 4, &sp));
  h1->st = reinterpret_cast<double*>(sp);
  g.stats = h1->st;
  TRY(run_gemm(c, g));
  if (second) {
    drop(c, part);
    c.skip_part.erase(p);
  }
  *done = true;
  return 0;
}

// ResnetBlock (models/ddpm/diffusion.py:151-170) on the virtual concat (x0|x1)
int resblock(Ctx& c, const std::string& p, const Act& x0, const Act* x1, Act* out) {
  asyrp_engine* e = c.e;
  const int Cin = x0.C + (x1 ? x1->C : 0);
  const int Cout = (int)e->specs[e->spec_idx.at(p + ".conv1.bias")].shape[0];
  float *sc1, *sh1, *sc2, *sh2, *mr1 = nullptr, *mr2 = nullptr;
  TRY(gn(c, x0, x1, p + ".norm1", 1e-6f, &sc1, &sh1, nullptr, nullptr, 0, c.tape ? &mr1 : nullptr));
  Act h1;
  bool shared = false;
  if (c.skip_share && x1 && !c.tape && e->math == MATH_F16X3)
    TRY(conv1_shared(c, p, p + ".conv1.weight", p + ".conv1.bias", x0, *x1, sc1, sh1, Cout, c.tproj + e->tproj_off.at(p), &h1,
                     &shared));
  if (!shared)
    TRY(conv(c, x0, x1, p + ".conv1.weight", p + ".conv1.bias", Cout, 3, 1, 0, sc1, sh1, 1,
             c.tproj + e->tproj_off.at(p), nullptr, &h1, true));
  e->pool.put(sc1); e->pool.put(sh1);
  TRY(gn(c, h1, nullptr, p + ".norm2", 1e-6f, &sc2, &sh2, nullptr, nullptr, 0, c.tape ? &mr2 : nullptr));
  if (c.tape) {   // (the pool defers every put while a tape is recorded: the pointers below stay valid until the backward)
    TapeRes t;
    t.p = p; t.x0 = x0; t.has_x1 = (x1 != nullptr);
    if (x1) t.x1 = *x1;
    t.h1 = h1; t.sc1 = sc1; t.sh1 = sh1; t.mr1 = mr1; t.sc2 = sc2; t.sh2 = sh2; t.mr2 = mr2;
    t.Cout = Cout; t.shortcut = (Cin != Cout);
    c.tape->order.push_back({0, (int)c.tape->res.size()});
    c.tape->res.push_back(t);
    e->pool.put(mr1); e->pool.put(mr2);
  }
  if (Cin != Cout) {
    // x + h with x = nin_shortcut(x): the 1x1 rides in conv2's K-loop when the launch runs on the fusing tile
    bool fused = false;
    TRY(conv(c, h1, nullptr, p + ".conv2.weight", p + ".conv2.bias", Cout, 3, 1, 0, sc2, sh2, 1, nullptr, nullptr, out, true,
             0, &x0, x1, &fused));
    if (!fused) {
      Act sc;
      TRY(conv(c, x0, x1, p + ".nin_shortcut.weight", p + ".nin_shortcut.bias", Cout, 1, 1, 0, nullptr, nullptr, 0, nullptr,
               nullptr, &sc));
      TRY(conv(c, h1, nullptr, p + ".conv2.weight", p + ".conv2.bias", Cout, 3, 1, 0, sc2, sh2, 1, nullptr, &sc, out, true));
      drop(c, sc);
    }
  } else {
    TRY(conv(c, h1, nullptr, p + ".conv2.weight", p + ".conv2.bias", Cout, 3, 1, 0, sc2, sh2, 1, nullptr, &x0, out, true));
  }
  e->pool.put(sc2); e->pool.put(sh2);
  drop(c, h1);
  return 0;
}

// softmax(Q K^T * scale) V over T tokens; qkv is [B][T][3C] (q|k|v, or per-head [q,k,v] blocks)
int attention_core(Ctx& c, const float* qkv, int C, int T, int heads, float scale, float* out /*[B][T][C]*/,
                   int force_unfused = 0, float** keep_P = nullptr) {
  const int Dh = C / heads;
  // fused kernel (no T x T tensor in HBM) for the f16x3 engine; the fp32-MFMA engine and shapes it does not cover
  // (T > 1024, heads wider than 512) keep the three-launch form below
  if (c.e->math == MATH_F16X3 && !force_unfused && attn_fused_supported(T, Dh, 3 * C, C)) {
    AttnArgs a;
    memset(&a, 0, sizeof a);
    a.qkv = qkv; a.ld = 3 * C; a.img_stride = (long long)T * 3 * C;
    a.head_stride = (heads == 1) ? 0 : 3 * Dh;
    a.q_off = 0; a.k_off = (heads == 1) ? C : Dh; a.v_off = (heads == 1) ? 2 * C : 2 * Dh;
    a.B = c.B; a.heads = heads; a.T = T; a.Dh = Dh; a.scale = scale;
    a.out = out; a.ldo = C; a.o_img_stride = (long long)T * C; a.o_head_stride = Dh;
    a.np = c.e->np;
    const double fl = 4.0 * T * (double)T * C * c.B, by = 4.0 * 4.0 * T * (double)C * c.B;
    return run_timed(c, 200000 + T, fl, by, [&]() { return launch_attention_fused(a, c.s); });
  }
  float* Sbuf = nullptr;
  TRY(c.e->pool.get((size_t)c.B * heads * T * T, &Sbuf));
  const long long img = (long long)T * 3 * C;
  const int q_off = 0, k_off = (heads == 1) ? C : Dh, v_off = (heads == 1) ? 2 * C : 2 * Dh;
  const long long head_stride = (heads == 1) ? 0 : 3 * Dh;
  GemmArgs g;
  memset(&g, 0, sizeof g);
  g.a0 = qkv + q_off; g.c0 = Dh; g.lda0 = 3 * C; g.a0_zo = img; g.a0_zi = head_stride;
  g.Hin = T; g.Win = 1; g.Hout = T; g.Wout = 1;
  g.Cin = Dh; g.Cout = T; g.ks = 1; g.stride = 1;
  g.w = qkv + k_off; g.ldb = 3 * C; g.bT = 1; g.w_zo = img; g.w_zi = head_stride;
  g.alpha = scale;
  g.out = Sbuf; g.ldo = T; g.o_zo = (long long)heads * T * T; g.o_zi = (long long)T * T;
  g.ZI = heads; g.Z = c.B * heads;
  TRY(run_gemm(c, g));
  HIPCHK(launch_softmax_rows(Sbuf, (long long)c.B * heads * T, T, c.s));
  memset(&g, 0, sizeof g);
  g.a0 = Sbuf; g.c0 = T; g.lda0 = T; g.a0_zo = (long long)heads * T * T; g.a0_zi = (long long)T * T;
  g.Hin = T; g.Win = 1; g.Hout = T; g.Wout = 1;
  g.Cin = T; g.Cout = Dh; g.ks = 1; g.stride = 1;
  g.w = qkv + v_off; g.ldb = 3 * C; g.bT = 0; g.w_zo = img; g.w_zi = head_stride;
  g.alpha = 1.0f;
  g.out = out; g.ldo = C; g.o_zo = (long long)T * C; g.o_zi = Dh;
  g.ZI = heads; g.Z = c.B * heads;
  TRY(run_gemm(c, g));
  if (keep_P) *keep_P = Sbuf;      // softmax probabilities [B][heads][T][T], read by the attention backward
  c.e->pool.put(Sbuf);
  return 0;
}

// ---- split-plane attention path (attention.hip, attn_planes_kernel): the q|k|v projection writes f16 hi/lo planes (v transposed)
// from its epilogue and the attention kernel multiplies them without any conversion or gather work.  ASYRP_ATTN=old keeps the
// round-2 kernel (fp32 q|k|v, operands split in registers) for A/B; recorded in bench.py's line like the other switches.
struct QkvPlanes { _Float16 *h = nullptr, *l = nullptr, *vth = nullptr, *vtl = nullptr; float* raw[4] = {nullptr, nullptr, nullptr, nullptr}; int ld16 = 0; };
static bool attn_planes_enabled() {
  static const bool on = [] { const char* e = ab_env("ASYRP_ATTN"); return !(e && e[0] == 'o'); }();
  return on;
}
void drop_planes(Ctx& c, QkvPlanes& pl) {
  for (float*& r : pl.raw) { if (r) c.e->pool.put(r); r = nullptr; }
}
// q|k|v = conv1x1(GN(x)) [B][T][3C] as planes; heads == 1: channel blocks q|k|v (DDPM AttnBlock), else the legacy per-head
// [q|k|v] order (QKVAttentionLegacy, improved_ddpm/unet.py:389)
int qkv_planes_conv(Ctx& c, const Act& x, const std::string& wname, const std::string& bname, int heads, const float* sc,
                    const float* sh, QkvPlanes* pl) {
  asyrp_engine* e = c.e;
  const int C = x.C, T = x.H * x.W, Dh = C / heads;
  auto it = e->xw.find(wname);
  if (it == e->xw.end()) return fail(ASYRP_EKEY, "missing f16x3 weight image " + wname);
  const size_t nqk = ((size_t)c.B * T * 3 * C + 1) / 2, nv = ((size_t)c.B * C * T + 1) / 2;     // halfs -> floats
  const bool lo = (e->np == 3);
  TRY(e->pool.get(nqk, &pl->raw[0]));
  TRY(e->pool.get(nv, &pl->raw[2]));
  if (lo) {
    TRY(e->pool.get(nqk, &pl->raw[1]));
    TRY(e->pool.get(nv, &pl->raw[3]));
  }
  pl->h = reinterpret_cast<_Float16*>(pl->raw[0]); pl->l = reinterpret_cast<_Float16*>(pl->raw[1]);
  pl->vth = reinterpret_cast<_Float16*>(pl->raw[2]); pl->vtl = reinterpret_cast<_Float16*>(pl->raw[3]);
  pl->ld16 = 3 * C;
  GemmArgs g;
  memset(&g, 0, sizeof g);
  g.nz = e->cfg.nominal_batch;
  g.a0 = x.p; g.c0 = C; g.lda0 = C; g.a0_zo = x.per_image();
  g.Hin = x.H; g.Win = x.W; g.Hout = x.H; g.Wout = x.W; g.Cin = C; g.Cout = 3 * C; g.ks = 1; g.stride = 1;
  g.pscale = sc; g.pshift = sh; g.silu = 0;
  g.w = P(c, wname); g.ldb = 3 * C; g.bias = P(c, bname);
  g.ZI = 1; g.Z = c.B; g.ldo = 3 * C;
  g.math = MATH_F16X3; g.np = e->np; g.wpk = it->second.p; g.cout_pad = it->second.cout_pad;
  g.alpha = 1.0f / (it->second.wscale * f16x3_act_scale());
  g.o16h = pl->h; g.o16l = pl->l; g.vth = pl->vth; g.vtl = pl->vtl; g.ld16 = 3 * C;
  g.o16_zo = (long long)T * 3 * C; g.vt_zo = (long long)C * T;
  if (heads == 1) { g.v_mod = 3 * C; g.v_off = 2 * C; g.v_dh = C; }
  else { g.v_mod = 3 * Dh; g.v_off = 2 * Dh; g.v_dh = Dh; }
  if (!g.bias) return fail(ASYRP_EKEY, "missing bias " + bname);
  try_gemm1x1(c, wname, g);
  return run_gemm(c, g);
}
int attention_planes(Ctx& c, const QkvPlanes& pl, int C, int T, int heads, float scale, float* out /*[B][T][C]*/) {
  const int Dh = C / heads;
  AttnArgs a;
  memset(&a, 0, sizeof a);
  a.qkh = pl.h; a.qkl = pl.l; a.vth = pl.vth; a.vtl = pl.vtl; a.ld16 = pl.ld16;
  a.head_stride = (heads == 1) ? 0 : 3 * Dh;
  a.q_off = 0; a.k_off = (heads == 1) ? C : Dh;
  a.B = c.B; a.heads = heads; a.T = T; a.Dh = Dh; a.scale = scale;
  a.out = out; a.ldo = C; a.o_img_stride = (long long)T * C; a.o_head_stride = Dh;
  a.np = c.e->np;
  const double fl = 4.0 * T * (double)T * C * c.B, by = 4.0 * 4.0 * T * (double)C * c.B;
  return run_timed(c, 210000 + T, fl, by, [&]() { return launch_attention_planes(a, c.s); });   // 21xxxx: the split-plane kernel
}
static bool use_attn_planes(const Ctx& c, int T, int Dh) {
  return c.e->math == MATH_F16X3 && !c.tape && attn_planes_enabled() && attn_planes_supported(T, Dh);
}

// AttnBlock (models/ddpm/diffusion.py:200-225): GN -> fused q|k|v 1x1 -> attention -> proj_out -> + x
int attnblock(Ctx& c, const std::string& p, const Act& x, Act* out) {
  float *sc, *sh, *mr = nullptr, *Pk = nullptr;
  TRY(gn(c, x, nullptr, p + ".norm", 1e-6f, &sc, &sh, nullptr, nullptr, 0, c.tape ? &mr : nullptr));
  if (use_attn_planes(c, x.H * x.W, x.C)) {
    QkvPlanes pl;
    TRY(qkv_planes_conv(c, x, p + ".qkv.weight", p + ".qkv.bias", 1, sc, sh, &pl));
    c.e->pool.put(sc); c.e->pool.put(sh);
    Act o;
    TRY(new_act(c, x.C, x.H, x.W, &o));
    TRY(attention_planes(c, pl, x.C, x.H * x.W, 1, 1.0f / std::sqrt((float)x.C), o.p));
    drop_planes(c, pl);
    TRY(conv(c, o, nullptr, p + ".proj_out.weight", p + ".proj_out.bias", x.C, 1, 1, 0, nullptr, nullptr, 0, nullptr, &x, out, true));
    drop(c, o);
    return 0;
  }
  Act qkv;
  TRY(conv(c, x, nullptr, p + ".qkv.weight", p + ".qkv.bias", 3 * x.C, 1, 1, 0, sc, sh, 0, nullptr, nullptr, &qkv));
  c.e->pool.put(sc); c.e->pool.put(sh);
  Act o;
  TRY(new_act(c, x.C, x.H, x.W, &o));
  // training: the three-launch form keeps the softmax probabilities for the backward pass
  TRY(attention_core(c, qkv.p, x.C, x.H * x.W, 1, 1.0f / std::sqrt((float)x.C), o.p, c.tape ? 1 : 0, c.tape ? &Pk : nullptr));
  if (c.tape) {
    TapeAttn t;
    t.p = p; t.x = x; t.qkv = qkv; t.sc = sc; t.sh = sh; t.mr = mr; t.P = Pk;
    c.tape->order.push_back({1, (int)c.tape->attn.size()});
    c.tape->attn.push_back(t);
    c.e->pool.put(mr);
  }
  drop(c, qkv);
  TRY(conv(c, o, nullptr, p + ".proj_out.weight", p + ".proj_out.bias", x.C, 1, 1, 0, nullptr, nullptr, 0, nullptr,
           &x, out, true));
  drop(c, o);
  return 0;
}

// DeltaBlock (models/ddpm/diffusion.py:250-263)
int deltablock(Ctx& c, const std::string& p, const Act& h, bool use_temb, Act* out) {
  Act d1;
  TRY(conv(c, h, nullptr, p + ".conv1.weight", p + ".conv1.bias", h.C, 1, 1, 0, nullptr, nullptr, 0,
           use_temb ? c.tproj + c.e->tproj_off.at(p) : nullptr, nullptr, &d1, true));
  float *sc, *sh, *mr = nullptr;
  TRY(gn(c, d1, nullptr, p + ".norm2", 1e-6f, &sc, &sh, nullptr, nullptr, 0, c.tape ? &mr : nullptr));
  if (c.tape) {
    c.tape->d_h = h; c.tape->d_d1 = d1; c.tape->d_sc = sc; c.tape->d_sh = sh; c.tape->d_mr = mr; c.tape->d_temb = use_temb;
    c.e->pool.put(mr);
  }
  TRY(conv(c, d1, nullptr, p + ".conv2.weight", p + ".conv2.bias", h.C, 1, 1, 0, sc, sh, 1, nullptr, nullptr, out));
  c.e->pool.put(sc); c.e->pool.put(sh);
  drop(c, d1);
  return 0;
}

int decoder(Ctx& c, const Act& hin, const std::vector<Act>& skips, Act* eps) {
  const asyrp_config& cf = c.e->cfg;
  const int L = cf.n_levels;
  int res = cf.resolution >> (L - 1);
  int k = (int)skips.size() - 1;
  Act h = hin;
  bool own = false;
  auto replace = [&](Act& nh) {
    if (own) drop(c, h);
    h = nh;
    own = true;
  };
  for (int i = L - 1; i >= 0; --i) {
    for (int j = 0; j < cf.num_res_blocks + 1; ++j) {
      Act o;
      TRY(resblock(c, S("up.%d.block.%d", i, j), h, &skips[k], &o));
      --k;
      replace(o);
      if (has_attn(cf, res)) {
        Act o2;
        TRY(attnblock(c, S("up.%d.attn.%d", i, j), h, &o2));
        replace(o2);
      }
    }
    if (i != 0) {
      Act o;
      TRY(conv(c, h, nullptr, S("up.%d.upsample.conv.weight", i), S("up.%d.upsample.conv.bias", i), h.C, 3, 1, 1,
               nullptr, nullptr, 0, nullptr, nullptr, &o, true));
      if (c.tape) {
        TapeUp t;
        t.p = S("up.%d.upsample.conv", i); t.C = h.C; t.H = h.H; t.W = h.W;
        c.tape->order.push_back({2, (int)c.tape->up.size()});
        c.tape->up.push_back(t);
      }
      replace(o);
      res *= 2;
    }
  }
  float *sc, *sh, *mr = nullptr;
  TRY(gn(c, h, nullptr, "norm_out", 1e-6f, &sc, &sh, nullptr, nullptr, 0, c.tape ? &mr : nullptr));
  TRY(conv(c, h, nullptr, "conv_out.weight", "conv_out.bias", cf.out_channels, 3, 1, 0, sc, sh, 1, nullptr, nullptr,
           eps));
  if (c.tape) {
    c.tape->out_h = h; c.tape->out_sc = sc; c.tape->out_sh = sh; c.tape->out_mr = mr;
    c.e->pool.put(mr);
  }
  c.e->pool.put(sc); c.e->pool.put(sh);
  if (own) drop(c, h);
  return 0;
}

// DDPM.forward (models/ddpm/diffusion.py:473-580) on NHWC buffers.  Outputs are pool buffers (NHWC) the
// caller drops.  et_mod.p == nullptr when no second decoder ran (index<0, or index>=0 without edit: ε̃ ≡ ε).
// ---------------------------------------------------------------------------------------------------
// iDDPM / ADM family (models/improved_ddpm/unet.py; models/guided_diffusion/unet.py is the same network)
// ---------------------------------------------------------------------------------------------------
constexpr float EPS_I = 1e-5f;   // GroupNorm32 default eps (improved_ddpm/nn.py:17-19, 100)

// ResBlock._forward (:278-298), use_scale_shift_norm=True: GN-SiLU-[up|down]-conv3x3, FiLM(GN(h))-SiLU-conv3x3, + skip(x)
int resblock_i(Ctx& c, const asyrp_engine::Layer& L, const Act& x0, const Act* x1, Act* out) {
  asyrp_engine* e = c.e;
  const std::string& p = L.p;
  const int Cin = x0.C + (x1 ? x1->C : 0), Cout = L.cout;
  if (Cin != L.cin) return fail(ASYRP_EINVAL, "channel mismatch entering " + p);
  float *sc1, *sh1, *sc2, *sh2, *mr1 = nullptr, *mr2 = nullptr;
  TRY(gn(c, x0, x1, p + ".in_layers.0", EPS_I, &sc1, &sh1, nullptr, nullptr, 0, c.tape ? &mr1 : nullptr));
  Act h1, xr;          // xr: the (pooled) input of the skip path when the block resamples
  bool own_xr = false;
  int rups = 0;
  const Act* skip_src = &x0;
  if (L.mode == 1) {   // down: 2x2 average of the ACTIVATED tensor and of x (:279-284, Downsample with use_conv=False)
    if (x1) return fail(ASYRP_EINVAL, "down-sampling ResBlock takes a single input");
    Act hp;
    TRY(new_act(c, x0.C, x0.H / 2, x0.W / 2, &hp));
    TRY(new_act(c, x0.C, x0.H / 2, x0.W / 2, &xr));
    own_xr = true;
    HIPCHK(launch_pool2(x0.p, c.B, x0.H, x0.W, x0.C, sc1, sh1, hp.p, xr.p, c.s));
    TRY(conv(c, hp, nullptr, p + ".in_layers.2.weight", p + ".in_layers.2.bias", Cout, 3, 1, 0, nullptr, nullptr, 0, nullptr,
             nullptr, &h1, true));
    drop(c, hp);
    skip_src = &xr;
  } else if (L.mode == 2) {   // up: nearest x2 of the activated tensor (folded into the conv's gather) and of x (epilogue)
    if (x1) return fail(ASYRP_EINVAL, "up-sampling ResBlock takes a single input");
    TRY(conv(c, x0, nullptr, p + ".in_layers.2.weight", p + ".in_layers.2.bias", Cout, 3, 1, 1, sc1, sh1, 1, nullptr, nullptr,
             &h1, true));
    rups = 1;
  } else {
    bool shared = false;   // dual-decoder steps: the skip half of the decoder blocks' first conv is shared (conv1_shared)
    if (c.skip_share && x1 && !c.tape && e->math == MATH_F16X3)
      TRY(conv1_shared(c, p, p + ".in_layers.2.weight", p + ".in_layers.2.bias", x0, *x1, sc1, sh1, Cout, nullptr, &h1, &shared));
    if (!shared)
      TRY(conv(c, x0, x1, p + ".in_layers.2.weight", p + ".in_layers.2.bias", Cout, 3, 1, 0, sc1, sh1, 1, nullptr, nullptr, &h1,
               true));
  }
  e->pool.put(sc1); e->pool.put(sh1);
  // h = GN(h) * (1 + scale) + shift, (scale, shift) = chunk(Linear(SiLU(emb)), 2)  (:290-294)
  const float* film = c.tproj + e->tproj_off.at(p);
  TRY(gn(c, h1, nullptr, p + ".out_layers.0", EPS_I, &sc2, &sh2, film, film + Cout, c.tproj_ld, c.tape ? &mr2 : nullptr));
  if (c.tape) {
    if (L.mode == 1) return fail(ASYRP_EINVAL, "down-sampling ResBlock inside the decoder");
    TapeRes t;
    t.p = p; t.x0 = x0; t.has_x1 = (x1 != nullptr);
    if (x1) t.x1 = *x1;
    t.h1 = h1; t.sc1 = sc1; t.sh1 = sh1; t.mr1 = mr1; t.sc2 = sc2; t.sh2 = sh2; t.mr2 = mr2;
    t.Cout = Cout; t.shortcut = (Cin != Cout); t.mode = L.mode;
    t.n1 = ".in_layers.0"; t.c1 = ".in_layers.2"; t.n2 = ".out_layers.0"; t.c2 = ".out_layers.3"; t.sk = ".skip_connection";
    t.film = film; t.ld_film = c.tproj_ld;
    c.tape->order.push_back({0, (int)c.tape->res.size()});
    c.tape->res.push_back(t);
    e->pool.put(mr1); e->pool.put(mr2);
  }
  if (Cin != Cout) {   // 1x1 skip_connection (never combined with up/down in the reference's arch dicts)
    if (L.mode) return fail(ASYRP_EINVAL, "resampling ResBlock with a channel change is not part of any reference config");
    bool fused = false;
    TRY(conv(c, h1, nullptr, p + ".out_layers.3.weight", p + ".out_layers.3.bias", Cout, 3, 1, 0, sc2, sh2, 1, nullptr, nullptr,
             out, true, 0, &x0, x1, &fused));
    if (!fused) {
      Act sk;
      TRY(conv(c, x0, x1, p + ".skip_connection.weight", p + ".skip_connection.bias", Cout, 1, 1, 0, nullptr, nullptr, 0,
               nullptr, nullptr, &sk));
      TRY(conv(c, h1, nullptr, p + ".out_layers.3.weight", p + ".out_layers.3.bias", Cout, 3, 1, 0, sc2, sh2, 1, nullptr, &sk,
               out, true));
      drop(c, sk);
    }
  } else {
    TRY(conv(c, h1, nullptr, p + ".out_layers.3.weight", p + ".out_layers.3.bias", Cout, 3, 1, 0, sc2, sh2, 1, nullptr, skip_src,
             out, true, rups));
  }
  e->pool.put(sc2); e->pool.put(sh2);
  drop(c, h1);
  if (own_xr) drop(c, xr);
  return 0;
}

// AttentionBlock + QKVAttentionLegacy (:341-347, 379-396): GN -> Conv1d(C,3C) -> per head [q|k|v] -> softmax(q.k/sqrt(ch)) v
// -> Conv1d(C,C) -> + x
int attnblock_i(Ctx& c, const asyrp_engine::Layer& L, const Act& x, Act* out) {
  const std::string& p = L.p;
  float *sc, *sh, *mr = nullptr, *Pk = nullptr;
  TRY(gn(c, x, nullptr, p + ".norm", EPS_I, &sc, &sh, nullptr, nullptr, 0, c.tape ? &mr : nullptr));
  const int nhc = c.e->cfg.num_head_channels;
  if (nhc <= 0 || x.C % nhc) return fail(ASYRP_EINVAL, "num_head_channels must divide the attention width");
  const int heads = x.C / nhc;
  if (use_attn_planes(c, x.H * x.W, nhc)) {
    QkvPlanes pl;
    TRY(qkv_planes_conv(c, x, p + ".qkv.weight", p + ".qkv.bias", heads, sc, sh, &pl));
    c.e->pool.put(sc); c.e->pool.put(sh);
    Act o;
    TRY(new_act(c, x.C, x.H, x.W, &o));
    TRY(attention_planes(c, pl, x.C, x.H * x.W, heads, 1.0f / std::sqrt((float)nhc), o.p));
    drop_planes(c, pl);
    TRY(conv(c, o, nullptr, p + ".proj_out.weight", p + ".proj_out.bias", x.C, 1, 1, 0, nullptr, nullptr, 0, nullptr, &x, out, true));
    drop(c, o);
    return 0;
  }
  Act qkv;
  TRY(conv(c, x, nullptr, p + ".qkv.weight", p + ".qkv.bias", 3 * x.C, 1, 1, 0, sc, sh, 0, nullptr, nullptr, &qkv));
  c.e->pool.put(sc); c.e->pool.put(sh);
  Act o;
  TRY(new_act(c, x.C, x.H, x.W, &o));
  TRY(attention_core(c, qkv.p, x.C, x.H * x.W, heads, 1.0f / std::sqrt((float)nhc), o.p, c.tape ? 1 : 0, c.tape ? &Pk : nullptr));
  if (c.tape) {
    TapeAttn t;
    t.p = p; t.x = x; t.qkv = qkv; t.sc = sc; t.sh = sh; t.mr = mr; t.P = Pk; t.heads = heads;
    c.tape->order.push_back({1, (int)c.tape->attn.size()});
    c.tape->attn.push_back(t);
    c.e->pool.put(mr);
  }
  drop(c, qkv);
  TRY(conv(c, o, nullptr, p + ".proj_out.weight", p + ".proj_out.bias", x.C, 1, 1, 0, nullptr, nullptr, 0, nullptr, &x, out,
           true));
  drop(c, o);
  return 0;
}

// DeltaBlock.forward (:835-853; use_scale_shift_norm=False): GN-SiLU-conv1x1, (+Linear(SiLU(emb))), GN-SiLU-conv1x1
int deltablock_i(Ctx& c, const std::string& p, const Act& h, bool use_emb, Act* out) {
  float *sc, *sh, *mr0 = nullptr, *mr = nullptr;
  TRY(gn(c, h, nullptr, p + ".in_layers.0", EPS_I, &sc, &sh, nullptr, nullptr, 0, c.tape ? &mr0 : nullptr));
  Act d1;
  TRY(conv(c, h, nullptr, p + ".in_layers.2.weight", p + ".in_layers.2.bias", h.C, 1, 1, 0, sc, sh, 1,
           use_emb ? c.tproj + c.e->tproj_off.at(p) : nullptr, nullptr, &d1, true));
  if (c.tape) { c.tape->d_sc0 = sc; c.tape->d_sh0 = sh; c.tape->d_mr0 = mr0; c.e->pool.put(mr0); }
  c.e->pool.put(sc); c.e->pool.put(sh);
  TRY(gn(c, d1, nullptr, p + ".out_layers.0", EPS_I, &sc, &sh, nullptr, nullptr, 0, c.tape ? &mr : nullptr));
  if (c.tape) {
    c.tape->d_h = h; c.tape->d_d1 = d1; c.tape->d_sc = sc; c.tape->d_sh = sh; c.tape->d_mr = mr; c.tape->d_temb = use_emb;
    c.e->pool.put(mr);
  }
  TRY(conv(c, d1, nullptr, p + ".out_layers.3.weight", p + ".out_layers.3.bias", h.C, 1, 1, 0, sc, sh, 1, nullptr, nullptr, out));
  c.e->pool.put(sc); c.e->pool.put(sh);
  drop(c, d1);
  return 0;
}

// one TimestepEmbedSequential: layers applied in order; the first may read the virtual concat (x0|x1)
int run_layers_i(Ctx& c, const std::vector<asyrp_engine::Layer>& ls, const Act& x0, const Act* x1, Act* out) {
  Act h = x0;
  bool own = false;
  for (size_t k = 0; k < ls.size(); ++k) {
    const asyrp_engine::Layer& L = ls[k];
    const Act* second = (k == 0) ? x1 : nullptr;
    Act o;
    if (L.type == 0) {
      TRY(conv(c, h, second, L.p + ".weight", L.p + ".bias", L.cout, 3, 1, 0, nullptr, nullptr, 0, nullptr, nullptr, &o, true));
    } else if (L.type == 1) {
      TRY(resblock_i(c, L, h, second, &o));
    } else {
      TRY(attnblock_i(c, L, h, &o));
    }
    if (own) drop(c, h);
    h = o;
    own = true;
  }
  *out = h;
  return 0;
}

int decoder_i(Ctx& c, const Act& hin, const std::vector<Act>& hs, Act* eps) {
  asyrp_engine* e = c.e;
  int k = (int)hs.size() - 1;
  Act h = hin;
  bool own = false;
  for (auto& blk : e->out_blocks) {
    Act o;
    TRY(run_layers_i(c, blk, h, &hs[k], &o));
    --k;
    if (own) drop(c, h);
    h = o;
    own = true;
  }
  float *sc, *sh, *mr = nullptr;
  TRY(gn(c, h, nullptr, "out.0", EPS_I, &sc, &sh, nullptr, nullptr, 0, c.tape ? &mr : nullptr));
  TRY(conv(c, h, nullptr, "out.2.weight", "out.2.bias", e->cfg.out_channels, 3, 1, 0, sc, sh, 1, nullptr, nullptr, eps));
  if (c.tape) {
    c.tape->out_h = h; c.tape->out_sc = sc; c.tape->out_sh = sh; c.tape->out_mr = mr;
    e->pool.put(mr);
  }
  e->pool.put(sc); e->pool.put(sh);
  if (own) drop(c, h);
  return 0;
}

// UNetModel.forward (models/improved_ddpm/unet.py:676-752) on NHWC buffers; same output contract as unet_core_ddpm
int unet_core_iddpm(Ctx& c, const float* x_nhwc, const float* t_dev, int index, int apply_edit, const float* coeff,
                    int ignore_t, Act* et, Act* et_mod, Act* last_delta, Act* middle) {
  asyrp_engine* e = c.e;
  const asyrp_config& cf = e->cfg;
  et_mod->p = nullptr;
  last_delta->p = nullptr;
  const bool pre = c.tproj_pre && !c.tape;      // this step's row was computed before the loops (asyrp_run_edit)
  float *temb = nullptr, *temb_act = nullptr;
  if (pre) {
    c.tproj = const_cast<float*>(c.tproj_pre);
    c.tproj_ld = 0;
  } else {
    TRY(e->pool.get((size_t)c.B * e->temb_ch, &temb));
    TRY(e->pool.get((size_t)c.B * e->temb_ch, &temb_act));
    TRY(e->pool.get((size_t)c.B * e->tproj_total, &c.tproj));
    c.tproj_ld = e->tproj_total;
    // timestep_embedding [cos|sin] (nn.py:103-121) -> time_embed = Linear, SiLU, Linear (:513-517); every block's
    // emb_layers = Linear(SiLU(emb)) in one launch
    TRY(timestep_rows(c, t_dev, c.B, temb, temb_act, c.tproj));
  }
  Tape* const tape = c.tape;     // recording is limited to the DeltaBlock and decoder #2 below
  c.tape = nullptr;
  if (tape) tape->temb_act = temb_act;
  if (temb) e->pool.put(temb);
  // training forward: the pool HOLDS (instead of recycling) exactly what the backward pass reads -- swish(temb), the skip
  // tensors, the bottleneck h, the timestep projections and everything the DeltaBlock + decoder #2 window returns; the encoder's
  // and decoder #1's intermediates are recycled as in inference
  e->pool.defer = (tape != nullptr);
  if (temb_act) e->pool.put(temb_act);
  e->pool.defer = false;
  std::vector<Act> hs;
  Act xin;
  xin.p = const_cast<float*>(x_nhwc); xin.C = cf.in_channels; xin.H = cf.resolution; xin.W = cf.resolution;
  for (size_t n = 0; n < e->in_blocks.size(); ++n) {
    Act o;
    TRY(run_layers_i(c, e->in_blocks[n], n == 0 ? xin : hs.back(), nullptr, &o));
    hs.push_back(o);
  }
  Act h;
  TRY(run_layers_i(c, e->mid_block, hs.back(), nullptr, &h));
  *middle = h;
  c.skip_share = (index >= 0 && apply_edit && !tape && skip_share_enabled());   // both decoders run: see conv1_shared
  if (index >= 0 && apply_edit && c.dh_in) {   // injected delta_h tensor: unet.py:708-731
    Act h2;
    TRY(new_act(c, h.C, h.H, h.W, &h2));
    HIPCHK(launch_slerp_mix(h.p, c.dh_in, (float)(1.0 - (double)coeff[0]), c.use_mask, c.B, h.H, h.W, h.C, h2.p, c.s));
    TRY(decoder_i(c, h2, hs, et_mod));
    drop(c, h2);
  } else if (index >= 0 && apply_edit) {   // :697-704
    std::vector<Act> deltas(index + 1);
    const float* dptr[4] = {nullptr, nullptr, nullptr, nullptr};
    if (index + 1 > 4) return fail(ASYRP_EINVAL, "at most 4 DeltaBlocks can be summed");
    if (tape && index != 0) return fail(ASYRP_EINVAL, "the training step supports one DeltaBlock (index 0)");
    c.tape = tape;
    e->pool.defer = (tape != nullptr);
    for (int i = 0; i <= index; ++i) {
      TRY(deltablock_i(c, S("layer_%d", i), h, !ignore_t, &deltas[i]));
      dptr[i] = deltas[i].p;
    }
    Act h2;
    TRY(new_act(c, h.C, h.H, h.W, &h2));
    if (c.coeff_per_image) HIPCHK(launch_mix_per_image(h.p, dptr, coeff, index + 1, h2.p, c.B, h.per_image(), c.s));
    else HIPCHK(launch_mix(h.p, dptr, coeff, index + 1, h2.p, (long long)c.B * h.per_image(), c.s));
    for (int i = 0; i < index; ++i) drop(c, deltas[i]);
    *last_delta = deltas[index];
    if (tape) tape->c1 = coeff[1];
    TRY(decoder_i(c, h2, hs, et_mod));
    c.tape = nullptr;
    drop(c, h2);
    e->pool.defer = false;
  }
  TRY(decoder_i(c, h, hs, et));
  c.skip_share = false;
  for (auto& kv : c.skip_part) drop(c, kv.second);   // (empty unless a pass stopped early)
  c.skip_part.clear();
  e->pool.defer = (tape != nullptr);                 // skips and timestep projections: read by the backward pass
  for (auto& a : hs) drop(c, a);
  if (!pre) e->pool.put(c.tproj);
  e->pool.defer = false;
  c.tproj = nullptr;
  return 0;
}

int unet_core_ddpm(Ctx& c, const float* x_nhwc, const float* t_dev, int index, int apply_edit, const float* coeff,
                   int ignore_t, Act* et, Act* et_mod, Act* last_delta, Act* middle);

int unet_core(Ctx& c, const float* x_nhwc, const float* t_dev, int index, int apply_edit, const float* coeff,
              int ignore_t, Act* et, Act* et_mod, Act* last_delta, Act* middle) {
  if (c.e->cfg.family == ASYRP_FAMILY_IDDPM)
    return unet_core_iddpm(c, x_nhwc, t_dev, index, apply_edit, coeff, ignore_t, et, et_mod, last_delta, middle);
  return unet_core_ddpm(c, x_nhwc, t_dev, index, apply_edit, coeff, ignore_t, et, et_mod, last_delta, middle);
}

int unet_core_ddpm(Ctx& c, const float* x_nhwc, const float* t_dev, int index, int apply_edit, const float* coeff,
                   int ignore_t, Act* et, Act* et_mod, Act* last_delta, Act* middle) {
  asyrp_engine* e = c.e;
  const asyrp_config& cf = e->cfg;
  const int L = cf.n_levels, R = cf.resolution;
  et_mod->p = nullptr;
  last_delta->p = nullptr;
  // timestep embedding + every block's Linear(swish(temb)) in one launch
  const bool pre = c.tproj_pre && !c.tape;      // this step's row was computed before the loops (asyrp_run_edit)
  float *temb = nullptr, *temb_act = nullptr;
  if (pre) {
    c.tproj = const_cast<float*>(c.tproj_pre);
    c.tproj_ld = 0;
  } else {
    TRY(e->pool.get((size_t)c.B * e->temb_ch, &temb));
    TRY(e->pool.get((size_t)c.B * e->temb_ch, &temb_act));
    TRY(e->pool.get((size_t)c.B * e->tproj_total, &c.tproj));
    c.tproj_ld = e->tproj_total;
    TRY(timestep_rows(c, t_dev, c.B, temb, temb_act, c.tproj));
  }
  Tape* const tape = c.tape;     // recording is limited to the DeltaBlock and decoder #2 below
  c.tape = nullptr;
  if (tape) tape->temb_act = temb_act;
  if (temb) e->pool.put(temb);
  // training forward: the pool HOLDS (instead of recycling) exactly what the backward pass reads -- swish(temb), the skip
  // tensors, the bottleneck h, the timestep projections and everything the DeltaBlock + decoder #2 window returns; the encoder's
  // and decoder #1's intermediates are recycled as in inference
  e->pool.defer = (tape != nullptr);
  if (temb_act) e->pool.put(temb_act);
  e->pool.defer = false;

  // encoder (:485-495)
  std::vector<Act> skips;
  Act xin;
  xin.p = const_cast<float*>(x_nhwc); xin.C = cf.in_channels; xin.H = R; xin.W = R;
  {
    Act h0;
    TRY(conv(c, xin, nullptr, "conv_in.weight", "conv_in.bias", cf.ch, 3, 1, 0, nullptr, nullptr, 0, nullptr, nullptr,
             &h0, true));
    skips.push_back(h0);
  }
  int res = R;
  for (int i = 0; i < L; ++i) {
    for (int j = 0; j < cf.num_res_blocks; ++j) {
      Act o;
      TRY(resblock(c, S("down.%d.block.%d", i, j), skips.back(), nullptr, &o));
      if (has_attn(cf, res)) {
        Act o2;
        TRY(attnblock(c, S("down.%d.attn.%d", i, j), o, &o2));
        drop(c, o);
        o = o2;
      }
      skips.push_back(o);
    }
    if (i != L - 1) {
      Act o;
      TRY(conv(c, skips.back(), nullptr, S("down.%d.downsample.conv.weight", i), S("down.%d.downsample.conv.bias", i),
               skips.back().C, 3, 2, 0, nullptr, nullptr, 0, nullptr, nullptr, &o, true));
      skips.push_back(o);
      res /= 2;
    }
  }
  // middle (:500-504)
  Act m1, m2, h;
  TRY(resblock(c, "mid.block_1", skips.back(), nullptr, &m1));
  TRY(attnblock(c, "mid.attn_1", m1, &m2));
  drop(c, m1);
  TRY(resblock(c, "mid.block_2", m2, nullptr, &h));
  drop(c, m2);
  *middle = h;

  // both decoders run (and no backward tape is being recorded): share the skip halves of the decoder conv1s between them
  c.skip_share = (index >= 0 && apply_edit && !tape && skip_share_enabled());
  if (index >= 0 && apply_edit && c.dh_in) {   // injected delta_h tensor: :518-539
    Act h2;
    TRY(new_act(c, h.C, h.H, h.W, &h2));
    HIPCHK(launch_slerp_mix(h.p, c.dh_in, (float)(1.0 - (double)coeff[0]), c.use_mask, c.B, h.H, h.W, h.C, h2.p, c.s));
    TRY(decoder(c, h2, skips, et_mod));
    drop(c, h2);
  } else if (index >= 0 && apply_edit) {   // :510-516
    std::vector<Act> deltas(index + 1);
    const float* dptr[4] = {nullptr, nullptr, nullptr, nullptr};
    if (index + 1 > 4) return fail(ASYRP_EINVAL, "at most 4 DeltaBlocks can be summed");
    if (tape && index != 0) return fail(ASYRP_EINVAL, "the training step supports one DeltaBlock (index 0)");
    c.tape = tape;
    e->pool.defer = (tape != nullptr);
    for (int i = 0; i <= index; ++i) {
      TRY(deltablock(c, S("layer_%d", i), h, !ignore_t, &deltas[i]));
      dptr[i] = deltas[i].p;
    }
    Act h2;
    TRY(new_act(c, h.C, h.H, h.W, &h2));
    if (c.coeff_per_image) HIPCHK(launch_mix_per_image(h.p, dptr, coeff, index + 1, h2.p, c.B, h.per_image(), c.s));
    else HIPCHK(launch_mix(h.p, dptr, coeff, index + 1, h2.p, (long long)c.B * h.per_image(), c.s));
    for (int i = 0; i < index; ++i) drop(c, deltas[i]);
    *last_delta = deltas[index];
    if (tape) tape->c1 = coeff[1];
    TRY(decoder(c, h2, skips, et_mod));
    c.tape = nullptr;
    drop(c, h2);
    e->pool.defer = false;
  }
  TRY(decoder(c, h, skips, et));
  c.skip_share = false;
  for (auto& kv : c.skip_part) drop(c, kv.second);   // (empty unless a pass stopped early)
  c.skip_part.clear();
  e->pool.defer = (tape != nullptr);                 // skips and timestep projections: read by the backward pass
  for (auto& a : skips) drop(c, a);
  if (!pre) e->pool.put(c.tproj);
  e->pool.defer = false;
  c.tproj = nullptr;
  return 0;
}

__global__ void fill_kernel(float* p, float v, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = v;
}
struct FillVals { float v[256]; };
__global__ void fill_values_kernel(float* p, const FillVals f, int n) {
  if ((int)threadIdx.x < n) p[threadIdx.x] = f.v[threadIdx.x];
}

// The timestep-only work of a whole loop, before the loop (asyrp_run_edit, asyrp_run_inversion): row k of `tproj` is what the two
// launches at the head of step k's UNet evaluation would have produced for every image of the batch (see Ctx::tproj_pre).
struct TimestepRows {
  float *t = nullptr, *temb = nullptr, *tact = nullptr, *tproj = nullptr;
  void release(asyrp_engine* e) {
    for (float* p : {t, temb, tact, tproj})
      if (p) e->pool.put(p);
    t = temb = tact = tproj = nullptr;
  }
};
int precompute_timestep_rows(Ctx& c, const std::vector<float>& ts, TimestepRows* r) {
  asyrp_engine* e = c.e;
  const int n_steps = (int)ts.size();
  if (n_steps < 1) return 0;
  TRY(e->pool.get((size_t)n_steps, &r->t));
  TRY(e->pool.get((size_t)n_steps * e->temb_ch, &r->temb));
  TRY(e->pool.get((size_t)n_steps * e->temb_ch, &r->tact));
  TRY(e->pool.get((size_t)n_steps * e->tproj_total, &r->tproj));
  for (int o = 0; o < n_steps; o += 256) {      // the values travel as kernel arguments (no host buffer outlives the call, no copy engine)
    FillVals fv;
    const int n = std::min(256, n_steps - o);
    for (int i = 0; i < n; ++i) fv.v[i] = ts[o + i];
    hipLaunchKernelGGL(fill_values_kernel, dim3(1), dim3(256), 0, c.s, r->t + o, fv, n);
  }
  return timestep_rows(c, r->t, n_steps, r->temb, r->tact, r->tproj);
}

// Workspace buffers are recycled without waiting, which is only safe on one in-order stream: when the caller switches
// streams between calls, the new stream first waits for everything the previous call enqueued.
int bind_stream(asyrp_engine* e, hipStream_t s) {
  if (e->has_bound && e->bound_stream != s) {
    if (!e->bind_ev) HIPCHK(hipEventCreateWithFlags(&e->bind_ev, hipEventDisableTiming));
    HIPCHK(hipEventRecord(e->bind_ev, e->bound_stream));
    HIPCHK(hipStreamWaitEvent(s, e->bind_ev, 0));
  }
  e->bound_stream = s;
  e->has_bound = true;
  return 0;
}

// RAII: every ABI compute call returns its outstanding workspace on ANY exit path
struct PoolGuard {
  asyrp_engine* e;
  ~PoolGuard() { if (e) e->pool.reclaim(); }
};

int check_ready(asyrp_engine* e, int B) {
  if (!e) return fail(ASYRP_EINVAL, "null engine");
  if (!e->finalized) return fail(ASYRP_ESTATE, "asyrp_finalize_params has not been called");
  if (B < 1 || B > e->max_batch) return fail(ASYRP_EINVAL, "batch size outside [1, max_batch]");
  if (!e->d_freqs) return fail(ASYRP_ESTATE, "asyrp_set_temb_freqs has not been called");
  return 0;
}

int ddim_apply(Ctx& c, const float* x, const Act& et, const Act& et_mod, const float* noise_nhwc, int t, int t_next,
               float eta, float dt_lambda, int dt_end, float* xn, float* x0t) {
  asyrp_engine* e = c.e;
  if (e->alphas.empty()) return fail(ASYRP_ESTATE, "asyrp_set_schedule has not been called");
  if (t < 0 || t >= (int)e->alphas.size() || t_next >= (int)e->alphas.size())
    return fail(ASYRP_EINVAL, "timestep outside the schedule");
  if (eta != 0.f && !noise_nhwc) return fail(ASYRP_EINVAL, "eta != 0 requires a noise tensor");
  DdimArgs a;
  memset(&a, 0, sizeof a);
  a.xt = x; a.et = et.p; a.et_mod = et_mod.p; a.ld_e = et.C;
  a.noise = noise_nhwc;
  a.at = e->alphas[t];
  a.at_next = (t_next < 0) ? 1.0f : e->alphas[t_next];   // utils/diffusion_utils.py:68-69
  a.eta = eta; a.dt_lambda = dt_lambda;
  a.apply_dt = (dt_lambda != 1.0f && t >= dt_end) ? 1 : 0;   // :99
  a.xt_next = xn; a.x0_t = x0t;
  a.npix = (long long)c.B * et.H * et.W;
  HIPCHK(launch_ddim(a, c.s));
  return 0;
}

}  // namespace

// =====================================================================================================
// Training step of the DeltaBlock (diffusion_latent.py:301-354, DDPM family):
//   forward  = one Asyrp step (dual decoder) that keeps decoder #2's activations ("tape");
//   backward = d(loss)/d(et_modified) -> data gradients through decoder #2 (transposed convolutions on the same
//              implicit-GEMM kernels, GroupNorm/SiLU/attention/upsample backward glue from backward.hip) -> d(h2)
//              -> DeltaBlock parameter gradients.  The loss itself (CLIP direction + L1) stays in the caller's PyTorch.
// =====================================================================================================
namespace {

// conv weight [Cout][Cin][k][k] -> the weight of the data-gradient convolution [Cin][Cout][k][k], taps rotated by 180 degrees
std::vector<float> transpose_conv_weight(const std::vector<float>& w, int cout, int cin, int k) {
  std::vector<float> o((size_t)cout * cin * k * k);
  const int kk = k * k;
  for (int co = 0; co < cout; ++co)
    for (int ci = 0; ci < cin; ++ci)
      for (int t = 0; t < kk; ++t) o[((size_t)ci * cout + co) * kk + (kk - 1 - t)] = w[((size_t)co * cin + ci) * kk + t];
  return o;
}

int add_bwd_weight(asyrp_engine* e, const std::string& name, const std::vector<float>& wT, int cout_t, int cin_t, int k) {
  TRY(upload(e, name, pack_conv(wT, cout_t, cin_t, k)));
  if (e->math == MATH_F16X3) TRY(pack_x3(e, name, wT, cout_t, cin_t, k));
  return 0;
}

// which parameters take part in the backward pass: every convolution of the decoder and the DeltaBlock's own 1x1 convs
bool is_bwd_weight(const asyrp_engine* e, const std::string& key) {
  if (e->cfg.family == ASYRP_FAMILY_IDDPM)
    return key.rfind("output_blocks.", 0) == 0 || key.rfind("out.2.", 0) == 0 || key == "layer_0.out_layers.3.weight" ||
           key == "layer_0.in_layers.2.weight";
  return key.rfind("up.", 0) == 0 || key.rfind("conv_out.", 0) == 0 || key == "layer_0.conv2.weight";
}

int build_bwd_weight(asyrp_engine* e, const ParamSpec& s) {
  const int cout = (int)s.shape[0], cin = (int)s.shape[1], k = (s.shape.size() == 4) ? (int)s.shape[2] : 1;
  return add_bwd_weight(e, s.key + "#T", transpose_conv_weight(hostp(e, s.key), cout, cin, k), cin, cout, k);
}

// transposed images of every decoder convolution (+ the DeltaBlock's convs); built on the first training call
int ensure_bwd_weights(asyrp_engine* e) {
  if (e->bwd_weights) return 0;
  HIPCHK(hipDeviceSynchronize());
  for (auto& s : e->specs) {
    if (!(s.shape.size() == 4 || s.shape.size() == 3) || !ends_with(s.key, ".weight") || !is_bwd_weight(e, s.key)) continue;
    const int cout = (int)s.shape[0], cin = (int)s.shape[1];
    const std::string p = s.key.substr(0, s.key.size() - strlen(".weight"));
    if (e->cfg.family == ASYRP_FAMILY_DDPM && (ends_with(p, ".k") || ends_with(p, ".v"))) continue;
    if (e->cfg.family == ASYRP_FAMILY_DDPM && ends_with(p, ".q")) {   // fused q|k|v [3C][C] -> [C][3C]
      const std::string ap = p.substr(0, p.size() - 2);
      std::vector<float> w3((size_t)3 * cout * cin);
      const char* names[3] = {".q", ".k", ".v"};
      for (int t = 0; t < 3; ++t) {
        const auto& wt = hostp(e, ap + names[t] + ".weight");
        std::copy(wt.begin(), wt.end(), w3.begin() + (size_t)t * cout * cin);
      }
      TRY(add_bwd_weight(e, ap + ".qkv.weight#T", transpose_conv_weight(w3, 3 * cout, cin, 1), cin, 3 * cout, 1));
      continue;
    }
    TRY(build_bwd_weight(e, s));
  }
  e->bwd_weights = true;
  return 0;
}

// GroupNorm (+SiLU) backward of y = act(GN(x)) given dA = dL/dy: returns dx for the leading Cd channels of x (+ `add`)
int act_gn_backward(Ctx& c, const Act& dA, const Act& x0, const Act* x1, const float* sc, const float* sh, const float* mr,
                    const std::string& norm, int silu, int Cd, const Act* add, Act* dx, const float* film = nullptr,
                    int ld_film = 0) {
  const int C = x0.C + (x1 ? x1->C : 0), HW = x0.H * x0.W;
  if (dA.C != C) return fail(ASYRP_EINVAL, "gradient / activation channel mismatch at " + norm);
  Act dy;
  TRY(new_act(c, C, x0.H, x0.W, &dy));
  const int nblk = act_bwd_nblk(HW);
  float *part, *coef;
  TRY(c.e->pool.get((size_t)c.B * nblk * C * 4, &part));
  TRY(c.e->pool.get((size_t)c.B * C * 3, &coef));
  ActBwdArgs a;
  memset(&a, 0, sizeof a);
  a.dA = dA.p; a.ldd = dA.C;
  a.x0 = x0.p; a.c0 = x0.C; a.ldx0 = x0.C; a.x0_z = x0.per_image();
  if (x1) { a.x1 = x1->p; a.c1 = x1->C; a.ldx1 = x1->C; a.x1_z = x1->per_image(); }
  a.scale = sc; a.shift = sh; a.silu = silu; a.dy = dy.p; a.partial = reinterpret_cast<double*>(part);
  a.HW = HW; a.N = c.B; a.C = C;
  HIPCHK(launch_act_bwd_partial(a, c.s));
  GnBwdFinArgs f;
  memset(&f, 0, sizeof f);
  f.partial = a.partial; f.nblk = nblk; f.gamma = P(c, norm + ".weight"); f.mr = mr; f.N = c.B; f.HW = HW; f.C = C; f.coef = coef;
  f.film_scale = film; f.ld_film = ld_film;
  if (!f.gamma) return fail(ASYRP_EKEY, "missing norm params " + norm);
  HIPCHK(launch_gn_bwd_finalize(f, c.s));
  TRY(new_act(c, Cd, x0.H, x0.W, dx));
  HIPCHK(launch_gn_bwd_apply(dy.p, C, x0.p, x0.C, x0.per_image(), coef, add ? add->p : nullptr, dx->p, Cd, HW, c.B, c.s));
  c.e->pool.put(part); c.e->pool.put(coef);
  drop(c, dy);
  return 0;
}

// transposed convolution of a gradient tensor: dX = conv(dY, W^T rot180) -- same geometry as the forward (stride 1)
int conv_bwd_data(Ctx& c, const Act& dY, const std::string& wname_fwd, int Cin_fwd, int ks, const Act* add, Act* dX,
                  int ldb_full = 0) {
  return conv(c, dY, nullptr, wname_fwd + "#T", "", Cin_fwd, ks, 1, 0, nullptr, nullptr, 0, nullptr, add, dX, false, 0, nullptr,
              nullptr, nullptr, ldb_full);
}

int resblock_backward(Ctx& c, const TapeRes& t, const Act& d_out, Act* d_in) {
  const int Cin = t.x0.C + (t.has_x1 ? t.x1.C : 0), Ch = t.x0.C;
  Act dA2, d_h1, dA1;
  TRY(conv_bwd_data(c, d_out, t.p + t.c2 + ".weight", t.Cout, 3, nullptr, &dA2));
  TRY(act_gn_backward(c, dA2, t.h1, nullptr, t.sc2, t.sh2, t.mr2, t.p + t.n2, 1, t.Cout, nullptr, &d_h1, t.film, t.ld_film));
  drop(c, dA2);
  TRY(conv_bwd_data(c, d_h1, t.p + t.c1 + ".weight", Cin, 3, nullptr, &dA1));
  drop(c, d_h1);
  Act d_short;
  bool own_short = false;
  const Act* add = &d_out;     // identity shortcut: x + h
  if (t.mode == 2) {           // ResBlock(up=True): both branches went through nearest x2 -> 2x2 sums on the way back
    if (t.shortcut || t.has_x1) return fail(ASYRP_EINVAL, "up-sampling ResBlock with a channel change");
    Act lo;
    TRY(new_act(c, Cin, t.x0.H, t.x0.W, &lo));
    HIPCHK(launch_sum2x2(dA1.p, lo.p, c.B, t.x0.H, t.x0.W, Cin, c.s));
    drop(c, dA1);
    dA1 = lo;
    TRY(new_act(c, Ch, t.x0.H, t.x0.W, &d_short));
    HIPCHK(launch_sum2x2(d_out.p, d_short.p, c.B, t.x0.H, t.x0.W, Ch, c.s));
    own_short = true;
    add = &d_short;
  } else if (t.shortcut) {     // 1x1 shortcut: only the gradient of the leading Ch input channels (the h part) is needed
    TRY(conv_bwd_data(c, d_out, t.p + t.sk + ".weight", Ch, 1, nullptr, &d_short, Cin));
    own_short = true;
    add = &d_short;
  } else if (Ch != t.Cout) {
    return fail(ASYRP_EINVAL, "identity shortcut with a channel change");
  }
  TRY(act_gn_backward(c, dA1, t.x0, t.has_x1 ? &t.x1 : nullptr, t.sc1, t.sh1, t.mr1, t.p + t.n1, 1, Ch, add, d_in));
  drop(c, dA1);
  if (own_short) drop(c, d_short);
  return 0;
}

// batched fp32-MFMA GEMM, z = (image, head): out[z][m][n] = sum_k A[z][m][k] * B[z][k][n]  (bT: B[z][k][n] = Bt[z][n][k])
int bgemm(Ctx& c, const float* A, int lda, long long a_zo, long long a_zi, const float* Bm, int ldb, long long b_zo, long long b_zi,
          int bT, int M, int N, int K, float* out, int ldo, long long o_zo, long long o_zi, int Zo, int ZI = 1) {
  GemmArgs g;
  memset(&g, 0, sizeof g);
  g.a0 = A; g.c0 = K; g.lda0 = lda; g.a0_zo = a_zo; g.a0_zi = a_zi;
  g.Hin = M; g.Win = 1; g.Hout = M; g.Wout = 1; g.Cin = K; g.Cout = N; g.ks = 1; g.stride = 1;
  g.w = Bm; g.ldb = ldb; g.bT = bT; g.w_zo = b_zo; g.w_zi = b_zi;
  g.alpha = 1.0f; g.out = out; g.ldo = ldo; g.o_zo = o_zo; g.o_zi = o_zi; g.ZI = ZI; g.Z = Zo * ZI; g.math = MATH_F32;
  return run_gemm(c, g);
}

// AttnBlock (one C-wide head, q|k|v blocks) and AttentionBlock + QKVAttentionLegacy (heads of Dh channels, per-head [q|k|v])
int attnblock_backward(Ctx& c, const TapeAttn& t, const Act& d_out, Act* d_in) {
  const int C = t.x.C, T = t.x.H * t.x.W, B = c.B, H = t.heads, Dh = C / H;
  const float scale = 1.0f / std::sqrt((float)Dh);
  const bool iddpm = c.e->cfg.family == ASYRP_FAMILY_IDDPM;
  Act d_o;
  TRY(conv_bwd_data(c, d_out, t.p + ".proj_out.weight", C, 1, nullptr, &d_o));
  const int q_off = 0, k_off = (H == 1) ? C : Dh, v_off = (H == 1) ? 2 * C : 2 * Dh;
  const long long hs = (H == 1) ? 0 : 3 * Dh;            // head stride inside a qkv row
  const float* q = t.qkv.p + q_off; const float* k = t.qkv.p + k_off; const float* v = t.qkv.p + v_off;
  const long long qz = (long long)T * 3 * C, tz = (long long)T * T, oz = (long long)T * C;
  float *dP, *tr;
  TRY(c.e->pool.get((size_t)B * H * T * T, &dP));
  TRY(c.e->pool.get((size_t)B * H * T * T, &tr));
  Act dqkv;
  TRY(new_act(c, 3 * C, t.x.H, t.x.W, &dqkv));
  // dP = d_o V^T ; dS = P * (dP - rowsum(dP * P)) * scale
  TRY(bgemm(c, d_o.p, C, oz, Dh, v, 3 * C, qz, hs, 1, T, T, Dh, dP, T, H * tz, tz, B, H));
  HIPCHK(launch_softmax_bwd(t.P, dP, (long long)B * H * T, T, scale, c.s));
  // dQ = dS K ; dK = dS^T Q ; dV = P^T d_o
  TRY(bgemm(c, dP, T, H * tz, tz, k, 3 * C, qz, hs, 0, T, Dh, T, dqkv.p + q_off, 3 * C, qz, hs, B, H));
  HIPCHK(launch_transpose(dP, T, tz, tr, T, T, tz, B * H, c.s));
  TRY(bgemm(c, tr, T, H * tz, tz, q, 3 * C, qz, hs, 0, T, Dh, T, dqkv.p + k_off, 3 * C, qz, hs, B, H));
  HIPCHK(launch_transpose(t.P, T, tz, tr, T, T, tz, B * H, c.s));
  TRY(bgemm(c, tr, T, H * tz, tz, d_o.p, C, oz, Dh, 0, T, Dh, T, dqkv.p + v_off, 3 * C, qz, hs, B, H));
  c.e->pool.put(dP); c.e->pool.put(tr);
  drop(c, d_o);
  Act d_n;
  TRY(conv_bwd_data(c, dqkv, t.p + ".qkv.weight", C, 1, nullptr, &d_n));
  drop(c, dqkv);
  (void)iddpm;
  TRY(act_gn_backward(c, d_n, t.x, nullptr, t.sc, t.sh, t.mr, t.p + ".norm", 0, C, &d_out, d_in));
  drop(c, d_n);
  return 0;
}

// gradient of a [Cout][Cin] matrix W used as y[m][co] = sum_ci x[m][ci] W[co][ci]:  dW = dY^T X  (M rows)
int weight_grad(Ctx& c, const float* dY, int Cout, const float* X, int Cin, long long M, float* dW /*device [Cout][Cin]*/) {
  float* dYt;
  TRY(c.e->pool.get((size_t)Cout * M, &dYt));
  HIPCHK(launch_transpose(dY, Cout, 0, dYt, (int)M, Cout, 0, 1, c.s));
  TRY(bgemm(c, dYt, (int)M, 0, 0, X, Cin, 0, 0, 0, Cout, Cin, (int)M, dW, Cin, 0, 0, 1));
  c.e->pool.put(dYt);
  return 0;
}

}  // namespace

namespace {
// weights of the 3x3 convolutions that read a nearest-x2 up-sampled tensor: DDPM `up.<i>.upsample.conv` (models/ddpm/diffusion.py:
// 84-87), iDDPM / ADM `in_layers.2` of a ResBlock(up=True) (models/improved_ddpm/unet.py:232-234, 281-284)
bool is_upsampled_conv(const asyrp_engine* e, const std::string& key) {
  if (e->cfg.family != ASYRP_FAMILY_IDDPM) return key.find(".upsample.conv.weight") != std::string::npos;
  for (const auto& blk : e->out_blocks)
    for (const auto& L : blk)
      if (L.type == 1 && L.mode == 2 && key == L.p + ".in_layers.2.weight") return true;
  return false;
}
}  // namespace

// =====================================================================================================
// C ABI
// =====================================================================================================
extern "C" {

int asyrp_abi_version(void) { return ASYRP_ABI_VERSION; }

const char* asyrp_last_error(void) { return g_err.c_str(); }

int asyrp_create(asyrp_engine** out, const asyrp_config* cfg, int max_batch, int device) {
  if (!out || !cfg) return fail(ASYRP_EINVAL, "null argument");
  if (cfg->family != ASYRP_FAMILY_DDPM && cfg->family != ASYRP_FAMILY_IDDPM) return fail(ASYRP_EINVAL, "unknown UNet family");
  if (cfg->family == ASYRP_FAMILY_IDDPM && (cfg->num_head_channels <= 0 || cfg->num_classes < 0))
    return fail(ASYRP_EINVAL, "iDDPM family needs num_head_channels > 0");
  if (cfg->n_levels < 1 || cfg->n_levels > ASYRP_MAX_LEVELS || cfg->ch % 32 != 0 || max_batch < 1 || cfg->n_delta < 0 ||
      cfg->n_delta > 4 || cfg->resolution % (1 << (cfg->n_levels - 1)) != 0)
    return fail(ASYRP_EINVAL, "unsupported configuration");
  if (!(cfg->nominal_batch == 0 || cfg->nominal_batch == 1 || cfg->nominal_batch == 2 || cfg->nominal_batch == 32))
    return fail(ASYRP_EINVAL, "nominal_batch must be 0 (= 32, the default class), 1, 2 (the small class) or 32: the batch classes that are tested");
  for (int i = 0; i < 5; ++i)
    if (cfg->reserved[i] != 0) return fail(ASYRP_EINVAL, "asyrp_config.reserved must be zero");
  // no device call here: the engine can be created (and its parameter inventory listed) without a GPU;
  // device memory is first touched by asyrp_set_temb_freqs / asyrp_finalize_params.
  asyrp_engine* e = new asyrp_engine();
  e->cfg = *cfg;
  if (cfg->conv_math != ASYRP_MATH_F16X3 && cfg->conv_math != ASYRP_MATH_F32 && cfg->conv_math != ASYRP_MATH_F16) {
    delete e;
    return fail(ASYRP_EINVAL, "unknown conv_math");
  }
  e->math = (cfg->conv_math == ASYRP_MATH_F32) ? MATH_F32 : MATH_F16X3;
  e->np = (cfg->conv_math == ASYRP_MATH_F16) ? 1 : 3;   // the single-product mode runs the f16x3 family's kernels with NP = 1
  e->max_batch = max_batch;
  e->device = device;
  if (cfg->family == ASYRP_FAMILY_IDDPM) build_specs_iddpm(e);
  else build_specs_ddpm(e);
  for (size_t i = 0; i < e->specs.size(); ++i) e->spec_idx[e->specs[i].key] = (int)i;
  e->host.resize(e->specs.size());
  e->loaded.assign(e->specs.size(), 0);
  e->dirty.assign(e->specs.size(), 0);
  *out = e;
  return 0;
}

void asyrp_destroy(asyrp_engine* e) {
  if (!e) return;
  if (e->finalized || e->d_freqs || !e->dev.empty()) {
    (void)hipSetDevice(e->device);
    (void)hipDeviceSynchronize();
  }
  for (auto& kv : e->dev) (void)hipFree(kv.second);
  for (auto& kv : e->xw) (void)hipFree(kv.second.p);
  if (e->d_freqs) (void)hipFree(e->d_freqs);
  if (e->d_t) (void)hipFree(e->d_t);
  if (e->bind_ev) (void)hipEventDestroy(e->bind_ev);
  for (auto& r : e->prof) { (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b); }
  for (auto& ev : e->ev_free) { (void)hipEventDestroy(ev.first); (void)hipEventDestroy(ev.second); }
  e->pool.destroy();
  delete e;
}

int asyrp_num_params(const asyrp_engine* e) { return e ? (int)e->specs.size() : 0; }

int asyrp_param_info(const asyrp_engine* e, int i, const char** key, int64_t shape[4], int* ndim) {
  if (!e || i < 0 || i >= (int)e->specs.size()) return fail(ASYRP_EINVAL, "param index out of range");
  const ParamSpec& s = e->specs[i];
  if (key) *key = s.key.c_str();
  if (ndim) *ndim = (int)s.shape.size();
  if (shape)
    for (size_t d = 0; d < 4; ++d) shape[d] = d < s.shape.size() ? s.shape[d] : 1;
  return 0;
}

int asyrp_load_param(asyrp_engine* e, const char* key, const float* host_data, const int64_t* shape, int ndim) {
  if (!e || !key || !host_data || !shape) return fail(ASYRP_EINVAL, "null argument");
  auto it = e->spec_idx.find(key);
  if (it == e->spec_idx.end()) return fail(ASYRP_EKEY, std::string("unexpected parameter key: ") + key);
  const ParamSpec& s = e->specs[it->second];
  if ((int)s.shape.size() != ndim) return fail(ASYRP_EKEY, std::string("rank mismatch for ") + key);
  for (int d = 0; d < ndim; ++d)
    if (s.shape[d] != shape[d]) return fail(ASYRP_EKEY, std::string("shape mismatch for ") + key);
  e->host[it->second].assign(host_data, host_data + s.numel());
  e->loaded[it->second] = 1;
  e->dirty[it->second] = 1;
  e->finalized = false;
  return 0;
}

int asyrp_set_schedule(asyrp_engine* e, const float* ab, int n) {
  if (!e || !ab || n < 1) return fail(ASYRP_EINVAL, "bad schedule");
  e->alphas.assign(ab, ab + n);
  return 0;
}

int asyrp_set_temb_freqs(asyrp_engine* e, const float* f, int n) {
  if (!e || !f || n != e->cfg.ch / 2) return fail(ASYRP_EINVAL, "freqs must have ch/2 entries");
  HIPCHK(hipSetDevice(e->device));
  if (!e->d_freqs) HIPCHK(hipMalloc(&e->d_freqs, sizeof(float) * n));
  HIPCHK(hipMemcpy(e->d_freqs, f, sizeof(float) * n, hipMemcpyHostToDevice));
  e->n_freqs = n;
  return 0;
}

int asyrp_finalize_params(asyrp_engine* e) {
  if (!e) return fail(ASYRP_EINVAL, "null engine");
  HIPCHK(hipSetDevice(e->device));
  for (size_t i = 0; i < e->specs.size(); ++i)
    if (!e->loaded[i]) return fail(ASYRP_EKEY, "parameter not loaded: " + e->specs[i].key);
  HIPCHK(hipDeviceSynchronize());
  if (!e->d_t) HIPCHK(hipMalloc(&e->d_t, sizeof(float) * e->max_batch));
  // Only tensors loaded since the previous finalize (and the images fused from them) are re-packed: a DeltaBlock update
  // (another checkpoint/*.pth, or one optimiser step) costs four small uploads, not a re-pack of the whole UNet.
  auto isd = [&](const std::string& key) {
    auto it = e->spec_idx.find(key);
    return it != e->spec_idx.end() && e->dirty[it->second];
  };
  // temb projections of every ResnetBlock / DeltaBlock -> one [O_total][temb_ch] matrix
  // per-block timestep projections: DDPM `<block>.temb_proj`, iDDPM `<block>.emb_layers.1`
  const char* tsuf = (e->cfg.family == ASYRP_FAMILY_IDDPM) ? ".emb_layers.1" : ".temb_proj";
  const std::string tw_suf = std::string(tsuf) + ".weight", tb_suf = std::string(tsuf) + ".bias";
  {
    bool any = e->tproj_off.empty();
    for (auto& s : e->specs)
      if ((ends_with(s.key, tw_suf) || ends_with(s.key, tb_suf)) && isd(s.key)) any = true;
    if (any) {
      std::vector<float> tw, tb;
      e->tproj_off.clear();
      int off = 0;
      for (auto& s : e->specs) {
        if (!ends_with(s.key, tw_suf)) continue;
        const std::string p = s.key.substr(0, s.key.size() - tw_suf.size());
        const auto& w = hostp(e, s.key);
        const auto& b = hostp(e, p + tb_suf);
        e->tproj_off[p] = off;
        tw.insert(tw.end(), w.begin(), w.end());
        tb.insert(tb.end(), b.begin(), b.end());
        off += (int)s.shape[0];
      }
      e->tproj_total = off;
      TRY(upload(e, "__tproj.weight", tw));
      TRY(upload(e, "__tproj.bias", tb));
    }
  }
  for (auto& s : e->specs) {
    const auto& v = hostp(e, s.key);
    if (ends_with(s.key, tw_suf) || ends_with(s.key, tb_suf)) continue;
    if (s.key == "label_emb.weight") continue;   // never read by forward (models/improved_ddpm/unet.py:676-688)
    if (s.shape.size() == 4 || s.shape.size() == 3) {   // Conv2d, or Conv1d with kernel 1 (attention qkv / proj_out)
      const int cout = (int)s.shape[0], cin = (int)s.shape[1], k = (s.shape.size() == 4) ? (int)s.shape[2] : 1;
      const std::string p = s.key.substr(0, s.key.size() - strlen(".weight"));
      if (ends_with(p, ".q")) {   // fuse q|k|v into one [Cin][3C] operand
        const std::string ap = p.substr(0, p.size() - 2);
        const char* names[3] = {".q", ".k", ".v"};
        bool any = false;
        for (int t = 0; t < 3; ++t) any = any || isd(ap + names[t] + ".weight") || isd(ap + names[t] + ".bias");
        if (!any) continue;
        std::vector<float> w((size_t)cin * 3 * cout), b((size_t)3 * cout);
        for (int t = 0; t < 3; ++t) {
          const auto& wt = hostp(e, ap + names[t] + ".weight");
          const auto& bt = hostp(e, ap + names[t] + ".bias");
          for (int co = 0; co < cout; ++co) {
            for (int ci = 0; ci < cin; ++ci) w[(size_t)ci * 3 * cout + t * cout + co] = wt[(size_t)co * cin + ci];
            b[(size_t)t * cout + co] = bt[co];
          }
        }
        TRY(upload(e, ap + ".qkv.weight", w));
        TRY(upload(e, ap + ".qkv.bias", b));
        if (e->math == MATH_F16X3) {
          std::vector<float> w3((size_t)3 * cout * cin);   // [3C][Cin] = q|k|v rows, PyTorch conv layout
          for (int t = 0; t < 3; ++t) {
            const auto& wt = hostp(e, ap + names[t] + ".weight");
            std::copy(wt.begin(), wt.end(), w3.begin() + (size_t)t * cout * cin);
          }
          TRY(pack_x3(e, ap + ".qkv.weight", w3, 3 * cout, cin, 1));
          if (cin % 64 == 0) TRY(pack_g1(e, ap + ".qkv.weight", w3, 3 * cout, cin));
        }
      } else if (ends_with(p, ".k") || ends_with(p, ".v")) {
        continue;
      } else {
        if (isd(s.key)) {
          TRY(upload(e, s.key, pack_conv(v, cout, cin, k)));
          if (e->math == MATH_F16X3) TRY(pack_x3(e, s.key, v, cout, cin, k));
          if (e->math == MATH_F16X3 && k == 1 && cin % 64 == 0) TRY(pack_g1(e, s.key, v, cout, cin));   // gemm1x1.hip
          // convolutions applied to a nearest-x2 up-sampled tensor: the four phase-collapsed 2x2 images (polyphase form)
          if (e->math == MATH_F16X3 && k == 3 && cin % 32 == 0 && is_upsampled_conv(e, s.key)) TRY(pack_x3_up(e, s.key, v, cout, cin));
          // the UNet's last conv (conv_out / out.2): a second image with the 9 taps folded into N for conv_out.hip
          if (e->math == MATH_F16X3 && k == 3 && (s.key == "conv_out.weight" || s.key == "out.2.weight") && cout * 9 <= (conv_out_two_tiles() ? 64 : 32) &&
              cin % 16 == 0 && cin <= 256) {
            std::vector<float> w1((size_t)9 * cout * cin);
            for (int co = 0; co < cout; ++co)
              for (int ci = 0; ci < cin; ++ci)
                for (int t = 0; t < 9; ++t) w1[((size_t)(t * cout + co)) * cin + ci] = v[((size_t)co * cin + ci) * 9 + t];
            TRY(pack_x3(e, s.key + "#co", w1, 9 * cout, cin, 1));
          }
        }
        // a ResnetBlock / ResBlock with a 1x1 shortcut: fused image (second conv ++ shortcut) and fused bias
        const char* c2 = (e->cfg.family == ASYRP_FAMILY_IDDPM) ? ".out_layers.3" : ".conv2";
        const char* sk = (e->cfg.family == ASYRP_FAMILY_IDDPM) ? ".skip_connection" : ".nin_shortcut";
        if (e->math == MATH_F16X3 && k == 3 && ends_with(p, c2)) {
          const std::string blk = p.substr(0, p.size() - strlen(c2));
          auto sit = e->spec_idx.find(blk + sk + ".weight");
          if (sit != e->spec_idx.end() &&
              (isd(s.key) || isd(p + ".bias") || isd(blk + sk + ".weight") || isd(blk + sk + ".bias"))) {
            const ParamSpec& ss = e->specs[sit->second];
            TRY(pack_x3_fused(e, s.key, v, cout, cin, hostp(e, ss.key), (int)ss.shape[1]));
            std::vector<float> fb = hostp(e, p + ".bias");
            const auto& sb = hostp(e, blk + sk + ".bias");
            for (size_t i = 0; i < fb.size(); ++i) fb[i] += sb[i];
            TRY(upload(e, p + ".bias#sc", fb));
          }
        }
      }
    } else {
      const std::string p = s.key.substr(0, s.key.rfind('.'));
      if (ends_with(p, ".q") || ends_with(p, ".k") || ends_with(p, ".v")) continue;   // folded into qkv.bias
      if (isd(s.key)) TRY(upload(e, s.key, v));
    }
  }
  // transposed images of the training step: the DeltaBlock's second conv changes at every optimiser step, the decoder's
  // only when new base weights are loaded (then everything is rebuilt lazily by the next asyrp_train_forward)
  if (e->bwd_weights) {
    for (auto& s : e->specs)
      if (isd(s.key) && is_bwd_weight(e, s.key) && s.key.rfind("layer_0.", 0) != 0) e->bwd_weights = false;
    if (e->bwd_weights)
      for (auto& s : e->specs)
        if (isd(s.key) && is_bwd_weight(e, s.key) && s.key.rfind("layer_0.", 0) == 0) TRY(build_bwd_weight(e, s));
  }
  std::fill(e->dirty.begin(), e->dirty.end(), 0);
  e->finalized = true;
  return 0;
}

int64_t asyrp_device_bytes(const asyrp_engine* e) { return e ? (int64_t)(e->param_bytes + e->pool.total_bytes) : 0; }

static void bottleneck_shape(const asyrp_engine* e, int* C, int* R) {
  const asyrp_config& cf = e->cfg;
  *C = cf.ch * cf.ch_mult[cf.n_levels - 1];
  *R = cf.resolution >> (cf.n_levels - 1);
}

// Stage an injected delta-h tensor (NCHW at the boundary) into the engine's NHWC layout; nullptr in -> nullptr out.
static int stage_delta_in(Ctx& c, const float* delta_h_in, int use_mask, int index, int apply_edit, float** staged) {
  *staged = nullptr;
  if (!delta_h_in || index < 0 || !apply_edit) return 0;   // ignored below t_edit / without index, as in the reference
  int C = 0, R = 0;
  bottleneck_shape(c.e, &C, &R);
  if (use_mask && R < 6) return fail(ASYRP_EINVAL, "use_mask needs a bottleneck of at least 6x6 (rows 4..H-2 would be empty)");
  TRY(c.e->pool.get((size_t)c.B * C * R * R, staged));
  HIPCHK(launch_nchw_to_nhwc(delta_h_in, *staged, c.B, C, R * R, c.s));
  c.dh_in = *staged;
  c.use_mask = use_mask;
  return 0;
}

int asyrp_unet_forward(asyrp_engine* e, const float* x, const float* t, int B, int index, int apply_edit,
                       const float* hs_coeff_host, int n_coeff, int ignore_timestep, const float* delta_h_in, int use_mask,
                       float* et, float* et_mod, float* delta_h_out, float* middle_h, void* stream) {
  TRY(check_ready(e, B));
  if (!x || !t || !et) return fail(ASYRP_EINVAL, "null tensor");
  if (index >= e->cfg.n_delta) return fail(ASYRP_EINVAL, "index >= number of DeltaBlocks (setattr_layers)");
  if (index >= 0 && apply_edit && (!hs_coeff_host || n_coeff < (delta_h_in ? 1 : index + 2)))
    return fail(ASYRP_EINVAL, "hs_coeff needs index+2 entries (1 with an injected delta_h)");
  if (index >= 0 && !et_mod) return fail(ASYRP_EINVAL, "et_mod buffer required when index is given");
  HIPCHK(hipSetDevice(e->device));
  Ctx c{e, (hipStream_t)stream, B};
  PoolGuard guard{e};
  TRY(bind_stream(e, c.s));
  const asyrp_config& cf = e->cfg;
  const int HW = cf.resolution * cf.resolution;
  float* xn = nullptr;
  TRY(e->pool.get((size_t)B * HW * cf.in_channels, &xn));
  HIPCHK(launch_nchw_to_nhwc(x, xn, B, cf.in_channels, HW, c.s));
  float* dh_staged = nullptr;
  TRY(stage_delta_in(c, delta_h_in, use_mask, index, apply_edit, &dh_staged));
  Act a_et, a_em, a_dh, a_mid;
  TRY(unet_core(c, xn, t, index, apply_edit, hs_coeff_host, ignore_timestep, &a_et, &a_em, &a_dh, &a_mid));
  if (dh_staged) e->pool.put(dh_staged);
  HIPCHK(launch_nhwc_to_nchw(a_et.p, a_et.C, et, B, a_et.C, HW, c.s));
  if (index >= 0) {
    const Act& src = a_em.p ? a_em : a_et;   // no edit: ε̃ ≡ ε
    HIPCHK(launch_nhwc_to_nchw(src.p, src.C, et_mod, B, src.C, HW, c.s));
  }
  if (a_dh.p && delta_h_out) HIPCHK(launch_nhwc_to_nchw(a_dh.p, a_dh.C, delta_h_out, B, a_dh.C, a_dh.H * a_dh.W, c.s));
  if (middle_h) HIPCHK(launch_nhwc_to_nchw(a_mid.p, a_mid.C, middle_h, B, a_mid.C, a_mid.H * a_mid.W, c.s));
  drop(c, a_et);
  if (a_em.p) drop(c, a_em);
  if (a_dh.p) drop(c, a_dh);
  drop(c, a_mid);
  e->pool.put(xn);
  return 0;
}

int asyrp_ddim_step(asyrp_engine* e, const float* xt, int t, int t_next, int B, float eta, const float* noise,
                    int learn_sigma, int index, int apply_edit, const float* hs_coeff_host, int n_coeff,
                    int ignore_timestep, const float* delta_h_in, int use_mask, float dt_lambda, int dt_end,
                    float* xt_next, float* x0_t, float* delta_h_out, float* middle_h, void* stream) {
  TRY(check_ready(e, B));
  if (!xt || !xt_next) return fail(ASYRP_EINVAL, "null tensor");
  if (index >= e->cfg.n_delta) return fail(ASYRP_EINVAL, "index >= number of DeltaBlocks (setattr_layers)");
  if (index >= 0 && apply_edit && (!hs_coeff_host || n_coeff < (delta_h_in ? 1 : index + 2)))
    return fail(ASYRP_EINVAL, "hs_coeff needs index+2 entries (1 with an injected delta_h)");
  const asyrp_config& cf = e->cfg;
  if ((learn_sigma ? cf.out_channels / 2 : cf.out_channels) != 3 || cf.in_channels != 3)
    return fail(ASYRP_EINVAL, "DDIM step expects 3 image channels");
  HIPCHK(hipSetDevice(e->device));
  Ctx c{e, (hipStream_t)stream, B};
  PoolGuard guard{e};
  TRY(bind_stream(e, c.s));
  const int HW = cf.resolution * cf.resolution;
  float *xn, *xo, *x0o, *nz = nullptr;
  TRY(e->pool.get((size_t)B * HW * 3, &xn));
  TRY(e->pool.get((size_t)B * HW * 3, &xo));
  TRY(e->pool.get((size_t)B * HW * 3, &x0o));
  HIPCHK(launch_nchw_to_nhwc(xt, xn, B, 3, HW, c.s));
  if (eta != 0.f) {
    if (!noise) return fail(ASYRP_EINVAL, "eta != 0 requires a noise tensor");
    TRY(e->pool.get((size_t)B * HW * 3, &nz));
    HIPCHK(launch_nchw_to_nhwc(noise, nz, B, 3, HW, c.s));
  }
  hipLaunchKernelGGL(fill_kernel, dim3((B + 63) / 64), dim3(64), 0, c.s, e->d_t, (float)t, B);
  float* dh_staged = nullptr;
  TRY(stage_delta_in(c, delta_h_in, use_mask, index, apply_edit, &dh_staged));
  Act a_et, a_em, a_dh, a_mid;
  TRY(unet_core(c, xn, e->d_t, index, apply_edit, hs_coeff_host, ignore_timestep, &a_et, &a_em, &a_dh, &a_mid));
  if (dh_staged) e->pool.put(dh_staged);
  TRY(ddim_apply(c, xn, a_et, a_em, nz, t, t_next, eta, dt_lambda, dt_end, xo, x0o));
  HIPCHK(launch_nhwc_to_nchw(xo, 3, xt_next, B, 3, HW, c.s));
  if (x0_t) HIPCHK(launch_nhwc_to_nchw(x0o, 3, x0_t, B, 3, HW, c.s));
  if (a_dh.p && delta_h_out) HIPCHK(launch_nhwc_to_nchw(a_dh.p, a_dh.C, delta_h_out, B, a_dh.C, a_dh.H * a_dh.W, c.s));
  if (middle_h) HIPCHK(launch_nhwc_to_nchw(a_mid.p, a_mid.C, middle_h, B, a_mid.C, a_mid.H * a_mid.W, c.s));
  drop(c, a_et);
  if (a_em.p) drop(c, a_em);
  if (a_dh.p) drop(c, a_dh);
  drop(c, a_mid);
  e->pool.put(xn); e->pool.put(xo); e->pool.put(x0o);
  if (nz) e->pool.put(nz);
  return 0;
}

int asyrp_run_edit(asyrp_engine* e, const float* x0, int B, const int32_t* seq_inv, int n_inv, const int32_t* seq_gen,
                   int n_gen, int t_edit, int t_addnoise, int index, const float* hs_coeff_host, int n_coeff,
                   int learn_sigma, const float* noise, int n_noise, float* x_T, float* x_edit, void* stream) {
  TRY(check_ready(e, B));
  if (!x0 || !x_edit || n_gen < 1 || !seq_gen || (n_inv > 0 && !seq_inv)) return fail(ASYRP_EINVAL, "bad argument");
  if (index >= e->cfg.n_delta) return fail(ASYRP_EINVAL, "index >= number of DeltaBlocks (setattr_layers)");
  // n_coeff < 0 (round 5, ABI v8): -n_coeff == B * (index + 2) entries, ONE TUPLE PER IMAGE (an editing-strength sweep as batch entries)
  const bool per_image = n_coeff < 0;
  if (per_image) {
    if (index < 0 || !hs_coeff_host || -n_coeff != B * (index + 2)) return fail(ASYRP_EINVAL, "per-image hs_coeff: n_coeff must be -(B * (index + 2))");
    if (B > MIX_MAX_IMAGES) return fail(ASYRP_EINVAL, "per-image hs_coeff: at most 128 images per call");
  } else if (index >= 0 && (!hs_coeff_host || n_coeff < index + 2)) {
    return fail(ASYRP_EINVAL, "hs_coeff needs index+2 entries");
  }
  const asyrp_config& cf = e->cfg;
  if ((learn_sigma ? cf.out_channels / 2 : cf.out_channels) != 3 || cf.in_channels != 3)
    return fail(ASYRP_EINVAL, "edit loop expects 3 image channels");
  HIPCHK(hipSetDevice(e->device));
  Ctx c{e, (hipStream_t)stream, B};
  c.coeff_per_image = per_image ? 1 : 0;
  PoolGuard guard{e};
  TRY(bind_stream(e, c.s));
  const int HW = cf.resolution * cf.resolution;
  const size_t nx = (size_t)B * HW * 3;
  float *xa, *xb, *nz = nullptr;
  TRY(e->pool.get(nx, &xa));
  TRY(e->pool.get(nx, &xb));
  HIPCHK(launch_nchw_to_nhwc(x0, xa, B, 3, HW, c.s));
  // Every timestep of the edit is known here and all images of the batch share it: the timestep embedding + MLP and every block's
  // Linear(swish(temb)) of ALL steps are two launches now instead of two per UNet evaluation (158 per 39 + 40-step edit); step k
  // reads row k (Ctx::tproj_pre, row pitch 0 over the images).  Same arithmetic per row as the per-step launches: bit-identical.
  std::vector<float> ts;
  for (int k = 1; k < n_inv; ++k) ts.push_back((float)seq_inv[k - 1]);
  for (int k = n_gen - 1; k >= 0; --k) ts.push_back((float)seq_gen[k]);
  TimestepRows rows;
  TRY(precompute_timestep_rows(c, ts, &rows));
  float* const tproj_all = rows.tproj;
  int step_row = 0;
  // loop A — DDIM inversion (diffusion_latent.py:1034-1045): (t, t_next) = (seq[k-1], seq[k]), k = 1..n_inv-1
  for (int k = 1; k < n_inv; ++k) {
    const int t = seq_inv[k - 1], tn = seq_inv[k];
    c.tproj_pre = tproj_all + (size_t)(step_row++) * e->tproj_total;
    Act a_et, a_em, a_dh, a_mid;
    TRY(unet_core(c, xa, e->d_t, -1, 0, nullptr, 0, &a_et, &a_em, &a_dh, &a_mid));
    TRY(ddim_apply(c, xa, a_et, a_em, nullptr, t, tn, 0.f, 1.f, 999, xb, nullptr));
    drop(c, a_et);
    drop(c, a_mid);
    std::swap(xa, xb);
  }
  if (x_T) HIPCHK(launch_nhwc_to_nchw(xa, 3, x_T, B, 3, HW, c.s));
  // loop B — Asyrp generation (diffusion_latent.py:503-520): t descending, last t_next = -1
  int used_noise = 0;
  for (int k = n_gen - 1; k >= 0; --k) {
    const int t = seq_gen[k], tn = (k > 0) ? seq_gen[k - 1] : -1;
    const int edit = (index >= 0 && t >= t_edit) ? 1 : 0;
    const float eta = (t < t_addnoise) ? 1.0f : 0.0f;
    const float* nzp = nullptr;
    if (eta != 0.f) {
      if (!noise || used_noise >= n_noise) return fail(ASYRP_EINVAL, "not enough noise tensors for the eta=1 steps");
      if (!nz) TRY(e->pool.get(nx, &nz));
      HIPCHK(launch_nchw_to_nhwc(noise + (size_t)used_noise * nx, nz, B, 3, HW, c.s));
      ++used_noise;
      nzp = nz;
    }
    c.tproj_pre = tproj_all + (size_t)(step_row++) * e->tproj_total;
    Act a_et, a_em, a_dh, a_mid;
    TRY(unet_core(c, xa, e->d_t, index, edit, hs_coeff_host, 0, &a_et, &a_em, &a_dh, &a_mid));
    TRY(ddim_apply(c, xa, a_et, a_em, nzp, t, tn, eta, 1.f, 999, xb, nullptr));
    drop(c, a_et);
    if (a_em.p) drop(c, a_em);
    if (a_dh.p) drop(c, a_dh);
    drop(c, a_mid);
    std::swap(xa, xb);
  }
  HIPCHK(launch_nhwc_to_nchw(xa, 3, x_edit, B, 3, HW, c.s));
  e->pool.put(xa); e->pool.put(xb);
  if (nz) e->pool.put(nz);
  rows.release(e);
  return 0;
}

int asyrp_train_forward(asyrp_engine* e, const float* xt, int t, int t_next, int B, int learn_sigma,
                        const float* hs_coeff_host, int n_coeff, int ignore_timestep, float* xt_next, float* x0_t,
                        float* delta_h_out, float* middle_h, int64_t* tape_id, void* stream) {
  TRY(check_ready(e, B));
  if (!xt || !xt_next || !x0_t || !tape_id) return fail(ASYRP_EINVAL, "null tensor");
  if (e->cfg.n_delta < 1) return fail(ASYRP_EINVAL, "no DeltaBlock (setattr_layers)");
  if (!hs_coeff_host || n_coeff < 2) return fail(ASYRP_EINVAL, "hs_coeff needs 2 entries");
  const asyrp_config& cf = e->cfg;
  if ((learn_sigma ? cf.out_channels / 2 : cf.out_channels) != 3 || cf.in_channels != 3)
    return fail(ASYRP_EINVAL, "training step expects 3 image channels");
  if (e->np != 3) return fail(ASYRP_EINVAL, "the training step needs conv_math f16x3 or f32 (the single-product f16 mode is inference only)");
  HIPCHK(hipSetDevice(e->device));
  TRY(ensure_bwd_weights(e));
  Ctx c{e, (hipStream_t)stream, B};
  TRY(bind_stream(e, c.s));
  // a previous tape that was never consumed is discarded (its id stops being valid)
  e->pool.release_held();
  e->tape.clear();
  struct Guard {   // on any exit: stop deferring; on failure also drop what was held
    asyrp_engine* e; bool ok = false;
    ~Guard() { e->pool.defer = false; if (!ok) { e->pool.reclaim(); e->pool.release_held(); e->tape.clear(); } }
  } guard{e};
  c.tape = &e->tape;        // unet_core holds (pool.defer) what the backward pass reads, and only that
  e->tape.B = B;
  const int HW = cf.resolution * cf.resolution;
  float *xn, *xo, *x0o;
  TRY(e->pool.get((size_t)B * HW * 3, &xn));
  TRY(e->pool.get((size_t)B * HW * 3, &xo));
  TRY(e->pool.get((size_t)B * HW * 3, &x0o));
  HIPCHK(launch_nchw_to_nhwc(xt, xn, B, 3, HW, c.s));
  hipLaunchKernelGGL(fill_kernel, dim3((B + 63) / 64), dim3(64), 0, c.s, e->d_t, (float)t, B);
  Act a_et, a_em, a_dh, a_mid;
  TRY(unet_core(c, xn, e->d_t, 0, 1, hs_coeff_host, ignore_timestep, &a_et, &a_em, &a_dh, &a_mid));
  TRY(ddim_apply(c, xn, a_et, a_em, nullptr, t, t_next, 0.f, 1.f, 999, xo, x0o));
  HIPCHK(launch_nhwc_to_nchw(xo, 3, xt_next, B, 3, HW, c.s));
  HIPCHK(launch_nhwc_to_nchw(x0o, 3, x0_t, B, 3, HW, c.s));
  if (a_dh.p && delta_h_out) HIPCHK(launch_nhwc_to_nchw(a_dh.p, a_dh.C, delta_h_out, B, a_dh.C, a_dh.H * a_dh.W, c.s));
  if (middle_h) HIPCHK(launch_nhwc_to_nchw(a_mid.p, a_mid.C, middle_h, B, a_mid.C, a_mid.H * a_mid.W, c.s));
  e->pool.defer = false;      // eps, eps~, the DeltaBlock's output and the image buffers are not read by the backward pass
  drop(c, a_et);
  if (a_em.p) drop(c, a_em);
  if (a_dh.p) drop(c, a_dh);
  e->pool.put(xn); e->pool.put(xo); e->pool.put(x0o);
  e->pool.defer = true;       // the bottleneck h is (tape.d_h)
  drop(c, a_mid);
  e->pool.defer = false;
  e->pool.reclaim();          // (nothing should be live: every buffer was put -> free or held)
  e->tape.valid = true;
  e->tape.id = ++e->tape_gen;
  *tape_id = e->tape.id;
  guard.ok = true;
  return 0;
}

int asyrp_train_backward(asyrp_engine* e, int64_t tape_id, const float* d_et_mod, int n_grads, const char* const* keys,
                         float* const* grads, void* stream) {
  if (!e || !e->finalized) return fail(ASYRP_ESTATE, "asyrp_train_backward on an engine that is not ready");
  if (!e->tape.valid) return fail(ASYRP_ESTATE, "asyrp_train_backward without a preceding asyrp_train_forward");
  if (tape_id != e->tape.id)
    return fail(ASYRP_ESTATE, "asyrp_train_backward: stale tape id (a later asyrp_train_forward replaced the recorded step; the "
                              "engine keeps ONE pending step)");
  if (!d_et_mod || n_grads < 0 || (n_grads && (!keys || !grads))) return fail(ASYRP_EINVAL, "bad argument");
  {   // every requested key must be a parameter of layer_0 this pass produces a gradient for (checked BEFORE the tape is touched: a
      // refused call leaves the pending step intact -- ADVICE r03)
    const bool idd = e->cfg.family == ASYRP_FAMILY_IDDPM;
    static const char* const kd[] = {"layer_0.conv1.weight", "layer_0.conv1.bias", "layer_0.temb_proj.weight", "layer_0.temb_proj.bias",
                                     "layer_0.norm2.weight", "layer_0.norm2.bias", "layer_0.conv2.weight", "layer_0.conv2.bias"};
    static const char* const ki[] = {"layer_0.in_layers.0.weight", "layer_0.in_layers.0.bias", "layer_0.in_layers.2.weight",
                                     "layer_0.in_layers.2.bias", "layer_0.emb_layers.1.weight", "layer_0.emb_layers.1.bias",
                                     "layer_0.out_layers.0.weight", "layer_0.out_layers.0.bias", "layer_0.out_layers.3.weight",
                                     "layer_0.out_layers.3.bias"};
    for (int i = 0; i < n_grads; ++i) {
      bool known = false;
      if (keys[i] && grads[i])
        for (const char* k : idd ? std::vector<const char*>(ki, ki + 10) : std::vector<const char*>(kd, kd + 8))
          known = known || !strcmp(k, keys[i]);
      if (!known) return fail(ASYRP_EKEY, std::string("asyrp_train_backward: no gradient for key ") + (keys[i] ? keys[i] : "(null)"));
    }
  }
  Tape& tp = e->tape;
  const int B = tp.B;
  HIPCHK(hipSetDevice(e->device));
  Ctx c{e, (hipStream_t)stream, B};
  TRY(bind_stream(e, c.s));
  struct Guard {   // from here on the tape is consumed whatever happens
    asyrp_engine* e;
    ~Guard() { e->pool.reclaim(); e->pool.release_held(); e->tape.clear(); }
  } guard{e};
  auto out_ptr = [&](const std::string& key) -> float* {
    for (int i = 0; i < n_grads; ++i)
      if (key == keys[i]) return grads[i];
    return nullptr;
  };
  const asyrp_config& cf = e->cfg;
  const int R = cf.resolution, HW = R * R;
  // d(eps~) NCHW -> NHWC
  Act g;
  TRY(new_act(c, cf.out_channels, R, R, &g));
  HIPCHK(launch_nchw_to_nhwc(d_et_mod, g.p, B, cf.out_channels, HW, c.s));
  // conv_out^T, then norm_out + SiLU backward  (iDDPM: out.2 / out.0)
  const bool iddpm = cf.family == ASYRP_FAMILY_IDDPM;
  Act dA, d;
  TRY(conv_bwd_data(c, g, iddpm ? "out.2.weight" : "conv_out.weight", tp.out_h.C, 3, nullptr, &dA));
  drop(c, g);
  TRY(act_gn_backward(c, dA, tp.out_h, nullptr, tp.out_sc, tp.out_sh, tp.out_mr, iddpm ? "out.0" : "norm_out", 1, tp.out_h.C,
                      nullptr, &d));
  drop(c, dA);
  for (int k = (int)tp.order.size() - 1; k >= 0; --k) {
    const TapeEntry& en = tp.order[k];
    Act nd;
    if (en.type == 0) {
      TRY(resblock_backward(c, tp.res[en.idx], d, &nd));
    } else if (en.type == 1) {
      TRY(attnblock_backward(c, tp.attn[en.idx], d, &nd));
    } else {
      const TapeUp& u = tp.up[en.idx];     // y = conv3x3(nearest_x2(h)): transposed conv at the fine resolution, then 2x2 sums
      Act fine;
      TRY(conv_bwd_data(c, d, u.p + ".weight", u.C, 3, nullptr, &fine));
      TRY(new_act(c, u.C, u.H, u.W, &nd));
      HIPCHK(launch_sum2x2(fine.p, nd.p, B, u.H, u.W, u.C, c.s));
      drop(c, fine);
    }
    drop(c, d);
    d = nd;
  }
  // d = dL/dh2 [B, Rb, Rb, Cb];  h2 = c0*h + c1*Delta  ->  dDelta = c1 * d
  const int Cb = tp.d_h.C, Mb = B * tp.d_h.H * tp.d_h.W, HWb = tp.d_h.H * tp.d_h.W;
  Act dD;
  TRY(new_act(c, Cb, tp.d_h.H, tp.d_h.W, &dD));
  HIPCHK(launch_scale(d.p, tp.c1, dD.p, (long long)Mb * Cb, c.s));
  drop(c, d);
  // DeltaBlock.  DDPM (models/ddpm/diffusion.py:250-263):   d1 = conv1(h) + temb_proj(swish(temb)); a = swish(GN(d1)); Delta = conv2(a)
  //             iDDPM (models/improved_ddpm/unet.py:835-853): d1 = conv1(swish(GN0(h))) + emb_layers(emb);   a = swish(GN(d1)); Delta = conv2(a)
  const char* k_c1 = iddpm ? "layer_0.in_layers.2" : "layer_0.conv1";
  const char* k_c2 = iddpm ? "layer_0.out_layers.3" : "layer_0.conv2";
  const char* k_n2 = iddpm ? "layer_0.out_layers.0" : "layer_0.norm2";
  const char* k_tp = iddpm ? "layer_0.emb_layers.1" : "layer_0.temb_proj";
  auto K = [](const char* a, const char* b) { return std::string(a) + b; };
  Act a_act;
  TRY(new_act(c, Cb, tp.d_h.H, tp.d_h.W, &a_act));
  HIPCHK(launch_act_apply(tp.d_d1.p, tp.d_sc, tp.d_sh, 1, a_act.p, B, HWb, Cb, c.s));
  if (float* o = out_ptr(K(k_c2, ".weight"))) TRY(weight_grad(c, dD.p, Cb, a_act.p, Cb, Mb, o));
  if (float* o = out_ptr(K(k_c2, ".bias"))) HIPCHK(launch_colsum(dD.p, Cb, Mb, Cb, o, c.s));
  drop(c, a_act);
  Act da;
  TRY(conv_bwd_data(c, dD, K(k_c2, ".weight"), Cb, 1, nullptr, &da));
  drop(c, dD);
  // GroupNorm + SiLU backward on d1, keeping the partial sums for the norm's own parameter gradients
  {
    const int nblk = act_bwd_nblk(HWb);
    Act dy, d_d1;
    float *part, *coef;
    TRY(new_act(c, Cb, tp.d_h.H, tp.d_h.W, &dy));
    TRY(e->pool.get((size_t)B * nblk * Cb * 4, &part));
    TRY(e->pool.get((size_t)B * Cb * 3, &coef));
    ActBwdArgs a;
    memset(&a, 0, sizeof a);
    a.dA = da.p; a.ldd = Cb; a.x0 = tp.d_d1.p; a.c0 = Cb; a.ldx0 = Cb; a.x0_z = tp.d_d1.per_image();
    a.scale = tp.d_sc; a.shift = tp.d_sh; a.silu = 1; a.dy = dy.p; a.partial = reinterpret_cast<double*>(part);
    a.HW = HWb; a.N = B; a.C = Cb;
    HIPCHK(launch_act_bwd_partial(a, c.s));
    float* dgam = out_ptr(K(k_n2, ".weight"));
    float* dbet = out_ptr(K(k_n2, ".bias"));
    if (dgam || dbet) {   // the kernel writes the pair: an unrequested half goes to scratch
      float* scratch = nullptr;
      if (!dgam || !dbet) TRY(e->pool.get((size_t)Cb, &scratch));
      HIPCHK(launch_gn_param_grad(a.partial, nblk, tp.d_mr, B, Cb, dgam ? dgam : scratch, dbet ? dbet : scratch, c.s));
      if (scratch) e->pool.put(scratch);
    }
    GnBwdFinArgs f;
    memset(&f, 0, sizeof f);
    f.partial = a.partial; f.nblk = nblk; f.gamma = P(c, K(k_n2, ".weight")); f.mr = tp.d_mr; f.N = B; f.HW = HWb; f.C = Cb;
    f.coef = coef;
    HIPCHK(launch_gn_bwd_finalize(f, c.s));
    TRY(new_act(c, Cb, tp.d_h.H, tp.d_h.W, &d_d1));
    HIPCHK(launch_gn_bwd_apply(dy.p, Cb, tp.d_d1.p, Cb, tp.d_d1.per_image(), coef, nullptr, d_d1.p, Cb, HWb, B, c.s));
    drop(c, dy);
    drop(c, da);
    // conv1's input: h itself (DDPM) or swish(GN0(h)) (iDDPM)
    if (iddpm) {
      Act a0;
      TRY(new_act(c, Cb, tp.d_h.H, tp.d_h.W, &a0));
      HIPCHK(launch_act_apply(tp.d_h.p, tp.d_sc0, tp.d_sh0, 1, a0.p, B, HWb, Cb, c.s));
      if (float* o = out_ptr(K(k_c1, ".weight"))) TRY(weight_grad(c, d_d1.p, Cb, a0.p, Cb, Mb, o));
      drop(c, a0);
      // GN0 parameter gradients: dA0 = conv1^T(d_d1), then the partial sums of dA0 * swish'(GN0(h)) against h
      float* dg0 = out_ptr("layer_0.in_layers.0.weight");
      float* db0 = out_ptr("layer_0.in_layers.0.bias");
      if (dg0 || db0) {
        Act dA0, dy0;
        float* scratch0 = nullptr;
        if (!dg0 || !db0) TRY(e->pool.get((size_t)Cb, &scratch0));
        TRY(conv_bwd_data(c, d_d1, K(k_c1, ".weight"), Cb, 1, nullptr, &dA0));
        TRY(new_act(c, Cb, tp.d_h.H, tp.d_h.W, &dy0));
        ActBwdArgs a2;
        memset(&a2, 0, sizeof a2);
        a2.dA = dA0.p; a2.ldd = Cb; a2.x0 = tp.d_h.p; a2.c0 = Cb; a2.ldx0 = Cb; a2.x0_z = tp.d_h.per_image();
        a2.scale = tp.d_sc0; a2.shift = tp.d_sh0; a2.silu = 1; a2.dy = dy0.p; a2.partial = reinterpret_cast<double*>(part);
        a2.HW = HWb; a2.N = B; a2.C = Cb;
        HIPCHK(launch_act_bwd_partial(a2, c.s));
        HIPCHK(launch_gn_param_grad(a2.partial, nblk, tp.d_mr0, B, Cb, dg0 ? dg0 : scratch0, db0 ? db0 : scratch0, c.s));
        if (scratch0) e->pool.put(scratch0);
        drop(c, dA0);
        drop(c, dy0);
      }
    } else {
      if (float* o = out_ptr(K(k_c1, ".weight"))) TRY(weight_grad(c, d_d1.p, Cb, tp.d_h.p, Cb, Mb, o));
    }
    e->pool.put(part); e->pool.put(coef);
    if (float* o = out_ptr(K(k_c1, ".bias"))) HIPCHK(launch_colsum(d_d1.p, Cb, Mb, Cb, o, c.s));
    float* dtw = out_ptr(K(k_tp, ".weight"));
    float* dtb = out_ptr(K(k_tp, ".bias"));
    if (tp.d_temb) {
      if (dtb) HIPCHK(launch_colsum(d_d1.p, Cb, Mb, Cb, dtb, c.s));
      if (dtw) {   // d_tp[b][co] = sum_pix d_d1[b][pix][co];  dW[co][k] = sum_b d_tp[b][co] * swish(temb)[b][k]
        float* dtp;
        TRY(e->pool.get((size_t)B * Cb, &dtp));
        for (int b = 0; b < B; ++b)
          HIPCHK(launch_colsum(d_d1.p + (size_t)b * HWb * Cb, Cb, HWb, Cb, dtp + (size_t)b * Cb, c.s));
        TRY(weight_grad(c, dtp, Cb, tp.temb_act, e->temb_ch, B, dtw));
        e->pool.put(dtp);
      }
    } else {      // ignore_timestep: temb is None (:253-254), the projection does not take part
      if (dtw) HIPCHK(hipMemsetAsync(dtw, 0, sizeof(float) * (size_t)Cb * e->temb_ch, c.s));
      if (dtb) HIPCHK(hipMemsetAsync(dtb, 0, sizeof(float) * Cb, c.s));
    }
    drop(c, d_d1);
  }
  return 0;
}

void asyrp_train_discard(asyrp_engine* e, int64_t tape_id) {
  if (!e || !e->tape.valid) return;
  if (tape_id >= 0 && tape_id != e->tape.id) return;   // a newer step replaced it: nothing of `tape_id` is held any more
  e->pool.release_held();
  e->tape.clear();
}

// DDIM inversion with a per-step read-out: the engine half of the reference's LPIPS(t) table builder
// (diffusion_latent.py:1239-1276 runs denoising_step over seq_inv and feeds x and x0_t of EVERY step to LPIPS).
int asyrp_run_inversion(asyrp_engine* e, const float* x0, int B, const int32_t* seq_inv, int n_inv, int learn_sigma,
                        int tap_first, int tap_count, float* x_tap, float* x0t_tap, float* x_last, void* stream) {
  TRY(check_ready(e, B));
  if (!x0 || !seq_inv || n_inv < 2) return fail(ASYRP_EINVAL, "bad argument");
  if (tap_count < 0 || tap_first < 0 || tap_first + tap_count > n_inv - 1)
    return fail(ASYRP_EINVAL, "tap window outside the n_inv-1 inversion steps");
  if (tap_count > 0 && !x_tap && !x0t_tap) return fail(ASYRP_EINVAL, "tap window without a tap buffer");
  const asyrp_config& cf = e->cfg;
  if ((learn_sigma ? cf.out_channels / 2 : cf.out_channels) != 3 || cf.in_channels != 3)
    return fail(ASYRP_EINVAL, "inversion loop expects 3 image channels");
  HIPCHK(hipSetDevice(e->device));
  Ctx c{e, (hipStream_t)stream, B};
  PoolGuard guard{e};
  TRY(bind_stream(e, c.s));
  const int HW = cf.resolution * cf.resolution;
  const size_t nx = (size_t)B * HW * 3;
  float *xa, *xb, *x0o = nullptr;
  TRY(e->pool.get(nx, &xa));
  TRY(e->pool.get(nx, &xb));
  if (x0t_tap) TRY(e->pool.get(nx, &x0o));
  HIPCHK(launch_nchw_to_nhwc(x0, xa, B, 3, HW, c.s));
  std::vector<float> ts;
  for (int k = 1; k < n_inv; ++k) ts.push_back((float)seq_inv[k - 1]);
  TimestepRows rows;
  TRY(precompute_timestep_rows(c, ts, &rows));
  for (int k = 1; k < n_inv; ++k) {
    const int t = seq_inv[k - 1], tn = seq_inv[k];
    c.tproj_pre = rows.tproj + (size_t)(k - 1) * e->tproj_total;
    Act a_et, a_em, a_dh, a_mid;
    TRY(unet_core(c, xa, e->d_t, -1, 0, nullptr, 0, &a_et, &a_em, &a_dh, &a_mid));
    const int slot = (k - 1) - tap_first;
    const bool tap = slot >= 0 && slot < tap_count;
    TRY(ddim_apply(c, xa, a_et, a_em, nullptr, t, tn, 0.f, 1.f, 999, xb, (tap && x0t_tap) ? x0o : nullptr));
    drop(c, a_et);
    drop(c, a_mid);
    std::swap(xa, xb);
    if (tap && x_tap) HIPCHK(launch_nhwc_to_nchw(xa, 3, x_tap + (size_t)slot * nx, B, 3, HW, c.s));
    if (tap && x0t_tap) HIPCHK(launch_nhwc_to_nchw(x0o, 3, x0t_tap + (size_t)slot * nx, B, 3, HW, c.s));
  }
  if (x_last) HIPCHK(launch_nhwc_to_nchw(xa, 3, x_last, B, 3, HW, c.s));
  rows.release(e);
  return 0;
}

// DDPM.get_temb (models/ddpm/diffusion.py:464-470): temb = dense1(swish(dense0(sinusoid(t)))); for the iDDPM family the
// same network is `time_embed(timestep_embedding(t))` (models/improved_ddpm/unet.py:688).
int asyrp_get_temb(asyrp_engine* e, const float* t, int B, float* temb_out, void* stream) {
  TRY(check_ready(e, B));
  if (!t || !temb_out) return fail(ASYRP_EINVAL, "null tensor");
  HIPCHK(hipSetDevice(e->device));
  Ctx c{e, (hipStream_t)stream, B};
  PoolGuard guard{e};
  TRY(bind_stream(e, c.s));
  float* act = nullptr;
  TRY(e->pool.get((size_t)B * e->temb_ch, &act));
  const bool iddpm = e->cfg.family == ASYRP_FAMILY_IDDPM;
  const char* n0 = iddpm ? "time_embed.0" : "temb.dense.0";
  const char* n1 = iddpm ? "time_embed.2" : "temb.dense.1";
  HIPCHK(launch_temb_mlp(t, e->d_freqs, e->n_freqs, iddpm ? 0 : 1, P(c, std::string(n0) + ".weight"),
                         P(c, std::string(n0) + ".bias"), P(c, std::string(n1) + ".weight"), P(c, std::string(n1) + ".bias"),
                         e->cfg.ch, e->temb_ch, temb_out, act, B, c.s));
  return 0;
}

int asyrp_profile_enable(asyrp_engine* e, int on) {
  if (!e) return fail(ASYRP_EINVAL, "null engine");
  e->prof_on = on != 0;
  return 0;
}

int asyrp_profile_read(asyrp_engine* e, int* variant, double* ms, int64_t* launches, double* flops, double* bytes,
                       double* all_ms, double* all_flops) {
  if (!e) return fail(ASYRP_EINVAL, "null engine");
  HIPCHK(hipSetDevice(e->device));
  HIPCHK(hipDeviceSynchronize());
  struct Acc { double ms = 0, fl = 0, by = 0; int64_t n = 0; };
  std::map<int, Acc> acc;
  double tms = 0, tfl = 0;
  for (auto& r : e->prof) {
    float t = 0.f;
    HIPCHK(hipEventElapsedTime(&t, r.a, r.b));
    Acc& a = acc[r.variant];
    a.ms += t; a.fl += r.flops; a.by += r.bytes; a.n += 1;
    tms += t; tfl += r.flops;
    e->ev_free.emplace_back(r.a, r.b);
  }
  e->prof.clear();
  int best = 0;
  Acc b;
  for (auto& kv : acc)
    if (kv.second.ms > b.ms) { b = kv.second; best = kv.first; }
  if (variant) *variant = best;
  if (ms) *ms = b.ms;
  if (launches) *launches = b.n;
  if (flops) *flops = b.fl;
  if (bytes) *bytes = b.by;
  if (all_ms) *all_ms = tms;
  if (all_flops) *all_flops = tfl;
  return 0;
}

// Every kernel family recorded since the last read, one row each (bench.py's per-family table).  Does NOT reset the
// record: call asyrp_profile_read afterwards.  Returns the number of rows written (<= max_rows), negative on error.
int asyrp_profile_table(asyrp_engine* e, int max_rows, int* variants, double* ms, int64_t* launches, double* flops,
                        double* bytes) {
  if (!e || max_rows < 0) return fail(ASYRP_EINVAL, "bad argument");
  HIPCHK(hipSetDevice(e->device));
  HIPCHK(hipDeviceSynchronize());
  struct Acc { double ms = 0, fl = 0, by = 0; int64_t n = 0; };
  std::map<int, Acc> acc;
  for (auto& r : e->prof) {
    float t = 0.f;
    HIPCHK(hipEventElapsedTime(&t, r.a, r.b));
    Acc& a = acc[r.variant];
    a.ms += t; a.fl += r.flops; a.by += r.bytes; a.n += 1;
  }
  int n = 0;
  for (auto& kv : acc) {
    if (n >= max_rows) break;
    if (variants) variants[n] = kv.first;
    if (ms) ms[n] = kv.second.ms;
    if (launches) launches[n] = kv.second.n;
    if (flops) flops[n] = kv.second.fl;
    if (bytes) bytes[n] = kv.second.by;
    ++n;
  }
  return n;
}

// ---------------------------------------------------------------------------------------------------
// op-level test hooks (synchronous; allocate scratch with hipMalloc)
// ---------------------------------------------------------------------------------------------------
__global__ void pack_conv_weight_kernel(const float* src, float* dst, int cout, int cin, int kk) {
  const long long total = (long long)cout * cin * kk;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int t = (int)(i % kk);
    const long long r = i / kk;
    const int ci = (int)(r % cin), co = (int)(r / cin);
    dst[((long long)t * cin + ci) * cout + co] = src[i];
  }
}

int asyrp_op_conv2d(int device, const float* x0, int C0, const float* x1, int C1, int B, int H, int W,
                    const float* weight, const float* bias, int Cout, int ksize, int stride, int upsample,
                    const float* gn_weight, const float* gn_bias, float gn_eps, int silu, const float* chan_add,
                    const float* residual, float* y, int conv_math, int tile, void* stream) {
  if (!x0 || !weight || !y || B < 1) return fail(ASYRP_EINVAL, "bad argument");
  HIPCHK(hipSetDevice(device));
  hipStream_t s = (hipStream_t)stream;
  const int Cin = C0 + (x1 ? C1 : 0), HW = H * W;
  int Ho = H, Wo = W;
  if (upsample) { Ho *= 2; Wo *= 2; }
  if (stride == 2) { Ho /= 2; Wo /= 2; }
  std::vector<void*> tmp;
  auto dalloc = [&](size_t nfloats, float** p) -> int {
    HIPCHK(hipMalloc(p, std::max<size_t>(nfloats, 1) * sizeof(float)));
    tmp.push_back(*p);
    return 0;
  };
  float *a0, *a1 = nullptr, *wp, *yo, *rs = nullptr, *sc = nullptr, *sh = nullptr;
  TRY(dalloc((size_t)B * HW * C0, &a0));
  HIPCHK(launch_nchw_to_nhwc(x0, a0, B, C0, HW, s));
  if (x1) {
    TRY(dalloc((size_t)B * HW * C1, &a1));
    HIPCHK(launch_nchw_to_nhwc(x1, a1, B, C1, HW, s));
  }
  TRY(dalloc((size_t)ksize * ksize * Cin * Cout, &wp));
  hipLaunchKernelGGL(pack_conv_weight_kernel, dim3(256), dim3(256), 0, s, weight, wp, Cout, Cin, ksize * ksize);
  TRY(dalloc((size_t)B * Ho * Wo * Cout, &yo));
  if (residual) {
    TRY(dalloc((size_t)B * Ho * Wo * Cout, &rs));
    HIPCHK(launch_nchw_to_nhwc(residual, rs, B, Cout, Ho * Wo, s));
  }
  if (gn_weight) {
    float* part;
    TRY(dalloc((size_t)B * Cin, &sc));
    TRY(dalloc((size_t)B * Cin, &sh));
    TRY(dalloc(gn_partial_doubles(B, Cin, HW) * 2, &part));
    GnArgs a;
    memset(&a, 0, sizeof a);
    a.a0 = a0; a.c0 = C0; a.lda0 = C0; a.a0_z = (long long)HW * C0;
    if (x1) { a.a1 = a1; a.c1 = C1; a.lda1 = C1; a.a1_z = (long long)HW * C1; }
    a.HW = HW; a.N = B; a.C = Cin; a.gamma = gn_weight; a.beta = gn_bias; a.eps = gn_eps;
    a.scale = sc; a.shift = sh; a.partial = reinterpret_cast<double*>(part);
    HIPCHK(launch_gn(a, s));
  }
  GemmArgs g;
  memset(&g, 0, sizeof g);
  g.a0 = a0; g.c0 = C0; g.lda0 = C0; g.a0_zo = (long long)HW * C0;
  if (x1) { g.a1 = a1; g.c1 = C1; g.lda1 = C1; g.a1_zo = (long long)HW * C1; }
  g.Hin = H; g.Win = W; g.Hout = Ho; g.Wout = Wo; g.Cin = Cin; g.Cout = Cout;
  g.ks = ksize; g.stride = stride; g.ups = upsample; g.pad = (ksize == 3 && stride == 1) ? 1 : 0;
  g.pscale = sc; g.pshift = sh; g.silu = silu;
  g.w = wp; g.ldb = Cout; g.bias = bias;
  g.chan_add = chan_add; g.ld_chan_add = Cout;
  if (rs) { g.resid = rs; g.ldr = Cout; g.r_zo = (long long)Ho * Wo * Cout; }
  g.alpha = 1.f; g.out = yo; g.ldo = Cout; g.o_zo = (long long)Ho * Wo * Cout; g.ZI = 1; g.Z = B;
  g.math = MATH_F32;
  g.tile = tile;
  if (conv_math == ASYRP_MATH_F16X3 || conv_math == ASYRP_MATH_F16) {
    g.np = (conv_math == ASYRP_MATH_F16) ? 1 : 3;
    std::vector<float> hw((size_t)Cout * Cin * ksize * ksize);
    HIPCHK(hipMemcpy(hw.data(), weight, hw.size() * sizeof(float), hipMemcpyDeviceToHost));
    float mx = 0.f;
    for (float v : hw) mx = std::max(mx, std::fabs(v));
    const float wscale = (mx > 0.f && std::isfinite(mx)) ? std::ldexp(1.0f, 10 - (int)std::floor(std::log2(mx))) : 1.f;
    float* xp;
    TRY(dalloc((f16x3_packed_halfs(Cout, Cin, ksize) + 1) / 2, &xp));
    HIPCHK(launch_pack_f16x3(weight, xp, Cout, Cin, ksize, wscale, s));
    g.math = MATH_F16X3;
    g.wpk = xp;
    g.cout_pad = ((Cout + 127) / 128) * 128;
    g.alpha = 1.0f / (wscale * f16x3_act_scale());
    if (tile == XT_256x128K32UP) {   // polyphase form of "nearest x2 then 3x3": the four phase-collapsed 2x2 images
      if (ksize != 3 || stride != 1 || !upsample || (Cin & 31) || rs) {
        for (void* p : tmp) (void)hipFree(p);
        return fail(ASYRP_EINVAL, "shape not covered by the polyphase tile (3x3, stride 1, upsample, Cin % 32 == 0, no residual)");
      }
      const std::vector<float> wp = polyphase_weights(hw.data(), Cout, Cin);
      float mp = 0.f;
      for (float v : wp) mp = std::max(mp, std::fabs(v));
      const float ps = (mp > 0.f && std::isfinite(mp)) ? std::ldexp(1.0f, 10 - (int)std::floor(std::log2(mp))) : 1.f;
      const size_t ph = f16x3_packed_halfs(Cout, Cin, 2);
      float *wpd, *xpu;
      TRY(dalloc(wp.size(), &wpd));
      HIPCHK(hipMemcpy(wpd, wp.data(), wp.size() * sizeof(float), hipMemcpyHostToDevice));
      TRY(dalloc((4 * ph + 1) / 2, &xpu));
      for (int q = 0; q < 4; ++q)
        HIPCHK(launch_pack_f16x3(wpd + (size_t)q * Cout * Cin * 4, reinterpret_cast<char*>(xpu) + (size_t)q * ph * 2, Cout, Cin, 2, ps, s));
      g.poly = 1; g.ups = 0; g.Hout = H; g.Wout = W; g.tile = 0;
      g.wpk = xpu; g.w_phase = (long long)ph * 2;
      g.alpha = 1.0f / (ps * f16x3_act_scale());
    }
    if (tile == XT_G1_256 || tile == XT_G1_128) {   // gemm1x1.hip: its fragment-major weight image
      if (!gemm1x1_ok(g)) {
        for (void* p : tmp) (void)hipFree(p);
        return fail(ASYRP_EINVAL, "shape not covered by the 1x1 kernel (1x1, stride 1, Cin % 64 == 0, concat split % 32 == 0)");
      }
      float* xg;
      TRY(dalloc((gemm1x1_packed_halfs(Cout, Cin) + 1) / 2, &xg));
      HIPCHK(launch_gemm1x1_pack(weight, xg, Cout, Cin, wscale, s));
      g.wpk = xg;
    }
    if (tile == XT_CONV_IN) {   // the first-convolution stencil (conv_in.hip): fp32 weights [tap][Cin][Cout] = `wp`
      g.tile = 0; g.alpha = 1.f; g.cin_wmul = wscale;
      if (!conv_in_supported(g)) {
        for (void* p : tmp) (void)hipFree(p);
        return fail(ASYRP_EINVAL, "shape not covered by the conv_in kernel (3 input channels, 3x3, no prologue / residual / channel vector)");
      }
      hipError_t le = launch_conv_in(g, s);
      if (le == hipSuccess) le = launch_nhwc_to_nchw(yo, Cout, y, B, Cout, Ho * Wo, s);
      hipError_t se = hipStreamSynchronize(s);
      for (void* p : tmp) (void)hipFree(p);
      if (le != hipSuccess) return fail(ASYRP_EHIP, std::string("conv_in launch: ") + hipGetErrorString(le));
      if (se != hipSuccess) return fail(ASYRP_EHIP, std::string("conv_in sync: ") + hipGetErrorString(se));
      return 0;
    }
    if (tile == 13) {   // the taps-in-N kernel of the UNet's last conv (conv_out.hip): its own weight image
      std::vector<float> w1((size_t)9 * Cout * Cin);
      for (int co = 0; co < Cout; ++co)
        for (int ci = 0; ci < Cin; ++ci)
          for (int t = 0; t < 9; ++t) w1[((size_t)(t * Cout + co)) * Cin + ci] = hw[((size_t)co * Cin + ci) * 9 + t];
      float *w1d, *xp1;
      TRY(dalloc(w1.size(), &w1d));
      HIPCHK(hipMemcpy(w1d, w1.data(), w1.size() * sizeof(float), hipMemcpyHostToDevice));
      TRY(dalloc((f16x3_packed_halfs(9 * Cout, Cin, 1) + 1) / 2, &xp1));
      HIPCHK(launch_pack_f16x3(w1d, xp1, 9 * Cout, Cin, 1, wscale, s));
      g.wpk = xp1;
      g.cout_pad = ((9 * Cout + 127) / 128) * 128;
      g.tile = 0;
      if (ksize != 3 || !conv_out_supported(g)) {
        for (void* p : tmp) (void)hipFree(p);
        return fail(ASYRP_EINVAL, "shape not covered by the conv_out kernel");
      }
      hipError_t le = launch_conv_out(g, s);
      if (le == hipSuccess) le = launch_nhwc_to_nchw(yo, Cout, y, B, Cout, Ho * Wo, s);
      hipError_t se = hipStreamSynchronize(s);
      for (void* p : tmp) (void)hipFree(p);
      if (le != hipSuccess) return fail(ASYRP_EHIP, std::string("conv_out launch: ") + hipGetErrorString(le));
      if (se != hipSuccess) return fail(ASYRP_EHIP, std::string("conv_out sync: ") + hipGetErrorString(se));
      return 0;
    }
  }
  // the launcher's own choice includes split-K for the 8 x 8 layers (as in the engine's conv()): tile 0 only
  const int sk = (tile == 0 && g.math == MATH_F16X3) ? splitk_factor(g) : 1;
  if (sk > 1) {
    g.sk = sk;
    g.tile = splitk_tile(g);
    TRY(dalloc((size_t)sk * B * Ho * Wo * Cout, &g.part));
  }
  hipError_t le = launch_gemm(g, s);
  if (le == hipSuccess && sk > 1) le = launch_splitk_reduce(g, s);
  if (le == hipSuccess) le = launch_nhwc_to_nchw(yo, Cout, y, B, Cout, Ho * Wo, s);
  hipError_t se = hipStreamSynchronize(s);
  for (void* p : tmp) (void)hipFree(p);
  if (le != hipSuccess) return fail(ASYRP_EHIP, std::string("conv launch: ") + hipGetErrorString(le));
  if (se != hipSuccess) return fail(ASYRP_EHIP, std::string("conv sync: ") + hipGetErrorString(se));
  return 0;
}

// conv (f16x3) with the fused GroupNorm-statistics epilogue, then the finalize kernel: y and the per-(image,channel)
// scale/shift the NEXT layer's GroupNorm32(gamma, beta, eps) would apply to y
int asyrp_op_conv2d_stats(int device, const float* x, int Cin, int B, int H, int W, const float* weight,
                          const float* bias, int Cout, int ksize, int tile, const float* gamma, const float* beta,
                          float eps, float* y, float* scale_out, float* shift_out, void* stream) {
  if (!x || !weight || !y || !gamma || !beta || !scale_out || !shift_out || B < 1) return fail(ASYRP_EINVAL, "bad argument");
  HIPCHK(hipSetDevice(device));
  hipStream_t s = (hipStream_t)stream;
  const int HW = H * W;
  std::vector<void*> tmp;
  auto dalloc = [&](size_t nfloats, float** p) -> int {
    HIPCHK(hipMalloc(p, std::max<size_t>(nfloats, 1) * sizeof(float)));
    tmp.push_back(*p);
    return 0;
  };
  float *a0, *yo, *xp, *st;
  TRY(dalloc((size_t)B * HW * Cin, &a0));
  HIPCHK(launch_nchw_to_nhwc(x, a0, B, Cin, HW, s));
  TRY(dalloc((size_t)B * HW * Cout, &yo));
  std::vector<float> hw((size_t)Cout * Cin * ksize * ksize);
  HIPCHK(hipMemcpy(hw.data(), weight, hw.size() * sizeof(float), hipMemcpyDeviceToHost));
  float mx = 0.f;
  for (float v : hw) mx = std::max(mx, std::fabs(v));
  const float wscale = (mx > 0.f && std::isfinite(mx)) ? std::ldexp(1.0f, 10 - (int)std::floor(std::log2(mx))) : 1.f;
  TRY(dalloc((f16x3_packed_halfs(Cout, Cin, ksize) + 1) / 2, &xp));
  HIPCHK(launch_pack_f16x3(weight, xp, Cout, Cin, ksize, wscale, s));
  GemmArgs g;
  memset(&g, 0, sizeof g);
  g.a0 = a0; g.c0 = Cin; g.lda0 = Cin; g.a0_zo = (long long)HW * Cin;
  g.Hin = H; g.Win = W; g.Hout = H; g.Wout = W; g.Cin = Cin; g.Cout = Cout;
  g.ks = ksize; g.stride = 1; g.pad = ksize == 3 ? 1 : 0;
  g.bias = bias; g.out = yo; g.ldo = Cout; g.o_zo = (long long)HW * Cout; g.ZI = 1; g.Z = B;
  g.math = MATH_F16X3; g.tile = tile; g.wpk = xp; g.cout_pad = ((Cout + 127) / 128) * 128;
  g.alpha = 1.0f / (wscale * f16x3_act_scale());
  if (tile == XT_G1_256 || tile == XT_G1_128) {
    if (!gemm1x1_ok(g)) {
      for (void* p : tmp) (void)hipFree(p);
      return fail(ASYRP_EINVAL, "shape not covered by the 1x1 kernel");
    }
    float* xg;
    TRY(dalloc((gemm1x1_packed_halfs(Cout, Cin) + 1) / 2, &xg));
    HIPCHK(launch_gemm1x1_pack(weight, xg, Cout, Cin, wscale, s));
    g.wpk = xg;
  }
  const bool cin_kernel = (tile == XT_CONV_IN);   // conv_in.hip: fp32 weights [tap][Cin][Cout], its own statistics rows
  if (cin_kernel) {
    float* wp;
    TRY(dalloc((size_t)ksize * ksize * Cin * Cout, &wp));
    hipLaunchKernelGGL(pack_conv_weight_kernel, dim3(256), dim3(256), 0, s, weight, wp, Cout, Cin, ksize * ksize);
    g.w = wp; g.ldb = Cout; g.tile = 0; g.alpha = 1.f; g.cin_wmul = wscale;
    if (!conv_in_supported(g)) {
      for (void* p : tmp) (void)hipFree(p);
      return fail(ASYRP_EINVAL, "shape not covered by the conv_in kernel");
    }
  }
  const int nblk = cin_kernel ? conv_in_stat_blocks(g) : gemm_mblocks(g);
  TRY(dalloc((size_t)B * nblk * Cout * 4, &st));
  g.stats = reinterpret_cast<double*>(st);
  hipError_t le = cin_kernel ? launch_conv_in(g, s) : launch_gemm(g, s);
  GnFin2Args f;
  memset(&f, 0, sizeof f);
  f.p0 = g.stats; f.nblk0 = nblk; f.C0 = Cout; f.N = B; f.HW = HW; f.gamma = gamma; f.beta = beta; f.eps = eps;
  f.scale = scale_out; f.shift = shift_out;
  if (le == hipSuccess) le = launch_gn_finalize2(f, s);
  if (le == hipSuccess) le = launch_nhwc_to_nchw(yo, Cout, y, B, Cout, HW, s);
  hipError_t se = hipStreamSynchronize(s);
  for (void* p : tmp) (void)hipFree(p);
  if (le != hipSuccess) return fail(ASYRP_EHIP, std::string("conv+stats launch: ") + hipGetErrorString(le));
  if (se != hipSuccess) return fail(ASYRP_EHIP, std::string("conv+stats sync: ") + hipGetErrorString(se));
  return 0;
}

// y = conv3x3(swish(GroupNorm32(h))) + b3 + conv1x1(cat(x0, x1)) + b1 in ONE launch (fused shortcut on the main f16x3 tile):
// the tail of a ResnetBlock with nin_shortcut (models/ddpm/diffusion.py:159-170)
int asyrp_op_resblock_tail(int device, const float* h, int Ch, const float* x0, int C0, const float* x1, int C1, int B, int H,
                           int W, const float* w3, const float* b3, const float* w1, const float* b1, int Cout,
                           const float* gn_weight, const float* gn_bias, float gn_eps, float* y, void* stream) {
  if (!h || !x0 || !w3 || !w1 || !y || !gn_weight || !gn_bias || B < 1) return fail(ASYRP_EINVAL, "bad argument");
  HIPCHK(hipSetDevice(device));
  hipStream_t s = (hipStream_t)stream;
  const int HW = H * W, Cx = C0 + (x1 ? C1 : 0);
  std::vector<void*> tmp;
  auto dalloc = [&](size_t nfloats, float** p) -> int {
    HIPCHK(hipMalloc(p, std::max<size_t>(nfloats, 1) * sizeof(float)));
    tmp.push_back(*p);
    return 0;
  };
  float *hn, *a0, *a1 = nullptr, *yo, *xp, *sc, *sh, *part, *fb;
  TRY(dalloc((size_t)B * HW * Ch, &hn));
  HIPCHK(launch_nchw_to_nhwc(h, hn, B, Ch, HW, s));
  TRY(dalloc((size_t)B * HW * C0, &a0));
  HIPCHK(launch_nchw_to_nhwc(x0, a0, B, C0, HW, s));
  if (x1) {
    TRY(dalloc((size_t)B * HW * C1, &a1));
    HIPCHK(launch_nchw_to_nhwc(x1, a1, B, C1, HW, s));
  }
  TRY(dalloc((size_t)B * HW * Cout, &yo));
  TRY(dalloc((size_t)B * Ch, &sc));
  TRY(dalloc((size_t)B * Ch, &sh));
  TRY(dalloc(gn_partial_doubles(B, Ch, HW) * 2, &part));
  GnArgs ga;
  memset(&ga, 0, sizeof ga);
  ga.a0 = hn; ga.c0 = Ch; ga.lda0 = Ch; ga.a0_z = (long long)HW * Ch; ga.HW = HW; ga.N = B; ga.C = Ch;
  ga.gamma = gn_weight; ga.beta = gn_bias; ga.eps = gn_eps; ga.scale = sc; ga.shift = sh;
  ga.partial = reinterpret_cast<double*>(part);
  HIPCHK(launch_gn(ga, s));
  std::vector<float> hw3((size_t)Cout * Ch * 9), hw1((size_t)Cout * Cx), hb3(Cout), hb1(Cout);
  HIPCHK(hipMemcpy(hw3.data(), w3, hw3.size() * sizeof(float), hipMemcpyDeviceToHost));
  HIPCHK(hipMemcpy(hw1.data(), w1, hw1.size() * sizeof(float), hipMemcpyDeviceToHost));
  HIPCHK(hipMemcpy(hb3.data(), b3, Cout * sizeof(float), hipMemcpyDeviceToHost));
  HIPCHK(hipMemcpy(hb1.data(), b1, Cout * sizeof(float), hipMemcpyDeviceToHost));
  float mx = 0.f;
  for (float v : hw3) mx = std::max(mx, std::fabs(v));
  for (float v : hw1) mx = std::max(mx, std::fabs(v));
  const float wscale = (mx > 0.f && std::isfinite(mx)) ? std::ldexp(1.0f, 10 - (int)std::floor(std::log2(mx))) : 1.f;
  const size_t h3 = f16x3_packed_halfs(Cout, Ch, 3), h1 = f16x3_packed_halfs(Cout, Cx, 1);
  TRY(dalloc((h3 + h1 + 1) / 2, &xp));
  HIPCHK(launch_pack_f16x3(w3, xp, Cout, Ch, 3, wscale, s));
  HIPCHK(launch_pack_f16x3(w1, reinterpret_cast<char*>(xp) + h3 * 2, Cout, Cx, 1, wscale, s));
  for (int i = 0; i < Cout; ++i) hb3[i] += hb1[i];
  TRY(dalloc(Cout, &fb));
  HIPCHK(hipMemcpyAsync(fb, hb3.data(), Cout * sizeof(float), hipMemcpyHostToDevice, s));
  HIPCHK(hipStreamSynchronize(s));
  GemmArgs g;
  memset(&g, 0, sizeof g);
  g.a0 = hn; g.c0 = Ch; g.lda0 = Ch; g.a0_zo = (long long)HW * Ch;
  g.Hin = H; g.Win = W; g.Hout = H; g.Wout = W; g.Cin = Ch; g.Cout = Cout;
  g.ks = 3; g.stride = 1; g.pad = 1; g.pscale = sc; g.pshift = sh; g.silu = 1;
  g.bias = fb; g.out = yo; g.ldo = Cout; g.o_zo = (long long)HW * Cout; g.ZI = 1; g.Z = B;
  // the main tile: its 16x16x32 form when the channel counts allow it (Ch and C0 + C1 multiples of 32), else the 32x32x16 form
  // (images of 16 x 16 pixels and less exercise the kernel's 128-pixel form, 8 x 8 images its 64-pixel form)
  g.math = MATH_F16X3; g.tile = (HW <= 64) ? XT_64x128K32 : (HW <= 256) ? XT_128x128K32 : XT_256x128K32; g.wpk = xp; g.cout_pad = ((Cout + 127) / 128) * 128;
  g.alpha = 1.0f / (wscale * f16x3_act_scale());
  g.s0 = a0; g.sc0 = C0; g.lds0 = C0; g.s0_zo = (long long)HW * C0;
  if (x1) { g.s1 = a1; g.sc1 = C1; g.lds1 = C1; g.s1_zo = (long long)HW * C1; }
  g.Cin2 = Cx;
  hipError_t le = gemm_can_fuse_shortcut(g) ? launch_gemm(g, s) : hipErrorInvalidValue;
  if (le == hipSuccess) le = launch_nhwc_to_nchw(yo, Cout, y, B, Cout, HW, s);
  hipError_t se = hipStreamSynchronize(s);
  for (void* p : tmp) (void)hipFree(p);
  if (le != hipSuccess) return fail(ASYRP_EHIP, std::string("fused resblock tail launch: ") + hipGetErrorString(le));
  if (se != hipSuccess) return fail(ASYRP_EHIP, std::string("fused resblock tail sync: ") + hipGetErrorString(se));
  return 0;
}


int asyrp_op_attention(int device, const float* qkv, int B, int C, int T, int heads, int fused, float* out, void* stream) {
  if (!qkv || !out || B < 1 || heads < 1 || C % heads) return fail(ASYRP_EINVAL, "bad argument");
  if (fused && !attn_fused_supported(T, C / heads, 3 * C, C)) return fail(ASYRP_EINVAL, "shape not covered by the fused attention kernel");
  HIPCHK(hipSetDevice(device));
  hipStream_t s = (hipStream_t)stream;
  asyrp_engine tmp_e;   // only the pool (and the math switch) is used
  tmp_e.math = fused ? MATH_F16X3 : MATH_F32;
  tmp_e.np = (fused == 2) ? 1 : 3;
  Ctx c{&tmp_e, s, B};
  float *q2 = nullptr, *o2 = nullptr;
  int rc = tmp_e.pool.get((size_t)B * T * 3 * C, &q2);
  if (!rc) rc = tmp_e.pool.get((size_t)B * T * C, &o2);
  hipError_t le = hipSuccess;
  if (!rc && fused >= 3) {   // the split-plane kernel (3: two-term split, 4: single product), planes from the standalone splitter
    const int Dh = C / heads;
    if (!attn_planes_supported(T, Dh)) { tmp_e.pool.destroy(); return fail(ASYRP_EINVAL, "shape not covered by attn_planes_kernel"); }
    tmp_e.np = (fused == 4) ? 1 : 3;
    QkvPlanes pl;
    const size_t nqk = ((size_t)B * T * 3 * C + 1) / 2, nv = ((size_t)B * C * T + 1) / 2;
    for (int i = 0; i < 4 && !rc; ++i) rc = tmp_e.pool.get((i < 2) ? nqk : nv, &pl.raw[i]);
    if (!rc) {
      pl.h = reinterpret_cast<_Float16*>(pl.raw[0]); pl.l = reinterpret_cast<_Float16*>(pl.raw[1]);
      pl.vth = reinterpret_cast<_Float16*>(pl.raw[2]); pl.vtl = reinterpret_cast<_Float16*>(pl.raw[3]);
      pl.ld16 = 3 * C;
      le = launch_nchw_to_nhwc(qkv, q2, B, 3 * C, T, s);
      if (le == hipSuccess)
        le = launch_qkv_to_planes(q2, 3 * C, B, T, 3 * C, heads == 1 ? 3 * C : 3 * Dh, heads == 1 ? 2 * C : 2 * Dh, heads == 1 ? C : Dh,
                                  pl.h, pl.l, pl.vth, pl.vtl, s);
      if (le == hipSuccess) rc = attention_planes(c, pl, C, T, heads, 1.0f / std::sqrt((float)Dh), o2);
      if (!rc && le == hipSuccess) le = launch_nhwc_to_nchw(o2, C, out, B, C, T, s);
    }
  } else if (!rc) {
    le = launch_nchw_to_nhwc(qkv, q2, B, 3 * C, T, s);
    const int Dh = C / heads;
    if (le == hipSuccess) rc = attention_core(c, q2, C, T, heads, 1.0f / std::sqrt((float)Dh), o2);
    if (!rc && le == hipSuccess) le = launch_nhwc_to_nchw(o2, C, out, B, C, T, s);
  }
  hipError_t se = hipStreamSynchronize(s);
  tmp_e.pool.destroy();
  if (rc) return rc;
  if (le != hipSuccess) return fail(ASYRP_EHIP, std::string("attention launch: ") + hipGetErrorString(le));
  if (se != hipSuccess) return fail(ASYRP_EHIP, std::string("attention sync: ") + hipGetErrorString(se));
  return 0;
}

// Step arithmetic of the vendored samplers (GaussianDiffusion.p_sample / ddim_sample / ddim_reverse_sample,
// models/guided_diffusion/gaussian_diffusion.py:402-446, 544-630) as one elementwise launch per 32 images: see
// sampler_update_kernel.  Stateless (no engine): the schedule lives in the caller's float64 tables, which it folds into the
// per-image rows coef_host[b] = {a, b, p, q, r, lo, hi, clip}.  Asynchronous on `stream`.
int asyrp_sampler_update(int device, const float* x, const float* model_out, int out_channels, int B, int C, int HW,
                         const float* coef_host, const float* noise, float* sample, float* pred_xstart, float* log_variance,
                         void* stream) {
  if (!x || !model_out || !coef_host || B < 1 || C < 1 || HW < 1) return fail(ASYRP_EINVAL, "bad argument");
  if (out_channels != C && out_channels != 2 * C) return fail(ASYRP_EINVAL, "model output must have C or 2C channels");
  if (!sample && !pred_xstart && !log_variance) return fail(ASYRP_EINVAL, "no output requested");
  if (log_variance && out_channels != 2 * C) return fail(ASYRP_EINVAL, "log_variance needs the learned variance channels");
  HIPCHK(hipSetDevice(device));
  const long long per = (long long)C * HW;
  for (int b0 = 0; b0 < B; b0 += SamplerArgs::MAXB) {
    SamplerArgs a;
    memset(&a, 0, sizeof a);
    a.nb = std::min<int>(SamplerArgs::MAXB, B - b0);
    a.C = C; a.HW = HW; a.eps_img = (long long)out_channels * HW;
    a.x = x + b0 * per;
    a.eps = model_out + b0 * a.eps_img;
    a.var = (out_channels == 2 * C) ? a.eps + per : nullptr;
    a.noise = noise ? noise + b0 * per : nullptr;
    a.out = sample ? sample + b0 * per : nullptr;
    a.x0 = pred_xstart ? pred_xstart + b0 * per : nullptr;
    a.logvar = log_variance ? log_variance + b0 * per : nullptr;
    for (int i = 0; i < a.nb; ++i) {
      const float* r = coef_host + (size_t)(b0 + i) * 8;
      a.k[i] = SamplerCoef{r[0], r[1], r[2], r[3], r[4], r[5], r[6], r[7] != 0.f ? 1 : 0};
    }
    HIPCHK(launch_sampler_update(a, (hipStream_t)stream));
  }
  return 0;
}

}  // extern "C"
