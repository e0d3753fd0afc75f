// Module-level entry points: the launch plans of Decoder.forward and DiffusionSampler.forward in C++.
//
// One st2_engine holds one model's packed weights on one device (the single device allocation this file makes) and
// issues, per forward call, a straight-line sequence of the kernel entry points declared in st2.h on the caller's
// stream, inside a caller-owned workspace.  No allocation, no synchronisation, no host read of device data in the
// forward calls: they are legal under hipStreamBeginCapture.  The same plans exist in Python (styletts2_amd/decoder.py,
// diffusion.py: the per-kernel path kept for tap-point work and A-B runs); both issue the same kernels with the same
// arguments, so their results are bitwise equal (tests/test_engine_gpu.py).
//
// Every kernel / memory call goes through a function table (st2_debug_set_backend): tests substitute CPU contracts
// for the HIP kernels and run these plans on host memory, which validates wiring, packing and workspace aliasing
// without a GPU.
#include <math.h>
#include <stdio.h>
#include <string.h>

#include <algorithm>
#include <cmath>
#include <string>
#include <unordered_map>
#include <vector>

#include "st2_common.h"

namespace {

// ------------------------------------------------------------------------------------------------------------------
// backend table
// ------------------------------------------------------------------------------------------------------------------
struct Backend {
  decltype(&st2_conv1d_f16s) conv1d_f16s;
  decltype(&st2_conv1d_xs) conv1d_xs;
  decltype(&st2_act_split) act_split;
  decltype(&st2_stats_finalize) stats_finalize;
  decltype(&st2_conv1d_direct) conv1d_direct;
  decltype(&st2_phase_split) phase_split;
  decltype(&st2_instnorm_stats) instnorm_stats;
  decltype(&st2_colnorm_stats) colnorm_stats;
  decltype(&st2_style_fc) style_fc;
  decltype(&st2_convt_interleave_stats) convt_interleave_stats;
  decltype(&st2_adain_leaky_pool) adain_leaky_pool;
  decltype(&st2_har_source) har_source;
  decltype(&st2_stft_mag_phase) stft_mag_phase;
  decltype(&st2_istft) istft;
  decltype(&st2_attention_keylen) attention_keylen;
  decltype(&st2_add_chanvec) add_chanvec;
  decltype(&st2_mean_tokens_len) mean_tokens_len;
  decltype(&st2_axpbypcz) axpbypcz;
  decltype(&st2_time_features) time_features;
  decltype(&st2_tokens_to_channels) tokens_to_channels;
  decltype(&st2_broadcast_cols) broadcast_cols;
  decltype(&st2_copy_ncl) copy_ncl;
  decltype(&st2_expand_by_durations) expand_by_durations;
  // bidirectional LSTM recurrence with a scratch buffer: the HIP entry tries the cooperative kernel and falls back to
  // the single-CU one when the device cannot hold its workgroups co-resident (same policy as ops.lstm_bidir)
  int (*lstm_bidir)(const float*, int64_t, int32_t, const float*, const int32_t*, int32_t, int32_t, int32_t, float*, int64_t,
                    int32_t, void*, int64_t, void*);
  decltype(&st2_colnorm_apply) colnorm_apply;
  decltype(&st2_duration_head) duration_head;
  decltype(&st2_mask_tail) mask_tail;
  decltype(&st2_embed_tokens) embed_tokens;
  decltype(&st2_dwconv3x3s2) dwconv3x3s2;
  decltype(&st2_avgpool2x2) avgpool2x2;
  void* (*dev_alloc)(int64_t);
  void (*dev_free)(void*);
  int (*upload)(void*, const void*, int64_t);
};

void* hip_alloc(int64_t n) {
  void* p = nullptr;
  if (hipMalloc(&p, (size_t)n) != hipSuccess) {
    (void)hipGetLastError();
    return nullptr;
  }
  return p;
}
void hip_free(void* p) { (void)hipFree(p); }
int hip_upload(void* d, const void* s, int64_t n) {
  return hipMemcpy(d, s, (size_t)n, hipMemcpyHostToDevice) == hipSuccess ? 0 : 1;
}

int hip_lstm(const float* G, int64_t g_bs, int32_t g_cs, const float* whh_t, const int32_t* lengths, int32_t B, int32_t H,
             int32_t N, float* Y, int64_t y_bs, int32_t y_cs, void* scratch, int64_t scratch_bytes, void* stream) {
  // No sticky "refused" state: whether a cooperative launch fits depends on the device AND the batch (a refusal at B = 48
  // says nothing about the latency-critical B = 1 long-form launches, nor about another device of the process).  The
  // refusal itself is a host-side occupancy comparison -- nothing was launched -- so asking every time costs nothing.
  // The cooperative launch carries its own safety net (round 5): the single-CU kernel is queued behind it in its conditional
  // form and re-runs the call into the same output if a group was not co-resident in time (two queues sharing the chip) --
  // a latency cost and ST2_STATUS_LSTM_RECOVERED instead of a batch of bad audio.
  if (scratch && scratch_bytes > 0) {
    if (st2_lstm_bidir_coop_recovering(G, g_bs, g_cs, whh_t, lengths, B, H, N, Y, y_bs, y_cs, scratch, scratch_bytes, stream) == 0)
      return 0;
    const char* msg = st2_last_error();
    if (!msg || !strstr(msg, "co-resident")) return 1;
  }
  return st2_lstm_bidir(G, g_bs, g_cs, whh_t, lengths, B, H, N, Y, y_bs, y_cs, stream);
}

const Backend kHipBackend = {st2_conv1d_f16s, st2_conv1d_xs, st2_act_split, st2_stats_finalize, st2_conv1d_direct,
                             st2_phase_split, st2_instnorm_stats, st2_colnorm_stats, st2_style_fc,
                             st2_convt_interleave_stats, st2_adain_leaky_pool, st2_har_source, st2_stft_mag_phase,
                             st2_istft, st2_attention_keylen, st2_add_chanvec, st2_mean_tokens_len, st2_axpbypcz,
                             st2_time_features, st2_tokens_to_channels, st2_broadcast_cols, st2_copy_ncl,
                             st2_expand_by_durations, hip_lstm, st2_colnorm_apply, st2_duration_head, st2_mask_tail,
                             st2_embed_tokens, st2_dwconv3x3s2, st2_avgpool2x2, hip_alloc, hip_free, hip_upload};
Backend g_be = kHipBackend;

// ------------------------------------------------------------------------------------------------------------------
// workspace arena, views
// ------------------------------------------------------------------------------------------------------------------
constexpr int XS_HALO = 32;        // == styletts2_amd.ops.XS_HALO
constexpr int FUSED_MAX_C = 64, FUSED_K3_MAX_C = 128;  // ops.prefer_fused
constexpr int XS_MIN_L = 256;      // shorter rows stay on the fused kernel
constexpr int XS_MIN_C_PLAIN = 64;  // prologue-free convs take the xs pair from this many input channels on
constexpr int CVT_TILE = 1024;     // positions per st2_convt_interleave_stats partial sum

struct Arena {
  char* base = nullptr;
  int64_t off = 0, cap = 0, peak = 0;
  bool dry = false;  // size query: hand out fake addresses, count the peak
  bool overflow = false;
  void* alloc(int64_t bytes) {
    off = (off + 255) & ~(int64_t)255;
    void* p = dry ? reinterpret_cast<void*>((uintptr_t)0x10000 + (uintptr_t)off) : static_cast<void*>(base + off);
    off += bytes;
    peak = std::max(peak, off);
    if (!dry && off > cap) overflow = true;
    return p;
  }
  float* f32(int64_t n) { return static_cast<float*>(alloc(n * 4)); }
};

struct View {  // NCL view, strides in elements
  float* p = nullptr;
  int64_t bs = 0;
  int cs = 0;
  int B = 0, C = 0, L = 0;
  View rows(int c0, int c1) const {
    View v = *this;
    v.p = p + (int64_t)c0 * cs;
    v.C = c1 - c0;
    return v;
  }
  bool ok() const { return p != nullptr; }
};

int pitch_of(int L) { return (L + 31) / 32 * 32; }  // rows of the big activations start 128-byte aligned

struct Ctx {
  Arena a;
  void* stream = nullptr;
  int rc = 0;
  bool dry = false;
};

View new_ncl(Ctx& c, int B, int C, int L, bool padded = true) {
  View v;
  v.B = B; v.C = C; v.L = L;
  v.cs = padded ? pitch_of(L) : L;
  v.bs = (int64_t)C * v.cs;
  v.p = c.a.f32((int64_t)B * v.bs);
  return v;
}

View wrap(const float* p, int B, int C, int L) {
  View v;
  v.p = const_cast<float*>(p);
  v.B = B; v.C = C; v.L = L; v.cs = L; v.bs = (int64_t)C * L;
  return v;
}

#define RUN(c, expr)                                   \
  do {                                                 \
    if (!(c).dry && (c).rc == 0 && !(c).a.overflow) {  \
      (c).rc = (expr);                                 \
    }                                                  \
  } while (0)

// ------------------------------------------------------------------------------------------------------------------
// packed weights (host side: offsets into one blob; device side: base + offset)
// ------------------------------------------------------------------------------------------------------------------
struct HostTensor {
  std::vector<int64_t> shape;
  std::vector<float> data;
  int64_t numel() const { return (int64_t)data.size(); }
};

struct ConvSite {  // one split-f16 packed conv weight = one calibration site (st2_calibration_read)
  std::string name;    // the reference state_dict key it was packed from
  int C_in = 0, C_out = 0, ks = 0;
};

struct Blob {
  std::vector<char> host;
  std::vector<ConvSite> sites;  // in packing order: the index is the site id (a function of the model layout alone)
  int64_t add(const void* src, int64_t bytes) {
    int64_t off = ((int64_t)host.size() + 255) & ~(int64_t)255;
    host.resize((size_t)(off + bytes));
    if (src) memcpy(host.data() + off, src, (size_t)bytes);
    return off;
  }
  int64_t add_f32(const std::vector<float>& v) { return add(v.data(), (int64_t)v.size() * 4); }
};

struct SplitW {  // st2.h: split-f16 packed conv weight
  int64_t wq = -1, row_scale = -1;
  int C_in = 0, C_out = 0, ks = 0, co_pad = 0, cin_pad = 0;
  int site = -1;  // index into st2_engine::sites
};

struct PConv {
  SplitW w;
  int64_t bias = -1;
  int c_out = 0, ks = 0;
};

struct PResBlock1 {  // AdaINResBlock1 (Modules/istftnet.py:27-81)
  int channels = 0, ks = 0;
  int dil[3] = {1, 3, 5};
  PConv c1[3], c2[3];
  int64_t a1[3] = {-1, -1, -1}, a2[3] = {-1, -1, -1};  // alpha
  int ad1[3] = {0, 0, 0}, ad2[3] = {0, 0, 0};          // style-bank offsets of adain1 / adain2
};

struct PAdainResBlk {  // AdainResBlk1d (Modules/istftnet.py:410-454)
  int dim_in = 0, dim_out = 0;
  bool upsample = false, learned_sc = false;
  PConv conv1, conv2, sc;
  int64_t pool_w = -1, pool_b = -1;
  int n1 = 0, n2 = 0;  // style-bank offsets of norm1 / norm2
};

int f16s_chunk(int ks) { return ks <= 3 ? 32 : 16; }
int f16s_co_block(int C_out) { return C_out > 64 ? 128 : (C_out > 32 ? 64 : 32); }

// weights.pack_conv_f16s, bit for bit: per-row power-of-two scale, hi = f16(w*s), lo = f16(w*s - hi),
// layout [ci/16][tap][k-half][co][hi8 | lo8]
SplitW pack_split(Blob& blob, const std::string& name, const float* w, int C_out, int C_in, int ks) {
  SplitW r;
  r.C_in = C_in; r.C_out = C_out; r.ks = ks;
  r.site = (int)blob.sites.size();
  blob.sites.push_back({name, C_in, C_out, ks});
  const int cb = f16s_chunk(ks), rb = f16s_co_block(C_out);
  r.cin_pad = (C_in + cb - 1) / cb * cb;
  r.co_pad = (C_out + rb - 1) / rb * rb;
  const int n16 = r.cin_pad / 16;
  std::vector<_Float16> q((size_t)n16 * ks * 2 * r.co_pad * 16, (_Float16)0.0f);
  std::vector<float> rs((size_t)r.co_pad, 1.0f);
  for (int co = 0; co < C_out; ++co) {
    const float* wr = w + (int64_t)co * C_in * ks;
    float amax = 0.f;
    for (int i = 0; i < C_in * ks; ++i) amax = std::max(amax, fabsf(wr[i]));
    float scale = 1.0f;
    if (amax > 0.f) {
      int e = 0;
      (void)frexpf(amax, &e);
      scale = ldexpf(1.0f, std::min(14 - e, 126));  // clamp: rows with amax < 2^-112 must not get scale = inf (weights.py)
    }
    rs[co] = 1.0f / scale;
    for (int ci = 0; ci < C_in; ++ci)
      for (int t = 0; t < ks; ++t) {
        const float v = wr[(int64_t)ci * ks + t] * scale;
        const _Float16 hi = (_Float16)v;
        const _Float16 lo = (_Float16)(v - (float)hi);
        const int i16 = ci / 16, kg = (ci % 16) / 8, e8 = ci % 8;
        const size_t base = ((((size_t)i16 * ks + t) * 2 + kg) * r.co_pad + co) * 16;
        q[base + e8] = hi;
        q[base + 8 + e8] = lo;
      }
  }
  r.wq = blob.add(q.data(), (int64_t)q.size() * 2);
  r.row_scale = blob.add_f32(rs);
  return r;
}

// weights.polyphase_convt: ConvTranspose1d weight [C_in][C_out][K = 2*stride] -> Conv1d weight [stride*C_out][C_in][2]
std::vector<float> polyphase_convt(const float* w, int C_in, int C_out, int stride) {
  const int K = 2 * stride;
  std::vector<float> wp((size_t)stride * C_out * C_in * 2);
  for (int r = 0; r < stride; ++r)
    for (int co = 0; co < C_out; ++co)
      for (int ci = 0; ci < C_in; ++ci) {
        const size_t o = (((size_t)r * C_out + co) * C_in + ci) * 2;
        wp[o + 0] = w[((int64_t)ci * C_out + co) * K + r + stride];
        wp[o + 1] = w[((int64_t)ci * C_out + co) * K + r];
      }
  return wp;
}

// weights.polyphase_strided_conv: [C_out][C_in][K = 2*stride] -> [C_out][C_in*stride][2]
std::vector<float> polyphase_strided(const float* w, int C_out, int C_in, int stride) {
  const int K = 2 * stride;
  std::vector<float> wp((size_t)C_out * C_in * stride * 2);
  for (int co = 0; co < C_out; ++co)
    for (int ci = 0; ci < C_in; ++ci)
      for (int r = 0; r < stride; ++r)
        for (int j = 0; j < 2; ++j)
          wp[(((size_t)co * C_in * stride) + (size_t)ci * stride + r) * 2 + j] = w[((int64_t)co * C_in + ci) * K + j * stride + r];
  return wp;
}

struct PGenerator {
  int64_t lin_w = -1, lin_b = -1;
  std::vector<SplitW> noise_wt;
  std::vector<int> noise_stride;
  std::vector<int64_t> noise_b;
  std::vector<PResBlock1> noise_res, resblocks;
  std::vector<SplitW> ups_wt;
  std::vector<int64_t> ups_b;
  PConv post;
  std::vector<int64_t> alphas;  // hifigan
  std::vector<int> channels;
};

struct PDecoder {
  bool ready = false;
  int J = 0;                    // style bank width
  int64_t bank_wt = -1, bank_b = -1;
  PAdainResBlk encode, decode[4];
  int64_t f0_w = -1, f0_b = -1, n_w = -1, n_b = -1;
  PConv asr_res;
  PGenerator gen;
};

struct PBlock {
  int64_t n_w = -1, n_b = -1, nc_w = -1, nc_b = -1;  // LayerNorm affine (single speaker)
  SplitW q, kv, o, f1, f2;
  int64_t o_b = -1, f1_b = -1, f2_b = -1;
  int f1_out = 0;
};

struct PDenoiser {
  bool ready = false;
  int features = 0;  // channels + embedding
  int64_t time_w = -1, time_lin = -1, time_b = -1, map0 = -1, map0_b = -1, map2 = -1, map2_b = -1;
  int64_t feat = -1, feat_b = -1, ada_wt = -1, ada_b = -1;
  int64_t out_t = -1, out_b = -1, fixed = -1;
  std::vector<PBlock> blocks;
};

struct PLstm {  // nn.LSTM(1 layer, bidirectional): input projection as a k = 1 conv, recurrence weights transposed
  SplitW w_ih;          // [8H][I]: forward gates then reverse gates
  int64_t bias = -1;    // [8H] = b_ih + b_hh per direction
  int64_t whh_t = -1;   // [2][H][4H]
  int H = 0;
};

struct PDuration {  // DurationEncoder (models.py:517-569) + duration LSTM + duration_proj (models.py:450-451)
  bool ready = false;
  std::vector<PLstm> lstms;             // nlayers
  std::vector<int64_t> ada_wt, ada_b;   // AdaLayerNorm fc per layer: [style][2C] (st2_style_fc layout), [2C]
  PLstm dur_lstm;
  int64_t proj_w = -1, proj_b = -1;     // duration_proj.linear_layer: [max_dur][d_hid], [max_dur]
  int max_dur = 0;
};

struct PText {  // TextEncoder (models.py:284-345)
  bool ready = false;
  int64_t emb = -1;
  int V = 0, C = 0;
  std::vector<PConv> convs;
  std::vector<int64_t> ln_g, ln_b;
  PLstm lstm;
};

struct PBert {  // PL-BERT: HF AlbertModel, one shared layer (Utils/PLBERT/util.py:6-20) + the bert_encoder Linear (models.py:689)
  int64_t word = -1, pos = -1, tok0 = -1, eln_w = -1, eln_b = -1;
  int V = 0, P = 0, E = 0, H = 0, I = 0;
  SplitW map, qkv, dense, ffn, out, enc;
  int64_t map_b = -1, qkv_b = -1, dense_b = -1, aln_w = -1, aln_b = -1, ffn_b = -1, out_b = -1, fln_w = -1, fln_b = -1,
          enc_b = -1;
  bool has_enc = false, ready = false;
};

struct PStyleBlk {  // ResBlk(normalize=False, downsample='half'), models.py:97-137
  int c_in = 0, c_out = 0;
  SplitW w1, w2, wsc;
  int64_t b1 = -1, b2 = -1, wd = -1, bd = -1;
  bool has_sc = false;
};
struct PStyleEnc {  // StyleEncoder, models.py:139-164 (spectral-norm convs folded by the caller)
  int64_t w0 = -1, b0 = -1;
  int c0 = 0, c_last = 0, style_dim = 0;
  std::vector<PStyleBlk> blocks;
  SplitW w5, wl;
  int64_t b5 = -1, bl = -1;
  bool ready = false;
};

struct PPredictor {  // ProsodyPredictor.F0Ntrain (models.py:497-510)
  bool ready = false;
  int J = 0;
  int64_t bank_wt = -1, bank_b = -1;
  PLstm shared;
  PAdainResBlk f0[3], n[3];
  int64_t f0p_w = -1, f0p_b = -1, np_w = -1, np_b = -1;
};

}  // namespace

struct st2_engine {
  st2_model_config cfg;
  std::unordered_map<std::string, HostTensor> host;
  char* wbase = nullptr;  // device blob
  int64_t wbytes = 0;
  PDecoder dec;
  PDenoiser dn;
  PPredictor pred;
  PDuration dur;
  PText text;
  PBert bert;
  PStyleEnc style[2];  // 0 = style_encoder (acoustic), 1 = predictor_encoder (prosodic)
  int64_t zeros = -1;  // 4096 zero floats (map borders)
  // Calibration sites = the split-f16 conv weights in packing order, and the calibrated power-of-two operand scale of each
  // (st2_calibrate / st2_calibration_write; 0 = not calibrated: the rule x_scale_for(pro)).  seen = max |pro(x)| at calibration.
  std::vector<ConvSite> sites;
  std::vector<float> site_scale, site_seen;
  float x_scale(int site, float by_rule) const {
    return site >= 0 && site < (int)site_scale.size() && site_scale[(size_t)site] > 0.f ? site_scale[(size_t)site] : by_rule;
  }
  template <class T>
  T* P(int64_t off) const { return off < 0 ? nullptr : reinterpret_cast<T*>(wbase + off); }
  const float* F(int64_t off) const { return P<const float>(off); }
};

namespace {

// ------------------------------------------------------------------------------------------------------------------
// packing
// ------------------------------------------------------------------------------------------------------------------
struct Packer {
  st2_engine& e;
  Blob& blob;
  bool ok = true;
  std::string missing;
  const HostTensor* get(const std::string& name) {
    auto it = e.host.find(name);
    if (it == e.host.end()) {
      if (ok) missing = name;
      ok = false;
      return nullptr;
    }
    return &it->second;
  }
  bool has(const std::string& name) const { return e.host.find(name) != e.host.end(); }
  // A checkpoint whose layout differs from st2_model_config must fail HERE, with a message -- never index a host tensor
  // past its end.  `want` lists the expected dimensions, -1 = any; trailing dimensions of size 1 may be absent
  // (nn.Linear weights stand in for k = 1 convs, [C][1][1] gains for [C] vectors).
  void fail(const std::string& why) {
    if (ok) missing = why;
    ok = false;
  }
  static std::string shape_str(const std::vector<int64_t>& s) {
    std::string r = "[";
    for (size_t i = 0; i < s.size(); ++i) r += (i ? ", " : "") + std::to_string(s[i]);
    return r + "]";
  }
  const HostTensor* get(const std::string& name, std::initializer_list<int64_t> want) {
    const HostTensor* t = get(name);
    if (!t) return nullptr;
    std::vector<int64_t> w(want);
    bool good = t->shape.size() <= w.size();
    for (size_t i = 0; good && i < w.size(); ++i) {
      const int64_t have = i < t->shape.size() ? t->shape[i] : 1;
      good = w[i] < 0 ? have > 0 : have == w[i];
    }
    if (!good) {
      fail(name + " has shape " + shape_str(t->shape) + ", the model configuration needs " + shape_str(w));
      return nullptr;
    }
    return t;
  }
  int64_t vec(const std::string& name, int64_t numel = -1) {  // flat fp32 copy (numel >= 0: exact element count)
    const HostTensor* t = get(name);
    if (!t) return -1;
    if (numel >= 0 && t->numel() != numel) {
      fail(name + " has " + std::to_string(t->numel()) + " elements, the model configuration needs " + std::to_string(numel));
      return -1;
    }
    return blob.add_f32(t->data);
  }
  // nn.Linear weight [out][in] -> [in][out] (the st2_style_fc layout)
  int64_t lin_t(const std::string& name, int64_t out = -1, int64_t in = -1) {
    const HostTensor* t = get(name, {out, in});
    if (!t) return -1;
    if (t->shape.size() != 2) { fail(name + " (2-D expected)"); return -1; }
    const int64_t O = t->shape[0], I = t->shape[1];
    std::vector<float> v((size_t)(O * I));
    for (int64_t o = 0; o < O; ++o)
      for (int64_t i = 0; i < I; ++i) v[(size_t)(i * O + o)] = t->data[(size_t)(o * I + i)];
    return blob.add_f32(v);
  }
  // [C_out][C_in][ks] (or nn.Linear [out][in] as ks = 1); co / ci / ks >= 0 pin the expected geometry
  SplitW conv_w(const std::string& name, int co = -1, int ci = -1, int ks = -1) {
    const HostTensor* t = get(name, {co, ci, ks});
    if (!t) return SplitW();
    if (t->shape.size() < 2) { fail(name + " has shape " + shape_str(t->shape) + ", a conv / linear weight is at least 2-D"); return SplitW(); }
    const int C_out = (int)t->shape[0], C_in = (int)t->shape[1], k = t->shape.size() > 2 ? (int)t->shape[2] : 1;
    return pack_split(blob, name, t->data.data(), C_out, C_in, k);
  }
  PConv conv(const std::string& prefix, bool bias = true, int co = -1, int ci = -1, int ks = -1) {
    PConv c;
    c.w = conv_w(prefix + ".weight", co, ci, ks);
    c.c_out = c.w.C_out;
    c.ks = c.w.ks;
    if (bias && has(prefix + ".bias")) c.bias = vec(prefix + ".bias", c.w.C_out > 0 ? c.w.C_out : -1);
    return c;
  }
};

struct Bank {  // decoder.StyleBank: every AdaIN fc of the module in one [style][J] matrix
  std::vector<const HostTensor*> ws, bs;
  std::vector<int> chans;
  std::vector<std::string> names;
  int J = 0;
  int add(Packer& pk, const std::string& prefix, int channels) {
    names.push_back(prefix + ".fc.weight");
    ws.push_back(pk.get(prefix + ".fc.weight"));
    bs.push_back(pk.get(prefix + ".fc.bias"));
    chans.push_back(channels);
    const int off = J;
    J += 2 * channels;
    return off;
  }
  void pack(Packer& pk, int style_dim, int64_t* wt_off, int64_t* b_off) {
    std::vector<float> wt((size_t)style_dim * J), b((size_t)J);
    int off = 0;
    for (size_t m = 0; m < ws.size(); ++m) {
      const int n = 2 * chans[m];
      if (!ws[m] || !bs[m]) return;
      if (ws[m]->numel() != (int64_t)n * style_dim || bs[m]->numel() != n) {  // AdaIN fc: Linear(style_dim, 2 * channels)
        pk.fail("AdaIN fc " + names[m] + " has shape " + Packer::shape_str(ws[m]->shape) + " / bias " + Packer::shape_str(bs[m]->shape) +
                ", the model configuration needs [" + std::to_string(n) + ", " + std::to_string(style_dim) + "]");
        return;
      }
      for (int j = 0; j < n; ++j) {
        for (int k = 0; k < style_dim; ++k) wt[(size_t)k * J + off + j] = ws[m]->data[(size_t)j * style_dim + k];
        b[(size_t)off + j] = bs[m]->data[(size_t)j];
      }
      off += n;
    }
    *wt_off = pk.blob.add_f32(wt);
    *b_off = pk.blob.add_f32(b);
  }
};

PResBlock1 pack_resblock1(Packer& pk, const std::string& prefix, int channels, int ks, const int* dil) {
  PResBlock1 r;
  r.channels = channels;
  r.ks = ks;
  for (int i = 0; i < 3; ++i) {
    const std::string si = std::to_string(i);
    r.dil[i] = dil[i];
    r.c1[i] = pk.conv(prefix + ".convs1." + si, true, channels, channels, ks);
    r.c2[i] = pk.conv(prefix + ".convs2." + si, true, channels, channels, ks);
    r.a1[i] = pk.vec(prefix + ".alpha1." + si, channels);
    r.a2[i] = pk.vec(prefix + ".alpha2." + si, channels);
  }
  return r;
}

PAdainResBlk pack_adain_resblk(Packer& pk, const std::string& prefix, int dim_in, int dim_out, bool upsample) {
  PAdainResBlk r;
  r.dim_in = dim_in; r.dim_out = dim_out; r.upsample = upsample; r.learned_sc = dim_in != dim_out;
  r.conv1 = pk.conv(prefix + ".conv1", true, dim_out, dim_in, 3);
  r.conv2 = pk.conv(prefix + ".conv2", true, dim_out, dim_out, 3);
  if (r.learned_sc) r.sc = pk.conv(prefix + ".conv1x1", false, dim_out, dim_in, 1);
  if (upsample) {
    r.pool_w = pk.vec(prefix + ".pool.weight", (int64_t)dim_in * 3);  // [C][1][3] folded
    r.pool_b = pk.vec(prefix + ".pool.bias", dim_in);
  }
  return r;
}

int prod_from(const int32_t* v, int lo, int hi) {
  int p = 1;
  for (int i = lo; i < hi; ++i) p *= v[i];
  return p;
}

int pack_decoder(st2_engine& e, Blob& blob, std::string* err) {
  const st2_model_config& cfg = e.cfg;
  Packer pk{e, blob};
  PDecoder d;
  Bank bank;
  const std::string D = "decoder.";
  const int sty = cfg.style_dim, Cin = cfg.dim_in;
  // registration order of decoder.Decoder._prepare: encode, decode[0..3], then the generator's noise_res + resblocks
  d.encode = pack_adain_resblk(pk, D + "encode", Cin + 2, 1024, false);
  d.encode.n1 = bank.add(pk, D + "encode.norm1", Cin + 2);
  d.encode.n2 = bank.add(pk, D + "encode.norm2", 1024);
  for (int i = 0; i < 4; ++i) {
    const std::string p = D + "decode." + std::to_string(i);
    const bool up = i == 3;
    d.decode[i] = pack_adain_resblk(pk, p, 1024 + 2 + 64, up ? 512 : 1024, up);
    d.decode[i].n1 = bank.add(pk, p + ".norm1", 1024 + 2 + 64);
    d.decode[i].n2 = bank.add(pk, p + ".norm2", up ? 512 : 1024);
  }
  d.f0_w = pk.vec(D + "F0_conv.weight", 3); d.f0_b = pk.vec(D + "F0_conv.bias", 1);  // Conv1d(1, 1, 3, stride 2)
  d.n_w = pk.vec(D + "N_conv.weight", 3);   d.n_b = pk.vec(D + "N_conv.bias", 1);
  d.asr_res = pk.conv(D + "asr_res.0", true, 64, Cin, 1);

  PGenerator& g = d.gen;
  const std::string G = D + "generator.";
  const int nu = cfg.n_upsamples, nk = cfg.n_resblock_kernels;
  const bool ist = cfg.decoder_kind == 0;
  const int c0 = cfg.upsample_initial_channel;
  g.lin_w = pk.vec(G + "m_source.l_linear.weight", 9);  // Linear(harmonic_num + 1 = 9, 1), Modules/istftnet.py:283
  g.lin_b = pk.vec(G + "m_source.l_linear.bias", 1);
  static const int dil135[3] = {1, 3, 5};
  for (int i = 0; i < nu; ++i) g.channels.push_back(c0 >> (i + 1));
  // noise_res first, then resblocks (Generator.register)
  for (int i = 0; i < nu; ++i) {
    const bool last = i + 1 == nu;
    PResBlock1 r = pack_resblock1(pk, G + "noise_res." + std::to_string(i), g.channels[i], last ? 11 : 7, dil135);
    g.noise_res.push_back(r);
  }
  for (int i = 0; i < nu; ++i)
    for (int k = 0; k < nk; ++k) {
      PResBlock1 r = pack_resblock1(pk, G + "resblocks." + std::to_string(i * nk + k), g.channels[i],
                                    cfg.resblock_kernel_sizes[k], cfg.resblock_dilations[k]);
      g.resblocks.push_back(r);
    }
  for (auto* list : {&g.noise_res, &g.resblocks}) {
    const std::string pre = list == &g.noise_res ? G + "noise_res." : G + "resblocks.";
    for (size_t i = 0; i < list->size(); ++i) {
      PResBlock1& r = (*list)[i];
      for (int j = 0; j < 3; ++j) r.ad1[j] = bank.add(pk, pre + std::to_string(i) + ".adain1." + std::to_string(j), r.channels);
      for (int j = 0; j < 3; ++j) r.ad2[j] = bank.add(pk, pre + std::to_string(i) + ".adain2." + std::to_string(j), r.channels);
    }
  }
  if (pk.ok) bank.pack(pk, sty, &d.bank_wt, &d.bank_b);
  d.J = bank.J;
  // noise convs: kernel = 2*stride -> polyphase k = 2 conv; the last one is k = 1
  for (int i = 0; i < nu; ++i) {
    const int stride_f0 = i + 1 < nu ? prod_from(cfg.upsample_rates, i + 1, nu) : 1;
    const HostTensor* w = pk.get(G + "noise_convs." + std::to_string(i) + ".weight", {g.channels[i], -1, -1});
    g.noise_stride.push_back(stride_f0);
    if (w && w->shape.size() != 3) { pk.fail(G + "noise_convs." + std::to_string(i) + ".weight must be 3-D"); w = nullptr; }
    if (w) {
      const int C_out = (int)w->shape[0], C_in = (int)w->shape[1], K = (int)w->shape[2];
      if (stride_f0 > 1) {
        if (K != 2 * stride_f0) { *err = "noise_convs kernel must be 2*stride"; return 1; }
        std::vector<float> wp = polyphase_strided(w->data.data(), C_out, C_in, stride_f0);
        g.noise_wt.push_back(pack_split(blob, G + "noise_convs." + std::to_string(i) + ".weight", wp.data(), C_out, C_in * stride_f0, 2));
      } else {
        g.noise_wt.push_back(pack_split(blob, G + "noise_convs." + std::to_string(i) + ".weight", w->data.data(), C_out, C_in, K));
      }
    } else {
      g.noise_wt.push_back(SplitW());
    }
    g.noise_b.push_back(pk.vec(G + "noise_convs." + std::to_string(i) + ".bias", g.channels[i]));
  }
  for (int i = 0; i < nu; ++i) {
    const int u = cfg.upsample_rates[i];
    const HostTensor* w = pk.get(G + "ups." + std::to_string(i) + ".weight", {c0 >> i, c0 >> (i + 1), -1});  // [C_in][C_out][K]
    if (w && w->shape.size() != 3) { pk.fail(G + "ups." + std::to_string(i) + ".weight must be 3-D"); w = nullptr; }
    if (w) {
      const int C_in = (int)w->shape[0], C_out = (int)w->shape[1], K = (int)w->shape[2];
      if (K != 2 * u) { *err = "ups kernel must be 2*stride"; return 1; }
      std::vector<float> wp = polyphase_convt(w->data.data(), C_in, C_out, u);
      g.ups_wt.push_back(pack_split(blob, G + "ups." + std::to_string(i) + ".weight", wp.data(), u * C_out, C_in, 2));
    } else {
      g.ups_wt.push_back(SplitW());
    }
    g.ups_b.push_back(pk.vec(G + "ups." + std::to_string(i) + ".bias", c0 >> (i + 1)));
  }
  g.post = pk.conv(G + "conv_post", true, ist ? cfg.gen_istft_n_fft + 2 : 1, g.channels.back(), 7);
  if (!ist)
    for (int i = 0; i <= nu; ++i) g.alphas.push_back(pk.vec(G + "alphas." + std::to_string(i), c0 >> i));
  if (!pk.ok) {
    *err = "decoder weights (missing or malformed): " + pk.missing;
    return 1;
  }
  d.ready = true;
  e.dec = d;
  return 0;
}

int pack_denoiser(st2_engine& e, Blob& blob, std::string* err) {
  const st2_model_config& cfg = e.cfg;
  Packer pk{e, blob};
  PDenoiser d;
  const std::string N = "denoiser.";
  d.features = cfg.dn_channels + cfg.dn_embedding;
  d.time_w = pk.vec(N + "to_time.0.0.weights");
  d.time_lin = pk.lin_t(N + "to_time.0.1.weight"); d.time_b = pk.vec(N + "to_time.0.1.bias");
  d.map0 = pk.lin_t(N + "to_mapping.0.weight");    d.map0_b = pk.vec(N + "to_mapping.0.bias");
  d.map2 = pk.lin_t(N + "to_mapping.2.weight");    d.map2_b = pk.vec(N + "to_mapping.2.bias");
  const int F = d.features;
  if (cfg.multispeaker) {
    d.feat = pk.lin_t(N + "to_features.0.weight"); d.feat_b = pk.vec(N + "to_features.0.bias");
    // every AdaLayerNorm fc of the net in one [style][J] matrix: per block norm (2F) then norm_context (2F)
    const int Fc = cfg.dn_context_features, J = cfg.dn_layers * 4 * F;
    std::vector<float> wt((size_t)Fc * J), b((size_t)J);
    int off = 0;
    for (int i = 0; i < cfg.dn_layers; ++i)
      for (const char* nm : {".attention.norm", ".attention.norm_context"}) {
        const HostTensor* w = pk.get(N + "blocks." + std::to_string(i) + nm + ".fc.weight", {2 * F, Fc});
        const HostTensor* bb = pk.get(N + "blocks." + std::to_string(i) + nm + ".fc.bias", {2 * F});
        if (w && bb)
          for (int j = 0; j < 2 * F; ++j) {
            for (int k = 0; k < Fc; ++k) wt[(size_t)k * J + off + j] = w->data[(size_t)j * Fc + k];
            b[(size_t)off + j] = bb->data[(size_t)j];
          }
        off += 2 * F;
      }
    d.ada_wt = blob.add_f32(wt);
    d.ada_b = blob.add_f32(b);
  }
  for (int i = 0; i < cfg.dn_layers; ++i) {
    const std::string B = N + "blocks." + std::to_string(i);
    PBlock b;
    if (!cfg.multispeaker) {
      b.n_w = pk.vec(B + ".attention.norm.weight", F);          b.n_b = pk.vec(B + ".attention.norm.bias", F);
      b.nc_w = pk.vec(B + ".attention.norm_context.weight", F); b.nc_b = pk.vec(B + ".attention.norm_context.bias", F);
    }
    const int HD = cfg.dn_heads * cfg.dn_head_features, FM = F * cfg.dn_multiplier;
    b.q = pk.conv_w(B + ".attention.to_q.weight", HD, F, 1);
    b.kv = pk.conv_w(B + ".attention.to_kv.weight", 2 * HD, F, 1);
    b.o = pk.conv_w(B + ".attention.attention.to_out.weight", F, HD, 1); b.o_b = pk.vec(B + ".attention.attention.to_out.bias", F);
    b.f1 = pk.conv_w(B + ".feed_forward.0.weight", FM, F, 1);            b.f1_b = pk.vec(B + ".feed_forward.0.bias", FM);
    b.f1_out = b.f1.C_out;
    b.f2 = pk.conv_w(B + ".feed_forward.2.weight", F, FM, 1);            b.f2_b = pk.vec(B + ".feed_forward.2.bias", F);
    d.blocks.push_back(b);
  }
  {  // to_out.1: Conv1d(F, channels, 1) applied to the token mean -> [F][channels] for st2_style_fc
    const HostTensor* w = pk.get(N + "to_out.1.weight", {cfg.dn_channels, F, 1});
    if (w) {
      const int O = (int)w->shape[0], I = (int)w->shape[1];
      std::vector<float> v((size_t)O * I);
      for (int o = 0; o < O; ++o)
        for (int i = 0; i < I; ++i) v[(size_t)i * O + o] = w->data[(size_t)o * I + i];
      d.out_t = blob.add_f32(v);
    }
    d.out_b = pk.vec(N + "to_out.1.bias", cfg.dn_channels);
  }
  d.fixed = pk.vec(N + "fixed_embedding.embedding.weight", (int64_t)cfg.dn_max_length * cfg.dn_embedding);
  if (!pk.ok) {
    *err = "denoiser weights (missing or malformed): " + pk.missing;
    return 1;
  }
  d.ready = true;
  e.dn = d;
  return 0;
}

// ------------------------------------------------------------------------------------------------------------------
// conv dispatch == styletts2_amd.ops.conv1d for split-f16 weights
// ------------------------------------------------------------------------------------------------------------------
struct ConvOpt {
  int dil = 1, pad_left = 0;
  const float* bias = nullptr;
  int pro = ST2_PRO_NONE;
  float slope = 0.f;
  const float* stats = nullptr;
  const float* gamma = nullptr;
  const float* beta = nullptr;
  int64_t gb_bs = 0;
  int gb_seg = 0;  // > 0: affine row = column / gb_seg (token-merged view, per-utterance AdaLayerNorm; xs path only)
  int gamma_plus_one = 0;
  const float* alpha = nullptr;
  View res; int res_shift = 0;
  View res2;
  float div = 1.0f;
  int act = ST2_ACT_NONE, act_split = 0;
  float act_slope = 0.f;
  float* stats_out = nullptr;  // want_stats: [B][C_out][2]
};

float x_scale_for(int pro) {
  return (pro == ST2_PRO_ADAIN_LEAKY || pro == ST2_PRO_ADAIN_SNAKE || pro == ST2_PRO_COLNORM) ? 8.0f : 1.0f;
}
int xs_row_slots(int L) { return XS_HALO + (std::max(L, 1) + 1 + 511) / 512 * 512 + 96; }

void conv(Ctx& c, const st2_engine& e, const View& x, const SplitW& w, const View& y, const ConvOpt& o) {
  st2_conv_desc d;
  memset(&d, 0, sizeof(d));
  d.B = x.B; d.C_in = x.C; d.C_out = y.C; d.L_in = x.L; d.L_out = y.L; d.ks = w.ks; d.dil = o.dil; d.pad_left = o.pad_left;
  d.wq = e.P<const void>(w.wq); d.wq_co_pad = w.co_pad; d.wq_cin_pad = w.cin_pad;
  d.x_scale = e.x_scale(w.site, x_scale_for(o.pro));  // calibrated per layer, else the rule (both powers of two)
  d.out_scale = 1.0f / d.x_scale;
  if (!c.dry) st2_headroom_set_site(&e, w.site);  // telemetry / calibration: which conv the next launches belong to
  d.w_row_scale = e.F(w.row_scale);
  d.bias = o.bias;
  d.y = y.p; d.y_bs = y.bs; d.y_cs = y.cs;
  if (o.res.ok()) { d.res = o.res.p; d.res_bs = o.res.bs; d.res_cs = o.res.cs; d.res_shift = o.res_shift; }
  if (o.res2.ok()) { d.res2 = o.res2.p; d.res2_bs = o.res2.bs; d.res2_cs = o.res2.cs; }
  d.div = o.div;
  d.act = o.act; d.act_split = o.act_split; d.act_slope = o.act_slope;
  if (w.C_in != x.C || w.C_out != y.C) {
    if (c.rc == 0) { st2_set_error("engine: conv weight is %d->%d, call has %d->%d", w.C_in, w.C_out, x.C, y.C); c.rc = 1; }
    return;
  }
  // == styletts2_amd.ops.prefer_fused: HBM-bound layers (C <= 64; k = 3 at C <= 128) skip the activation pass
  const bool prefer_fused = o.pro != ST2_PRO_NONE && (x.C <= FUSED_MAX_C || (w.ks <= 3 && x.C <= FUSED_K3_MAX_C));
  const bool use_xs = o.pad_left <= XS_HALO && x.L >= XS_MIN_L && (o.pro != ST2_PRO_NONE || x.C >= XS_MIN_C_PLAIN) &&
                      !prefer_fused;
  const int64_t mark = c.a.off;
  if (use_xs) {
    const int cg = (x.C + 31) / 32 * 32 / 8;
    const int Lp = xs_row_slots(x.L);
    void* xs = c.a.alloc((int64_t)x.B * 2 * cg * Lp * 16);
    RUN(c, g_be.act_split(x.p, x.bs, x.cs, x.B, x.C, x.L, o.pro, o.slope, o.stats, o.gamma, o.beta, o.gb_bs, o.gb_seg,
                          o.gamma_plus_one, o.alpha, d.x_scale, xs, cg, Lp, XS_HALO, c.stream));
    d.xs = xs; d.xs_cg = cg; d.xs_lp = Lp; d.xs_halo = XS_HALO;
    float* part = nullptr;
    int nt = 0;
    if (o.stats_out) {
      // small grids (one utterance) run 64 / 32-column tiles, one partial-sum slot per tile: the library says which
      const int pc = st2_conv1d_xs_part_cols(&d);
      nt = (y.L + pc - 1) / pc;
      part = c.a.f32((int64_t)y.B * y.C * nt * 3);  // (sum, sum of squares) per slot, then the slots' shifts
      d.part = part; d.part_nt = nt; d.part_cols = pc;
    }
    RUN(c, g_be.conv1d_xs(&d, c.stream));
    if (o.stats_out)
      RUN(c, g_be.stats_finalize(part, y.B * y.C, nt, y.L, 1e-5f, o.stats_out, d.part_cols, c.stream));
  } else {
    if (o.gb_seg > 0) {
      if (c.rc == 0) { st2_set_error("engine: a per-segment affine (gb_seg) needs the act_split + xs path"); c.rc = 1; }
      return;
    }
    d.x = x.p; d.x_bs = x.bs; d.x_cs = x.cs;
    d.pro = o.pro; d.slope = o.slope;
    d.stats = o.stats; d.gamma = o.gamma; d.beta = o.beta; d.gb_bs = o.gb_bs; d.gamma_plus_one = o.gamma_plus_one;
    d.alpha = o.alpha;
    float* part = nullptr;
    int nt = 0;
    if (o.stats_out) {  // statistics of the output from the epilogue's per-tile partial sums
      nt = (y.L + 127) / 128;
      part = c.a.f32((int64_t)y.B * y.C * nt * 3);
      d.part = part; d.part_nt = nt;
    } else {
      const int64_t skb = st2_conv1d_f16s_splitk_bytes(&d);  // skinny layers run split-K inside the workspace
      if (skb > 0) {
        d.splitk_ws = c.a.alloc(skb);
        d.splitk_ws_bytes = skb;
      }
    }
    RUN(c, g_be.conv1d_f16s(&d, c.stream));
    if (o.stats_out)
      RUN(c, g_be.stats_finalize(part, y.B * y.C, nt, y.L, 1e-5f, o.stats_out, 128, c.stream));
  }
  c.a.off = mark;  // planes / partial sums are dead once the launches are queued (stream order protects reuse)
  if (!c.dry) st2_headroom_set_site(nullptr, -1);
}

float* new_stats(Ctx& c, int B, int C) { return c.a.f32((int64_t)B * C * 2); }

// ------------------------------------------------------------------------------------------------------------------
// decoder plan == styletts2_amd/decoder.py
// ------------------------------------------------------------------------------------------------------------------
struct DecRun {
  Ctx& c;
  const st2_engine& e;
  const float* h;  // style bank output [B][J]
  int J;
  const float* gamma(int off) const { return h + off; }
  const float* beta(int off, int channels) const { return h + off + channels; }
};

// AdaINResBlock1.forward (Modules/istftnet.py:66-75); the MRF sum / division ride in the last conv's epilogue
View run_resblock1(DecRun& r, const PResBlock1& p, View x, const float* x_stats, View mrf_acc, bool mrf_last, int n_mrf,
                   View out) {
  Ctx& c = r.c;
  const int C = p.channels, ks = p.ks, B = x.B, L = x.L;
  const float* st = x_stats;
  if (!st) {
    float* s0 = new_stats(c, B, C);
    RUN(c, g_be.instnorm_stats(x.p, x.bs, x.cs, B, C, L, 1e-5f, s0, c.stream));
    st = s0;
  }
  View xt = new_ncl(c, B, C, L);
  View ping[2] = {new_ncl(c, B, C, L), new_ncl(c, B, C, L)};
  float* st2 = new_stats(c, B, C);
  float* stn[2] = {new_stats(c, B, C), new_stats(c, B, C)};
  for (int i = 0; i < 3; ++i) {
    const int d = p.dil[i];
    ConvOpt o1;
    o1.dil = d; o1.pad_left = (ks * d - d) / 2; o1.bias = r.e.F(p.c1[i].bias); o1.pro = ST2_PRO_ADAIN_SNAKE;
    o1.stats = st; o1.gamma = r.gamma(p.ad1[i]); o1.beta = r.beta(p.ad1[i], C); o1.gb_bs = r.J;
    o1.alpha = r.e.F(p.a1[i]); o1.stats_out = st2;
    conv(c, r.e, x, p.c1[i].w, xt, o1);
    const bool last = i == 2;
    ConvOpt o2;
    o2.dil = 1; o2.pad_left = (ks - 1) / 2; o2.bias = r.e.F(p.c2[i].bias); o2.pro = ST2_PRO_ADAIN_SNAKE;
    o2.stats = st2; o2.gamma = r.gamma(p.ad2[i]); o2.beta = r.beta(p.ad2[i], C); o2.gb_bs = r.J;
    o2.alpha = r.e.F(p.a2[i]); o2.res = x;
    if (last) {
      o2.res2 = mrf_acc;
      o2.div = mrf_last ? (float)n_mrf : 1.0f;
      conv(c, r.e, xt, p.c2[i].w, out, o2);
      x = out;
    } else {
      o2.stats_out = stn[i & 1];
      conv(c, r.e, xt, p.c2[i].w, ping[i & 1], o2);
      x = ping[i & 1];
      st = stn[i & 1];
    }
  }
  return x;
}

// AdainResBlk1d.forward (Modules/istftnet.py:435-454): (residual(x, s) + shortcut(x)) / sqrt(2)
void run_adain_resblk(DecRun& r, const PAdainResBlk& p, const View& x, const View& out) {
  Ctx& c = r.c;
  const int B = x.B, L = x.L;
  const int64_t mark = c.a.off;
  float* st1 = new_stats(c, B, p.dim_in);
  RUN(c, g_be.instnorm_stats(x.p, x.bs, x.cs, B, p.dim_in, L, 1e-5f, st1, c.stream));
  float* st2 = new_stats(c, B, p.dim_out);
  const int Lo = p.upsample ? 2 * L : L;
  View t1 = new_ncl(c, B, p.dim_out, Lo);
  if (p.upsample) {
    View u = new_ncl(c, B, p.dim_in, 2 * L);
    RUN(c, g_be.adain_leaky_pool(x.p, x.bs, x.cs, st1, r.gamma(p.n1), r.beta(p.n1, p.dim_in), r.J, 0.2f,
                                 r.e.F(p.pool_w), r.e.F(p.pool_b), u.p, u.bs, u.cs, B, p.dim_in, L, c.stream));
    ConvOpt o;
    o.pad_left = 1; o.bias = r.e.F(p.conv1.bias); o.stats_out = st2;
    conv(c, r.e, u, p.conv1.w, t1, o);
  } else {
    ConvOpt o;
    o.pad_left = 1; o.bias = r.e.F(p.conv1.bias); o.pro = ST2_PRO_ADAIN_LEAKY; o.slope = 0.2f; o.stats = st1;
    o.gamma = r.gamma(p.n1); o.beta = r.beta(p.n1, p.dim_in); o.gb_bs = r.J; o.stats_out = st2;
    conv(c, r.e, x, p.conv1.w, t1, o);
  }
  View sc = x;
  if (p.learned_sc) {  // the 1x1 shortcut commutes with nearest x2 up-sampling: it runs at the low rate
    sc = new_ncl(c, B, p.dim_out, L);
    ConvOpt o;
    conv(c, r.e, x, p.sc.w, sc, o);
  }
  ConvOpt o;
  o.pad_left = 1; o.bias = r.e.F(p.conv2.bias); o.pro = ST2_PRO_ADAIN_LEAKY; o.slope = 0.2f; o.stats = st2;
  o.gamma = r.gamma(p.n2); o.beta = r.beta(p.n2, p.dim_out); o.gb_bs = r.J; o.res = sc;
  o.res_shift = p.upsample ? 1 : 0;
  o.div = (float)sqrt(2.0);  // python passes math.sqrt(2) through a c_float: the same fp32 value
  conv(c, r.e, t1, p.conv2.w, out, o);
  c.a.off = mark;
}

void tap(Ctx& c, const View& v, float* dst) {
  if (!dst) return;
  RUN(c, g_be.copy_ncl(v.p, v.bs, v.cs, dst, (int64_t)v.C * v.L, v.L, v.B, v.C, v.L, c.stream));
}

int decoder_plan(Ctx& c, const st2_engine& e, const float* asr_p, const float* f0_p, const float* n_p, const float* s_p,
                 const float* sine_noise, const float* har_inject, int B, int T, float* wave,
                 const st2_decoder_taps* taps) {
  const st2_model_config& cfg = e.cfg;
  const PDecoder& d = e.dec;
  const PGenerator& g = d.gen;
  const bool ist = cfg.decoder_kind == 0;
  const int Cin = cfg.dim_in, nu = cfg.n_upsamples, nk = cfg.n_resblock_kernels;
  const int T2 = 2 * T;
  st2_decoder_taps notaps;
  memset(&notaps, 0, sizeof(notaps));
  if (!taps) taps = &notaps;

  float* h = c.a.f32((int64_t)B * d.J);
  RUN(c, g_be.style_fc(s_p, B, cfg.style_dim, e.F(d.bank_wt), e.F(d.bank_b), d.J, ST2_ACT_NONE, h, c.stream));
  DecRun r{c, e, h, d.J};

  View asr = wrap(asr_p, B, Cin, T);
  View f0 = wrap(f0_p, B, 1, T2), nn = wrap(n_p, B, 1, T2);
  // [x(1024) | asr_res(64) | F0 | N] lives in one buffer; producers write their channel slices in place
  View cat = new_ncl(c, B, 1024 + 64 + 2, T, false);
  View cat0 = new_ncl(c, B, Cin + 2, T, false);
  RUN(c, g_be.copy_ncl(asr.p, asr.bs, asr.cs, cat0.p, cat0.bs, cat0.cs, B, Cin, T, c.stream));
  {
    View a = cat0.rows(Cin, Cin + 1), b = cat0.rows(Cin + 1, Cin + 2);
    RUN(c, g_be.conv1d_direct(f0.p, f0.bs, f0.cs, e.F(d.f0_w), e.F(d.f0_b), a.p, a.bs, a.cs, B, 1, 1, T2, T, 3, 2, 1, c.stream));
    RUN(c, g_be.conv1d_direct(nn.p, nn.bs, nn.cs, e.F(d.n_w), e.F(d.n_b), b.p, b.bs, b.cs, B, 1, 1, T2, T, 3, 2, 1, c.stream));
    View src = cat0.rows(Cin, Cin + 2), dst = cat.rows(1088, 1090);
    RUN(c, g_be.copy_ncl(src.p, src.bs, src.cs, dst.p, dst.bs, dst.cs, B, 2, T, c.stream));
  }
  {
    ConvOpt o;
    o.bias = e.F(d.asr_res.bias);
    conv(c, e, asr, d.asr_res.w, cat.rows(1024, 1088), o);
  }
  run_adain_resblk(r, d.encode, cat0, cat.rows(0, 1024));
  tap(c, cat.rows(0, 1024), taps->encode);
  View x = new_ncl(c, B, 512, T2);
  for (int i = 0; i < 4; ++i) {
    if (d.decode[i].upsample)
      run_adain_resblk(r, d.decode[i], cat, x);
    else
      run_adain_resblk(r, d.decode[i], cat, cat.rows(0, 1024));
  }
  tap(c, x, taps->front);

  // ---- generator (istftnet.py:350-380 / hifigan.py:321-347) ---------------------------------------------------------
  int up_scale = prod_from(cfg.upsample_rates, 0, nu) * (ist ? cfg.gen_istft_hop : 1);
  const int L = T2 * up_scale;  // samples
  const int n_fft = cfg.gen_istft_n_fft, hop = cfg.gen_istft_hop;
  View har;
  if (har_inject) {
    har = ist ? wrap(har_inject, B, n_fft + 2, L / hop + 1) : wrap(har_inject, B, 1, L);
  } else {
    float* scratch = c.a.f32((int64_t)B * 9 * T2);
    float* hs = c.a.f32((int64_t)B * L);
    RUN(c, g_be.har_source(f0_p, B, T2, up_scale, 9, sine_noise, e.F(g.lin_w), e.F(g.lin_b), 0.1f, 0.003f, 10.0f,
                           24000.0f, scratch, hs, c.stream));
    if (taps->har_source) RUN(c, g_be.copy_ncl(hs, L, L, taps->har_source, L, L, B, 1, L, c.stream));
    if (ist) {
      har = new_ncl(c, B, n_fft + 2, L / hop + 1, false);
      RUN(c, g_be.stft_mag_phase(hs, B, L, n_fft, hop, har.p, har.bs, har.cs, c.stream));
    } else {
      har = wrap(hs, B, 1, L);
    }
  }
  if (ist) tap(c, har, taps->har);

  for (int i = 0; i < nu; ++i) {
    const int u = cfg.upsample_rates[i], k = cfg.upsample_kernel_sizes[i], C = g.channels[i];
    const bool last = i + 1 == nu;
    const int L_in = x.L;
    int pad, L_raw;
    if (ist) {
      pad = (k - u) / 2;
      L_raw = (L_in - 1) * u - 2 * ((k - u) / 2) + k;
    } else {
      pad = u / 2 + u % 2;
      L_raw = (L_in - 1) * u - 2 * (u / 2 + u % 2) + k + u % 2;
    }
    const bool reflect = ist && last;
    const int L_out = L_raw + (reflect ? 1 : 0);
    // persistent across the stage: the stage output and the MRF accumulators
    View x_next = new_ncl(c, B, C, L_out);
    const int64_t stage_mark = c.a.off;
    // harmonic-source branch (istftnet.py:361-362 / hifigan.py:330-331)
    View xs_src = new_ncl(c, B, C, L_out);
    {
      const int64_t m = c.a.off;
      View xs0 = new_ncl(c, B, C, L_out);
      const int stride_f0 = g.noise_stride[i];
      ConvOpt o;
      o.bias = e.F(g.noise_b[i]);
      if (stride_f0 > 1) {
        const int pad_f0 = (stride_f0 + 1) / 2;
        const int L_ns = (har.L + 2 * pad_f0 - 2 * stride_f0) / stride_f0 + 1;
        View harp = new_ncl(c, B, har.C * stride_f0, L_ns + 1, false);
        RUN(c, g_be.phase_split(har.p, har.bs, har.cs, B, har.C, har.L, stride_f0, pad_f0, harp.p, harp.bs, harp.cs,
                                L_ns + 1, c.stream));
        if (L_ns != L_out && c.rc == 0) { st2_set_error("engine: noise conv length %d != stage length %d", L_ns, L_out); c.rc = 1; }
        conv(c, e, harp, g.noise_wt[i], xs0, o);
      } else {
        conv(c, e, har, g.noise_wt[i], xs0, o);
      }
      View nul;
      run_resblock1(r, g.noise_res[i], xs0, nullptr, nul, false, nk, xs_src);
      // xs_src was allocated before m: keep it, drop the branch temporaries
      c.a.off = m;
    }
    // up-sampling ConvTranspose1d as polyphase GEMM + interleave (istftnet.py:360,364-368)
    View xu = new_ncl(c, B, C, L_out);
    float* st = new_stats(c, B, C);
    {
      const int64_t m = c.a.off;
      View Y = new_ncl(c, B, u * C, L_in + 1);
      ConvOpt o;
      o.pad_left = 1;
      if (ist) { o.pro = ST2_PRO_LEAKY; o.slope = 0.1f; } else { o.pro = ST2_PRO_SNAKE; o.alpha = e.F(g.alphas[i]); }
      conv(c, e, x, g.ups_wt[i], Y, o);
      const int nt = (L_out + CVT_TILE - 1) / CVT_TILE;
      float* part = c.a.f32((int64_t)B * C * nt * 3);
      RUN(c, g_be.convt_interleave_stats(Y.p, Y.bs, Y.cs, L_in + 1, e.F(g.ups_b[i]), xs_src.p, xs_src.bs, xs_src.cs,
                                         xu.p, xu.bs, xu.cs, B, C, u, pad, L_raw, reflect ? 1 : 0, part, nt, c.stream));
      RUN(c, g_be.stats_finalize(part, B * C, nt, L_out, 1e-5f, st, CVT_TILE, c.stream));
      c.a.off = m;
    }
    // multi-receptive-field fusion (istftnet.py:369-375): ((r0 + r1) + r2) / n in the last convs' epilogues
    View acc[2] = {new_ncl(c, B, C, L_out), new_ncl(c, B, C, L_out)};
    View prev;
    for (int j = 0; j < nk; ++j) {
      const int64_t m = c.a.off;
      const bool lastk = j + 1 == nk;
      View out = lastk ? x_next : acc[j & 1];
      run_resblock1(r, g.resblocks[(size_t)i * nk + j], xu, st, prev, lastk, nk, out);
      prev = out;
      c.a.off = m;
    }
    x = x_next;
    c.a.off = stage_mark;
    if (i < 4) tap(c, x, taps->stage[i]);
  }
  if (ist) {
    const int nb = n_fft / 2 + 1;
    View sp = new_ncl(c, B, n_fft + 2, x.L, false);
    ConvOpt o;
    o.pad_left = 3; o.bias = e.F(g.post.bias); o.pro = ST2_PRO_LEAKY; o.slope = 0.01f; o.act = ST2_ACT_EXP_SIN;
    o.act_split = nb;
    conv(c, e, x, g.post.w, sp, o);
    tap(c, sp, taps->spec_phase);
    RUN(c, g_be.istft(sp.p, sp.bs, sp.cs, B, x.L, n_fft, hop, wave, (int64_t)hop * (x.L - 1), c.stream));
  } else {
    View w = wrap(wave, B, 1, x.L);
    ConvOpt o;
    o.pad_left = 3; o.bias = e.F(g.post.bias); o.pro = ST2_PRO_SNAKE; o.alpha = e.F(g.alphas[(size_t)nu]);
    o.act = ST2_ACT_TANH;
    conv(c, e, x, g.post.w, w, o);
  }
  return c.rc;
}

// ------------------------------------------------------------------------------------------------------------------
// sampler plan == styletts2_amd/diffusion.py (DiffusionSampler.forward + _Transformer sessions)
// ------------------------------------------------------------------------------------------------------------------
struct Sess {
  Ctx& c;
  const st2_engine& e;
  int B, N;
  bool merged;
  const int32_t* key_len;
  View bases[2];
  int nbases;
  const float* feat_map;  // [B][F] or null
  const float* ada;       // [B][layers*4F] or null
  double scale;  // embedding_scale (classifier-free guidance weight)
};

// [B][C][N] activation buffer; in the merged layout a strided view of a [C][B*N] block
View dn_alloc(Sess& s, int C) {
  View v;
  v.B = s.B; v.C = C; v.L = s.N;
  v.p = s.c.a.f32((int64_t)s.B * C * s.N);
  if (s.merged) { v.bs = s.N; v.cs = s.B * s.N; } else { v.bs = (int64_t)C * s.N; v.cs = s.N; }
  return v;
}
// the view the k=1 convs consume: [1][C][B*N] (merged) or [B][C][N]
View dn_cv(const Sess& s, const View& t) {
  if (!s.merged) return t;
  View v = t;
  v.B = 1; v.L = s.B * s.N; v.bs = (int64_t)t.C * v.L; v.cs = v.L;
  return v;
}

const float* dn_mapping(Sess& s, float c_noise) {
  Ctx& c = s.c;
  const st2_engine& e = s.e;
  const PDenoiser& d = e.dn;
  const int F = d.features, H2 = e.cfg.dn_channels / 2;
  float* four = c.a.f32((int64_t)s.B * (1 + 2 * H2));
  RUN(c, g_be.time_features(c_noise, e.F(d.time_w), H2, s.B, four, c.stream));
  float* m = c.a.f32((int64_t)s.B * F);
  RUN(c, g_be.style_fc(four, s.B, 1 + 2 * H2, e.F(d.time_lin), e.F(d.time_b), F, ST2_ACT_GELU, m, c.stream));
  if (s.feat_map) {
    float* m2 = c.a.f32((int64_t)s.B * F);
    RUN(c, g_be.axpbypcz(m, 1.0f, s.feat_map, 1.0f, nullptr, 0.0f, m2, (int64_t)s.B * F, c.stream));
    m = m2;
  }
  float* m3 = c.a.f32((int64_t)s.B * F);
  RUN(c, g_be.style_fc(m, s.B, F, e.F(d.map0), e.F(d.map0_b), F, ST2_ACT_GELU, m3, c.stream));
  float* m4 = c.a.f32((int64_t)s.B * F);
  RUN(c, g_be.style_fc(m3, s.B, F, e.F(d.map2), e.F(d.map2_b), F, ST2_ACT_GELU, m4, c.stream));
  return m4;
}

// one denoiser evaluation on base buffer `base`: x [B][C] -> out [B][C]   (_Transformer._run)
void dn_run(Sess& s, View base, const float* x, const float* m, float* out) {
  Ctx& c = s.c;
  const st2_engine& e = s.e;
  const PDenoiser& d = e.dn;
  const st2_model_config& cfg = e.cfg;
  const int B = s.B, N = s.N, Fz = d.features, C = cfg.dn_channels;
  const int mid = cfg.dn_heads * cfg.dn_head_features;
  const int64_t mark = c.a.off;
  View b0 = base.rows(0, C);
  RUN(c, g_be.broadcast_cols(x, C, b0.p, b0.bs, b0.cs, B, C, N, c.stream));
  View X = dn_alloc(s, Fz);
  RUN(c, g_be.add_chanvec(base.p, base.bs, base.cs, m, Fz, X.p, X.bs, X.cs, B, Fz, N, c.stream));
  const int nblk = (int)d.blocks.size();
  for (int i = 0; i < nblk; ++i) {
    const PBlock& b = d.blocks[(size_t)i];
    float* st = c.a.f32((int64_t)B * N * 2);
    RUN(c, g_be.colnorm_stats(X.p, X.bs, X.cs, B, Fz, N, 1e-5f, st, c.stream));
    View qkv = dn_alloc(s, 3 * mid);
    ConvOpt o1, o2;
    o1.pro = o2.pro = ST2_PRO_COLNORM;
    o1.stats = o2.stats = st;  // [B][N][2] == [1][B*N][2] in the merged view
    if (cfg.multispeaker) {
      const int64_t J = (int64_t)nblk * 4 * Fz;
      const float* a = s.ada + (int64_t)4 * Fz * i;
      o1.gamma = a;          o1.beta = a + Fz;     o1.gb_bs = J; o1.gamma_plus_one = 1;
      o2.gamma = a + 2 * Fz; o2.beta = a + 3 * Fz; o2.gb_bs = J; o2.gamma_plus_one = 1;
    } else {
      o1.gamma = e.F(b.n_w);  o1.beta = e.F(b.n_b);
      o2.gamma = e.F(b.nc_w); o2.beta = e.F(b.nc_b);
    }
    // The multispeaker net's AdaLayerNorm affine is per utterance.  Its q / kv convs still run as ONE GEMM over the B*N
    // merged columns: st2_act_split picks the affine row of a column from its utterance (gb_seg = N).  That needs the xs
    // pair (>= XS_MIN_L columns); smaller calls (one sentence of the long-form loop) keep the [B][F][N] view.  Measured
    // at B = 32, N = 100: 2 x 99 us (fused kernel on 32 rows of 100 columns, 0.04-0.08 of the roof) -> ~65 us per layer.
    // ... and conv()'s routing rule: nets of <= FUSED_K3_MAX_C features (small / test configurations) send k = 1 convs with
    // a prologue to the fused kernel, which has no per-segment affine: they keep the per-utterance view (advisor, round 3).
    const bool seg = cfg.multispeaker && s.merged && B > 1 && (int64_t)B * N >= XS_MIN_L && Fz > FUSED_K3_MAX_C;
    if (seg) o1.gb_seg = o2.gb_seg = N;
    const bool per_utt = cfg.multispeaker && !seg;
    View Xv = per_utt ? X : dn_cv(s, X), qv = per_utt ? qkv : dn_cv(s, qkv);
    conv(c, e, Xv, b.q, qv.rows(0, mid), o1);
    conv(c, e, Xv, b.kv, qv.rows(mid, 3 * mid), o2);
    View att = dn_alloc(s, mid);
    View q = qkv.rows(0, mid), k = qkv.rows(mid, 2 * mid), v = qkv.rows(2 * mid, 3 * mid);
    RUN(c, g_be.attention_keylen(q.p, k.p, v.p, q.bs, q.cs, att.p, att.bs, att.cs, B, cfg.dn_heads, cfg.dn_head_features,
                                 N, (float)pow((double)cfg.dn_head_features, -0.5), s.key_len, c.stream));
    View X1 = dn_alloc(s, Fz), hmid = dn_alloc(s, b.f1_out), X2 = dn_alloc(s, Fz);
    ConvOpt oo;
    oo.bias = e.F(b.o_b); oo.res = dn_cv(s, X);
    conv(c, e, dn_cv(s, att), b.o, dn_cv(s, X1), oo);
    ConvOpt of1;
    of1.bias = e.F(b.f1_b); of1.act = ST2_ACT_GELU;
    conv(c, e, dn_cv(s, X1), b.f1, dn_cv(s, hmid), of1);
    ConvOpt of2;
    of2.bias = e.F(b.f2_b); of2.res = dn_cv(s, X1);
    conv(c, e, dn_cv(s, hmid), b.f2, dn_cv(s, X2), of2);
    if (i + 1 < nblk) {
      View Xn = dn_alloc(s, Fz);
      RUN(c, g_be.add_chanvec(X2.p, X2.bs, X2.cs, m, Fz, Xn.p, Xn.bs, Xn.cs, B, Fz, N, c.stream));
      X = Xn;
    } else {
      X = X2;
    }
  }
  float* mean = c.a.f32((int64_t)B * Fz);
  RUN(c, g_be.mean_tokens_len(X.p, X.bs, X.cs, mean, Fz, B, Fz, N, s.key_len, c.stream));
  RUN(c, g_be.style_fc(mean, B, Fz, e.F(d.out_t), e.F(d.out_b), C, ST2_ACT_NONE, out, c.stream));
  c.a.off = mark;
}

// KDiffusion.denoise_fn at one sigma (sampler.py:184-208): out = c_skip * x + c_out * net(c_in * x, c_noise)
void dn_denoise(Sess& s, const float* x, const double* w4, float* out) {
  Ctx& c = s.c;
  const int C = s.e.cfg.dn_channels;
  const int64_t n = (int64_t)s.B * C;
  const int64_t mark = c.a.off;
  const float c_skip = (float)w4[0], c_out = (float)w4[1], c_in = (float)w4[2], c_noise = (float)w4[3];
  float* x_in = c.a.f32(n);
  RUN(c, g_be.axpbypcz(x, c_in, nullptr, 0.f, nullptr, 0.f, x_in, n, c.stream));
  const float* m = dn_mapping(s, c_noise);
  float* pred = c.a.f32(n);
  dn_run(s, s.bases[0], x_in, m, pred);
  if (s.nbases > 1) {  // classifier-free guidance, modules.py:418-423
    float* pm = c.a.f32(n);
    dn_run(s, s.bases[1], x_in, m, pm);
    float* mix = c.a.f32(n);
    RUN(c, g_be.axpbypcz(pm, (float)(1.0 - s.scale), pred, (float)s.scale, nullptr, 0.f, mix, n, c.stream));
    pred = mix;
  }
  RUN(c, g_be.axpbypcz(x, c_skip, pred, c_out, nullptr, 0.f, out, n, c.stream));
  c.a.off = mark;
}

int sampler_plan(Ctx& c, const st2_engine& e, const float* noise, const float* embedding, const View* emb_cm,
                 const float* features, const float* step_noise, const int32_t* lengths, int B, int N, int steps, double scale,
                 const double* table, double sigma0, float* out, float* step_taps) {
  const st2_model_config& cfg = e.cfg;
  const PDenoiser& d = e.dn;
  const int C = cfg.dn_channels, E = cfg.dn_embedding, Fz = d.features;
  Sess s{c, e, B, N, true, lengths, {}, 1, nullptr, nullptr, scale};  // token-merged storage for both denoisers
  // session: everything constant across the 2*(steps-1) net calls
  s.bases[0] = dn_alloc(s, Fz);
  {
    View dst = s.bases[0].rows(C, Fz);
    if (emb_cm)  // already channel-major on the device (front plan: PL-BERT's merged output)
      RUN(c, g_be.copy_ncl(emb_cm->p, emb_cm->bs, emb_cm->cs, dst.p, dst.bs, dst.cs, B, E, N, c.stream));
    else
      RUN(c, g_be.tokens_to_channels(embedding, (int64_t)N * E, B, N, E, dst.p, dst.bs, dst.cs, c.stream));
  }
  if (scale != 1.0) {
    s.nbases = 2;
    s.bases[1] = dn_alloc(s, Fz);
    View dst = s.bases[1].rows(C, Fz);
    RUN(c, g_be.tokens_to_channels(e.F(d.fixed), 0, B, N, E, dst.p, dst.bs, dst.cs, c.stream));
  }
  if (cfg.multispeaker) {
    if (!features) { st2_set_error("st2_sampler_run: the multispeaker denoiser needs `features`"); return 1; }
    float* fm = c.a.f32((int64_t)B * Fz);
    RUN(c, g_be.style_fc(features, B, cfg.dn_context_features, e.F(d.feat), e.F(d.feat_b), Fz, ST2_ACT_GELU, fm, c.stream));
    const int J = cfg.dn_layers * 4 * Fz;
    float* ada = c.a.f32((int64_t)B * J);
    RUN(c, g_be.style_fc(features, B, cfg.dn_context_features, e.F(d.ada_wt), e.F(d.ada_b), J, ST2_ACT_NONE, ada, c.stream));
    s.feat_map = fm;
    s.ada = ada;
  }
  const int64_t n = (int64_t)B * C;
  float* xa = c.a.f32(n);
  float* xb = c.a.f32(n);
  float* x = xa;
  RUN(c, g_be.axpbypcz(noise, (float)sigma0, nullptr, 0.f, nullptr, 0.f, x, n, c.stream));
  for (int i = 0; i + 1 < steps; ++i) {
    const double* row = table + (int64_t)i * ST2_SAMPLER_TABLE_COLS;
    const int64_t mark = c.a.off;
    float* den = c.a.f32(n);
    dn_denoise(s, x, row + 0, den);
    // d = (x - den) / sigma ; x_mid = x + d * (sigma_mid - sigma)
    const double k = row[8];
    float* x_mid = c.a.f32(n);
    RUN(c, g_be.axpbypcz(x, (float)(1.0 + k), den, (float)(-k), nullptr, 0.f, x_mid, n, c.stream));
    float* den_mid = c.a.f32(n);
    dn_denoise(s, x_mid, row + 4, den_mid);
    // d_mid = (x_mid - den_mid) / sigma_mid ; x = x + d_mid * (sigma_down - sigma) + eps * sigma_up
    const double k2 = row[9];
    float* d_mid = c.a.f32(n);
    RUN(c, g_be.axpbypcz(x_mid, (float)k2, den_mid, (float)(-k2), nullptr, 0.f, d_mid, n, c.stream));
    float* xn = (x == xa) ? xb : xa;
    RUN(c, g_be.axpbypcz(x, 1.0f, d_mid, 1.0f, step_noise + (int64_t)i * n, (float)row[10], xn, n, c.stream));
    x = xn;
    if (step_taps) RUN(c, g_be.axpbypcz(x, 1.0f, nullptr, 0.f, nullptr, 0.f, step_taps + (int64_t)i * n, n, c.stream));
    c.a.off = mark;
  }
  RUN(c, g_be.axpbypcz(x, 1.0f, nullptr, 0.f, nullptr, 0.f, out, n, c.stream));
  return c.rc;
}

// ------------------------------------------------------------------------------------------------------------------
// prosody plan == the notebooks' alignment expansion + ProsodyPredictor.F0Ntrain (styletts2_amd/text.py, pipeline.py)
// ------------------------------------------------------------------------------------------------------------------
PLstm pack_lstm(Packer& pk, const std::string& prefix) {
  PLstm l;
  const HostTensor* wf = pk.get(prefix + ".weight_ih_l0");
  const HostTensor* wr = pk.get(prefix + ".weight_ih_l0_reverse");
  const HostTensor* hf = pk.get(prefix + ".weight_hh_l0");
  const HostTensor* hr = pk.get(prefix + ".weight_hh_l0_reverse");
  const HostTensor* bif = pk.get(prefix + ".bias_ih_l0");
  const HostTensor* bhf = pk.get(prefix + ".bias_hh_l0");
  const HostTensor* bir = pk.get(prefix + ".bias_ih_l0_reverse");
  const HostTensor* bhr = pk.get(prefix + ".bias_hh_l0_reverse");
  if (!wf || !wr || !hf || !hr || !bif || !bhf || !bir || !bhr) return l;
  if (wf->shape.size() != 2 || wf->shape[0] % 4 != 0) { pk.fail(prefix + ".weight_ih_l0 must be [4H, I]"); return l; }
  const int G4 = (int)wf->shape[0], I = (int)wf->shape[1], H = G4 / 4;
  for (const HostTensor* t : {wr})
    if (t->shape != wf->shape) { pk.fail(prefix + ": the two directions' input weights differ in shape"); return l; }
  for (const HostTensor* t : {hf, hr})
    if (t->shape.size() != 2 || t->shape[0] != G4 || t->shape[1] != H) { pk.fail(prefix + ".weight_hh_l0* must be [4H, H]"); return l; }
  for (const HostTensor* t : {bif, bhf, bir, bhr})
    if (t->numel() != G4) { pk.fail(prefix + ".bias_* must have 4H elements"); return l; }
  l.H = H;
  std::vector<float> w((size_t)2 * G4 * I);  // cat([W_ih, W_ih_reverse]) as a k = 1 conv weight [8H][I][1]
  std::copy(wf->data.begin(), wf->data.end(), w.begin());
  std::copy(wr->data.begin(), wr->data.end(), w.begin() + (size_t)G4 * I);
  l.w_ih = pack_split(pk.blob, prefix + ".weight_ih_l0", w.data(), 2 * G4, I, 1);
  std::vector<float> b((size_t)2 * G4);
  for (int i = 0; i < G4; ++i) {
    b[(size_t)i] = bif->data[(size_t)i] + bhf->data[(size_t)i];
    b[(size_t)G4 + i] = bir->data[(size_t)i] + bhr->data[(size_t)i];
  }
  l.bias = pk.blob.add_f32(b);
  std::vector<float> t((size_t)2 * H * G4);  // stack([W_hh.t(), W_hh_reverse.t()]): [2][H][4H]
  for (int dir = 0; dir < 2; ++dir) {
    const HostTensor* h = dir ? hr : hf;
    for (int r = 0; r < G4; ++r)
      for (int k = 0; k < H; ++k) t[((size_t)dir * H + k) * G4 + r] = h->data[(size_t)r * H + k];
  }
  l.whh_t = pk.blob.add_f32(t);
  return l;
}

int pack_predictor(st2_engine& e, Blob& blob, std::string* err) {
  const st2_model_config& cfg = e.cfg;
  Packer pk{e, blob};
  PPredictor p;
  Bank bank;
  const std::string P = "predictor.";
  const int dh = cfg.pred_hidden;
  if (dh <= 0) { *err = "st2_model_config.pred_hidden is not set"; return 1; }
  p.shared = pack_lstm(pk, P + "shared");
  // registration order of text.ProsodyPredictor._prepare: F0[0..2] then N[0..2], norm1 then norm2 each
  const int din[3] = {dh, dh, dh / 2}, dout[3] = {dh, dh / 2, dh / 2};
  for (int path = 0; path < 2; ++path) {
    for (int i = 0; i < 3; ++i) {
      const std::string pre = P + (path ? "N." : "F0.") + std::to_string(i);
      PAdainResBlk& r = path ? p.n[i] : p.f0[i];
      r = pack_adain_resblk(pk, pre, din[i], dout[i], i == 1);
      r.n1 = bank.add(pk, pre + ".norm1", din[i]);
      r.n2 = bank.add(pk, pre + ".norm2", dout[i]);
    }
  }
  p.f0p_w = pk.vec(P + "F0_proj.weight", dh / 2); p.f0p_b = pk.vec(P + "F0_proj.bias", 1);  // Conv1d(d_hid / 2, 1, 1)
  p.np_w = pk.vec(P + "N_proj.weight", dh / 2);   p.np_b = pk.vec(P + "N_proj.bias", 1);
  if (pk.ok && p.shared.H * 2 != dh) pk.fail(P + "shared: hidden size does not match st2_model_config.pred_hidden");
  if (!pk.ok) { *err = "predictor parameter missing or malformed: " + pk.missing; return 1; }
  p.J = bank.J;
  bank.pack(pk, cfg.style_dim, &p.bank_wt, &p.bank_b);
  p.ready = true;
  e.pred = p;
  return 0;
}

// asr[B][dim_in][T] = expand(t_en), (F0, N)[B][2T] = F0Ntrain(expand(d), s); d_cm [B][pred_hidden + style_dim][N] is the
// duration encoder's output channel-major, dur int64 [B][N] with rows summing to T
int prosody_plan(Ctx& c, const st2_engine& e, const float* d_cm, const float* t_en, const int64_t* dur, const float* s_p,
                 int B, int N, int T, int shift, float* asr, float* f0, float* nn) {
  const st2_model_config& cfg = e.cfg;
  const PPredictor& p = e.pred;
  const int dh = cfg.pred_hidden, Cd = dh + cfg.style_dim, Ct = cfg.dim_in;
  View en = new_ncl(c, B, Cd, T, false);
  RUN(c, g_be.expand_by_durations(d_cm, (int64_t)Cd * N, N, dur, B, Cd, N, T, shift, en.p, en.bs, en.cs, c.stream));
  RUN(c, g_be.expand_by_durations(t_en, (int64_t)Ct * N, N, dur, B, Ct, N, T, shift, asr, (int64_t)Ct * T, T, c.stream));
  // shared BiLSTM: input projection of every frame as one k = 1 conv, then the recurrence
  const int H = p.shared.H;
  View G = new_ncl(c, B, 8 * H, T, false);
  {
    ConvOpt o;
    o.bias = e.F(p.shared.bias);
    conv(c, e, en, p.shared.w_ih, G, o);
  }
  View y = new_ncl(c, B, 2 * H, T, false);
  const int64_t sb = st2_lstm_coop_scratch_bytes(B);
  void* scratch = sb > 0 ? c.a.alloc(sb) : nullptr;
  RUN(c, g_be.lstm_bidir(G.p, G.bs, G.cs, e.F(p.shared.whh_t), nullptr, B, H, T, y.p, y.bs, y.cs, scratch, sb, c.stream));
  float* h = c.a.f32((int64_t)B * p.J);
  RUN(c, g_be.style_fc(s_p, B, cfg.style_dim, e.F(p.bank_wt), e.F(p.bank_b), p.J, ST2_ACT_NONE, h, c.stream));
  DecRun r{c, e, h, p.J};
  for (int path = 0; path < 2; ++path) {
    const PAdainResBlk* blks = path ? p.n : p.f0;
    View t = y;
    for (int i = 0; i < 3; ++i) {
      View out = new_ncl(c, B, blks[i].dim_out, blks[i].upsample ? 2 * t.L : t.L, false);
      run_adain_resblk(r, blks[i], t, out);
      t = out;
    }
    float* dst = path ? nn : f0;
    RUN(c, g_be.conv1d_direct(t.p, t.bs, t.cs, e.F(path ? p.np_w : p.f0p_w), e.F(path ? p.np_b : p.f0p_b), dst,
                              (int64_t)t.L, t.L, B, t.C, 1, t.L, t.L, 1, 1, 0, c.stream));
  }
  return c.rc;
}

// ------------------------------------------------------------------------------------------------------------------
// duration plan == DurationEncoder.forward + pipeline.predict_durations (styletts2_amd/text.py, pipeline.py)
// ------------------------------------------------------------------------------------------------------------------
int pack_duration(st2_engine& e, Blob& blob, std::string* err) {
  Packer pk{e, blob};
  PDuration d;
  const std::string T = "predictor.text_encoder.lstms.";
  for (int i = 0; pk.has(T + std::to_string(2 * i) + ".weight_ih_l0"); ++i) {
    d.lstms.push_back(pack_lstm(pk, T + std::to_string(2 * i)));
    const int dm = 2 * d.lstms.back().H;  // AdaLayerNorm(style_dim, d_model): fc = Linear(style_dim, 2 * d_model)
    d.ada_wt.push_back(pk.lin_t(T + std::to_string(2 * i + 1) + ".fc.weight", 2 * dm, e.cfg.style_dim));
    d.ada_b.push_back(pk.vec(T + std::to_string(2 * i + 1) + ".fc.bias", 2 * dm));
  }
  if (d.lstms.empty()) { *err = "missing predictor parameter predictor.text_encoder.lstms.0.weight_ih_l0"; return 1; }
  d.dur_lstm = pack_lstm(pk, "predictor.lstm");
  const HostTensor* pw = pk.get("predictor.duration_proj.linear_layer.weight", {-1, 2 * d.dur_lstm.H});
  d.proj_w = pk.vec("predictor.duration_proj.linear_layer.weight");
  if (pw) d.proj_b = pk.vec("predictor.duration_proj.linear_layer.bias", pw->shape[0]);
  if (!pk.ok || !pw) { *err = "predictor parameter missing or malformed: " + pk.missing; return 1; }
  d.max_dur = (int)pw->shape[0];
  d.ready = true;
  e.dur = d;
  return 0;
}

View lstm_run(Ctx& c, const st2_engine& e, const PLstm& l, const View& x, const int32_t* lens) {
  View G = new_ncl(c, x.B, 8 * l.H, x.L, false);
  ConvOpt o;
  o.bias = e.F(l.bias);
  conv(c, e, x, l.w_ih, G, o);
  View y = new_ncl(c, x.B, 2 * l.H, x.L, false);
  const int64_t sb = st2_lstm_coop_scratch_bytes(x.B);
  void* scratch = sb > 0 ? c.a.alloc(sb) : nullptr;
  RUN(c, g_be.lstm_bidir(G.p, G.bs, G.cs, e.F(l.whh_t), lens, x.B, l.H, x.L, y.p, y.bs, y.cs, scratch, sb, c.stream));
  return y;
}

int duration_plan(Ctx& c, const st2_engine& e, const View& d_en, const float* s_p, const int32_t* lens, int B, int N,
                  int tail, float* d_cm, int64_t* durations) {
  const st2_model_config& cfg = e.cfg;
  const PDuration& d = e.dur;
  const int dh = cfg.pred_hidden, sty = cfg.style_dim, Cd = dh + sty;
  const int nl = (int)d.lstms.size();
  View h = new_ncl(c, B, Cd, N, false);
  {  // [x | style broadcast over the tokens], pad positions zeroed
    View src = d_en, dst = h.rows(0, dh), st = h.rows(dh, Cd);
    RUN(c, g_be.copy_ncl(src.p, src.bs, src.cs, dst.p, dst.bs, dst.cs, B, dh, N, c.stream));
    RUN(c, g_be.broadcast_cols(s_p, sty, st.p, st.bs, st.cs, B, sty, N, c.stream));
    if (lens) RUN(c, g_be.mask_tail(h.p, h.bs, h.cs, B, Cd, N, lens, c.stream));
  }
  for (int i = 0; i < nl; ++i) {
    View y = lstm_run(c, e, d.lstms[(size_t)i], h, lens);  // [B][dh][N]
    // AdaLayerNorm (models.py:418-438): (1 + gamma) * LayerNorm_c(y) + beta, gamma | beta = fc(style)
    float* gb = c.a.f32((int64_t)B * 2 * dh);
    RUN(c, g_be.style_fc(s_p, B, sty, e.F(d.ada_wt[(size_t)i]), e.F(d.ada_b[(size_t)i]), 2 * dh, ST2_ACT_NONE, gb, c.stream));
    float* st = c.a.f32((int64_t)B * N * 2);
    RUN(c, g_be.colnorm_stats(y.p, y.bs, y.cs, B, dh, N, 1e-5f, st, c.stream));
    const bool last = i + 1 == nl;
    View nh = last ? wrap(d_cm, B, Cd, N) : new_ncl(c, B, Cd, N, false);
    View top = nh.rows(0, dh), bot = nh.rows(dh, Cd);
    RUN(c, g_be.colnorm_apply(y.p, y.bs, y.cs, st, gb, gb + dh, 2 * dh, 1, ST2_ACT_NONE, 0.f, lens, top.p, top.bs, top.cs, B,
                              dh, N, c.stream));
    RUN(c, g_be.broadcast_cols(s_p, sty, bot.p, bot.bs, bot.cs, B, sty, N, c.stream));
    if (lens) RUN(c, g_be.mask_tail(bot.p, bot.bs, bot.cs, B, sty, N, lens, c.stream));
    h = nh;
  }
  if (durations) {
    View x = lstm_run(c, e, d.dur_lstm, h, lens);
    RUN(c, g_be.duration_head(x.p, x.bs, x.cs, e.F(d.proj_w), e.F(d.proj_b), B, dh, d.max_dur, N, lens, tail, durations,
                              nullptr, c.stream));
  }
  return c.rc;
}

// ------------------------------------------------------------------------------------------------------------------
// text-encoder plan == TextEncoder.forward (styletts2_amd/text.py)
// ------------------------------------------------------------------------------------------------------------------
int pack_text(st2_engine& e, Blob& blob, std::string* err) {
  Packer pk{e, blob};
  PText t;
  const std::string T = "text_encoder.";
  const HostTensor* emb = pk.get(T + "embedding.weight");
  if (!emb || emb->shape.size() != 2) { *err = "missing text-encoder parameter text_encoder.embedding.weight"; return 1; }
  t.V = (int)emb->shape[0];
  t.C = (int)emb->shape[1];
  t.emb = blob.add_f32(emb->data);
  for (int i = 0; pk.has(T + "cnn." + std::to_string(i) + ".0.weight"); ++i) {
    const std::string p = T + "cnn." + std::to_string(i);
    t.convs.push_back(pk.conv(p + ".0", true, t.C, t.C, -1));
    t.ln_g.push_back(pk.vec(p + ".1.gamma", t.C));
    t.ln_b.push_back(pk.vec(p + ".1.beta", t.C));
  }
  t.lstm = pack_lstm(pk, T + "lstm");
  if (!pk.ok || t.convs.empty()) { *err = "text-encoder parameter missing or malformed: " + pk.missing; return 1; }
  t.ready = true;
  e.text = t;
  return 0;
}

int text_plan(Ctx& c, const st2_engine& e, const int64_t* tokens, const int32_t* lens, int B, int N, float* t_en) {
  const PText& t = e.text;
  View h = new_ncl(c, B, t.C, N, false);
  RUN(c, g_be.embed_tokens(tokens, B, N, e.F(t.emb), t.V, t.C, nullptr, nullptr, lens, h.p, h.bs, h.cs, c.stream));
  for (size_t i = 0; i < t.convs.size(); ++i) {
    const PConv& pc = t.convs[i];
    View y = new_ncl(c, B, pc.c_out, N, false);
    ConvOpt o;
    o.pad_left = (pc.ks - 1) / 2; o.bias = e.F(pc.bias);
    conv(c, e, h, pc.w, y, o);
    // LayerNorm over channels + LeakyReLU(0.2) + masked_fill in one pass (models.py:270-282, 308-312)
    float* st = c.a.f32((int64_t)B * N * 2);
    RUN(c, g_be.colnorm_stats(y.p, y.bs, y.cs, B, pc.c_out, N, 1e-5f, st, c.stream));
    View z = new_ncl(c, B, pc.c_out, N, false);
    RUN(c, g_be.colnorm_apply(y.p, y.bs, y.cs, st, e.F(t.ln_g[i]), e.F(t.ln_b[i]), 0, 0, ST2_ACT_LEAKY, 0.2f, lens, z.p, z.bs,
                              z.cs, B, pc.c_out, N, c.stream));
    h = z;
  }
  View y = lstm_run(c, e, t.lstm, h, lens);  // outputs past a sequence's end are zero (packed-sequence semantics)
  View dst = wrap(t_en, B, 2 * t.lstm.H, N);
  RUN(c, g_be.copy_ncl(y.p, y.bs, y.cs, dst.p, dst.bs, dst.cs, B, 2 * t.lstm.H, N, c.stream));
  return c.rc;
}

// ------------------------------------------------------------------------------------------------------------------
// PL-BERT plan == CustomAlbert.forward_engine (styletts2_amd/text.py); front plan == pipeline._front_core
// ------------------------------------------------------------------------------------------------------------------
int pack_bert(st2_engine& e, Blob& blob, std::string* err) {
  Packer pk{e, blob};
  PBert b;
  const std::string R = "bert.", L = R + "encoder.albert_layer_groups.0.albert_layers.0.";
  const HostTensor* w = pk.get(R + "embeddings.word_embeddings.weight");
  const HostTensor* p = pk.get(R + "embeddings.position_embeddings.weight");
  const HostTensor* tt = pk.get(R + "embeddings.token_type_embeddings.weight");
  const HostTensor* q = pk.get(L + "attention.query.weight");
  const HostTensor* k = pk.get(L + "attention.key.weight");
  const HostTensor* v = pk.get(L + "attention.value.weight");
  const HostTensor* qb = pk.get(L + "attention.query.bias");
  const HostTensor* kb = pk.get(L + "attention.key.bias");
  const HostTensor* vb = pk.get(L + "attention.value.bias");
  if (!pk.ok || w->shape.size() != 2 || p->shape.size() != 2 || tt->shape.size() != 2 || q->shape.size() != 2) {
    *err = "PL-BERT parameter missing or malformed: " + pk.missing;
    return 1;
  }
  b.V = (int)w->shape[0]; b.E = (int)w->shape[1]; b.P = (int)p->shape[0]; b.H = (int)q->shape[0];
  if (p->shape[1] != b.E || tt->shape[1] != b.E || q->shape[1] != b.H || k->numel() != q->numel() ||
      v->numel() != q->numel() || qb->numel() != b.H || kb->numel() != b.H || vb->numel() != b.H || b.H % 64 != 0 ||
      e.cfg.bert_layers <= 0) {
    *err = "PL-BERT: inconsistent shapes (64-wide heads, one shared layer) or cfg.bert_layers == 0";
    return 1;
  }
  b.word = blob.add_f32(w->data);
  b.pos = blob.add_f32(p->data);
  b.tok0 = blob.add_f32(std::vector<float>(tt->data.begin(), tt->data.begin() + b.E));  // token_type_ids == 0 everywhere
  b.eln_w = pk.vec(R + "embeddings.LayerNorm.weight", b.E); b.eln_b = pk.vec(R + "embeddings.LayerNorm.bias", b.E);
  b.map = pk.conv_w(R + "encoder.embedding_hidden_mapping_in.weight", b.H, b.E, 1);
  b.map_b = pk.vec(R + "encoder.embedding_hidden_mapping_in.bias", b.H);
  {  // q | k | v as one 3H-row Linear
    const size_t HH = (size_t)b.H * b.H;
    std::vector<float> cat(3 * HH), bias((size_t)3 * b.H);
    std::copy(q->data.begin(), q->data.end(), cat.begin());
    std::copy(k->data.begin(), k->data.end(), cat.begin() + HH);
    std::copy(v->data.begin(), v->data.end(), cat.begin() + 2 * HH);
    std::copy(qb->data.begin(), qb->data.end(), bias.begin());
    std::copy(kb->data.begin(), kb->data.end(), bias.begin() + b.H);
    std::copy(vb->data.begin(), vb->data.end(), bias.begin() + 2 * b.H);
    b.qkv = pack_split(blob, L + "attention.query|key|value.weight", cat.data(), 3 * b.H, b.H, 1);
    b.qkv_b = blob.add_f32(bias);
  }
  b.dense = pk.conv_w(L + "attention.dense.weight", b.H, b.H, 1); b.dense_b = pk.vec(L + "attention.dense.bias", b.H);
  b.aln_w = pk.vec(L + "attention.LayerNorm.weight", b.H);        b.aln_b = pk.vec(L + "attention.LayerNorm.bias", b.H);
  b.ffn = pk.conv_w(L + "ffn.weight", -1, b.H, 1);
  b.I = b.ffn.C_out;
  b.ffn_b = pk.vec(L + "ffn.bias", b.I > 0 ? b.I : -1);
  b.out = pk.conv_w(L + "ffn_output.weight", b.H, b.I > 0 ? b.I : -1, 1); b.out_b = pk.vec(L + "ffn_output.bias", b.H);
  b.fln_w = pk.vec(L + "full_layer_layer_norm.weight", b.H);      b.fln_b = pk.vec(L + "full_layer_layer_norm.bias", b.H);
  if (pk.has("bert_encoder.weight")) {
    b.enc = pk.conv_w("bert_encoder.weight", -1, b.H, 1);
    b.enc_b = pk.vec("bert_encoder.bias", b.enc.C_out > 0 ? b.enc.C_out : -1);
    b.has_enc = true;
  }
  if (!pk.ok) { *err = "PL-BERT parameter missing or malformed: " + pk.missing; return 1; }
  b.ready = true;
  e.bert = b;
  return 0;
}

// -> the last hidden state as a [B][H][N] view of token-merged storage ([H][B*N]); lives in the arena
View bert_plan(Ctx& c, const st2_engine& e, const int64_t* tokens, const int32_t* lens, int B, int N) {
  const PBert& p = e.bert;
  const float eps = e.cfg.bert_ln_eps;
  const int H = p.H, heads = H / 64;
  Sess s{c, e, B, N, true, lens, {}, 0, nullptr, nullptr, 1.0};
  View X = dn_alloc(s, H), Xn = dn_alloc(s, H);
  {
    const int64_t mark = c.a.off;
    View E = dn_alloc(s, p.E);
    RUN(c, g_be.embed_tokens(tokens, B, N, e.F(p.word), p.V, p.E, e.F(p.tok0), e.F(p.pos), nullptr, E.p, E.bs, E.cs,
                             c.stream));
    float* st = c.a.f32((int64_t)B * N * 2);
    RUN(c, g_be.colnorm_stats(E.p, E.bs, E.cs, B, p.E, N, eps, st, c.stream));
    ConvOpt o;  // embedding LayerNorm in the prologue of the E -> H mapping
    o.bias = e.F(p.map_b); o.pro = ST2_PRO_COLNORM; o.stats = st; o.gamma = e.F(p.eln_w); o.beta = e.F(p.eln_b);
    conv(c, e, dn_cv(s, E), p.map, dn_cv(s, X), o);
    c.a.off = mark;
  }
  for (int l = 0; l < e.cfg.bert_layers; ++l) {
    const int64_t mark = c.a.off;
    View qkv = dn_alloc(s, 3 * H);
    {
      ConvOpt o;
      o.bias = e.F(p.qkv_b);
      conv(c, e, dn_cv(s, X), p.qkv, dn_cv(s, qkv), o);
    }
    View ctx = dn_alloc(s, H);
    View q = qkv.rows(0, H), k = qkv.rows(H, 2 * H), v = qkv.rows(2 * H, 3 * H);
    RUN(c, g_be.attention_keylen(q.p, k.p, v.p, q.bs, q.cs, ctx.p, ctx.bs, ctx.cs, B, heads, 64, N, 0.125f, lens, c.stream));
    View Y = dn_alloc(s, H);
    {
      ConvOpt o;
      o.bias = e.F(p.dense_b); o.res = dn_cv(s, X);
      conv(c, e, dn_cv(s, ctx), p.dense, dn_cv(s, Y), o);
    }
    float* st1 = c.a.f32((int64_t)B * N * 2);
    RUN(c, g_be.colnorm_stats(Y.p, Y.bs, Y.cs, B, H, N, eps, st1, c.stream));
    View X1 = dn_alloc(s, H);
    RUN(c, g_be.colnorm_apply(Y.p, Y.bs, Y.cs, st1, e.F(p.aln_w), e.F(p.aln_b), 0, 0, ST2_ACT_NONE, 0.f, nullptr, X1.p, X1.bs,
                              X1.cs, B, H, N, c.stream));
    View Hm = dn_alloc(s, p.I);
    {
      ConvOpt o;
      o.bias = e.F(p.ffn_b); o.act = ST2_ACT_GELU_TANH;
      conv(c, e, dn_cv(s, X1), p.ffn, dn_cv(s, Hm), o);
    }
    View Z = dn_alloc(s, H);
    {
      ConvOpt o;
      o.bias = e.F(p.out_b); o.res = dn_cv(s, X1);
      conv(c, e, dn_cv(s, Hm), p.out, dn_cv(s, Z), o);
    }
    float* st2 = c.a.f32((int64_t)B * N * 2);
    RUN(c, g_be.colnorm_stats(Z.p, Z.bs, Z.cs, B, H, N, eps, st2, c.stream));
    RUN(c, g_be.colnorm_apply(Z.p, Z.bs, Z.cs, st2, e.F(p.fln_w), e.F(p.fln_b), 0, 0, ST2_ACT_NONE, 0.f, nullptr, Xn.p, Xn.bs,
                              Xn.cs, B, H, N, c.stream));
    std::swap(X, Xn);
    c.a.off = mark;
  }
  return X;
}

int front_plan(Ctx& c, const st2_engine& e, const st2_front_args& a) {
  const st2_model_config& cfg = e.cfg;
  const int B = a.B, N = a.N, sty = cfg.style_dim, C2 = cfg.dn_channels, dh = cfg.pred_hidden;
  text_plan(c, e, a.tokens, a.lengths, B, N, a.t_en);
  View X = bert_plan(c, e, a.tokens, a.lengths, B, N);
  Sess s0{c, e, B, N, true, a.lengths, {}, 0, nullptr, nullptr, 1.0};
  View D = dn_alloc(s0, dh);  // bert_encoder: Linear(H -> hidden_dim) over the merged tokens == d_en channel-major
  {
    ConvOpt o;
    o.bias = e.F(e.bert.enc_b);
    conv(c, e, dn_cv(s0, X), e.bert.enc, dn_cv(s0, D), o);
  }
  const int64_t n = (int64_t)B * C2;
  float* sp = c.a.f32(n);
  {
    const int64_t mark = c.a.off;
    if (sampler_plan(c, e, a.noise, nullptr, &X, a.ref_s, a.step_noise, a.lengths, B, N, a.steps, a.embedding_scale, a.table,
                     a.sigma0, sp, nullptr) != 0 && c.rc == 0)
      c.rc = 1;
    c.a.off = mark;
  }
  if (a.carry && B > 1) {
    // the rows are consecutive sentences of one passage: row k mixes with row k-1's MIXED style (the loop of LFinference,
    // Demo/Inference_LibriTTS.ipynb LFinference: `s_prev = s_pred` after the speaker mixing).  A scan of the same elementwise
    // launches the one-sentence call makes, on one row each: bitwise the sentence-by-sentence results, ~5 us per launch.
    float* mixed = a.s_pred_out ? a.s_pred_out : c.a.f32(n);
    float* m = c.a.f32(C2);
    float* ma = c.a.f32(C2);
    float* mb = c.a.f32(C2);
    for (int k = 0; k < B; ++k) {
      const float* prev = k ? mixed + (int64_t)(k - 1) * C2 : a.s_prev;
      const float* cur = sp + (int64_t)k * C2;
      if (prev) {
        RUN(c, g_be.axpbypcz(prev, (float)a.t, cur, (float)(1.0 - a.t), nullptr, 0.f, m, C2, c.stream));
        cur = m;
      }
      const float* ref_src = cur;
      const float* s_src = cur + sty;
      if (a.ref_s) {
        const float* rs = a.ref_s + (int64_t)k * C2;
        RUN(c, g_be.axpbypcz(cur, (float)a.alpha, rs, (float)(1.0 - a.alpha), nullptr, 0.f, ma, C2, c.stream));
        RUN(c, g_be.axpbypcz(cur, (float)a.beta, rs, (float)(1.0 - a.beta), nullptr, 0.f, mb, C2, c.stream));
        ref_src = ma;
        s_src = mb + sty;
      }
      RUN(c, g_be.copy_ncl(ref_src, C2, sty, mixed + (int64_t)k * C2, C2, sty, 1, 1, sty, c.stream));
      RUN(c, g_be.copy_ncl(s_src, C2, sty, mixed + (int64_t)k * C2 + sty, C2, sty, 1, 1, sty, c.stream));
    }
    RUN(c, g_be.copy_ncl(mixed, C2, sty, a.ref, sty, sty, B, 1, sty, c.stream));
    RUN(c, g_be.copy_ncl(mixed + sty, C2, sty, a.s, sty, sty, B, 1, sty, c.stream));
    duration_plan(c, e, D, a.s, a.lengths, B, N, a.tail, a.d_cm, a.durations);
    return c.rc;
  }
  const float* cur = sp;
  if (a.s_prev) {  // LFinference: convex combination of the previous and the current style
    float* m = c.a.f32(n);
    RUN(c, g_be.axpbypcz(a.s_prev, (float)a.t, cur, (float)(1.0 - a.t), nullptr, 0.f, m, n, c.stream));
    cur = m;
  }
  const float* ref_src = cur;
  const float* s_src = cur + sty;
  if (a.ref_s) {  // Demo/Inference_LibriTTS.ipynb:289-290
    float* ma = c.a.f32(n);
    float* mb = c.a.f32(n);
    RUN(c, g_be.axpbypcz(cur, (float)a.alpha, a.ref_s, (float)(1.0 - a.alpha), nullptr, 0.f, ma, n, c.stream));
    RUN(c, g_be.axpbypcz(cur, (float)a.beta, a.ref_s, (float)(1.0 - a.beta), nullptr, 0.f, mb, n, c.stream));
    ref_src = ma;
    s_src = mb + sty;
  }
  RUN(c, g_be.copy_ncl(ref_src, C2, sty, a.ref, sty, sty, B, 1, sty, c.stream));
  RUN(c, g_be.copy_ncl(s_src, C2, sty, a.s, sty, sty, B, 1, sty, c.stream));
  if (a.s_pred_out) {
    RUN(c, g_be.copy_ncl(ref_src, C2, sty, a.s_pred_out, C2, sty, B, 1, sty, c.stream));
    RUN(c, g_be.copy_ncl(s_src, C2, sty, a.s_pred_out + sty, C2, sty, B, 1, sty, c.stream));
  }
  duration_plan(c, e, D, a.s, a.lengths, B, N, a.tail, a.d_cm, a.durations);
  return c.rc;
}

// ------------------------------------------------------------------------------------------------------------------
// style-encoder plan == StyleEncoder.forward (styletts2_amd/style.py): feature maps stored (h, c, w) with one zero row
// above and below, every 3x3 Conv2d one split-f16 Conv1d over the width with 3 stacked rows as its input channels
// ------------------------------------------------------------------------------------------------------------------
constexpr int STYLE_ZEROS = 4096;

// Conv2d weight [Co][Ci][kh][kw] -> Conv1d weight [Co][kh*Ci][kw] on kh stacked image rows (channel index dh*Ci + ci)
std::vector<float> rows_as_channels(const HostTensor& t) {
  const int co = (int)t.shape[0], ci = (int)t.shape[1], kh = (int)t.shape[2], kw = (int)t.shape[3];
  std::vector<float> v((size_t)co * kh * ci * kw);
  for (int o = 0; o < co; ++o)
    for (int i = 0; i < ci; ++i)
      for (int dh = 0; dh < kh; ++dh)
        for (int x = 0; x < kw; ++x)
          v[(((size_t)o * kh + dh) * ci + i) * kw + x] = t.data[(((size_t)o * ci + i) * kh + dh) * kw + x];
  return v;
}

int pack_style(st2_engine& e, Blob& blob, int which, std::string* err) {
  Packer pk{e, blob};
  PStyleEnc s;
  const std::string R = which == 0 ? "style_encoder." : "predictor_encoder.";
  auto conv2d = [&](const std::string& name) -> SplitW {  // folded Conv2d -> packed Conv1d over stacked rows
    const HostTensor* t = pk.get(name);
    if (!t || t->shape.size() != 4) { pk.ok = false; if (pk.missing.empty()) pk.missing = name + " (4-D expected)"; return SplitW(); }
    const std::vector<float> v = rows_as_channels(*t);
    return pack_split(blob, name, v.data(), (int)t->shape[0], (int)(t->shape[1] * t->shape[2]), (int)t->shape[3]);
  };
  const HostTensor* first = pk.get(R + "shared.0.weight");
  if (!first || first->shape.size() != 4 || first->shape[1] != 1 || first->shape[2] != 3 || first->shape[3] != 3) {
    *err = "missing or malformed style-encoder parameter " + R + "shared.0.weight (spectral norm folded by the caller)";
    return 1;
  }
  s.c0 = (int)first->shape[0];
  s.w0 = blob.add_f32(rows_as_channels(*first));  // [C0][3][3]: plain OIK for st2_conv1d_direct
  s.b0 = pk.vec(R + "shared.0.bias");
  int i = 1;
  for (; pk.has(R + "shared." + std::to_string(i) + ".conv1.weight"); ++i) {
    const std::string Bn = R + "shared." + std::to_string(i);
    PStyleBlk b;
    const HostTensor* w1 = pk.get(Bn + ".conv1.weight");
    const HostTensor* w2 = pk.get(Bn + ".conv2.weight");
    const HostTensor* wd = pk.get(Bn + ".downsample_res.conv.weight");
    if (!w1 || !w2 || !wd) break;
    if (w1->shape.size() != 4 || w2->shape.size() != 4 || w1->shape[0] != w1->shape[1] || w2->shape[1] != w1->shape[0] ||
        wd->numel() != w1->shape[0] * 9) {
      *err = "malformed style-encoder block " + Bn + " (ResBlk: conv1 [C, C, 3, 3], conv2 [C', C, 3, 3], depthwise [C, 1, 3, 3])";
      return 1;
    }
    b.c_in = (int)w1->shape[1];
    b.c_out = (int)w2->shape[0];
    b.w1 = conv2d(Bn + ".conv1.weight");  b.b1 = pk.vec(Bn + ".conv1.bias");
    b.w2 = conv2d(Bn + ".conv2.weight");  b.b2 = pk.vec(Bn + ".conv2.bias");
    b.wd = blob.add_f32(wd->data);        b.bd = pk.vec(Bn + ".downsample_res.conv.bias");  // [C][1][3][3] == [C][3][3]
    if (pk.has(Bn + ".conv1x1.weight")) {
      b.wsc = conv2d(Bn + ".conv1x1.weight");
      b.has_sc = true;
    }
    s.blocks.push_back(b);
  }
  // shared.{i} = LeakyReLU, shared.{i+1} = the 5x5 valid conv (models.py:151-153)
  const std::string last = R + "shared." + std::to_string(i + 1);
  s.w5 = conv2d(last + ".weight");
  s.b5 = pk.vec(last + ".bias");
  s.c_last = s.w5.C_out;
  s.wl = pk.conv_w(R + "unshared.weight");
  s.bl = pk.vec(R + "unshared.bias");
  s.style_dim = s.wl.C_out;
  if (!pk.ok || s.c0 > STYLE_ZEROS || s.c_last > STYLE_ZEROS) {
    *err = "style-encoder parameter missing or malformed: " + pk.missing;
    return 1;
  }
  if (s.blocks.size() != 4 || s.w5.ks != 5 || s.w5.C_in != 5 * s.blocks.back().c_out) {  // 80 mel bins -> 5 rows -> 5x5 valid conv
    *err = "style encoder " + R + ": expected four down-sampling ResBlks and a 5x5 valid conv (models.py:139-164)";
    return 1;
  }
  s.ready = true;
  e.style[which] = s;
  return 0;
}

int style_plan(Ctx& c, const st2_engine& e, const PStyleEnc& s, const float* mel, int B, int H, int W, float* out) {
  // [B][h + 2][ch][w] map whose rows 0 and h + 1 are zero (the rows 1 .. h are written by the producing kernel)
  auto new_map = [&](int h, int ch, int w) -> float* {
    float* p = c.a.f32((int64_t)B * (h + 2) * ch * w);
    for (int r : {0, h + 1})
      RUN(c, g_be.broadcast_cols(e.F(e.zeros), 0, p + (int64_t)r * ch * w, (int64_t)(h + 2) * ch * w, w, B, ch, w, c.stream));
    return p;
  };
  // rows r0 .. r0 + k - 1 of utterance b's padded map stacked along the channels: [h][k * ch][w] (overlapping view)
  auto rows = [&](float* P, int b, int h, int ch, int w, int k, int r0) {
    View v;
    v.p = P + (int64_t)b * (h + 2) * ch * w + (int64_t)r0 * ch * w;
    v.B = h; v.C = k * ch; v.L = w; v.bs = (int64_t)ch * w; v.cs = w;  // k = 3 from row 0 / k = 1 from row 1: h image rows
    return v;
  };
  auto plain = [&](float* p, int b, int h, int ch, int w) {  // [h][ch][w] of utterance b in an unpadded [B][h][ch][w] buffer
    View v;
    v.p = p + (int64_t)b * h * ch * w;
    v.B = h; v.C = ch; v.L = w; v.bs = (int64_t)ch * w; v.cs = w;
    return v;
  };
  float* m0 = new_map(H, 1, W);
  RUN(c, g_be.copy_ncl(mel, (int64_t)H * W, W, m0 + W, (int64_t)(H + 2) * W, W, B, H, W, c.stream));
  int C = s.c0;
  float* P = new_map(H, C, W);
  for (int b = 0; b < B; ++b) {
    const View x = rows(m0, b, H, 1, W, 3, 0);
    RUN(c, g_be.conv1d_direct(x.p, x.bs, x.cs, e.F(s.w0), e.F(s.b0), P + (int64_t)b * (H + 2) * C * W + (int64_t)C * W,
                              (int64_t)C * W, W, H, 3, C, W, W, 3, 1, 1, c.stream));
  }
  for (const PStyleBlk& blk : s.blocks) {
    const int Co = blk.c_out, Ho = H / 2, Wo = (W + 1) / 2;
    // shortcut: 1x1 conv at full resolution, then the 2x2 average (models.py:118-123)
    float* SC = c.a.f32((int64_t)B * Ho * Co * Wo);
    if (blk.has_sc) {
      float* S = c.a.f32((int64_t)B * H * Co * W);
      for (int b = 0; b < B; ++b) conv(c, e, rows(P, b, H, C, W, 1, 1), blk.wsc, plain(S, b, H, Co, W), ConvOpt());
      RUN(c, g_be.avgpool2x2(S, (int64_t)H * Co * W, (int64_t)Co * W, W, B, Co, H, W, SC, (int64_t)Ho * Co * Wo, (int64_t)Co * Wo,
                             Wo, c.stream));
    } else {
      RUN(c, g_be.avgpool2x2(P + (int64_t)C * W, (int64_t)(H + 2) * C * W, (int64_t)C * W, W, B, C, H, W, SC,
                             (int64_t)Ho * Co * Wo, (int64_t)Co * Wo, Wo, c.stream));
    }
    // residual: leaky -> conv1 3x3 -> depthwise stride-2 3x3 -> leaky -> conv2 3x3 (models.py:125-135)
    float* R1 = c.a.f32((int64_t)B * H * C * W);
    for (int b = 0; b < B; ++b) {
      ConvOpt o;
      o.pad_left = 1; o.bias = e.F(blk.b1); o.pro = ST2_PRO_LEAKY; o.slope = 0.2f;
      conv(c, e, rows(P, b, H, C, W, 3, 0), blk.w1, plain(R1, b, H, C, W), o);
    }
    float* P2 = new_map(Ho, C, Wo);
    RUN(c, g_be.dwconv3x3s2(R1, (int64_t)H * C * W, (int64_t)C * W, W, e.F(blk.wd), e.F(blk.bd), B, C, H, W, P2 + (int64_t)C * Wo,
                            (int64_t)(Ho + 2) * C * Wo, (int64_t)C * Wo, Wo, c.stream));
    float* Pn = new_map(Ho, Co, Wo);
    for (int b = 0; b < B; ++b) {  // (shortcut + residual) / sqrt(2) in the epilogue
      ConvOpt o;
      o.pad_left = 1; o.bias = e.F(blk.b2); o.pro = ST2_PRO_LEAKY; o.slope = 0.2f;
      o.res = plain(SC, b, Ho, Co, Wo); o.div = (float)sqrt(2.0);
      View y = rows(Pn, b, Ho, Co, Wo, 1, 1);
      conv(c, e, rows(P2, b, Ho, C, Wo, 3, 0), blk.w2, y, o);
    }
    P = Pn; H = Ho; W = Wo; C = Co;
  }
  // LeakyReLU -> 5x5 valid conv -> global average -> LeakyReLU -> Linear (models.py:151-163)
  const int Cl = s.c_last, Wf = W - 4;
  float* Fm = c.a.f32((int64_t)B * Cl * Wf);
  for (int b = 0; b < B; ++b) {
    View x;
    x.p = P + (int64_t)b * (H + 2) * C * W + (int64_t)C * W;
    x.B = 1; x.C = 5 * C; x.L = W; x.bs = (int64_t)5 * C * W; x.cs = W;
    View y;
    y.p = Fm + (int64_t)b * Cl * Wf;
    y.B = 1; y.C = Cl; y.L = Wf; y.bs = (int64_t)Cl * Wf; y.cs = Wf;
    ConvOpt o;
    o.bias = e.F(s.b5); o.pro = ST2_PRO_LEAKY; o.slope = 0.2f;
    conv(c, e, x, s.w5, y, o);
  }
  float* m = c.a.f32((int64_t)B * Cl);
  RUN(c, g_be.mean_tokens_len(Fm, (int64_t)Cl * Wf, Wf, m, Cl, B, Cl, Wf, nullptr, c.stream));
  {
    View x, y;
    x.p = m; x.B = B; x.C = Cl; x.L = 1; x.bs = Cl; x.cs = 1;
    y.p = out; y.B = B; y.C = s.style_dim; y.L = 1; y.bs = s.style_dim; y.cs = 1;
    ConvOpt o;
    o.bias = e.F(s.bl); o.pro = ST2_PRO_LEAKY; o.slope = 0.2f;
    conv(c, e, x, s.wl, y, o);
  }
  return c.rc;
}

bool check_cfg(const st2_model_config& c) {
  return c.n_upsamples >= 1 && c.n_upsamples <= 4 && c.n_resblock_kernels >= 1 && c.n_resblock_kernels <= 4 &&
         (c.decoder_kind == 0 || c.decoder_kind == 1) && c.dim_in > 0 && c.style_dim > 0 && c.dn_layers >= 0;
}

}  // namespace

// ------------------------------------------------------------------------------------------------------------------
// C entry points
// ------------------------------------------------------------------------------------------------------------------
extern "C" int st2_debug_set_backend(void* const* table, int32_t entries) {
  if (!table) {
    g_be = kHipBackend;
    return 0;
  }
  ST2_REQUIRE(entries == ST2_BACKEND_ENTRIES, "st2_debug_set_backend: %d entries, expected %d", entries,
              (int)ST2_BACKEND_ENTRIES);
  for (int i = 0; i < entries; ++i) ST2_REQUIRE(table[i] != nullptr, "st2_debug_set_backend: entry %d is null", i);
#define SLOT(field, slot) g_be.field = reinterpret_cast<decltype(g_be.field)>(table[slot])
  SLOT(conv1d_f16s, ST2_BE_CONV1D_F16S); SLOT(conv1d_xs, ST2_BE_CONV1D_XS); SLOT(act_split, ST2_BE_ACT_SPLIT);
  SLOT(stats_finalize, ST2_BE_STATS_FINALIZE); SLOT(conv1d_direct, ST2_BE_CONV1D_DIRECT);
  SLOT(phase_split, ST2_BE_PHASE_SPLIT); SLOT(instnorm_stats, ST2_BE_INSTNORM_STATS);
  SLOT(colnorm_stats, ST2_BE_COLNORM_STATS); SLOT(style_fc, ST2_BE_STYLE_FC);
  SLOT(convt_interleave_stats, ST2_BE_CONVT_INTERLEAVE_STATS); SLOT(adain_leaky_pool, ST2_BE_ADAIN_LEAKY_POOL);
  SLOT(har_source, ST2_BE_HAR_SOURCE); SLOT(stft_mag_phase, ST2_BE_STFT_MAG_PHASE); SLOT(istft, ST2_BE_ISTFT);
  SLOT(attention_keylen, ST2_BE_ATTENTION_KEYLEN); SLOT(add_chanvec, ST2_BE_ADD_CHANVEC);
  SLOT(mean_tokens_len, ST2_BE_MEAN_TOKENS_LEN); SLOT(axpbypcz, ST2_BE_AXPBYPCZ);
  SLOT(time_features, ST2_BE_TIME_FEATURES); SLOT(tokens_to_channels, ST2_BE_TOKENS_TO_CHANNELS);
  SLOT(broadcast_cols, ST2_BE_BROADCAST_COLS); SLOT(copy_ncl, ST2_BE_COPY_NCL);
  SLOT(expand_by_durations, ST2_BE_EXPAND_BY_DURATIONS); SLOT(lstm_bidir, ST2_BE_LSTM_BIDIR);
  SLOT(colnorm_apply, ST2_BE_COLNORM_APPLY); SLOT(duration_head, ST2_BE_DURATION_HEAD); SLOT(mask_tail, ST2_BE_MASK_TAIL);
  SLOT(embed_tokens, ST2_BE_EMBED_TOKENS); SLOT(dwconv3x3s2, ST2_BE_DWCONV3X3S2); SLOT(avgpool2x2, ST2_BE_AVGPOOL2X2);
  SLOT(dev_alloc, ST2_BE_DEV_ALLOC); SLOT(dev_free, ST2_BE_DEV_FREE); SLOT(upload, ST2_BE_UPLOAD);
#undef SLOT
  return 0;
}

extern "C" int st2_create(const st2_model_config* cfg, st2_engine** out) {
  ST2_REQUIRE(cfg && out, "st2_create: null argument");
  ST2_REQUIRE(check_cfg(*cfg), "st2_create: invalid model configuration");
  st2_engine* e = new (std::nothrow) st2_engine();
  ST2_REQUIRE(e, "st2_create: out of memory");
  e->cfg = *cfg;
  *out = e;
  return 0;
}

extern "C" int st2_destroy(st2_engine* e) {
  if (!e) return 0;
  if (e->wbase) g_be.dev_free(e->wbase);
  delete e;
  return 0;
}

extern "C" int st2_load_weights(st2_engine* e, const char* name, const float* data, const int64_t* shape, int32_t ndim) {
  ST2_REQUIRE(e && name && data && shape && ndim >= 0 && ndim <= 4, "st2_load_weights: bad arguments");
  HostTensor t;
  int64_t n = 1;
  for (int i = 0; i < ndim; ++i) {
    ST2_REQUIRE(shape[i] > 0, "st2_load_weights: %s has a non-positive dimension", name);
    t.shape.push_back(shape[i]);
    n *= shape[i];
  }
  t.data.assign(data, data + n);
  e->host[std::string(name)] = std::move(t);
  return 0;
}

// Everything st2_finalize_weights produces: committed to the engine only after the new blob is on the device.
struct PackedSet {
  PDecoder dec;
  PDenoiser dn;
  PPredictor pred;
  PDuration dur;
  PText text;
  PBert bert;
  PStyleEnc style[2];
  int64_t zeros = -1;
  std::vector<ConvSite> sites;
  std::vector<float> site_scale, site_seen;
};

namespace {
PackedSet packed_of(const st2_engine& e) {
  PackedSet p;
  p.sites = e.sites; p.site_scale = e.site_scale; p.site_seen = e.site_seen;
  p.dec = e.dec; p.dn = e.dn; p.pred = e.pred; p.dur = e.dur; p.text = e.text; p.bert = e.bert;
  p.style[0] = e.style[0]; p.style[1] = e.style[1]; p.zeros = e.zeros;
  return p;
}
void commit_packed(st2_engine& e, const PackedSet& p) {
  e.sites = p.sites; e.site_scale = p.site_scale; e.site_seen = p.site_seen;
  e.dec = p.dec; e.dn = p.dn; e.pred = p.pred; e.dur = p.dur; e.text = p.text; e.bert = p.bert;
  e.style[0] = p.style[0]; e.style[1] = p.style[1]; e.zeros = p.zeros;
}
}  // namespace

// Transactional: the pack_* functions write offsets into the NEW host blob, so nothing they produce may become visible
// before that blob is on the device.  The engine's previous state (structs, ready flags, device blob) is saved first and
// restored on ANY failure -- a failed call leaves the engine exactly as it was, still usable with its old weights; the old
// device blob is freed only after the new allocation and upload have succeeded.
extern "C" int st2_finalize_weights(st2_engine* e, int32_t which) {
  ST2_REQUIRE(e && (which & 63) != 0, "st2_finalize_weights: bad arguments");
  const PackedSet saved = packed_of(*e);
  Blob blob;
  std::string err;
  auto pack_all = [&]() -> int {
    if (which & 1) { if (pack_decoder(*e, blob, &err)) return 1; }
    else e->dec.ready = false;
    if (which & 2) { if (pack_denoiser(*e, blob, &err)) return 1; }
    else e->dn.ready = false;
    if (which & 4) { if (pack_predictor(*e, blob, &err)) return 1; }
    else e->pred.ready = false;
    e->dur.ready = false;
    if ((which & 4) && e->host.count("predictor.text_encoder.lstms.0.weight_ih_l0"))
      if (pack_duration(*e, blob, &err)) return 1;
    if (which & 8) { if (pack_text(*e, blob, &err)) return 1; }
    else e->text.ready = false;
    if (which & 16) { if (pack_bert(*e, blob, &err)) return 1; }
    else e->bert.ready = false;
    e->style[0].ready = e->style[1].ready = false;
    if (which & 32) {
      int found = 0;
      for (int k = 0; k < 2; ++k)
        if (e->host.count(std::string(k == 0 ? "style_encoder." : "predictor_encoder.") + "shared.0.weight")) {
          if (pack_style(*e, blob, k, &err)) return 1;
          ++found;
        }
      if (!found) { err = "no style_encoder.* / predictor_encoder.* parameters were loaded"; return 1; }
      e->zeros = blob.add_f32(std::vector<float>((size_t)STYLE_ZEROS, 0.0f));
    }
    return 0;
  };
  if (pack_all() != 0) {
    commit_packed(*e, saved);  // roll back: old structs and flags, old device blob untouched
    st2_set_error("st2_finalize_weights: %s", err.c_str());
    return 1;
  }
  const int64_t bytes = (int64_t)blob.host.size();
  void* p = g_be.dev_alloc(bytes);
  if (!p) {
    commit_packed(*e, saved);
    st2_set_error("st2_finalize_weights: device allocation of %lld B failed", (long long)bytes);
    return 1;
  }
  if (g_be.upload(p, blob.host.data(), bytes) != 0) {
    g_be.dev_free(p);
    commit_packed(*e, saved);
    st2_set_error("st2_finalize_weights: upload failed");
    return 1;
  }
  if (e->wbase) g_be.dev_free(e->wbase);  // only now: the new blob is complete on the device
  e->wbase = static_cast<char*>(p);
  e->wbytes = bytes;
  // New weights: new sites, and NO operand scales.  A table belongs to the weights it was measured on: another checkpoint of the same
  // architecture has other activation magnitudes at every site, and scales that are too large clamp (ST2_STATUS_F16_RANGE) while
  // scales that are too small push the lo halves into f16 subnormals without any signal (advisor, round 5).  The caller
  // re-calibrates (st2_calibrate) or installs the table saved beside THAT checkpoint (st2_calibration_write).
  e->sites = blob.sites;
  e->site_scale.assign(e->sites.size(), 0.f);
  e->site_seen.assign(e->sites.size(), 0.f);
  return 0;
}

// ---- per-layer operand scales (st2.h: st2_calibrate) ---------------------------------------------------------------------
extern "C" float st2_calibration_scale(float max_abs, int32_t margin_bits) {
  if (!(max_abs > 0.f) || !std::isfinite(max_abs)) return 0.f;
  margin_bits = std::min(std::max(margin_bits, 0), 10);
  int ex = 0;
  (void)frexpf(max_abs, &ex);  // max_abs = f * 2^ex, f in [0.5, 1): max_abs * 2^(16 - margin - ex) lies in [2^(15-m), 2^(16-m))
  return ldexpf(1.0f, std::min(std::max(16 - margin_bits - ex, -60), 60));
}

extern "C" int st2_calibrate(st2_engine* e, int32_t margin_bits, int32_t* n_clamped) {
  if (!e) { st2_set_error("st2_calibrate: null engine"); return -1; }
  const int n = st2_debug_headroom_read(nullptr, 0);
  if (n < 0) return -1;
  std::vector<double> rows((size_t)std::max(n, 1) * ST2_HEADROOM_COLS);
  if (n > 0 && st2_debug_headroom_read(rows.data(), n) < 0) return -1;
  const double me = (double)(uintptr_t)e;
  std::vector<float> seen(e->sites.size(), 0.f);
  std::vector<char> free_range(e->sites.size(), 0);  // the site's operand does not come out of a normalising prologue
  int clamped = 0;
  for (int i = 0; i < n; ++i) {
    const double* r = rows.data() + (size_t)i * ST2_HEADROOM_COLS;
    const int site = (int)r[10];
    if (r[11] != me || site < 0 || site >= (int)seen.size() || !(r[5] > 0.0)) continue;
    if (r[6] >= 65504.0) ++clamped;  // the operand hit the clamp at the scale it ran with: this pass only bounds it from below
    seen[(size_t)site] = std::max(seen[(size_t)site], (float)(r[6] / r[5]));  // max |pro(x)| over every launch of the site
    if (x_scale_for((int)r[1]) == 1.0f) free_range[(size_t)site] = 1;
  }
  e->site_scale.resize(e->sites.size(), 0.f);
  e->site_seen.resize(e->sites.size(), 0.f);
  int set = 0;
  for (size_t i = 0; i < seen.size(); ++i) {
    if (!(seen[i] > 0.f)) continue;  // never launched in the recorded calls (or all-zero operand): keeps what it had
    // The maximum ACCUMULATES over calls (finalize / st2_calibration_write(n = 0) reset it): calibrating again on more utterances
    // can only widen a site's range, never forget what an earlier pass saw.
    e->site_seen[i] = std::max(e->site_seen[i], seen[i]);
    // Headroom above the largest operand seen: 2^margin_bits after a normalising prologue (AdaIN / LayerNorm outputs are bounded by
    // the affine's gain whatever the utterance), two bits more where the operand is free-ranging -- the F0 curve in Hz, generator
    // stage outputs, FFN intermediates differ between utterances far more than between passes of one (advisor, round 5: 8 x over
    // ONE synthetic step is thin for those; the lo half stays a normal f16 down to 2^-14 / x_scale, two bits cost nothing there).
    e->site_scale[i] = st2_calibration_scale(e->site_seen[i], margin_bits + (free_range[i] ? 2 : 0));
    ++set;
  }
  if (n_clamped) *n_clamped = clamped;
  return set;
}

extern "C" int st2_calibration_read(st2_engine* e, double* rows, int32_t cap_rows) {
  if (!e) return -1;
  const int n = (int)e->sites.size();
  for (int i = 0; rows && i < n && i < cap_rows; ++i) {
    double* r = rows + (size_t)i * ST2_CALIBRATION_COLS;
    r[0] = e->sites[(size_t)i].C_in; r[1] = e->sites[(size_t)i].C_out; r[2] = e->sites[(size_t)i].ks;
    r[3] = i < (int)e->site_scale.size() ? e->site_scale[(size_t)i] : 0.0;
    r[4] = i < (int)e->site_seen.size() ? e->site_seen[(size_t)i] : 0.0;
  }
  return n;
}

extern "C" int st2_calibration_site_name(st2_engine* e, int32_t site, char* name, int32_t cap) {
  ST2_REQUIRE(e && name && cap > 0 && site >= 0 && site < (int)e->sites.size(), "st2_calibration_site_name: bad arguments");
  snprintf(name, (size_t)cap, "%s", e->sites[(size_t)site].name.c_str());
  return 0;
}

extern "C" int st2_calibration_write(st2_engine* e, const float* scales, int32_t n) {
  ST2_REQUIRE(e && (n == 0 || scales) && n >= 0, "st2_calibration_write: bad arguments");
  if (n == 0) {  // back to the rule
    e->site_scale.assign(e->sites.size(), 0.f);
    e->site_seen.assign(e->sites.size(), 0.f);
    return 0;
  }
  ST2_REQUIRE(n == (int)e->sites.size(), "st2_calibration_write: %d scales for %d conv sites (another model layout?)", n,
              (int)e->sites.size());
  for (int i = 0; i < n; ++i) {
    int ex = 0;
    const bool pow2 = scales[i] == 0.f || (scales[i] > 0.f && std::isfinite(scales[i]) && frexpf(scales[i], &ex) == 0.5f);
    ST2_REQUIRE(pow2, "st2_calibration_write: scale %d = %g is not a power of two (0 = by rule)", i, (double)scales[i]);
  }
  e->site_scale.assign(scales, scales + n);
  e->site_seen.assign(e->sites.size(), 0.f);
  return 0;
}

extern "C" int64_t st2_decoder_workspace_bytes(st2_engine* e, int32_t B, int32_t T) {
  if (!e || !e->dec.ready || B <= 0 || T <= 0) return -1;
  Ctx c;
  c.dry = true;
  c.a.dry = true;
  decoder_plan(c, *e, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, B, T, nullptr, nullptr);
  return c.a.peak + 256;
}

extern "C" int st2_decoder_forward(st2_engine* e, const float* asr, const float* f0, const float* n, const float* s,
                                   const float* sine_noise, const float* har_inject, int32_t B, int32_t T, float* wave,
                                   void* workspace, int64_t workspace_bytes, const st2_decoder_taps* taps, void* stream) {
  ST2_REQUIRE(e && e->dec.ready, "st2_decoder_forward: decoder weights not finalized");
  ST2_REQUIRE(asr && f0 && n && s && wave && workspace && B > 0 && T > 0, "st2_decoder_forward: bad arguments");
  ST2_REQUIRE(sine_noise || har_inject, "st2_decoder_forward: sine_noise (or har_inject) is required");
  ST2_REQUIRE((reinterpret_cast<uintptr_t>(workspace) & 255) == 0, "st2_decoder_forward: workspace must be 256-byte aligned");
  Ctx c;
  c.stream = stream;
  c.a.base = static_cast<char*>(workspace);
  c.a.cap = workspace_bytes;
  const int rc = decoder_plan(c, *e, asr, f0, n, s, sine_noise, har_inject, B, T, wave, taps);
  ST2_REQUIRE(!c.a.overflow, "st2_decoder_forward: workspace of %lld B is too small (need %lld B, see "
              "st2_decoder_workspace_bytes)", (long long)workspace_bytes, (long long)c.a.peak);
  return rc;
}

extern "C" int64_t st2_text_workspace_bytes(st2_engine* e, int32_t B, int32_t N) {
  if (!e || !e->text.ready || B <= 0 || N <= 0) return -1;
  Ctx c;
  c.dry = true;
  c.a.dry = true;
  text_plan(c, *e, nullptr, nullptr, B, N, nullptr);
  return c.a.peak + 256;
}

extern "C" int st2_text_forward(st2_engine* e, const int64_t* tokens, const int32_t* lengths, int32_t B, int32_t N,
                                float* t_en, void* workspace, int64_t workspace_bytes, void* stream) {
  ST2_REQUIRE(e && e->text.ready, "st2_text_forward: text-encoder weights not finalized");
  ST2_REQUIRE(tokens && t_en && workspace && B > 0 && N > 0, "st2_text_forward: bad arguments");
  ST2_REQUIRE((reinterpret_cast<uintptr_t>(workspace) & 255) == 0, "st2_text_forward: workspace must be 256-byte aligned");
  Ctx c;
  c.stream = stream;
  c.a.base = static_cast<char*>(workspace);
  c.a.cap = workspace_bytes;
  const int rc = text_plan(c, *e, tokens, lengths, B, N, t_en);
  ST2_REQUIRE(!c.a.overflow, "st2_text_forward: workspace of %lld B is too small (need %lld B, see st2_text_workspace_bytes)",
              (long long)workspace_bytes, (long long)c.a.peak);
  return rc;
}

extern "C" int64_t st2_bert_workspace_bytes(st2_engine* e, int32_t B, int32_t N) {
  if (!e || !e->bert.ready || B <= 0 || N <= 0) return -1;
  Ctx c;
  c.dry = true;
  c.a.dry = true;
  bert_plan(c, *e, nullptr, nullptr, B, N);
  return c.a.peak + 256;
}

extern "C" int st2_bert_forward(st2_engine* e, const int64_t* tokens, const int32_t* lengths, int32_t B, int32_t N,
                                float* hidden_cm, void* workspace, int64_t workspace_bytes, void* stream) {
  ST2_REQUIRE(e && e->bert.ready, "st2_bert_forward: PL-BERT weights not finalized");
  ST2_REQUIRE(tokens && hidden_cm && workspace && B > 0 && N > 0, "st2_bert_forward: bad arguments");
  ST2_REQUIRE(N <= e->bert.P, "st2_bert_forward: N=%d tokens exceed the %d rows of the position table", N, e->bert.P);
  ST2_REQUIRE((reinterpret_cast<uintptr_t>(workspace) & 255) == 0, "st2_bert_forward: workspace must be 256-byte aligned");
  Ctx c;
  c.stream = stream;
  c.a.base = static_cast<char*>(workspace);
  c.a.cap = workspace_bytes;
  View X = bert_plan(c, *e, tokens, lengths, B, N);
  View dst = wrap(hidden_cm, B, e->bert.H, N);
  RUN(c, g_be.copy_ncl(X.p, X.bs, X.cs, dst.p, dst.bs, dst.cs, B, e->bert.H, N, c.stream));
  ST2_REQUIRE(!c.a.overflow, "st2_bert_forward: workspace of %lld B is too small (need %lld B, see st2_bert_workspace_bytes)",
              (long long)workspace_bytes, (long long)c.a.peak);
  return c.rc;
}

extern "C" int64_t st2_style_workspace_bytes(st2_engine* e, int32_t which, int32_t B, int32_t n_mels, int32_t T) {
  if (!e || which < 0 || which > 1 || !e->style[which].ready || B <= 0 || n_mels != 80 || T < 80) return -1;
  Ctx c;
  c.dry = true;
  c.a.dry = true;
  style_plan(c, *e, e->style[which], nullptr, B, n_mels, T, nullptr);
  return c.a.peak + 256;
}

extern "C" int st2_style_forward(st2_engine* e, int32_t which, const float* mel, int32_t B, int32_t n_mels, int32_t T,
                                 float* style, void* workspace, int64_t workspace_bytes, void* stream) {
  ST2_REQUIRE(e && which >= 0 && which <= 1 && e->style[which].ready, "st2_style_forward: style-encoder weights not finalized");
  ST2_REQUIRE(mel && style && workspace && B > 0, "st2_style_forward: bad arguments");
  ST2_REQUIRE(n_mels == 80 && T >= 80, "st2_style_forward: needs an 80-bin mel of >= 80 frames (four halvings, then the 5x5 "
              "valid conv), got %d x %d", n_mels, T);
  ST2_REQUIRE((reinterpret_cast<uintptr_t>(workspace) & 255) == 0, "st2_style_forward: workspace must be 256-byte aligned");
  Ctx c;
  c.stream = stream;
  c.a.base = static_cast<char*>(workspace);
  c.a.cap = workspace_bytes;
  const int rc = style_plan(c, *e, e->style[which], mel, B, n_mels, T, style);
  ST2_REQUIRE(!c.a.overflow, "st2_style_forward: workspace of %lld B is too small (need %lld B, see st2_style_workspace_bytes)",
              (long long)workspace_bytes, (long long)c.a.peak);
  return rc;
}

extern "C" int st2_sizeof_front_args(void) { return (int)sizeof(st2_front_args); }

namespace {
const char* front_ready(const st2_engine* e) {
  if (!e) return "null engine";
  if (!e->text.ready) return "text-encoder weights not finalized (bit 3)";
  if (!e->bert.ready || !e->bert.has_enc) return "PL-BERT / bert_encoder weights not finalized (bit 4)";
  if (!e->dn.ready) return "denoiser weights not finalized (bit 1)";
  if (!e->dur.ready) return "duration-encoder weights not finalized (bit 2)";
  if (e->cfg.dn_channels != 2 * e->cfg.style_dim) return "cfg.dn_channels != 2 * cfg.style_dim";
  if (e->cfg.dn_embedding != e->bert.H) return "cfg.dn_embedding != PL-BERT hidden size";
  if (e->bert.enc.C_out != e->cfg.pred_hidden) return "bert_encoder width != cfg.pred_hidden";
  return nullptr;
}
}  // namespace

extern "C" int64_t st2_front_workspace_bytes(st2_engine* e, const st2_front_args* a) {
  if (front_ready(e) || !a || a->B <= 0 || a->N <= 0 || a->steps < 2) return -1;
  Ctx c;
  c.dry = true;
  c.a.dry = true;
  std::vector<double> table((size_t)(a->steps - 1) * ST2_SAMPLER_TABLE_COLS, 0.0);
  static const float dummy = 0.f;
  static int64_t dummy_dur;
  st2_front_args q = *a;  // only B, N, steps, embedding_scale and which optional pointers are set matter for the size
  q.table = table.data();
  if (e->cfg.multispeaker) q.ref_s = &dummy;
  if (a->durations) q.durations = &dummy_dur;
  front_plan(c, *e, q);
  return c.a.peak + 256;
}

extern "C" int st2_front_forward(st2_engine* e, const st2_front_args* a, void* workspace, int64_t workspace_bytes,
                                 void* stream) {
  const char* why = front_ready(e);
  ST2_REQUIRE(!why, "st2_front_forward: %s", why);
  ST2_REQUIRE(a && a->tokens && a->noise && a->step_noise && a->table && a->t_en && a->d_cm && a->s && a->ref && workspace &&
              a->B > 0 && a->N > 0 && a->steps >= 2 && a->tail >= 0, "st2_front_forward: bad arguments");
  ST2_REQUIRE(!e->cfg.multispeaker || a->ref_s, "st2_front_forward: the multispeaker denoiser needs ref_s");
  ST2_REQUIRE(a->N <= e->bert.P && a->N <= e->cfg.dn_max_length && a->N <= 512,
              "st2_front_forward: N=%d tokens exceed the position / fixed-embedding tables", a->N);
  ST2_REQUIRE((reinterpret_cast<uintptr_t>(workspace) & 255) == 0, "st2_front_forward: workspace must be 256-byte aligned");
  Ctx c;
  c.stream = stream;
  c.a.base = static_cast<char*>(workspace);
  c.a.cap = workspace_bytes;
  const int rc = front_plan(c, *e, *a);
  ST2_REQUIRE(!c.a.overflow, "st2_front_forward: workspace of %lld B is too small (need %lld B, see "
              "st2_front_workspace_bytes)", (long long)workspace_bytes, (long long)c.a.peak);
  return rc;
}

extern "C" int64_t st2_duration_workspace_bytes(st2_engine* e, int32_t B, int32_t N) {
  if (!e || !e->dur.ready || B <= 0 || N <= 0) return -1;
  Ctx c;
  c.dry = true;
  c.a.dry = true;
  static int64_t dummy_dur;
  duration_plan(c, *e, wrap(nullptr, B, e->cfg.pred_hidden, N), nullptr, nullptr, B, N, 0, nullptr, &dummy_dur);
  return c.a.peak + 256;
}

extern "C" int st2_duration_forward(st2_engine* e, const float* d_en, const float* s, const int32_t* lengths, int32_t B,
                                    int32_t N, int32_t tail, float* d_cm, int64_t* durations, void* workspace,
                                    int64_t workspace_bytes, void* stream) {
  ST2_REQUIRE(e && e->dur.ready, "st2_duration_forward: duration-encoder weights not finalized");
  ST2_REQUIRE(d_en && s && d_cm && workspace && B > 0 && N > 0 && tail >= 0, "st2_duration_forward: bad arguments");
  ST2_REQUIRE(N <= 512, "st2_duration_forward: N=%d tokens exceed the 512 of PL-BERT's position table", N);
  ST2_REQUIRE((reinterpret_cast<uintptr_t>(workspace) & 255) == 0, "st2_duration_forward: workspace must be 256-byte aligned");
  Ctx c;
  c.stream = stream;
  c.a.base = static_cast<char*>(workspace);
  c.a.cap = workspace_bytes;
  const int rc = duration_plan(c, *e, wrap(d_en, B, e->cfg.pred_hidden, N), s, lengths, B, N, tail, d_cm, durations);
  ST2_REQUIRE(!c.a.overflow, "st2_duration_forward: workspace of %lld B is too small (need %lld B, see "
              "st2_duration_workspace_bytes)", (long long)workspace_bytes, (long long)c.a.peak);
  return rc;
}

extern "C" int64_t st2_prosody_workspace_bytes(st2_engine* e, int32_t B, int32_t N, int32_t T) {
  if (!e || !e->pred.ready || B <= 0 || N <= 0 || T <= 0) return -1;
  Ctx c;
  c.dry = true;
  c.a.dry = true;
  prosody_plan(c, *e, nullptr, nullptr, nullptr, nullptr, B, N, T, 0, nullptr, nullptr, nullptr);
  return c.a.peak + 256;
}

extern "C" int st2_prosody_forward(st2_engine* e, const float* d_cm, const float* t_en, const int64_t* durations,
                                   const float* s, int32_t B, int32_t N, int32_t T, int32_t shift, float* asr, float* f0,
                                   float* n, void* workspace, int64_t workspace_bytes, void* stream) {
  ST2_REQUIRE(e && e->pred.ready, "st2_prosody_forward: predictor weights not finalized");
  ST2_REQUIRE(d_cm && t_en && durations && s && asr && f0 && n && workspace && B > 0 && N > 0 && T > 0,
              "st2_prosody_forward: bad arguments");
  ST2_REQUIRE(N <= 512, "st2_prosody_forward: N=%d tokens exceed the 512 of PL-BERT's position table", N);
  ST2_REQUIRE((reinterpret_cast<uintptr_t>(workspace) & 255) == 0, "st2_prosody_forward: workspace must be 256-byte aligned");
  Ctx c;
  c.stream = stream;
  c.a.base = static_cast<char*>(workspace);
  c.a.cap = workspace_bytes;
  const int rc = prosody_plan(c, *e, d_cm, t_en, durations, s, B, N, T, shift, asr, f0, n);
  ST2_REQUIRE(!c.a.overflow, "st2_prosody_forward: workspace of %lld B is too small (need %lld B, see "
              "st2_prosody_workspace_bytes)", (long long)workspace_bytes, (long long)c.a.peak);
  return rc;
}

extern "C" int st2_sampler_table(int32_t steps, double sigma_min, double sigma_max, double rho, double sigma_data,
                                 double* table, double* sigma0) {
  ST2_REQUIRE(steps >= 2 && table && sigma0 && rho > 0, "st2_sampler_table: bad arguments");
  std::vector<float> sig((size_t)steps + 1, 0.f);
  const double rinv = 1.0 / rho;
  for (int i = 0; i < steps; ++i) {  // KarrasSchedule.forward in fp32 (sampler.py:328-337)
    const float frac = (float)i / (float)(steps - 1);
    const float base = (float)pow(sigma_max, rinv) + frac * (float)(pow(sigma_min, rinv) - pow(sigma_max, rinv));
    sig[(size_t)i] = powf(base, (float)rho);
  }
  *sigma0 = (double)sig[0];
  auto weights = [&](double sigma, double* w4) {  // KDiffusion.get_scale_weights in fp32 (sampler.py:184-191)
    const float s = (float)sigma, sd = (float)sigma_data;
    w4[3] = (double)(logf(s) * 0.25f);
    w4[0] = (double)((sd * sd) / (s * s + sd * sd));
    w4[1] = (double)(s * sd * (1.0f / sqrtf(sd * sd + s * s)));
    w4[2] = (double)(1.0f / sqrtf(s * s + sd * sd));
  };
  for (int i = 0; i + 1 < steps; ++i) {
    double* row = table + (int64_t)i * ST2_SAMPLER_TABLE_COLS;
    const double s = (double)sig[(size_t)i], sn = (double)sig[(size_t)i + 1];
    const double up = sqrt(sn * sn * (s * s - sn * sn) / (s * s));  // ADPM2Sampler.get_sigmas, rho = 1
    const double down = sqrt(sn * sn - up * up);
    const double mid = (s + down) / 2.0;
    weights(s, row + 0);
    weights(mid, row + 4);
    row[8] = (mid - s) / s;
    row[9] = (down - s) / mid;
    row[10] = up;
  }
  return 0;
}

extern "C" int64_t st2_sampler_workspace_bytes(st2_engine* e, int32_t B, int32_t N, int32_t steps, double embedding_scale) {
  if (!e || !e->dn.ready || B <= 0 || N <= 0 || steps < 2) return -1;
  Ctx c;
  c.dry = true;
  c.a.dry = true;
  std::vector<double> table((size_t)(steps - 1) * ST2_SAMPLER_TABLE_COLS, 0.0);
  static const float dummy = 0.f;
  sampler_plan(c, *e, nullptr, nullptr, nullptr, &dummy, nullptr, nullptr, B, N, steps, embedding_scale, table.data(), 1.0, nullptr,
               nullptr);
  return c.a.peak + 256;
}

extern "C" int st2_sampler_run(st2_engine* e, const float* noise, const float* embedding, const float* features,
                               const float* step_noise, const int32_t* lengths, int32_t B, int32_t N, int32_t steps,
                               double embedding_scale, const double* table, double sigma0, float* out, void* workspace,
                               int64_t workspace_bytes, float* step_taps, void* stream) {
  ST2_REQUIRE(e && e->dn.ready, "st2_sampler_run: denoiser weights not finalized");
  ST2_REQUIRE(noise && embedding && step_noise && table && out && workspace && B > 0 && N > 0 && steps >= 2,
              "st2_sampler_run: bad arguments");
  ST2_REQUIRE(N <= e->cfg.dn_max_length, "st2_sampler_run: N=%d exceeds the fixed-embedding length %d", N,
              e->cfg.dn_max_length);
  ST2_REQUIRE((reinterpret_cast<uintptr_t>(workspace) & 255) == 0, "st2_sampler_run: workspace must be 256-byte aligned");
  Ctx c;
  c.stream = stream;
  c.a.base = static_cast<char*>(workspace);
  c.a.cap = workspace_bytes;
  const int rc = sampler_plan(c, *e, noise, embedding, nullptr, features, step_noise, lengths, B, N, steps, embedding_scale, table,
                              sigma0, out, step_taps);
  ST2_REQUIRE(!c.a.overflow, "st2_sampler_run: workspace of %lld B is too small (need %lld B, see "
              "st2_sampler_workspace_bytes)", (long long)workspace_bytes, (long long)c.a.peak);
  return rc;
}
