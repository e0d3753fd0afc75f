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

#include "st2_plan_decoder.inc"
#include "st2_plan_sampler.inc"
#include "st2_plan_prosody.inc"
#include "st2_plan_duration.inc"
#include "st2_plan_text.inc"
#include "st2_plan_front.inc"
#include "st2_plan_style.inc"

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
