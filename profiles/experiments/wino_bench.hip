// EXPERIMENT (not part of the library): Toom-Cook F(M, R) forms -- F(3, 3) and F(4, 4) -- of the split-f16 MFMA conv for the
// vocoder's long filters (k = 11 / 7 as ceil(k / R) R-tap groups accumulated in the transform domain), DESIGN.md section 7
// item 0.  The tile step M equals the group width R, so ONE transformed copy of the input serves all groups.  Stand-alone binary: plain
// hipMalloc buffers, its own CPU fp64 check, HIP events.  Written at the end of round 2 without GPU minutes left -- the
// first thing to run in round 3:
//
//   tools/bin/wino_bench selftest        no GPU: host emulation of the kernels' data flow through the same checker
//   tools/bin/wino_bench check           small shapes (edge tiles included) against a direct fp64 conv on the host
//   tools/bin/wino_bench [k=11] [C=128] [L=48001] [B=32] [reps=10] [TN=2] [dil=1] [src_dil=1] [scheme=33] [occ=2] [WM=4]   timing
//                                          (TN = 32-tile blocks per wave; dil = the conv's dilation, src_dil = dilation of the
//                                           layer that produced x; scheme 33 = F(3,3) (TN 1 or 2), 44 = F(4,4) (TN 1, occ = 2 or 3 workgroups / CU), 66 = F(6,6) (TN 1, occ 2);
//                                           WM = waves along co: 4 -> 128 co x 32 TN tiles, 2 -> 64 co x 64 TN tiles per workgroup)
//
// Maths (P = M + R - 1 points: 0, +-1, 2, inf for F(3,3); 0, +-1, +-2, 1/2, inf for F(4,4); matrices built by toom() below,
// rounding studied on the CPU by tools/winograd_numerics.py):
//   y[M T + i] = sum_j w[j] a[M T + i + j - pad],  w zero-padded to R G taps, group g = taps R g .. R g + R - 1
//   V_p[ci][T'] = sum_n BT[p][n] a[ci][M T' - pad + n]               (input transform, fp32, then hi / lo f16 split)
//   U_{g,p}[co][ci] = sum_r Gm[p][r] w[co][ci][R g + r]              (weight transform, fp64 at pack time)
//   Y_p[co][T]  = sum_{g, ci} U_{g,p}[co][ci] V_p[ci][T + g]         (P independent G-tap convs over the TILE index: MFMA)
//   y[co][M T + i] = sum_p AT[i][p] Y_p[co][T]                       (inverse transform on the fp32 accumulators)
// i.e. P G / M MFMA-multiplies per output: k = 11: 6.67 (F(3,3)) / 5.25 (F(4,4)) vs 11; k = 7: 5 / 3.5 vs 7.
//
// Kernel structure = csrc/st2_conv1d_xs_impl.h with "taps" t = g * P + p: the packed-weight layout (st2.h) is reused with
// ks_eff = P G, the chunk image in LDS has one row set per point, the accumulators are acc[p][j] (transposed tile: lane =
// output row, registers = runs of 4 consecutive TILES = 4 M consecutive outputs -> M 16-byte stores).
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "wino_impl.h"

#define CK(x)                                                                  \
  do {                                                                         \
    hipError_t e_ = (x);                                                       \
    if (e_ != hipSuccess) {                                                    \
      fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                  \
      return 1;                                                                \
    }                                                                          \
  } while (0)

using namespace st2w;

// ---------------------------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------------------------
static uint32_t rng_state = 12345u;
static float frand() {  // uniform in [-1, 1)
  rng_state = rng_state * 1664525u + 1013904223u;
  return ((int)((rng_state >> 8) & 0xffff) - 32768) * (1.f / 32768.f);
}

// Toom-Cook F(M, R) with P - 1 finite points + infinity:  y = AT [(Gm g) * (BT d)]  (Vandermonde construction, fp64)
struct ToomD {
  int M, R, P;
  double AT[6][11], Gm[11][6], BT[11][11];
};
static ToomD toom(int M, int R) {
  ToomD t;
  memset(&t, 0, sizeof(t));
  t.M = M; t.R = R; t.P = M + R - 1;
  const int n = t.P;
  static const double pts33[] = {0, 1, -1, 2}, pts44[] = {0, 1, -1, 2, -2, 0.5},
                      pts66[] = {0, 1, -1, 2, -2, 0.5, -0.5, 3, -3, 1.0 / 3};
  const double* pts = (M == 3) ? pts33 : (M == 4 ? pts44 : pts66);
  for (int k = 0; k < n - 1; ++k) {
    double den = 1.0;
    for (int j = 0; j < n - 1; ++j)
      if (j != k) den *= pts[k] - pts[j];
    for (int i = 0; i < M; ++i) t.AT[i][k] = std::pow(pts[k], i);
    for (int r = 0; r < R; ++r) t.Gm[k][r] = std::pow(pts[k], r) / den;
    // BT row k: coefficients of prod_{j != k} (x - p_j), ascending powers
    double poly[12] = {1.0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    int deg = 0;
    for (int j = 0; j < n - 1; ++j) {
      if (j == k) continue;
      for (int z = deg + 1; z >= 1; --z) poly[z] = poly[z - 1] - pts[j] * poly[z];
      poly[0] = -pts[j] * poly[0];
      ++deg;
    }
    for (int z = 0; z <= deg; ++z) t.BT[k][z] = poly[z];
  }
  t.AT[M - 1][n - 1] = 1.0;
  t.Gm[n - 1][R - 1] = 1.0;
  {  // last BT row: prod_j (x - p_j)
    double poly[12] = {1.0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    int deg = 0;
    for (int j = 0; j < n - 1; ++j) {
      for (int z = deg + 1; z >= 1; --z) poly[z] = poly[z - 1] - pts[j] * poly[z];
      poly[0] = -pts[j] * poly[0];
      ++deg;
    }
    for (int z = 0; z <= deg; ++z) t.BT[n - 1][z] = poly[z];
  }
  return t;
}
static int toom_selfcheck(const ToomD& t) {  // AT [(Gm g) * (BT d)] == correlation of d with g
  double worst = 0.0;
  for (int trial = 0; trial < 20; ++trial) {
    double dd[11], g[6];
    for (int i = 0; i < t.P; ++i) dd[i] = frand();
    for (int i = 0; i < t.R; ++i) g[i] = frand();
    for (int i = 0; i < t.M; ++i) {
      double ref = 0.0, got = 0.0;
      for (int j = 0; j < t.R; ++j) ref += g[j] * dd[i + j];
      for (int p = 0; p < t.P; ++p) {
        double u = 0.0, v = 0.0;
        for (int r = 0; r < t.R; ++r) u += t.Gm[p][r] * g[r];
        for (int n = 0; n < t.P; ++n) v += t.BT[p][n] * dd[n];
        got += t.AT[i][p] * u * v;
      }
      worst = std::fmax(worst, std::fabs(got - ref));
    }
  }
  printf("toom F(%d,%d): max |transform-domain - direct| = %.2e\n", t.M, t.R, worst);
  return worst < (t.M <= 4 ? 1e-12 : 1e-9) ? 0 : 1;
}
static Toom toom_f32(const ToomD& t) {
  Toom f;
  memset(&f, 0, sizeof(f));
  for (int p = 0; p < t.P; ++p)
    for (int n = 0; n < t.P; ++n) f.BT[p][n] = (float)t.BT[p][n];
  for (int i = 0; i < t.M; ++i)
    for (int p = 0; p < t.P; ++p) f.AT[i][p] = (float)t.AT[i][p];
  return f;
}

struct Packed {
  std::vector<_Float16> q;
  std::vector<float> row_scale;
  int co_pad, cin_pad, ks_eff;
};

// U_{g,p} = sum_r Gm[p][r] w[R g + r] in fp64, then csrc/st2_engine.hip pack_split with ks = P G (per-row power-of-two scale,
// hi / lo f16, [(i16 * ks + t) * 2 + kg][co_pad][16])
static Packed pack_w3(const ToomD& tm, const std::vector<float>& w, int C_out, int C_in, int K, int G) {
  const int P = tm.P, R = tm.R;
  Packed r;
  r.ks_eff = P * G;
  r.cin_pad = (C_in + 15) / 16 * 16;
  r.co_pad = (C_out + 127) / 128 * 128;
  r.q.assign((size_t)(r.cin_pad / 16) * r.ks_eff * 2 * r.co_pad * 16, (_Float16)0.0f);
  r.row_scale.assign((size_t)r.co_pad, 1.0f);
  std::vector<float> u((size_t)C_in * r.ks_eff);
  for (int co = 0; co < C_out; ++co) {
    float amax = 0.f;
    for (int ci = 0; ci < C_in; ++ci)
      for (int g = 0; g < G; ++g)
        for (int p = 0; p < P; ++p) {
          double s = 0.0;
          for (int rr = 0; rr < R; ++rr) {
            const int j = R * g + rr;
            if (j < K) s += tm.Gm[p][rr] * (double)w[((size_t)co * C_in + ci) * K + j];
          }
          const float v = (float)s;
          u[(size_t)ci * r.ks_eff + g * P + p] = v;
          amax = std::fmax(amax, std::fabs(v));
        }
    float scale = 1.0f;
    if (amax > 0.f) {
      int e = 0;
      (void)frexpf(amax, &e);
      scale = ldexpf(1.0f, 14 - e);
    }
    r.row_scale[co] = 1.0f / scale;
    for (int ci = 0; ci < C_in; ++ci)
      for (int t = 0; t < r.ks_eff; ++t) {
        const float v = u[(size_t)ci * r.ks_eff + t] * scale;
        const _Float16 hi = (_Float16)v;
        const _Float16 lo = (_Float16)(v - (float)hi);
        const int i16 = ci / 16, kg = (ci % 16) / 8, e8 = ci % 8;
        const size_t base = ((((size_t)i16 * r.ks_eff + t) * 2 + kg) * r.co_pad + co) * 16;
        r.q[base + e8] = hi;
        r.q[base + 8 + e8] = lo;
      }
  }
  return r;
}

static double snake_ref(double v, double al) {
  const double s = std::sin(al * v);
  return v + s * s / al;
}

template <class S, int G, int TN, int OCC, int WM>
static int launch_conv(const WArgs& d, int B, int nblk) {  // nblk = wave blocks of 32 TN tiles (a multiple of WN)
  constexpr int WN = 4 / WM;
  constexpr int BT_ = 32 * TN * WN;
  constexpr int XW = BT_ + G - 1;
  constexpr int NS = (2 * S::P * CG * XW + NT - 1) / NT;
  const size_t smem = (size_t)2 * NS * NT * 16;
  static bool attr_done = false;
  if (!attr_done) {
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_w3_kernel<S, G, TN, OCC, WM>), hipFuncAttributeMaxDynamicSharedMemorySize,
                           160 * 1024));
    attr_done = true;
  }
  dim3 grid(nblk / WN, (d.C_out + 32 * WM - 1) / (32 * WM), B);
  hipLaunchKernelGGL((conv_w3_kernel<S, G, TN, OCC, WM>), grid, dim3(NT), smem, 0, d);
  CK(hipGetLastError());
  return 0;
}

struct Case {
  int M, R, P, TN, WM, K, C, L, B, dil, src_dil;
  bool adain;
  // derived
  int G, padq, Lq, n_tiles, BT_, nblk, Lt, cg_tot, pitch, pitch_q, pitch_s, VB;
  void derive() {
    P = M + R - 1;
    G = (K + R - 1) / R; padq = (K - 1) / 2;
    Lq = (L + dil - 1) / dil;                       // longest stride-d subsequence
    n_tiles = (Lq + M - 1) / M; BT_ = 32 * TN;      // BT_ = tiles per WAVE block (statistics granule)
    const int WN = 4 / WM;
    nblk = (n_tiles + BT_ * WN - 1) / (BT_ * WN) * WN;  // wave blocks, padded to whole workgroups
    Lt = nblk * BT_ + 8;                            // every workgroup stages WN BT_ + G - 1 <= WN BT_ + 3 tiles
    cg_tot = (C + 15) / 16 * 16 / 8;
    pitch = (L + 31) / 32 * 32;                     // natural rows (host reference, residual)
    pitch_q = (Lq + 31) / 32 * 32;                  // output rows: y[vb = b * dil + r][co][q]
    pitch_s = ((L + src_dil - 1) / src_dil + 31) / 32 * 32;  // input rows when x is residue-major
    VB = B * dil;
  }
};

struct HostData {
  std::vector<float> hx, hxsrc, hres, hw, hb, hst, hga, hbe, hal;  // hx natural [B][C][pitch]; hxsrc = what the device reads
};

static void make_data(const Case& c, HostData& h) {
  h.hx.assign((size_t)c.B * c.C * c.pitch, 0.f); h.hres.assign((size_t)c.B * c.C * c.pitch, 0.f);
  h.hw.resize((size_t)c.C * c.C * c.K); h.hb.resize(c.C); h.hst.resize((size_t)c.B * c.C * 2);
  h.hga.resize((size_t)c.B * c.C); h.hbe.resize((size_t)c.B * c.C); h.hal.resize(c.C);
  rng_state = 777u + c.K * 131 + c.L + 7 * c.dil + 3 * c.src_dil;
  for (auto& v : h.hx) v = 1.5f * frand();
  for (auto& v : h.hres) v = frand();
  const float wsc = 1.0f / std::sqrt((float)(c.C * c.K));
  for (auto& v : h.hw) v = 1.7f * wsc * frand();
  for (auto& v : h.hb) v = 0.3f * frand();
  for (size_t i = 0; i < h.hst.size(); i += 2) { h.hst[i] = 0.2f * frand(); h.hst[i + 1] = 1.0f + 0.3f * frand(); }
  for (auto& v : h.hga) v = 0.3f * frand();
  for (auto& v : h.hbe) v = 0.3f * frand();
  for (auto& v : h.hal) v = 1.0f + 0.5f * frand();
  if (c.src_dil > 1) {  // the residue-major layout a dilated layer leaves behind
    h.hxsrc.assign((size_t)c.B * c.src_dil * c.C * c.pitch_s, 0.f);
    for (int b = 0; b < c.B; ++b)
      for (int ci = 0; ci < c.C; ++ci)
        for (int l = 0; l < c.L; ++l)
          h.hxsrc[(((size_t)b * c.src_dil + l % c.src_dil) * c.C + ci) * c.pitch_s + l / c.src_dil] =
              h.hx[((size_t)b * c.C + ci) * c.pitch + l];
  } else {
    h.hxsrc = h.hx;
  }
}

// hy [VB][C][pitch_q], hpart [VB][C][nblk][2] (device layouts) against a direct fp64 dilated conv of the natural tensors
static int check_result(const Case& c, const HostData& h, const std::vector<float>& hy, const std::vector<float>& hpart,
                        const char* what) {
  const int C = c.C, L = c.L, K = c.K;
  std::vector<double> act((size_t)C * L);
  double err = 0.0, ymax = 0.0, perr = 0.0, pmax = 0.0;
  for (int b = 0; b < c.B; ++b) {
    for (int ci = 0; ci < C; ++ci)
      for (int l = 0; l < L; ++l) {
        double u = h.hx[((size_t)b * C + ci) * c.pitch + l];
        if (c.adain) {
          const double w = (u - h.hst[((size_t)b * C + ci) * 2]) * h.hst[((size_t)b * C + ci) * 2 + 1];
          u = snake_ref((1.0 + h.hga[(size_t)b * C + ci]) * w + h.hbe[(size_t)b * C + ci], h.hal[ci]);
        }
        act[(size_t)ci * L + l] = u;
      }
    for (int co = 0; co < C; ++co) {
      std::vector<double> row(L, (double)h.hb[co]);
      for (int ci = 0; ci < C; ++ci)
        for (int j = 0; j < K; ++j) {
          const double wv = h.hw[((size_t)co * C + ci) * K + j];
          const int off = (j - c.padq) * c.dil;
          const int lo = std::max(0, -off), hi = std::min(L, L - off);
          const double* ar = act.data() + (size_t)ci * L + off;
          for (int l = lo; l < hi; ++l) row[l] += wv * ar[l];
        }
      std::vector<double> s1((size_t)c.dil * c.nblk, 0.0), s2((size_t)c.dil * c.nblk, 0.0);
      for (int l = 0; l < L; ++l) {
        const int r = l % c.dil, q = l / c.dil;
        const double ref = row[l] + (c.dil == 1 ? (double)h.hres[((size_t)b * C + co) * c.pitch + l] : 0.0);
        const double got = hy[(((size_t)b * c.dil + r) * C + co) * c.pitch_q + q];
        err = std::fmax(err, std::fabs(got - ref));
        ymax = std::fmax(ymax, std::fabs(ref));
        s1[(size_t)r * c.nblk + q / (c.M * c.BT_)] += got;
        s2[(size_t)r * c.nblk + q / (c.M * c.BT_)] += got * got;
      }
      for (int r = 0; r < c.dil; ++r)
        for (int t = 0; t < c.nblk; ++t) {
          const float* pp = hpart.data() + ((((size_t)b * c.dil + r) * C + co) * c.nblk + t) * 2;
          perr = std::fmax(perr, std::fmax(std::fabs(pp[0] - s1[(size_t)r * c.nblk + t]), std::fabs(pp[1] - s2[(size_t)r * c.nblk + t])));
          pmax = std::fmax(pmax, std::fmax(std::fabs(s1[(size_t)r * c.nblk + t]), std::fabs(s2[(size_t)r * c.nblk + t])));
        }
    }
  }
  const bool ok = err < 2e-5 * ymax && perr < 1e-4 * pmax;
  printf("wino %s F(%d,%d) TN=%d k=%d dil=%d src_dil=%d C=%d L=%d B=%d adain=%d: max |y - ref| = %.3e of %.3e, partial sums "
         "%.3e of %.3e  -> %s\n", what, c.M, c.R, c.TN, K, c.dil, c.src_dil, C, L, c.B, (int)c.adain, err, ymax, perr, pmax,
         ok ? "OK" : "MISMATCH");
  return ok ? 0 : 1;
}

// the activation arithmetic of act_w3_kernel's staging loop on the host (sinf instead of the device's polynomial)
static float host_a_rq(const Case& c, const HostData& h, int b, int r, int ci, int q) {
  const int x_cs = c.src_dil > 1 ? c.pitch_s : c.pitch;
  const int64_t x_bs = (int64_t)c.C * x_cs;
  const int l = c.dil * q + r;
  if (!(q >= 0 && l < c.L && ci < c.C)) return 0.f;
  float u;
  if (c.src_dil > 1) {
    const int rs = l % c.src_dil, qs = l / c.src_dil;
    u = h.hxsrc[((int64_t)b * c.src_dil + rs) * x_bs + (int64_t)ci * x_cs + qs];
  } else {
    u = h.hxsrc[(int64_t)b * x_bs + (int64_t)ci * x_cs + l];
  }
  if (c.adain) {
    float w = (u - h.hst[((size_t)b * c.C + ci) * 2]) * h.hst[((size_t)b * c.C + ci) * 2 + 1];
    w = (1.0f + h.hga[(size_t)b * c.C + ci]) * w + h.hbe[(size_t)b * c.C + ci];
    const float al = h.hal[ci], sn = sinf(al * w);
    u = w + (1.0f / al) * (sn * sn);
  }
  return u * 8.f;
}
static void host_transform(const Case& c, const Toom& tm, const float* dd, float* v) {  // the device's fmaf order
  for (int p = 0; p < c.P; ++p) {
    float t = tm.BT[p][0] * dd[0];
    for (int n = 1; n < c.P; ++n) t = fmaf(tm.BT[p][n], dd[n], t);
    v[p] = t;
  }
}

// Host emulation of the two kernels' DATA FLOW (same plane / packed-weight / permutation index formulas, same transform
// constants, fp64 accumulation instead of MFMA): `wino_bench selftest` runs it through the same checker without a GPU, so
// that the packer, the layouts, the transforms and the checker itself are known to be consistent before the first GPU visit.
static void host_act(const Case& c, const HostData& h, const Toom& tm, std::vector<_Float16>& vs) {
  const int P = c.P;
  vs.assign((size_t)c.VB * 2 * P * c.cg_tot * c.Lt * 8, (_Float16)0.0f);
  for (int vb = 0; vb < c.VB; ++vb) {
    const int b = vb / c.dil, r = vb - b * c.dil;
    for (int cg = 0; cg < c.cg_tot; ++cg)
      for (int T = 0; T < c.Lt; ++T)
        for (int e = 0; e < 8; ++e) {
          float dd[11], v[11];
          for (int n = 0; n < P; ++n) dd[n] = host_a_rq(c, h, b, r, cg * 8 + e, c.M * T - c.padq + n);
          host_transform(c, tm, dd, v);
          for (int p = 0; p < P; ++p) {
            const float uc = v[p] > 65504.f ? 65504.f : (v[p] < -65504.f ? -65504.f : v[p]);
            const _Float16 hh = (_Float16)uc;
            vs[(((((size_t)vb * 2 + 0) * P + p) * c.cg_tot + cg) * c.Lt + T) * 8 + e] = hh;
            vs[(((((size_t)vb * 2 + 1) * P + p) * c.cg_tot + cg) * c.Lt + T) * 8 + e] = (_Float16)(uc - (float)hh);
          }
        }
  }
}

// inverse transform + epilogue arithmetic shared by the host emulations (the device's fmaf order on fp32 inputs)
static float host_out(const Case& c, const Toom& tm, const double* Y, int i) {
  float t = tm.AT[i][0] * (float)Y[0];
  for (int p = 1; p < c.P; ++p) t = fmaf(tm.AT[i][p], (float)Y[p], t);
  return t;
}

static void host_conv(const Case& c, const HostData& h, const Toom& tm, const Packed& pk, const std::vector<_Float16>& vs,
                      std::vector<float>& hy, std::vector<float>& hpart) {
  const int P = c.P;
  hy.assign((size_t)c.VB * c.C * c.pitch_q, 0.f);
  hpart.assign((size_t)c.VB * c.C * c.nblk * 2, 0.f);
  const int ks = pk.ks_eff;
  for (int vb = 0; vb < c.VB; ++vb) {
    const int b = vb / c.dil, r = vb - b * c.dil;
    const int L_eff = c.dil > 1 ? (c.L - r + c.dil - 1) / c.dil : c.L;
    for (int co = 0; co < c.C; ++co) {
      const float osc_r = (1.f / 8.f) * pk.row_scale[co];
      for (int blk = 0; blk < c.nblk; ++blk) {
        double s1 = 0.0, s2 = 0.0;
        for (int tt = 0; tt < c.BT_; ++tt) {
          const int T = blk * c.BT_ + tt;
          double Y[11] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
          for (int i16 = 0; i16 < pk.cin_pad / 16; ++i16)
            for (int g = 0; g < c.G; ++g)
              for (int p = 0; p < P; ++p)
                for (int kg = 0; kg < 2; ++kg) {
                  const size_t wb = ((((size_t)i16 * ks + (g * P + p)) * 2 + kg) * pk.co_pad + co) * 16;
                  const size_t xh = (((((size_t)vb * 2 + 0) * P + p) * c.cg_tot + (i16 * 2 + kg)) * c.Lt + T + g) * 8;
                  const size_t xl = (((((size_t)vb * 2 + 1) * P + p) * c.cg_tot + (i16 * 2 + kg)) * c.Lt + T + g) * 8;
                  for (int e8 = 0; e8 < 8; ++e8) {
                    const double wh = (double)(float)pk.q[wb + e8], wl = (double)(float)pk.q[wb + 8 + e8];
                    const double ah = (double)(float)vs[xh + e8], al = (double)(float)vs[xl + e8];
                    Y[p] += wh * ah + wh * al + wl * ah;
                  }
                }
          for (int i = 0; i < c.M; ++i) {
            const int q = c.M * T + i;
            if (q >= L_eff) continue;
            float t = fmaf(host_out(c, tm, Y, i), osc_r, h.hb[co]);
            if (c.dil == 1) t += h.hres[((size_t)b * c.C + co) * c.pitch + q];
            hy[((size_t)vb * c.C + co) * c.pitch_q + q] = t;
            s1 += t;
            s2 += (double)t * t;
          }
        }
        hpart[(((size_t)vb * c.C + co) * c.nblk + blk) * 2 + 0] = (float)s1;
        hpart[(((size_t)vb * c.C + co) * c.nblk + blk) * 2 + 1] = (float)s2;
      }
    }
  }
}

// Thread-level host twin of conv_w3_kernel (selftest only): the SAME index expressions as the device code -- staging
// offsets, LDS image, per-lane fragment addresses, weight pointer arithmetic, accumulator-register -> tile mapping, epilogue
// addresses -- with the MFMA replaced by its documented semantics (A operand: lane (m, kg) holds A[m][8 kg .. 8 kg + 7];
// B operand: lane (n, kg) holds B[8 kg .. 8 kg + 7][n]; D: lane (n, kg) register r holds D[8 (r / 4) + 4 kg + r % 4][n]).
template <class S, int G, int TN, int WM>
static void host_twin_conv(const Case& c, const HostData& h, const Toom& tm, const Packed& pk, const std::vector<_Float16>& vs,
                           std::vector<float>& hy, std::vector<float>& hpart) {
  constexpr int P = S::P, M = S::M;
  constexpr int WN = 4 / WM, BM = 32 * WM;
  constexpr int BT_ = 32 * TN * WN, XW = BT_ + G - 1, ROWS = 2 * P * CG, SS = ROWS * XW, NS = (SS + NT - 1) / NT, LBUF = NS * NT;
  constexpr int SPC = G * P;
  const int cg_tot = c.cg_tot, Lt = c.Lt, C_out = c.C;
  hy.assign((size_t)c.VB * c.C * c.pitch_q, 0.f);
  hpart.assign((size_t)c.VB * c.C * c.nblk * 2, 0.f);
  const int64_t pstride = (int64_t)cg_tot * Lt, plane_stride = (int64_t)P * pstride;
  const int64_t a_step = (int64_t)2 * pk.co_pad * 2;
  const int nchunk = pk.cin_pad / CI_T;
  auto slot_of = [&](const std::vector<_Float16>& arr, int64_t h8_index, int e) { return (double)(float)arr[(size_t)h8_index * 8 + e]; };
  std::vector<int64_t> image(LBUF);  // LDS image: global h8 index staged into each slot
  std::vector<double> D((size_t)4 * P * TN * 32 * 32);
  for (int b = 0; b < c.VB; ++b) {  // b = the grid's (virtual) batch index
    const int L_eff = c.dil > 1 ? (c.L - (b % c.dil) + c.dil - 1) / c.dil : c.L;
    for (int by = 0; by < (C_out + BM - 1) / BM; ++by)
      for (int bx = 0; bx < c.nblk / WN; ++bx) {
        const int t0 = bx * BT_, m0 = by * BM;
        const int64_t vsb = (int64_t)b * 2 * plane_stride + t0;
        std::fill(D.begin(), D.end(), 0.0);
        for (int ch = 0; ch < nchunk; ++ch) {
          for (int tid = 0; tid < NT; ++tid)
            for (int i = 0; i < NS; ++i) {
              const int slot = tid + i * NT;
              const int row = slot / XW, col = slot - row * XW;
              const int pl = row / (P * CG), rem = row % (P * CG), p = rem / CG, g8 = rem % CG;
              const int64_t soff = slot < SS ? (pl * plane_stride + p * pstride + (int64_t)g8 * Lt + col) : 0;
              image[tid + i * NT] = vsb + (int64_t)ch * CG * Lt + soff;
            }
          for (int i = 0; i < SPC; ++i) {
            const int g = i / P, p = i % P;
            const int64_t step = (int64_t)ch * SPC + i;
            for (int wave = 0; wave < 4; ++wave)
              for (int j = 0; j < TN; ++j)
                for (int m = 0; m < 32; ++m)      // A-operand lane l31 = m
                  for (int n = 0; n < 32; ++n) {  // B-operand lane l31 = n
                    double sum = 0.0;
                    const int wm = wave / WN, wn = wave % WN;
                    for (int kg = 0; kg < 2; ++kg) {
                      const int64_t xh = image[(p * CG + kg) * XW + g + wn * (32 * TN) + m + j * 32];
                      const int64_t xl = image[(p * CG + kg) * XW + g + wn * (32 * TN) + m + j * 32 + P * CG * XW];
                      const int co_a = m0 + wm * 32 + n;
                      const int64_t ap = ((int64_t)kg * pk.co_pad + co_a) * 2 + step * a_step;
                      for (int e = 0; e < 8; ++e) {
                        const double bh = slot_of(vs, xh, e), bl = slot_of(vs, xl, e);
                        const double ah = slot_of(pk.q, ap, e), al = slot_of(pk.q, ap + 1, e);
                        sum += bh * ah + bl * ah + bh * al;
                      }
                    }
                    D[((((size_t)wave * P + p) * TN + j) * 32 + m) * 32 + n] += sum;
                  }
          }
        }
        for (int wave = 0; wave < 4; ++wave)
          for (int lane = 0; lane < 64; ++lane) {
            const int kg = lane >> 5, l31 = lane & 31;
            const int wm = wave / WN, wn = wave % WN;
            const int tw = t0 + wn * (32 * TN);
            const int co = m0 + wm * 32 + l31;
            if (co >= C_out) continue;
            const float osc_r = (1.f / 8.f) * pk.row_scale[co];
            double s1 = 0.0, s2 = 0.0;
            for (int j = 0; j < TN; ++j)
              for (int q = 0; q < 4; ++q) {
                const int l0 = M * (tw + 32 * j + 8 * q + 4 * kg);
                for (int e = 0; e < 4; ++e) {
                  const int rr = 4 * q + e, m = 8 * (rr / 4) + 4 * kg + rr % 4;
                  double Y[11];
                  for (int p = 0; p < P; ++p) Y[p] = D[((((size_t)wave * P + p) * TN + j) * 32 + m) * 32 + l31];
                  for (int i = 0; i < M; ++i) {
                    const int l = l0 + M * e + i;
                    if (l >= L_eff) continue;
                    float t = fmaf(host_out(c, tm, Y, i), osc_r, h.hb[co]);
                    if (c.dil == 1) t += h.hres[((size_t)b * C_out + co) * c.pitch + l];
                    hy[((size_t)b * C_out + co) * c.pitch_q + l] = t;
                    s1 += t;
                    s2 += (double)t * t;
                  }
                }
              }
            float* pp = hpart.data() + (((size_t)b * C_out + co) * c.nblk + bx * WN + wn) * 2;  // lane + its kg partner
            pp[0] += (float)s1;
            pp[1] += (float)s2;
          }
      }
  }
}

// Thread-level host twin of act_w3_kernel: the same staging loop (idx -> (e, i) -> q -> l, source permutation) into an
// `sa` image and the same per-thread reads / destination index; compared with host_act on hi + lo (selftest only).
template <class S>
static int host_twin_act(const Case& c, const HostData& h, const Toom& tm, const std::vector<_Float16>& vs_ref) {
  constexpr int P = S::P, M = S::M, AT_TILES = ActGeom<S>::TILES, AT_POS = ActGeom<S>::POS, AT_PITCH = ActGeom<S>::PITCH;
  const int64_t pstride = (int64_t)c.cg_tot * c.Lt;
  std::vector<float> sa((size_t)8 * AT_PITCH);
  double worst = 0.0, scale = 0.0;
  for (int vb = 0; vb < c.VB; ++vb)
    for (int cg = 0; cg < c.cg_tot; ++cg)
      for (int bx = 0; bx < (c.Lt + AT_TILES - 1) / AT_TILES; ++bx) {
        const int tile0 = bx * AT_TILES;
        const int b = vb / c.dil, r = vb - b * c.dil;
        const int p0 = M * tile0 - c.padq;
        for (int tid = 0; tid < 256; ++tid)
          for (int idx = tid; idx < 8 * AT_POS; idx += 256) {
            const int e = idx / AT_POS, i = idx - e * AT_POS;
            sa[(size_t)e * AT_PITCH + i] = host_a_rq(c, h, b, r, cg * 8 + e, p0 + i);
          }
        for (int tid = 0; tid < AT_TILES; ++tid) {
          const int T = tile0 + tid;
          if (T >= c.Lt) continue;
          for (int e = 0; e < 8; ++e) {
            const float* sp = sa.data() + (size_t)e * AT_PITCH + M * tid;
            float v[11];
            host_transform(c, tm, sp, v);
            for (int p = 0; p < P; ++p) {
              const int64_t dst = (int64_t)vb * 2 * P * pstride + (int64_t)cg * c.Lt + T;  // h8 index of the hi slot of point 0
              const double ref = (double)(float)vs_ref[(size_t)(dst + p * pstride) * 8 + e] +
                                 (double)(float)vs_ref[(size_t)(dst + (P + p) * pstride) * 8 + e];
              worst = std::fmax(worst, std::fabs(ref - (double)v[p]));
              scale = std::fmax(scale, std::fabs((double)v[p]));
            }
          }
        }
      }
  const bool ok = worst < 1e-5 * scale;  // hi + lo carries ~22 bits of v
  printf("wino selftest (transform-pass twin) F(%d,%d) k=%d dil=%d src_dil=%d C=%d L=%d: max |hi + lo - v| = %.3e of %.3e -> %s\n",
         c.M, c.R, c.K, c.dil, c.src_dil, c.C, c.L, worst, scale, ok ? "OK" : "MISMATCH");
  return ok ? 0 : 1;
}

template <class S, int TN, class F>
static auto by_groups(int G, F&& f) {  // G = ceil(k / R): 4 or 3 for F(3,3), 3 or 2 for F(4,4)
  if (G == 4) return f(std::integral_constant<int, 4>{});
  if (G == 3) return f(std::integral_constant<int, 3>{});
  return f(std::integral_constant<int, 2>{});
}

// mode 0 = timing, 1 = GPU check, 2 = host selftest
template <class S, int TN, int OCC, int WM>
static int run_case(int K, int dil, int src_dil, int C, int L, int B, int reps, int mode, bool adain) {
  Case c;
  c.M = S::M; c.R = S::R;
  c.TN = TN; c.WM = WM; c.K = K; c.C = C; c.L = L; c.B = B; c.dil = dil; c.src_dil = src_dil; c.adain = adain;
  c.derive();
  const ToomD td = toom(S::M, S::R);
  const Toom tm = toom_f32(td);
  HostData h;
  make_data(c, h);
  Packed pk = pack_w3(td, h.hw, C, C, K, c.G);
  if (mode == 2) {
    std::vector<float> hy, hpart;
    std::vector<_Float16> hvs;
    host_act(c, h, tm, hvs);
    host_conv(c, h, tm, pk, hvs, hy, hpart);
    int bad = check_result(c, h, hy, hpart, "selftest (data flow)");
    if (TN == 1 || S::M == 3) bad |= host_twin_act<S>(c, h, tm, hvs);
    if (L <= 600) {  // thread-level twin of the conv kernel's index arithmetic (slow: small cases only)
      by_groups<S, TN>(c.G, [&](auto g_tag) { host_twin_conv<S, decltype(g_tag)::value, TN, WM>(c, h, tm, pk, hvs, hy, hpart); return 0; });
      bad |= check_result(c, h, hy, hpart, "selftest (thread-level twin)");
    }
    return bad;
  }
  float *x, *res, *y, *bias, *rsc, *part, *st, *ga, *be, *al;
  _Float16* wq;
  h8* vs;
  const size_t vs_slots = (size_t)c.VB * 2 * S::P * c.cg_tot * c.Lt;
  const size_t y_elems = (size_t)c.VB * C * c.pitch_q;
  CK(hipMalloc(&x, h.hxsrc.size() * 4)); CK(hipMalloc(&res, h.hres.size() * 4)); CK(hipMalloc(&y, y_elems * 4));
  CK(hipMalloc(&bias, (size_t)pk.co_pad * 4)); CK(hipMalloc(&rsc, (size_t)pk.co_pad * 4));
  CK(hipMalloc(&part, (size_t)c.VB * C * c.nblk * 8)); CK(hipMalloc(&wq, pk.q.size() * 2)); CK(hipMalloc(&vs, vs_slots * 16));
  CK(hipMalloc(&st, h.hst.size() * 4)); CK(hipMalloc(&ga, h.hga.size() * 4)); CK(hipMalloc(&be, h.hbe.size() * 4));
  CK(hipMalloc(&al, h.hal.size() * 4));
  CK(hipMemcpy(x, h.hxsrc.data(), h.hxsrc.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(res, h.hres.data(), h.hres.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemset(bias, 0, (size_t)pk.co_pad * 4));
  CK(hipMemcpy(bias, h.hb.data(), (size_t)C * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(rsc, pk.row_scale.data(), (size_t)pk.co_pad * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(wq, pk.q.data(), pk.q.size() * 2, hipMemcpyHostToDevice));
  CK(hipMemcpy(st, h.hst.data(), h.hst.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(ga, h.hga.data(), h.hga.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(be, h.hbe.data(), h.hbe.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(al, h.hal.data(), h.hal.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemset(vs, 0, vs_slots * 16));
  CK(hipMemset(y, 0, y_elems * 4));

  AArgs a;
  const int x_cs = src_dil > 1 ? c.pitch_s : c.pitch;
  a.x = x; a.x_bs = (int64_t)C * x_cs; a.x_cs = x_cs; a.C = C; a.L = L; a.pad = c.padq; a.pro = adain ? 1 : 0;
  a.stats = st; a.gamma = ga; a.beta = be; a.gb_bs = C; a.alpha = al; a.x_scale = 8.f;
  a.vs = vs; a.cg_tot = c.cg_tot; a.Lt = c.Lt; a.dil = dil; a.src_dil = src_dil; a.tm = tm;
  WArgs d;
  memset(&d, 0, sizeof(d));
  d.vs = vs; d.cg_tot = c.cg_tot; d.Lt = c.Lt;
  d.wq = reinterpret_cast<const h8*>(wq); d.co_pad = pk.co_pad; d.cin_pad = pk.cin_pad;
  d.row_scale = rsc; d.bias = bias; d.out_scale = 1.f / 8.f;
  d.y = y; d.y_bs = (int64_t)C * c.pitch_q; d.y_cs = c.pitch_q;
  if (dil == 1) { d.res = res; d.res_bs = (int64_t)C * c.pitch; d.res_cs = c.pitch; }  // the residual add sits after convs2
  d.part = part; d.part_nt = c.nblk;
  d.C_out = C; d.L_out = L; d.dil = dil; d.tm = tm;

  auto run_act = [&]() -> int {
    dim3 grid((c.Lt + ActGeom<S>::TILES - 1) / ActGeom<S>::TILES, c.cg_tot, c.VB);
    if (adain) hipLaunchKernelGGL((act_w3_kernel<S, 1>), grid, dim3(256), 0, 0, a);
    else hipLaunchKernelGGL((act_w3_kernel<S, 0>), grid, dim3(256), 0, 0, a);
    CK(hipGetLastError());
    return 0;
  };
  auto run_conv = [&]() -> int {
    return by_groups<S, TN>(c.G, [&](auto g_tag) { return launch_conv<S, decltype(g_tag)::value, TN, OCC, WM>(d, c.VB, c.nblk); });
  };
  if (run_act() || run_conv()) return 1;
  CK(hipDeviceSynchronize());

  if (mode == 1) {
    std::vector<float> hy(y_elems), hpart((size_t)c.VB * C * c.nblk * 2);
    CK(hipMemcpy(hy.data(), y, hy.size() * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(hpart.data(), part, hpart.size() * 4, hipMemcpyDeviceToHost));
    return check_result(c, h, hy, hpart, "check");
  }

  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  float ms_act = 0.f, ms_conv = 0.f;
  CK(hipEventRecord(e0, 0));
  for (int i = 0; i < reps; ++i)
    if (run_act()) return 1;
  CK(hipEventRecord(e1, 0));
  CK(hipEventSynchronize(e1));
  CK(hipEventElapsedTime(&ms_act, e0, e1));
  CK(hipEventRecord(e0, 0));
  for (int i = 0; i < reps; ++i)
    if (run_conv()) return 1;
  CK(hipEventRecord(e1, 0));
  CK(hipEventSynchronize(e1));
  CK(hipEventElapsedTime(&ms_conv, e0, e1));
  ms_act /= reps;
  ms_conv /= reps;
  const double flop = 2.0 * B * C * (double)C * K * L;
  printf("wino_bench F(%d,%d) TN=%d occ=%d WM=%d k=%d dil=%d src_dil=%d C=%d L=%d B=%d adain=%d: transform pass %.4f ms (%.2f TB/s of x read + "
         "planes written), conv %.4f ms = %.1f algorithmic TFLOP/s (%.3f of 833)\n", S::M, S::R, TN, OCC, WM, K, dil, src_dil, C, L,
         B, (int)adain, ms_act, ((double)B * C * L * 4 + (double)vs_slots * 16) / ms_act / 1e9, ms_conv, flop / ms_conv / 1e9,
         flop / ms_conv / 1e9 / (2500.0 / 3));
  return 0;
}

template <class S, int TN, int OCC, int WM>
static int check_all(int mode) {
  int bad = 0;
  bad |= run_case<S, TN, OCC, WM>(11, 1, 1, 128, 1000, 2, 1, mode, false);   // edge tiles along l
  bad |= run_case<S, TN, OCC, WM>(11, 1, 1, 128, 1152, 1, 1, mode, true);    // AdaIN + Snake prologue
  bad |= run_case<S, TN, OCC, WM>(7, 1, 1, 128, 777, 2, 1, mode, true);
  bad |= run_case<S, TN, OCC, WM>(7, 1, 1, 256, 389, 1, 1, mode, false);     // two co blocks
  bad |= run_case<S, TN, OCC, WM>(11, 1, 1, 96, 500, 1, 1, mode, false);     // C_out < 128: row guard
  bad |= run_case<S, TN, OCC, WM>(11, 3, 1, 128, 1000, 1, 1, mode, true);    // dilated (convs1): residue-major output, no residual
  bad |= run_case<S, TN, OCC, WM>(7, 5, 1, 128, 523, 2, 1, mode, false);
  bad |= run_case<S, TN, OCC, WM>(11, 1, 3, 128, 598, 1, 1, mode, true);     // convs2 behind a dilation-3 layer: residue-major input
  bad |= run_case<S, TN, OCC, WM>(7, 1, 5, 128, 1001, 1, 1, mode, true);
  return bad;
}

int main(int argc, char** argv) {
  if (argc > 2 && !strcmp(argv[1], "selftest") && !strcmp(argv[2], "quick")) {  // a few seconds: one small case per scheme
    int bad = toom_selfcheck(toom(3, 3)) | toom_selfcheck(toom(4, 4)) | toom_selfcheck(toom(6, 6));
    bad |= run_case<S66, 1, 2, 4>(11, 1, 1, 64, 260, 1, 1, 2, true);
    bad |= run_case<S33, 2, 2, 4>(11, 1, 1, 96, 200, 1, 1, 2, true);
    bad |= run_case<S44, 1, 2, 2>(7, 3, 1, 64, 230, 1, 1, 2, false);
    bad |= run_case<S44, 1, 2, 4>(11, 1, 5, 32, 150, 1, 1, 2, true);
    printf(bad ? "wino check: FAILED\n" : "wino check: all cases OK\n");
    return bad;
  }
  if (argc > 1 && (!strcmp(argv[1], "check") || !strcmp(argv[1], "selftest"))) {
    const int mode = !strcmp(argv[1], "check") ? 1 : 2;  // selftest: host emulation of the data flow, no GPU needed
    int bad = toom_selfcheck(toom(3, 3)) | toom_selfcheck(toom(4, 4)) | toom_selfcheck(toom(6, 6));
    bad |= check_all<S66, 1, 2, 4>(mode) | check_all<S66, 1, 2, 2>(mode);
    bad |= check_all<S33, 2, 2, 4>(mode) | check_all<S33, 1, 3, 4>(mode) | check_all<S44, 1, 2, 4>(mode) | check_all<S44, 1, 2, 2>(mode);
    if (mode == 1) bad |= check_all<S44, 1, 3, 4>(mode) | check_all<S44, 1, 3, 2>(mode) | check_all<S33, 1, 3, 2>(mode);  // other budgets / shapes: GPU only
    printf(bad ? "wino check: FAILED\n" : "wino check: all cases OK\n");
    return bad;
  }
  auto arg = [&](int i, int def) { return argc > i ? atoi(argv[i]) : def; };
  const int K = arg(1, 11), C = arg(2, 128), L = arg(3, 48001), B = arg(4, 32), reps = arg(5, 10), tn = arg(6, 2);
  const int dil = arg(7, 1), src_dil = arg(8, 1), scheme = arg(9, 33);
  if (K != 7 && K != 11) { fprintf(stderr, "k must be 7 or 11\n"); return 2; }
  const int occ = arg(10, 2), wm = arg(11, 4);
  if (scheme == 66)
    return wm == 2 ? run_case<S66, 1, 2, 2>(K, dil, src_dil, C, L, B, reps, 0, true) : run_case<S66, 1, 2, 4>(K, dil, src_dil, C, L, B, reps, 0, true);
  if (scheme == 44) {
    if (wm == 2) return occ == 3 ? run_case<S44, 1, 3, 2>(K, dil, src_dil, C, L, B, reps, 0, true) : run_case<S44, 1, 2, 2>(K, dil, src_dil, C, L, B, reps, 0, true);
    return occ == 3 ? run_case<S44, 1, 3, 4>(K, dil, src_dil, C, L, B, reps, 0, true) : run_case<S44, 1, 2, 4>(K, dil, src_dil, C, L, B, reps, 0, true);
  }
  if (tn == 1) return wm == 2 ? run_case<S33, 1, 3, 2>(K, dil, src_dil, C, L, B, reps, 0, true) : run_case<S33, 1, 3, 4>(K, dil, src_dil, C, L, B, reps, 0, true);
  return run_case<S33, 2, 2, 4>(K, dil, src_dil, C, L, B, reps, 0, true);
}
