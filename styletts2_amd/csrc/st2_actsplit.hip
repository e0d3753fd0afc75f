// Activation pass of the "xs" conv path and the InstanceNorm statistics finaliser.
//
//   st2_act_split: xs[b][plane][cg][pos][e] = split_f16( x_scale * pro(x)[b][cg*8+e][pos-halo] )
//
// One pass over the tensor: every element is read once (fp32, 4 B) and written once (f16 hi + f16 lo, 4 B), so the
// kernel is HBM-bound: 8 B per element.  A thread owns one position and the 8 channels of one 16-byte slot: its 8
// loads are each coalesced across the wave (lanes run along l), its two 16-byte stores are contiguous 1 KB per wave.
// The per-channel parameters (mean, rstd, gamma, beta, alpha) are wave-uniform and come through the scalar cache.
// Zero padding of the conv (halo columns, the tail up to Lp, channel padding) is materialised here, AFTER the
// activation as F.conv1d pads the activated tensor, so the MFMA kernel never tests a boundary.
// Arithmetic is op for op that of the fused prologue in st2_conv1d_f16s.hip (same helpers, st2_act.h).
#include "st2_common.h"
#include "st2_act.h"
#include <algorithm>
#include <cstring>
#include <type_traits>
#include <vector>

namespace {

struct ActArgs {
  const float* x; int64_t x_bs; int x_cs;
  int C, L;
  float slope;
  const float* stats; const float* gamma; const float* beta; int64_t gb_bs; int gb_seg; int gamma_plus_one;
  const float* alpha;
  float x_scale;
  st2_h8* xs; int xs_cg, Lp, halo;
  int* status;  // sticky status word (st2_status), may be null
};

#ifndef ST2_ACT_NT
#define ST2_ACT_NT 3  // bit 0 = nontemporal loads of x, bit 1 = nontemporal stores of the planes
#endif
template <int PRO>
__global__ __launch_bounds__(256) void act_split_kernel(const ActArgs a) {
  const int pos = blockIdx.x * 256 + threadIdx.x;
  if (pos >= a.Lp) return;
  const int cg = blockIdx.y;
  const int b = blockIdx.z;
  const int l = pos - a.halo;
  const bool lok = l >= 0 && l < a.L;
  const int lc = min(max(l, 0), a.L - 1);
  const float* xb = a.x + (int64_t)b * a.x_bs + lc;
  float v[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int ci = min(cg * 8 + e, a.C - 1);
    // read once, written once: nontemporal hints on both sides (-7 % at C = 128 / L = 48 001, -15 % at C = 256 / L = 8 000
    // alone, -0.35 ms per bench step; profiles/r04/r04aa_nt_*.log).  The conv's own stores and staging loads LOSE with the
    // same hint (+10 % / +2 %): its output is re-read as the next layer's residual and its tiles overlap in the halo.
#if ST2_ACT_NT & 1
    v[e] = __builtin_nontemporal_load(&xb[(int64_t)ci * a.x_cs]);
#else
    v[e] = xb[(int64_t)ci * a.x_cs];
#endif
  }
  float cmean = 0.f, crstd = 1.f;
  if constexpr (PRO == ST2_PRO_COLNORM) {
    const float* st = a.stats + ((int64_t)b * a.L + lc) * 2;
    cmean = st[0];
    crstd = st[1];
  }
  st2_h8 hi, lo;
  bool sat = false;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int ci = cg * 8 + e;
    const int cc = min(ci, a.C - 1);  // wave-uniform -> scalar loads
    float u = v[e];
    if constexpr (PRO == ST2_PRO_LEAKY) {
      u = leaky(u, a.slope);
    } else if constexpr (PRO == ST2_PRO_ADAIN_LEAKY || PRO == ST2_PRO_ADAIN_SNAKE) {
      const float* st = a.stats + ((int64_t)b * a.C + cc) * 2;
      const float g = 1.0f + a.gamma[(int64_t)b * a.gb_bs + cc];
      const float bt = a.beta[(int64_t)b * a.gb_bs + cc];
      float w = (u - st[0]) * st[1];
      w = g * w + bt;
      if constexpr (PRO == ST2_PRO_ADAIN_LEAKY) {
        u = leaky(w, a.slope);
      } else {
        const float al = a.alpha[cc];
        u = snake(w, al, 1.0f / al);
      }
    } else if constexpr (PRO == ST2_PRO_SNAKE) {
      const float al = a.alpha[cc];
      u = snake(u, al, 1.0f / al);
    } else if constexpr (PRO == ST2_PRO_COLNORM) {
      // affine row: the batch item (wave-uniform, scalar loads) or -- token-merged view of several utterances -- the
      // utterance this position belongs to (per lane; the rows are a few KB and L2 / L1 resident)
      const int64_t row = a.gb_seg > 0 ? (int64_t)(lc / a.gb_seg) : (int64_t)b;
      const float g0 = a.gamma[row * a.gb_bs + cc];
      const float g = a.gamma_plus_one ? 1.0f + g0 : g0;
      const float bt = a.beta[row * a.gb_bs + cc];
      const float w = (u - cmean) * crstd;
      u = w * g + bt;
    }
    u = (lok && ci < a.C) ? u * a.x_scale : 0.f;
    // saturate to the f16 range instead of overflowing to inf (hi) / NaN (lo = u - inf): the conv then stays finite
    // and the condition is reported through the sticky status word (so is a NaN operand, which becomes -65504)
    const float uc = st2_clamp_f16(u);
    sat |= uc != u;
    const _Float16 h = (_Float16)uc;
    hi[e] = h;
    lo[e] = (_Float16)(uc - (float)h);
  }
  if (__any(sat) && (threadIdx.x & 63) == 0) st2_raise_status(a.status, ST2_STATUS_F16_RANGE);
  const int64_t plane = (int64_t)a.xs_cg * a.Lp;
  st2_h8* dst = a.xs + ((int64_t)b * 2 * a.xs_cg + cg) * a.Lp + pos;
#if ST2_ACT_NT & 2
  __builtin_nontemporal_store(hi, &dst[0]);
  __builtin_nontemporal_store(lo, &dst[plane]);
#else
  dst[0] = hi;
  dst[plane] = lo;
#endif
}

// stats[row] = (mean, rstd) from per-slot partial sums; one wave per row, fp64, fixed order.  Slot i covers columns [i * cols,
// min((i + 1) * cols, L)) of row (b, c) and holds (sum, sum of squares) of (y - shift_i); the shifts (each slot's first stored value,
// what its producer subtracted) follow the sums in the same buffer: part = float2 [rows][nt], then float [rows][nt].  Slot mean =
// shift + s1 / n, slot M2 = s2 - s1^2 / n; the row's mean is the weighted mean of the slot means and its M2 = sum of slot M2 + sum
// n_i (mean_i - mean)^2 (Chan et al.) -- two wave reductions over coalesced reads.
__global__ __launch_bounds__(256) void stats_finalize_kernel(const float* __restrict__ part, int rows, int nt, int L,
                                                             float eps, float* __restrict__ stats, int cols) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float2* p = reinterpret_cast<const float2*>(part) + (int64_t)row * nt;
  const float* sh = part + (int64_t)rows * nt * 2 + (int64_t)row * nt;
  const int ns = min(nt, (L + cols - 1) / cols);  // slots that hold columns
  double wsum = 0.0;
  for (int i = lane; i < ns; i += 64) {
    const int n = min(cols, L - i * cols);
    wsum += (double)n * (double)sh[i] + (double)p[i].x;  // n_i * mean_i
  }
  wsum = st2_wave_sum(wsum);
  const double mean = __shfl(wsum, 0, 64) / (double)L;
  double m2 = 0.0;
  for (int i = lane; i < ns; i += 64) {
    const int n = min(cols, L - i * cols);
    const double s1 = (double)p[i].x, s2 = (double)p[i].y;
    const double mi = (double)sh[i] + s1 / (double)n;
    m2 += (s2 - s1 * s1 / (double)n) + (double)n * (mi - mean) * (mi - mean);
  }
  m2 = st2_wave_sum(m2);
  if (lane == 0) {
    double var = m2 / (double)L;  // biased, InstanceNorm1d
    if (var < 0.0) var = 0.0;
    stats[(int64_t)row * 2 + 0] = (float)mean;
    stats[(int64_t)row * 2 + 1] = (float)(1.0 / sqrt(var + (double)eps));
  }
}

// ---- operand-range telemetry (st2_debug_headroom, st2.h) --------------------------------------------------------------
// Both ends of the f16 range, from the planes one act_split launch wrote (u = hi + lo is the scaled operand the conv multiplies):
//   max |hi|                      the top: how close the layer comes to the clamp at 65504;
//   sum u^2                       operand energy;
//   sum ulp(lo)^2 / 12            energy of the split's rounding error: lo carries 11 bits while it is a normal f16 (|lo| >= 2^-14:
//                                 ulp = 2^(e - 10)), a fixed 2^-24 below that -- an operand below ~2^-3 keeps an ABSOLUTE error of
//                                 2^-25 instead of 2^-22 relative.  sqrt(this / sum u^2) is the relative RMS error the split adds to
//                                 the operand (fp32 storage itself: 2^-24 / sqrt(3) = 3.4e-8);
//   sum u^2 over |lo| < 2^-14     the share of the operand's energy carried by elements whose lo half is subnormal or zero.
// One record per launch: an atomicMax on the bit pattern of the non-negative float, three double atomicAdds per wave.
struct HeadroomSums {
  unsigned max_bits, pad;
  double su2, se2, ssub;
};
__global__ __launch_bounds__(256) void xs_probe_kernel(const st2_h8* __restrict__ xs, int64_t slots_per_item, int64_t item_stride,
                                                       HeadroomSums* out) {
  const st2_h8* hi_p = xs + (int64_t)blockIdx.y * item_stride;  // the hi plane of batch item blockIdx.y; lo follows it
  const st2_h8* lo_p = hi_p + slots_per_item;
  float m = 0.f;
  double su2 = 0.0, se2 = 0.0, ssub = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < slots_per_item; i += (int64_t)gridDim.x * 256) {
    const st2_h8 h = hi_p[i], l = lo_p[i];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float hf = (float)h[e], lf = (float)l[e];
      m = fmaxf(m, fabsf(hf));
      if (hf == 0.f && lf == 0.f) continue;  // padding / exact zeros: no energy, no error
      const float u = hf + lf;
      const float al = fabsf(lf);
      const bool sub = al < 6.103515625e-05f;  // 2^-14
      int ex = -14;
      if (!sub) {
        (void)frexpf(al, &ex);  // al = f * 2^ex, f in [0.5, 1)
        ex -= 1;
      }
      const float ulp = ldexpf(1.0f, ex - 10);
      su2 += (double)u * u;
      se2 += (double)ulp * ulp * (1.0 / 12.0);
      if (sub) ssub += (double)u * u;
    }
  }
  for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_down(m, off, 64));
  su2 = st2_wave_sum(su2);
  se2 = st2_wave_sum(se2);
  ssub = st2_wave_sum(ssub);
  if ((threadIdx.x & 63) == 0) {
    atomicMax(&out->max_bits, __float_as_uint(m));
    atomicAdd(&out->su2, su2);
    atomicAdd(&out->se2, se2);
    atomicAdd(&out->ssub, ssub);
  }
}

struct HeadroomRecord {
  int kind, pro, B, C, L;  // kind 0 = st2_act_split (xs path), 1 = st2_conv1d_f16s (prologue inside the conv)
  float x_scale;
  const void* engine;      // the st2_engine whose plan issued the launch and its conv site (st2_headroom_set_site), or null / -1
  int site;
};
bool g_headroom = false;
std::vector<HeadroomRecord> g_hr;
HeadroomSums* g_hr_dev = nullptr;  // one per record
constexpr int HR_CAP = 4096;
void* g_hr_scratch = nullptr;  // planes of the fused-path convs (debug mode only)
size_t g_hr_scratch_bytes = 0;
thread_local const void* t_site_engine = nullptr;  // set by the engine's conv() around its launches (same thread)
thread_local int t_site = -1;

void headroom_record(int kind, const ActArgs& a, int B, hipStream_t s) {
  if ((int)g_hr.size() >= HR_CAP || !g_hr_dev) return;
  // never under stream capture: a recorded probe would re-run at every replay -- and, for the fused convs, read scratch
  // planes that st2_debug_headroom(0) has freed.  Calibrate / report on eager calls, record graphs afterwards.
  hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(s, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone) {
    (void)hipGetLastError();
    return;
  }
  const int64_t plane = (int64_t)a.xs_cg * a.Lp;
  hipLaunchKernelGGL(xs_probe_kernel, dim3((unsigned)std::min<int64_t>((plane + 255) / 256, 1024), B), dim3(256), 0, s, a.xs, plane,
                     2 * plane, g_hr_dev + g_hr.size());
  g_hr.push_back({kind, 0, B, a.C, a.L, a.x_scale, t_site_engine, t_site});
}

template <int PRO>
void launch_act(const ActArgs& a, int B, hipStream_t s) {
  hipLaunchKernelGGL((act_split_kernel<PRO>), dim3(st2_cdiv(a.Lp, 256), a.xs_cg, B), dim3(256), 0, s, a);
}

}  // namespace

extern "C" int st2_act_split(const float* x, int64_t x_bs, int32_t x_cs, int32_t B, int32_t C, int32_t L, int32_t pro,
                             float slope, const float* stats, const float* gamma, const float* beta, int64_t gb_bs,
                             int32_t gb_seg, int32_t gamma_plus_one, const float* alpha, float x_scale, void* xs, int32_t xs_cg,
                             int32_t Lp, int32_t halo, void* stream) {
  ST2_REQUIRE(x && xs && B > 0 && C > 0 && L > 0, "st2_act_split: bad arguments");
  ST2_REQUIRE(B <= 65535 && xs_cg <= 65535, "st2_act_split: grid too large");
  ST2_REQUIRE(xs_cg * 8 >= C, "st2_act_split: xs_cg=%d groups cannot hold C=%d channels", xs_cg, C);
  ST2_REQUIRE(halo >= 0 && Lp >= L + halo, "st2_act_split: Lp=%d too small for L=%d + halo=%d", Lp, L, halo);
  ST2_REQUIRE((reinterpret_cast<uintptr_t>(xs) & 15) == 0, "st2_act_split: xs must be 16-byte aligned");
  ST2_REQUIRE(pro >= ST2_PRO_NONE && pro <= ST2_PRO_COLNORM, "st2_act_split: prologue %d not supported", pro);
  if (pro == ST2_PRO_ADAIN_LEAKY || pro == ST2_PRO_ADAIN_SNAKE || pro == ST2_PRO_COLNORM)
    ST2_REQUIRE(stats && gamma && beta, "st2_act_split: prologue %d needs stats/gamma/beta", pro);
  if (pro == ST2_PRO_ADAIN_SNAKE || pro == ST2_PRO_SNAKE) ST2_REQUIRE(alpha, "st2_act_split: snake needs alpha");
  ST2_REQUIRE(x_scale > 0.f, "st2_act_split: x_scale must be set");
  ST2_REQUIRE(gb_seg >= 0 && (gb_seg == 0 || pro == ST2_PRO_COLNORM), "st2_act_split: gb_seg=%d is for ST2_PRO_COLNORM", gb_seg);
  ActArgs a;
  a.x = x; a.x_bs = x_bs; a.x_cs = x_cs; a.C = C; a.L = L; a.slope = slope;
  a.stats = stats; a.gamma = gamma; a.beta = beta; a.gb_bs = gb_bs; a.gb_seg = gb_seg; a.gamma_plus_one = gamma_plus_one;
  a.alpha = alpha; a.x_scale = x_scale;
  a.xs = reinterpret_cast<st2_h8*>(xs); a.xs_cg = xs_cg; a.Lp = Lp; a.halo = halo;
  a.status = st2_status_device_ptr();
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  switch (pro) {
    case ST2_PRO_LEAKY: launch_act<ST2_PRO_LEAKY>(a, B, s); break;
    case ST2_PRO_ADAIN_LEAKY: launch_act<ST2_PRO_ADAIN_LEAKY>(a, B, s); break;
    case ST2_PRO_ADAIN_SNAKE: launch_act<ST2_PRO_ADAIN_SNAKE>(a, B, s); break;
    case ST2_PRO_SNAKE: launch_act<ST2_PRO_SNAKE>(a, B, s); break;
    case ST2_PRO_COLNORM: launch_act<ST2_PRO_COLNORM>(a, B, s); break;
    default: launch_act<ST2_PRO_NONE>(a, B, s); break;
  }
  ST2_CHECK_LAUNCH("st2_act_split");
  if (g_headroom) {
    headroom_record(0, a, B, s);
    if (!g_hr.empty()) g_hr.back().pro = pro;
  }
  return 0;
}

// Telemetry for the fused conv (st2_conv1d_f16s.hip calls this in debug mode): the same prologue into scratch planes.
int st2_headroom_of_fused_conv(const st2_conv_desc& d, hipStream_t s) {
  if (!g_headroom || !d.x) return 0;
  {
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;  // see headroom_record: no probes (no hipMalloc either) under capture
    if (hipStreamIsCapturing(s, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone) {
      (void)hipGetLastError();
      return 0;
    }
  }
  const int cg = (d.C_in + 31) / 32 * 4, halo = 0;
  const int Lp = (d.L_in + 7) / 8 * 8;
  const size_t bytes = (size_t)d.B * 2 * cg * Lp * 16;
  if (bytes > g_hr_scratch_bytes) {
    if (g_hr_scratch) (void)hipFree(g_hr_scratch);
    g_hr_scratch = nullptr;
    g_hr_scratch_bytes = 0;
    if (hipMalloc(&g_hr_scratch, bytes) != hipSuccess) {
      (void)hipGetLastError();
      return 0;
    }
    g_hr_scratch_bytes = bytes;
  }
  const size_t before = g_hr.size();
  const int rc = st2_act_split(d.x, d.x_bs, d.x_cs, d.B, d.C_in, d.L_in, d.pro, d.slope, d.stats, d.gamma, d.beta, d.gb_bs, 0,
                               d.gamma_plus_one, d.alpha, d.x_scale, g_hr_scratch, cg, Lp, halo, s);
  if (rc == 0 && g_hr.size() == before + 1) g_hr.back().kind = 1;
  return rc;
}

extern "C" int st2_debug_headroom(int enable) {
  if (enable) {
    g_hr.clear();
    if (!g_hr_dev && hipMalloc(&g_hr_dev, HR_CAP * sizeof(HeadroomSums)) != hipSuccess) {
      (void)hipGetLastError();
      g_hr_dev = nullptr;
      st2_set_error("st2_debug_headroom: cannot allocate the record buffer");
      return 1;
    }
    if (hipMemset(g_hr_dev, 0, HR_CAP * sizeof(HeadroomSums)) != hipSuccess) {
      st2_set_error("st2_debug_headroom: %s", hipGetErrorString(hipGetLastError()));
      return 1;
    }
  } else if (g_hr_scratch) {
    (void)hipDeviceSynchronize();
    (void)hipFree(g_hr_scratch);
    g_hr_scratch = nullptr;
    g_hr_scratch_bytes = 0;
  }
  g_headroom = enable != 0;
  return 0;
}

void st2_headroom_set_site(const void* engine, int site) {
  t_site_engine = engine;
  t_site = site;
}

extern "C" int st2_debug_headroom_read(double* rows, int32_t cap_rows) {
  ST2_REQUIRE(!g_headroom, "st2_debug_headroom_read: stop the recording first (st2_debug_headroom(0))");
  const int n = (int)g_hr.size();
  if (!rows || n == 0) return n;
  std::vector<HeadroomSums> h(n);
  if (hipDeviceSynchronize() != hipSuccess ||
      hipMemcpy(h.data(), g_hr_dev, n * sizeof(HeadroomSums), hipMemcpyDeviceToHost) != hipSuccess) {
    st2_set_error("st2_debug_headroom_read: %s", hipGetErrorString(hipGetLastError()));
    return -1;
  }
  for (int i = 0; i < n && i < cap_rows; ++i) {
    float m;
    memcpy(&m, &h[i].max_bits, 4);
    double* r = rows + (int64_t)i * ST2_HEADROOM_COLS;
    r[0] = g_hr[i].kind; r[1] = g_hr[i].pro; r[2] = g_hr[i].B; r[3] = g_hr[i].C; r[4] = g_hr[i].L; r[5] = g_hr[i].x_scale;
    r[6] = m; r[7] = m / 65504.0;
    r[8] = h[i].su2 > 0.0 ? sqrt(h[i].se2 / h[i].su2) : 0.0;
    r[9] = h[i].su2 > 0.0 ? h[i].ssub / h[i].su2 : 0.0;
    r[10] = g_hr[i].site;
    r[11] = (double)(uintptr_t)g_hr[i].engine;  // identity only (an address fits a double's 53 bits on this platform)
  }
  return n;
}

extern "C" int st2_stats_finalize(const float* part, int32_t rows, int32_t nt, int32_t L, float eps, float* stats, int32_t cols,
                                  void* stream) {
  ST2_REQUIRE(part && stats && rows > 0 && nt > 0 && L > 0, "st2_stats_finalize: bad arguments");
  ST2_REQUIRE(cols > 0 && (int64_t)nt * cols >= L, "st2_stats_finalize: %d slots of %d columns do not cover L=%d", nt, cols, L);
  ST2_REQUIRE((reinterpret_cast<uintptr_t>(part) & 7) == 0, "st2_stats_finalize: part must be 8-byte aligned");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(stats_finalize_kernel, dim3(st2_cdiv(rows, 4)), dim3(256), 0, s, part, rows, nt, L, eps, stats, cols);
  ST2_CHECK_LAUNCH("st2_stats_finalize");
  return 0;
}
