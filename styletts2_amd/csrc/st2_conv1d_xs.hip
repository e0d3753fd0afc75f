// C entry points of the xs conv family; the kernels live in st2_conv1d_xs_impl.h and are instantiated in
// st2_conv1d_xs_k{0,1,2}.hip.
#include "st2_conv1d_xs_impl.h"

#include <vector>

extern template int st2xs::launch_by_cout<1, 32>(const st2_conv_desc&, hipStream_t);
extern template int st2xs::launch_by_cout<2, 32>(const st2_conv_desc&, hipStream_t);
extern template int st2xs::launch_by_cout<3, 32>(const st2_conv_desc&, hipStream_t);
extern template int st2xs::launch_by_cout<5, 16>(const st2_conv_desc&, hipStream_t);
extern template int st2xs::launch_by_cout<7, 16>(const st2_conv_desc&, hipStream_t);
extern template int st2xs::launch_by_cout<11, 16>(const st2_conv_desc&, hipStream_t);


// Measurement hook (bench.py's roofline leg): HIP events around every st2_conv1d_xs launch, on the launch stream,
// whoever issues it (the C++ plans of st2_engine.hip or the Python per-kernel plans).  Not thread safe, not legal under
// stream capture: enabled around the bench's timed region only.
namespace {
struct TimedLaunch {
  int ks, c_in, c_out, L, B;
  hipEvent_t e0, e1;
};
std::vector<TimedLaunch> g_timed;
bool g_timing = false;
int launch_xs(const st2_conv_desc& d, hipStream_t s);
}  // namespace

extern "C" int st2_conv_timing(int enable) {
  if (enable) {
    for (auto& t : g_timed) {
      (void)hipEventDestroy(t.e0);
      (void)hipEventDestroy(t.e1);
    }
    g_timed.clear();
  }
  g_timing = enable != 0;
  return 0;
}

extern "C" int st2_conv_timing_read(double* rows, int32_t cap_rows) {
  ST2_REQUIRE(!g_timing, "st2_conv_timing_read: stop the recording first (st2_conv_timing(0))");
  int n = 0;
  for (auto& t : g_timed) {
    float ms = 0.f;
    if (hipEventSynchronize(t.e1) != hipSuccess || hipEventElapsedTime(&ms, t.e0, t.e1) != hipSuccess) {
      st2_set_error("st2_conv_timing_read: %s", hipGetErrorString(hipGetLastError()));
      return -1;
    }
    if (rows && n < cap_rows) {
      double* r = rows + (int64_t)n * 6;
      r[0] = t.ks; r[1] = t.c_in; r[2] = t.c_out; r[3] = t.L; r[4] = t.B; r[5] = ms;
    }
    ++n;
  }
  return n;
}

extern "C" int st2_conv1d_xs(const st2_conv_desc* dp, void* stream) {
  ST2_REQUIRE(dp != nullptr, "st2_conv1d_xs: null descriptor");
  const st2_conv_desc& d = *dp;
  ST2_REQUIRE(d.B > 0 && d.C_in > 0 && d.C_out > 0 && d.L_out > 0,
              "st2_conv1d_xs: empty geometry B=%d C_in=%d C_out=%d L_out=%d", d.B, d.C_in, d.C_out, d.L_out);
  ST2_REQUIRE(d.xs && d.wq && d.y, "st2_conv1d_xs: null tensor pointer");
  ST2_REQUIRE((reinterpret_cast<uintptr_t>(d.wq) & 15) == 0 && (reinterpret_cast<uintptr_t>(d.xs) & 15) == 0,
              "st2_conv1d_xs: wq and xs must be 16-byte aligned");
  ST2_REQUIRE(d.dil >= 1 && d.dil <= 8, "st2_conv1d_xs: dil=%d out of range", d.dil);
  ST2_REQUIRE(d.pad_left >= 0 && d.pad_left <= d.xs_halo, "st2_conv1d_xs: pad_left=%d exceeds the xs halo %d",
              d.pad_left, d.xs_halo);
  ST2_REQUIRE(d.res_shift >= 0 && d.res_shift <= 1, "st2_conv1d_xs: res_shift must be 0 or 1");
  ST2_REQUIRE(d.out_scale > 0.f, "st2_conv1d_xs: out_scale must be set");
  ST2_REQUIRE(d.B <= 65535, "st2_conv1d_xs: grid too large");
  ST2_REQUIRE((int64_t)d.C_out * d.y_cs < (1ll << 31) && (!d.res || (int64_t)d.C_out * d.res_cs < (1ll << 31)) &&
                  (!d.res2 || (int64_t)d.C_out * d.res2_cs < (1ll << 31)),
              "st2_conv1d_xs: a batch item of y / res / res2 must span < 2^31 elements");
  if (d.part) ST2_REQUIRE((reinterpret_cast<uintptr_t>(d.part) & 7) == 0, "st2_conv1d_xs: part must be 8-byte aligned");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  if (g_timing && d.C_in >= 64 && d.L_out >= 256 && g_timed.size() < 65536) {
    TimedLaunch t{d.ks, d.C_in, d.C_out, d.L_out, d.B, nullptr, nullptr};
    if (hipEventCreate(&t.e0) == hipSuccess && hipEventCreate(&t.e1) == hipSuccess) {
      (void)hipEventRecord(t.e0, s);
      const int rc = launch_xs(d, s);
      (void)hipEventRecord(t.e1, s);
      g_timed.push_back(t);
      return rc;
    }
  }
  return launch_xs(d, s);
}

namespace {
int launch_xs(const st2_conv_desc& d, hipStream_t s) {
  switch (d.ks) {
    case 1:
      return st2xs::launch_by_cout<1, 32>(d, s);
    case 2:
      return st2xs::launch_by_cout<2, 32>(d, s);
    case 3:
      return st2xs::launch_by_cout<3, 32>(d, s);
    case 5:
      return st2xs::launch_by_cout<5, 16>(d, s);
    case 7:
      return st2xs::launch_by_cout<7, 16>(d, s);
    case 11:
      return st2xs::launch_by_cout<11, 16>(d, s);
    default:
      st2_set_error("st2_conv1d_xs: unsupported kernel size %d (have 1,2,3,5,7,11)", d.ks);
      return 1;
  }
}
}  // namespace
