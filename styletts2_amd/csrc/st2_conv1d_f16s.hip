// C entry points of the f16s conv family; the kernels live in st2_conv1d_f16s_impl.h and are instantiated in
// st2_conv1d_f16s_k{0,1,2}.hip.
#include "st2_conv1d_f16s_impl.h"

extern template int st2f16s::launch_by_cout<1, 32>(const st2_conv_desc&, hipStream_t);
extern template int st2f16s::launch_by_cout<2, 32>(const st2_conv_desc&, hipStream_t);
extern template int st2f16s::launch_by_cout<3, 32>(const st2_conv_desc&, hipStream_t);
extern template int st2f16s::launch_by_cout<5, 16>(const st2_conv_desc&, hipStream_t);
extern template int st2f16s::launch_by_cout<7, 16>(const st2_conv_desc&, hipStream_t);
extern template int st2f16s::launch_by_cout<11, 16>(const st2_conv_desc&, hipStream_t);


int st2f16s::g_variant = 0;
int st2_headroom_of_fused_conv(const st2_conv_desc& d, hipStream_t s);  // st2_actsplit.hip: no-op unless st2_debug_headroom(1)

extern "C" void st2_conv1d_f16s_set_variant(int v) { st2f16s::g_variant = (v == 1 || v == 2) ? v : 0; }
int st2f16s::g_splitk_max = 8, st2f16s::g_splitk_min_chunks = 4;
extern "C" void st2_conv1d_f16s_set_splitk(int max_slices, int min_chunks) {
  st2f16s::g_splitk_max = max_slices >= 1 && max_slices <= 32 ? max_slices : 8;
  st2f16s::g_splitk_min_chunks = min_chunks >= 1 && min_chunks <= 16 ? min_chunks : 4;
}

extern "C" int st2_conv1d_f16s_chunk(int ks) { return ks <= 3 ? 32 : 16; }

extern "C" int st2_conv1d_f16s_co_block(int C_out) { return C_out > 64 ? 128 : (C_out > 32 ? 64 : 32); }

extern "C" int64_t st2_conv1d_f16s_splitk_bytes(const st2_conv_desc* dp) {
  if (!dp || dp->B <= 0 || dp->C_in <= 0 || dp->C_out <= 0 || dp->L_out <= 0) return 0;
  const int s = ksplit_for_geometry(*dp);
  return s > 1 ? splitk_bytes_for(*dp, s) : 0;
}

extern "C" int st2_conv1d_f16s(const st2_conv_desc* dp, void* stream) {
  ST2_REQUIRE(dp != nullptr, "st2_conv1d_f16s: null descriptor");
  const st2_conv_desc& d = *dp;
  ST2_REQUIRE(d.B > 0 && d.C_in > 0 && d.C_out > 0 && d.L_in > 0 && d.L_out > 0,
              "st2_conv1d_f16s: empty geometry B=%d C_in=%d C_out=%d L_in=%d L_out=%d", d.B, d.C_in, d.C_out,
              d.L_in, d.L_out);
  ST2_REQUIRE(d.x && d.wq && d.y, "st2_conv1d_f16s: null tensor pointer");
  ST2_REQUIRE((reinterpret_cast<uintptr_t>(d.wq) & 15) == 0, "st2_conv1d_f16s: wq must be 16-byte aligned");
  ST2_REQUIRE(d.dil >= 1 && d.dil <= 8, "st2_conv1d_f16s: dil=%d out of range", d.dil);
  ST2_REQUIRE(d.pro >= ST2_PRO_NONE && d.pro <= ST2_PRO_COLNORM, "st2_conv1d_f16s: prologue %d not supported", d.pro);
  if (d.pro == ST2_PRO_ADAIN_LEAKY || d.pro == ST2_PRO_ADAIN_SNAKE || d.pro == ST2_PRO_COLNORM)
    ST2_REQUIRE(d.stats && d.gamma && d.beta, "st2_conv1d_f16s: prologue %d needs stats/gamma/beta", d.pro);
  if (d.pro == ST2_PRO_ADAIN_SNAKE || d.pro == ST2_PRO_SNAKE)
    ST2_REQUIRE(d.alpha, "st2_conv1d_f16s: snake prologue needs alpha");
  ST2_REQUIRE(d.res_shift >= 0 && d.res_shift <= 1, "st2_conv1d_f16s: res_shift must be 0 or 1");
  ST2_REQUIRE(d.x_scale > 0.f && d.out_scale > 0.f, "st2_conv1d_f16s: x_scale / out_scale must be set");
  ST2_REQUIRE(d.B <= 65535, "st2_conv1d_f16s: grid too large");
  ST2_REQUIRE((int64_t)d.C_out * d.y_cs < (1ll << 31) && (!d.res || (int64_t)d.C_out * d.res_cs < (1ll << 31)) &&
                  (!d.res2 || (int64_t)d.C_out * d.res2_cs < (1ll << 31)),
              "st2_conv1d_f16s: a batch item of y / res / res2 must span < 2^31 elements");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  if (st2_headroom_of_fused_conv(d, s) != 0) return 1;
  switch (d.ks) {
    case 1:
      return st2f16s::launch_by_cout<1, 32>(d, s);
    case 2:
      return st2f16s::launch_by_cout<2, 32>(d, s);
    case 3:
      return st2f16s::launch_by_cout<3, 32>(d, s);
    case 5:
      return st2f16s::launch_by_cout<5, 16>(d, s);
    case 7:
      return st2f16s::launch_by_cout<7, 16>(d, s);
    case 11:
      return st2f16s::launch_by_cout<11, 16>(d, s);
    default:
      st2_set_error("st2_conv1d_f16s: unsupported kernel size %d (have 1,2,3,5,7,11)", d.ks);
      return 1;
  }
}
