// Fused Conv1d as an implicit GEMM on the CDNA4 fp32 matrix pipe.
//
//   y[b,co,l] = epi( bias[co] + sum_{ci,t} W[co,ci,t] * pro(x)[b,ci, l + t*dil - pad_left] )
//
// GEMM view per batch item:  M = C_out (rows), N = L_out (cols), K = C_in * ks.
// One workgroup (4 waves) owns a 128(co) x 128(l) output tile; each wave owns 64 x 64 as a
// 2 x 2 grid of v_mfma_f32_32x32x2_f32 tiles (64 accumulator VGPRs).  The K loop walks the
// input channels in chunks of CI_T: the activated input tile  [CI_T][128 + (ks-1)*dil]  and the
// weight slab [CI_T*ks][128] are staged in LDS, then every (channel pair, tap) issues 4 MFMAs.
//
// MFMA operand maps (cdna_hip_programming.md section 3): lane l supplies A[i = l&31][k = l>>5] and
// B[k = l>>5][j = l&31]; the accumulator register r of lane l is D[i = (r&3) + 8*(r>>2) + 4*(l>>5)]
// [j = l&31].  Here i = output channel, j = output position, k = one of two adjacent input
// channels at a fixed tap, so consecutive lanes touch consecutive LDS words (conflict free) and the
// epilogue stores 128 contiguous bytes per half wave.
//
// The fp32 MFMA is a k-ordered fmaf chain (bitwise), so results are exact-fp32 class: this is what
// keeps the vocoder inside the 1e-4 waveform tolerance while running on the matrix pipe.
#include "st2_common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {

constexpr int BM = 128;  // output channels per workgroup
constexpr int BN = 128;  // output positions per workgroup
constexpr int NT = 256;  // threads per workgroup

struct RowPar {  // per input-channel prologue parameters
  float mean, rstd, g, beta, alpha, inv_alpha;
};

__device__ __forceinline__ float leaky(float v, float slope) { return v >= 0.f ? v : v * slope; }

__device__ __forceinline__ float snake(float v, float alpha, float inv_alpha) {
  // x + (1/a) * sin(a*x)^2, op order of Modules/istftnet.py:69
  float s = sinf(alpha * v);
  return v + inv_alpha * (s * s);
}

__device__ __forceinline__ float gelu_erf(float v) {
  return 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f));
}

template <int KS, int CI_T>
__global__ __launch_bounds__(NT) void conv1d_mfma_kernel(const st2_conv_desc d) {
  extern __shared__ __attribute__((aligned(16))) float smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int half = lane >> 5;
  const int l31 = lane & 31;
  const int wm = wave >> 1;
  const int wn = wave & 1;
  const int n0 = blockIdx.x * BN;
  const int m0 = blockIdx.y * BM;
  const int b = blockIdx.z;

  const int XW = BN + (KS - 1) * d.dil;  // staged input width
  const int XS = (XW + 3) & ~3;          // LDS row stride (floats)
  float* xs = smem;
  float* ws = smem + CI_T * XS;

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const float* xb = d.x + (int64_t)b * d.x_bs;
  const int lin0 = n0 - d.pad_left;  // input position held by LDS column 0
  const int pro = d.pro;

  constexpr int TPR = NT / CI_T;  // loader threads per input row
  const int xr = tid / TPR;
  const int xc0 = tid % TPR;
  const int KK = d.C_in * KS;

  for (int c0 = 0; c0 < d.C_in; c0 += CI_T) {
    // ---- stage the activated input rows --------------------------------------------------
    {
      const int ci = c0 + xr;
      const bool rowok = ci < d.C_in;
      RowPar p = {0.f, 1.f, 1.f, 0.f, 1.f, 1.f};
      if (rowok) {
        if (pro == ST2_PRO_ADAIN_LEAKY || pro == ST2_PRO_ADAIN_SNAKE) {
          const float* st = d.stats + ((int64_t)b * d.C_in + ci) * 2;
          p.mean = st[0];
          p.rstd = st[1];
          p.g = 1.0f + d.gamma[(int64_t)b * d.gb_bs + ci];
          p.beta = d.beta[(int64_t)b * d.gb_bs + ci];
        } else if (pro == ST2_PRO_COLNORM) {
          float g = d.gamma[(int64_t)b * d.gb_bs + ci];
          p.g = d.gamma_plus_one ? 1.0f + g : g;
          p.beta = d.beta[(int64_t)b * d.gb_bs + ci];
        }
        if (pro == ST2_PRO_ADAIN_SNAKE || pro == ST2_PRO_SNAKE) {
          p.alpha = d.alpha[ci];
          p.inv_alpha = 1.0f / p.alpha;
        }
      }
      const float* xrow = xb + (int64_t)ci * d.x_cs;
      float* xsrow = xs + xr * XS;
      for (int c = xc0; c < XW; c += TPR) {
        const int l = lin0 + c;
        float v = 0.f;
        if (rowok && l >= 0 && l < d.L_in) {
          v = xrow[l];
          switch (pro) {
            case ST2_PRO_LEAKY:
              v = leaky(v, d.slope);
              break;
            case ST2_PRO_ADAIN_LEAKY: {
              float u = (v - p.mean) * p.rstd;
              u = p.g * u + p.beta;
              v = leaky(u, d.slope);
            } break;
            case ST2_PRO_ADAIN_SNAKE: {
              float u = (v - p.mean) * p.rstd;
              u = p.g * u + p.beta;
              v = snake(u, p.alpha, p.inv_alpha);
            } break;
            case ST2_PRO_SNAKE:
              v = snake(v, p.alpha, p.inv_alpha);
              break;
            case ST2_PRO_COLNORM: {
              const float* st = d.stats + ((int64_t)b * d.L_in + l) * 2;
              float u = (v - st[0]) * st[1];
              v = u * p.g + p.beta;
            } break;
            default:
              break;
          }
        }
        xsrow[c] = v;
      }
    }
    // ---- stage the weight slab: rows kk = c0*KS .. (c0+CI_T)*KS, columns m0 .. m0+BM ------
    {
      constexpr int ROWS = CI_T * KS;
      constexpr int V4 = BM / 4;
      const int kk0 = c0 * KS;
      for (int e = tid; e < ROWS * V4; e += NT) {
        const int r = e / V4;
        const int c4 = (e % V4) * 4;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (kk0 + r < KK && m0 + c4 < d.w_ld)
          v = *reinterpret_cast<const float4*>(d.wt + (int64_t)(kk0 + r) * d.w_ld + m0 + c4);
        *reinterpret_cast<float4*>(ws + r * BM + c4) = v;
      }
    }
    __syncthreads();

    // ---- MFMA phase ---------------------------------------------------------------------
    {
      const float* xbase = xs + half * XS + wn * 64 + l31;
      const float* wbase = ws + half * (KS * BM) + wm * 64 + l31;
#pragma unroll
      for (int cp = 0; cp < CI_T / 2; ++cp) {
#pragma unroll
        for (int t = 0; t < KS; ++t) {
          const float* xp = xbase + (2 * cp) * XS + t * d.dil;
          const float* wp = wbase + ((2 * cp) * KS + t) * BM;
          const float b0 = xp[0];
          const float b1 = xp[32];
          const float a0 = wp[0];
          const float a1 = wp[32];
          acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
          acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
          acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
          acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
        }
      }
    }
    __syncthreads();
  }

  // ---- epilogue ---------------------------------------------------------------------------
  float* yb = d.y + (int64_t)b * d.y_bs;
  const float* rb = d.res ? d.res + (int64_t)b * d.res_bs : nullptr;
  const float* r2b = d.res2 ? d.res2 + (int64_t)b * d.res2_bs : nullptr;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int col = n0 + wn * 64 + j * 32 + l31;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        if (row < d.C_out && col < d.L_out) {
          float v = acc[i][j][r];
          if (d.bias) v += d.bias[row];
          if (rb) v += rb[(int64_t)row * d.res_cs + (col >> d.res_shift)];
          if (r2b) v = r2b[(int64_t)row * d.res2_cs + col] + v;
          if (d.div != 1.0f) v = v / d.div;
          switch (d.act) {
            case ST2_ACT_GELU:
              v = gelu_erf(v);
              break;
            case ST2_ACT_EXP_SIN:
              v = row < d.act_split ? expf(v) : sinf(v);
              break;
            case ST2_ACT_TANH:
              v = tanhf(v);
              break;
            case ST2_ACT_LEAKY:
              v = leaky(v, d.act_slope);
              break;
            case ST2_ACT_GELU_TANH:
              v = 0.5f * v * (1.0f + tanhf(0.7978845608028654f * (v + 0.044715f * (v * v * v))));
              break;
            default:
              break;
          }
          yb[(int64_t)row * d.y_cs + col] = v;
        }
      }
    }
  }
}

template <int KS, int CI_T>
int launch(const st2_conv_desc& d, hipStream_t s) {
  const int XW = BN + (KS - 1) * d.dil;
  const int XS = (XW + 3) & ~3;
  const size_t smem = (size_t)(CI_T * XS + CI_T * KS * BM) * sizeof(float);
  ST2_REQUIRE(smem <= 64 * 1024, "st2_conv1d: tile needs %zu B of LDS (ks=%d dil=%d)", smem, KS, d.dil);
  dim3 grid(st2_cdiv(d.L_out, BN), st2_cdiv(d.C_out, BM), d.B);
  hipLaunchKernelGGL((conv1d_mfma_kernel<KS, CI_T>), grid, dim3(NT), smem, s, d);
  ST2_CHECK_LAUNCH("st2_conv1d");
  return 0;
}

}  // namespace

extern "C" int st2_conv1d(const st2_conv_desc* dp, void* stream) {
  ST2_REQUIRE(dp != nullptr, "st2_conv1d: null descriptor");
  const st2_conv_desc& d = *dp;
  ST2_REQUIRE(d.B > 0 && d.C_in > 0 && d.C_out > 0 && d.L_in > 0 && d.L_out > 0,
              "st2_conv1d: empty geometry B=%d C_in=%d C_out=%d L_in=%d L_out=%d", d.B, d.C_in, d.C_out,
              d.L_in, d.L_out);
  ST2_REQUIRE(d.x && d.wt && d.y, "st2_conv1d: null tensor pointer");
  ST2_REQUIRE(d.w_ld >= d.C_out && (d.w_ld & 3) == 0, "st2_conv1d: w_ld=%d must be >= C_out=%d and %%4==0",
              d.w_ld, d.C_out);
  ST2_REQUIRE((reinterpret_cast<uintptr_t>(d.wt) & 15) == 0, "st2_conv1d: wt must be 16-byte aligned");
  ST2_REQUIRE(d.dil >= 1 && d.dil <= 8, "st2_conv1d: dil=%d out of range", d.dil);
  ST2_REQUIRE(d.pro >= ST2_PRO_NONE && d.pro <= ST2_PRO_COLNORM, "st2_conv1d: bad prologue %d", d.pro);
  if (d.pro == ST2_PRO_ADAIN_LEAKY || d.pro == ST2_PRO_ADAIN_SNAKE || d.pro == ST2_PRO_COLNORM)
    ST2_REQUIRE(d.stats && d.gamma && d.beta, "st2_conv1d: prologue %d needs stats/gamma/beta", d.pro);
  if (d.pro == ST2_PRO_ADAIN_SNAKE || d.pro == ST2_PRO_SNAKE)
    ST2_REQUIRE(d.alpha, "st2_conv1d: snake prologue needs alpha");
  ST2_REQUIRE(d.res_shift >= 0 && d.res_shift <= 1, "st2_conv1d: res_shift must be 0 or 1");
  ST2_REQUIRE(d.B <= 65535 && st2_cdiv(d.C_out, BM) <= 65535, "st2_conv1d: grid too large");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  switch (d.ks) {
    case 1:
      return launch<1, 32>(d, s);
    case 2:
      return launch<2, 16>(d, s);
    case 3:
      return launch<3, 16>(d, s);
    case 5:
      return launch<5, 8>(d, s);
    case 7:
      return launch<7, 8>(d, s);
    case 11:
      return launch<11, 8>(d, s);
    default:
      st2_set_error("st2_conv1d: unsupported kernel size %d (have 1,2,3,5,7,11)", d.ks);
      return 1;
  }
}
