// Unmasked multi-head attention for the style-diffusion denoiser (N <= 512 tokens, 8 heads x 64), exact fp32 on
// the matrix pipe (v_mfma_f32_32x32x2_f32 == a k-ordered fmaf chain).
// Tensors are channel-major ([B][H*D][N]): exactly the operand layout the 32x32x2 MFMA wants, so neither Q, K
// nor V is ever transposed.
//
//   S^T tile [32 keys x 32 queries] = K_tile . Q_tile^T      A = K[d][m] (LDS), B = Q[d][q] (registers)
//   O^T tile [32 d    x 32 queries] += V[d][m] . P^T[m][q]   A = V[d][m] (LDS), B = the lane's own P registers
//
// The C/D layout of S^T puts query (lane & 31) in every lane with 16 of the 32 keys in its registers -- keys
// (r&3)+8(r>>2)+4(lane>>5) -- which is exactly the (k = lane>>5) operand split the next MFMA needs for P^T, so
// softmax probabilities feed the PV product without leaving their registers.  The online-softmax state is
// per query: the running max is shared between lanes l and l^32 with one shuffle per key tile.
#include "st2_common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {

constexpr int AD = 64;         // head features (fixed by the reference config)
constexpr int AKT = 128;       // keys per LDS chunk
constexpr int ALD = AKT + 1;   // LDS row stride (odd: the V^T reads walk d across lanes)
constexpr int AQB = 128;       // queries per workgroup (4 waves x 32)

__global__ __launch_bounds__(256) void attention_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                        const float* __restrict__ v, int64_t bs, int cs,
                                                        float* __restrict__ o, int64_t o_bs, int o_cs, int N,
                                                        float scale, const int* __restrict__ key_len) {
  extern __shared__ __attribute__((aligned(16))) float att_smem[];  // 2 x 64 x 129 floats = 66 KB (dynamic: > 64 KB)
  float* ks = att_smem;
  float* vs = att_smem + AD * ALD;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int half = lane >> 5;
  const int l31 = lane & 31;
  const int h = blockIdx.y;
  const int b = blockIdx.z;
  const int q0 = blockIdx.x * AQB + wave * 32;
  const int64_t base = (int64_t)b * bs + (int64_t)h * AD * cs;
  const int nq = q0 + l31;
  const bool wave_live = q0 < N;  // a wave whose 32 queries are all out of range still helps staging
  const int NK = key_len ? max(1, min(key_len[b], N)) : N;  // keys >= NK are padding (excluded from the softmax)

  // B operand of the QK product: Q[d = 2i + half][q0 + l31], pre-multiplied by nothing (scale is applied to S)
  float qr[AD / 2];
#pragma unroll
  for (int i = 0; i < AD / 2; ++i)
    qr[i] = (wave_live && nq < N) ? q[base + (int64_t)(2 * i + half) * cs + nq] : 0.f;

  f32x16 oacc[2];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) oacc[t][r] = 0.f;
  float mx = -INFINITY, den = 0.f;

  for (int c0 = 0; c0 < NK; c0 += AKT) {
    const int ct = min(AKT, NK - c0);
    __syncthreads();
    // K / V chunk -> LDS.  The loads of a batch of 8 rows are issued together (16 in flight per thread, unconditional with
    // a clamped column) before any of them is consumed: one load + one LDS store per iteration made this loop 32 serial
    // L2 round trips -- most of the 37 us a 100-token launch took (40 launches per bench step).
    constexpr int ST_U = 8;
    static_assert((AD * AKT) % (256 * ST_U) == 0, "whole batches");
    for (int e0 = tid; e0 < AD * AKT; e0 += 256 * ST_U) {
      float kr[ST_U], vr[ST_U];
#pragma unroll
      for (int u = 0; u < ST_U; ++u) {
        const int e = e0 + u * 256;
        const int d = e / AKT, mm = e % AKT;
        const int64_t off = base + (int64_t)d * cs + c0 + min(mm, ct - 1);
        kr[u] = k[off];
        vr[u] = v[off];
      }
#pragma unroll
      for (int u = 0; u < ST_U; ++u) {
        const int e = e0 + u * 256;
        const int d = e / AKT, mm = e % AKT;
        const bool ok = mm < ct;
        ks[d * ALD + mm] = ok ? kr[u] : 0.f;
        vs[d * ALD + mm] = ok ? vr[u] : 0.f;
      }
    }
    __syncthreads();
    if (!wave_live) continue;
    for (int m0 = 0; m0 < ct; m0 += 32) {
      // ---- S^T = K . Q^T -------------------------------------------------------------------------------
      f32x16 s;
#pragma unroll
      for (int r = 0; r < 16; ++r) s[r] = 0.f;
#pragma unroll
      for (int i = 0; i < AD / 2; ++i) {
        const float a = ks[(2 * i + half) * ALD + m0 + l31];
        s = __builtin_amdgcn_mfma_f32_32x32x2f32(a, qr[i], s, 0, 0, 0);
      }
      // ---- online softmax over this tile's keys (rows), per query (column) -------------------------------
      float tmx = -INFINITY;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int key = m0 + (r & 3) + 8 * (r >> 2) + 4 * half;
        s[r] = key < ct ? s[r] * scale : -INFINITY;
        tmx = fmaxf(tmx, s[r]);
      }
      tmx = fmaxf(tmx, __shfl_xor(tmx, 32, 64));  // key m0 is always in range: tmx is finite
      const float nmx = fmaxf(mx, tmx);
      const float corr = expf(mx - nmx);  // 0 on the first tile (mx = -inf)
      mx = nmx;
      float psum = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        s[r] = expf(s[r] - nmx);
        psum += s[r];
      }
      den = den * corr + psum;
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[t][r] *= corr;
      // ---- O^T += V . P^T: k-step r pairs key (r&3)+8(r>>2) (lanes < 32) with the same +4 (lanes >= 32) ----
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int key = m0 + (r & 3) + 8 * (r >> 2) + 4 * half;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          const float a = vs[(t * 32 + l31) * ALD + key];
          oacc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, s[r], oacc[t], 0, 0, 0);
        }
      }
    }
  }
  if (!wave_live) return;
  den += __shfl_xor(den, 32, 64);
  if (nq < N) {
    const float inv = 1.0f / den;
    const int64_t ob = (int64_t)b * o_bs + (int64_t)h * AD * o_cs;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int d = t * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        o[ob + (int64_t)d * o_cs + nq] = oacc[t][r] * inv;
      }
  }
}

}  // namespace

extern "C" int st2_attention(const float* q, const float* k, const float* v, int64_t bs, int32_t cs, float* o,
                             int64_t o_bs, int32_t o_cs, int32_t B, int32_t H, int32_t D, int32_t N, float scale,
                             void* stream) {
  return st2_attention_keylen(q, k, v, bs, cs, o, o_bs, o_cs, B, H, D, N, scale, nullptr, stream);
}

extern "C" int st2_attention_keylen(const float* q, const float* k, const float* v, int64_t bs, int32_t cs, float* o,
                                    int64_t o_bs, int32_t o_cs, int32_t B, int32_t H, int32_t D, int32_t N,
                                    float scale, const int32_t* key_len, void* stream) {
  ST2_REQUIRE(q && k && v && o && B > 0 && H > 0 && N > 0, "st2_attention: bad arguments");
  ST2_REQUIRE(D == AD, "st2_attention: head_features=%d unsupported (built for %d)", D, AD);
  ST2_REQUIRE(B <= 65535 && H <= 65535, "st2_attention: grid too large");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  constexpr size_t smem = (size_t)2 * AD * ALD * sizeof(float);
  static std::atomic<uint64_t> attr_done{0};  // one bit per device ordinal
  st2_once_per_device(attr_done, [&] {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&attention_kernel),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
  });
  hipLaunchKernelGGL(attention_kernel, dim3(st2_cdiv(N, AQB), H, B), dim3(256), smem, s, q, k, v, bs, cs, o, o_bs,
                     o_cs, N, scale, reinterpret_cast<const int*>(key_len));
  ST2_CHECK_LAUNCH("st2_attention");
  return 0;
}
