// Epilogue shared by the split-f16 MFMA conv kernels (st2_conv1d_xs_impl.h, st2_conv1d_f16s_impl.h): out_scale, bias,
// residual(s), divide, activation, stores, and the per-128-column InstanceNorm partial sums of the stored values.
//
// Accumulator layout: both kernels feed the activations as the MFMA's A operand and the weights as its B operand, so
// the 32 x 32 accumulator block is the TRANSPOSED tile: lane (l31, kg) owns output row co = m0 + wm*32 + l31 and, in
// acc[j][4*q + e], the position l = n0 + wn*32*TN + j*32 + 8*q + 4*kg + e -- four CONSECUTIVE positions per (j, q).
#pragma once
#include "st2_common.h"
#include "st2_act.h"
#include <type_traits>

typedef float st2_f32x16 __attribute__((ext_vector_type(16)));
typedef float st2_f32x4 __attribute__((ext_vector_type(4)));

namespace {

// Compile-time loop: f(integral_constant<int, 0>) ... f(integral_constant<int, N - 1>).  The accumulator array must only
// ever be indexed by constants (a run-time index would move all of it to scratch); `#pragma unroll` is a request the
// optimizer may decline for a large body, this is not.
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (N > 0) {
    static_for<N - 1>(f);
    f(std::integral_constant<int, N - 1>{});
  }
}

template <int TN, int WM, int WN>
__device__ __forceinline__ void st2_conv_epilogue(const st2_conv_desc& d, st2_f32x16 (&acc)[TN], int b, int m0, int n0,
                                                  int wm, int wn, int l31, int kg) {
  typedef st2_f32x4 f32x4;
  constexpr int BM = 32 * WM;
  // Interior tiles of 16-byte aligned tensors therefore store / load 16 bytes per lane and instruction -- 16 stores per
  // wave tile instead of 64 (the store tail of an MFMA kernel is bound by the number of store instructions, not by
  // their bytes: cdna_hip_programming.md T21; measured here: the epilogue cost 0.23 ms of a 1.70 ms k = 11 launch and
  // 0.35 of 0.85 ms at k = 3 with 4-byte stores), bias and weight scale are per lane, and the InstanceNorm partial
  // sums of a row are a lane's own 64 values plus its kg partner's: one cross-lane move.
  float* yb = d.y + (int64_t)b * d.y_bs;
  const float* rb = d.res ? d.res + (int64_t)b * d.res_bs : nullptr;
  const float* r2b = d.res2 ? d.res2 + (int64_t)b * d.res2_bs : nullptr;
  const float osc = d.out_scale;
  // per-row weight scale: unconditional load + select (the packed weights serve as a valid address without one)
  const float* rsc = d.w_row_scale ? d.w_row_scale : reinterpret_cast<const float*>(d.wq);
  const bool want_part = d.part != nullptr;
  // 128-column partial-sum tiles per wave tile.  TN < 4 bodies are (i) the narrow token-GEMM LAUNCHES, which never carry
  // d.part (checked by the launcher), and (ii) the row-end QUARTER bodies of a wide launch (st2_conv1d_xs_impl.h: a last
  // tile whose valid columns fit TN / 4 blocks runs the body instantiated with TN / 4), which DO produce partial sums:
  // write_part below closes their unfinished 128-column group.  Held against the CPU reduction of the stored output for
  // the aligned (L = 400), unaligned (L = 777) and wide (L = 8 000) row ends in tests/test_ops_gpu.py
  // (test_conv1d_xs_epilogue_stats; advisor, round 4).
  constexpr int NPT = TN >= 4 ? TN / 4 : 1;
  constexpr int JB = TN >= 4 ? 4 : TN;       // column blocks per residual batch
  // first partial-sum slot of this wave: slots are 128 columns wide, or -- small-grid launches of 64 / 32-column tiles, whose
  // tiles each own ONE slot (d.part_cols, checked by the launcher) -- 64 / 32
  const int ptile = (n0 + wn * (32 * TN)) >> (d.part_cols == 64 ? 6 : (d.part_cols == 32 ? 5 : 7));
  const int co = m0 + wm * 32 + l31;       // this lane's output row (< wq_co_pad by construction of the packing)
  const int lw = n0 + wn * (32 * TN) + 4 * kg;  // first position of this lane's (j = 0, q = 0) quad
  // 16-byte accesses need every row of y / res / res2 to start 16-byte aligned (workgroup-uniform, set by the plans'
  // padded row pitch); the residual must not be sub-sampled
  const bool vec_ok = ((reinterpret_cast<uintptr_t>(d.y) | (uintptr_t)(d.y_bs * 4) | (uintptr_t)(d.y_cs * 4)) & 15) == 0 &&
                      (!rb || (((reinterpret_cast<uintptr_t>(d.res) | (uintptr_t)(d.res_bs * 4) | (uintptr_t)(d.res_cs * 4)) & 15) == 0 &&
                               d.res_shift == 0)) &&
                      (!r2b || ((reinterpret_cast<uintptr_t>(d.res2) | (uintptr_t)(d.res2_bs * 4) | (uintptr_t)(d.res2_cs * 4)) & 15) == 0);
  // Which of this WAVE's 32-column blocks are complete.  A tile at the end of a row used to take the generic build as a
  // whole -- 4-byte predicated accesses for all TN blocks, 3.5 x the cycles of the straight-line build on every box and
  // 10-12 x on some -- and, where the tile count per row is a multiple of 8 (L = 8 000 in 256-column tiles), every such
  // tile of a launch lands on the same XCD and the same shader engine, which then finishes at twice the time of the other
  // seven (profiles/r04/r04h1_*, r04j_*: what rounds 1-3 knew as "the slow box class").  Now blocks [0, jn_full) take the
  // straight-line build, blocks past the row end are skipped, and only a block that STRADDLES the row end (none at L =
  // 8 000; one column's worth at L = 48 001) takes the generic code.  Same values, same order of the partial sums.
  const int avail = d.L_out - (n0 + wn * (32 * TN));  // valid columns of this wave's part of the tile (may be <= 0)
  const int jn_full = avail >= 32 * TN ? TN : (avail > 0 ? avail / 32 : 0);
  const int j_strad = (jn_full < TN && avail > jn_full * 32) ? jn_full : -1;
  const bool rows_ok = m0 + BM <= d.C_out;
  // The epilogue comes in straight-line builds.  Which terms exist (residual, MRF accumulator, divide) is uniform per
  // launch; tested per element it turns the loop into thousands of one-store basic blocks whose residual loads are
  // each waited for on the spot.  So interior tiles -- every tile but the last along l / co -- of aligned tensors
  // dispatch ONCE to a build with those terms as compile-time constants: no bounds tests, 32-bit offsets from scalar
  // bases, a column block's residual loads issued together ahead of its arithmetic.  Edge tiles, unaligned tensors and
  // rare combinations take the generic build (MODE < 0: run-time flags, per-element bounds, 4-byte accesses).
  // Running (sum, sum of squares) of this lane's stored values of the current slot, SHIFTED by the slot's first stored value
  // (csh = y[b][co][first column of the slot]: lane kg = 0 owns it, its kg = 1 partner gets it by one cross-lane move): the
  // finaliser gets that value with the sums and combines the slots with Chan's formula in fp64.  Unshifted sums lose the variance
  // of a channel whose mean dominates it -- E[x^2] - mean^2 with fp32 partial sums: |mean| / std = 100 (a bias-dominated channel of
  // a residual stream) costs 1e-8 x 1e4 = 1e-4 of rstd, three decades above fp32 (round 5, tools/stress.py stats_precision) --
  // shifted ones do not: the shifted mean is within a few std of zero whatever the channel's offset.
  float s1 = 0.f, s2 = 0.f, csh = 0.f;
  float s1d[NPT] = {}, s2d[NPT] = {}, cshd[NPT] = {};  // the finished slots' sums and shifts: shared by the builds a tile may combine
  auto epilogue_as = [&](auto act_tag, auto mode_tag, const int j_lo, const int j_hi) __attribute__((always_inline)) {
    constexpr int ACT = decltype(act_tag)::value;
    constexpr int MODE = decltype(mode_tag)::value;  // < 0: generic; else bit 0 = res, bit 1 = res2, bit 2 = div
    constexpr bool FULL = MODE >= 0;
    const bool use_res = FULL ? (MODE & 1) != 0 : rb != nullptr;
    const bool use_res2 = FULL ? (MODE & 2) != 0 : r2b != nullptr;
    const bool use_div = FULL ? (MODE & 4) != 0 : d.div != 1.0f;
    const bool rok = FULL || co < d.C_out;
    const int coc = FULL ? co : min(co, d.C_out - 1);
    // unconditional load + select (a branch here would split the code into separate basic blocks); without a bias the
    // packed weights serve as a valid address
    const float braw = (d.bias ? d.bias : reinterpret_cast<const float*>(d.wq))[coc];
    const float bias_r = d.bias ? braw : 0.f;
    const float sraw = rsc[co];
    const float osc_r = d.w_row_scale ? osc * sraw : osc;
    // 32-bit element offsets from the (scalar) per-batch bases (a batch item is < 2^31 elements, checked at launch)
    const int yo = coc * d.y_cs + lw;
    const int ro = coc * d.res_cs;   // + (l >> res_shift)
    const int r2o = coc * d.res2_cs + lw;
    auto finish = [&](float v) __attribute__((always_inline)) -> float {
      if (use_div) v = v / d.div;
      if constexpr (ACT == -1) {  // generic build: one body for every activation (run-time switch, wave-uniform)
        switch (d.act) {
          case ST2_ACT_GELU: v = gelu_erf(v); break;
          case ST2_ACT_EXP_SIN: v = co < d.act_split ? expf(v) : sin_acc(v); break;
          case ST2_ACT_TANH: v = tanhf(v); break;
          case ST2_ACT_LEAKY: v = leaky(v, d.act_slope); break;
          case ST2_ACT_GELU_TANH: v = gelu_tanh(v); break;
          default: break;
        }
      } else if constexpr (ACT == ST2_ACT_GELU) {
        v = gelu_erf(v);
      } else if constexpr (ACT == ST2_ACT_EXP_SIN) {
        v = co < d.act_split ? expf(v) : sin_acc(v);
      } else if constexpr (ACT == ST2_ACT_TANH) {
        v = tanhf(v);
      } else if constexpr (ACT == ST2_ACT_LEAKY) {
        v = leaky(v, d.act_slope);
      } else if constexpr (ACT == ST2_ACT_GELU_TANH) {
        v = gelu_tanh(v);
      }
      return v;
    };
    if constexpr (FULL) {
      // ALL residual quads of 4 column blocks are requested before the first one is used: 16 x 16 bytes per lane in
      // flight (64 VGPRs -- the k loop's fragment and staging registers are dead here).  Issued per (j, q) pair they
      // cost one HBM round trip each, 8 in series per tile: measured 27 000 cycles of epilogue against a 101 000-cycle
      // k loop (per-workgroup s_memtime stamps, tools/xs_bench.hip), i.e. a fifth of every workgroup slot's time.
#pragma unroll
      for (int jh = 0; jh < TN; jh += JB) {
        if (jh >= j_hi) break;  // (wave-uniform) blocks past the row end: nothing to load, store or sum
        f32x4 rv[JB][4];
        if (use_res) {
#pragma unroll
          for (int jj = 0; jj < JB; ++jj)
            if (jh + jj < j_hi) {
#pragma unroll
              for (int q = 0; q < 4; ++q)
                rv[jj][q] = *reinterpret_cast<const f32x4*>(rb + ro + lw + (jh + jj) * 32 + 8 * q);
            }
        }
#pragma unroll
        for (int jj = 0; jj < JB; ++jj) {
          const int j = jh + jj;
          if (j >= j_hi) break;
          f32x4 r2v[4];
          if (use_res2) {
#pragma unroll
            for (int q = 0; q < 4; ++q) r2v[q] = *reinterpret_cast<const f32x4*>(r2b + r2o + j * 32 + 8 * q);
          }
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            f32x4 v;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              // osc_r is a power of two: acc * osc_r is exact, so the fused form rounds exactly like mul + add
              float t = fmaf(acc[j][4 * q + e], osc_r, bias_r);
              if (use_res) t += rv[jj][q][e];
              if (use_res2) t = r2v[q][e] + t;
              t = finish(t);
              v[e] = t;
              if ((j & 3) == 0 && q == 0 && e == 0) {  // (compile time after unrolling) first value of a slot: its shift
                const float other = __shfl_xor(t, 32, 64);
                csh = kg == 0 ? t : other;
              }
              const float dv = t - csh;
              s1 += dv;
              s2 = fmaf(dv, dv, s2);
            }
            *reinterpret_cast<f32x4*>(yb + yo + j * 32 + 8 * q) = v;
          }
          // pin the running sums here: otherwise the compiler sinks the whole accumulation below the (wave-uniform)
          // `want_part` test, keeps all 64 stored values alive for it and spills
          asm volatile("" : "+v"(s1), "+v"(s2));
          __builtin_amdgcn_sched_barrier(0);
          if ((j & 3) == 3) {
            s1d[j >> 2] = s1;
            s2d[j >> 2] = s2;
            cshd[j >> 2] = csh;
            s1 = 0.f;
            s2 = 0.f;
          }
        }
      }
    } else {
      static_for<TN * 16>([&](auto idx_tag) __attribute__((always_inline)) {
        constexpr int idx = decltype(idx_tag)::value;
        constexpr int j = idx / 16, q = (idx % 16) / 4, e = idx % 4;
        const int l = lw + j * 32 + 8 * q + e;
        const bool ok = rok && l < d.L_out && j >= j_lo && j < j_hi;
        float t = fmaf(acc[j][4 * q + e], osc_r, bias_r);
        if (use_res) t += ok ? rb[ro + (l >> d.res_shift)] : 0.f;
        if (use_res2) t = (ok ? r2b[r2o + j * 32 + 8 * q + e] : 0.f) + t;
        t = finish(t);
        if constexpr ((j & 3) == 0 && q == 0 && e == 0) {
          if (j >= j_lo && j < j_hi) {  // (wave-uniform) this pass starts the slot: same shift as the straight-line build takes
            const float other = __shfl_xor(t, 32, 64);
            csh = kg == 0 ? t : other;
          }
        }
        if (ok) {
          yb[yo + j * 32 + 8 * q + e] = t;
          const float dv = t - csh;
          s1 += dv;
          s2 = fmaf(dv, dv, s2);
        }
        if constexpr (e == 3) asm volatile("" : "+v"(s1), "+v"(s2));
        if constexpr (idx % 64 == 63) {
          if (j >= j_lo && j < j_hi) {  // this pass owns the end of the 128-column group (wave-uniform)
            s1d[j >> 2] = s1;
            s2d[j >> 2] = s2;
            cshd[j >> 2] = csh;
            s1 = 0.f;
            s2 = 0.f;
          }
        }
      });
    }
  };
  // j_last = the last column block any pass processed: a 128-column group that ends inside the row is closed by the pass that
  // reaches its fourth block; the group the row ends in is closed here (same running sum, same order as one generic pass)
  auto write_part = [&](const int j_last) __attribute__((always_inline)) {
    if (want_part) {  // wave-uniform: (sum, sumsq) of row co over 128 columns = this lane + its kg partner
#pragma unroll
      for (int t = 0; t < NPT; ++t) {
        if (j_last >= 0 && (j_last & 3) != 3 && (j_last >> 2) == t) {
          s1d[t] = s1;
          s2d[t] = s2;
          cshd[t] = csh;
        }
        const float a1 = s1d[t] + __shfl_xor(s1d[t], 32, 64);
        const float a2 = s2d[t] + __shfl_xor(s2d[t], 32, 64);
        if (kg == 0 && co < d.C_out && ptile + t < d.part_nt) {
          const int64_t slot = ((int64_t)b * d.C_out + co) * d.part_nt + ptile + t;
          reinterpret_cast<float2*>(d.part)[slot] = make_float2(a1, a2);
          // the slot's shift rides behind the sums ([B * C_out][part_nt] floats): the finaliser reads it from there, coalesced,
          // instead of gathering one value per slot from y (59 us per launch on the 240 000-sample HiFi-GAN rows, r05n)
          d.part[(int64_t)d.B * d.C_out * d.part_nt * 2 + slot] = cshd[t];
        }
      }
    }
  };
  auto epilogue = [&](auto act_tag) __attribute__((always_inline)) {
    constexpr int ACT = decltype(act_tag)::value;
    const int mode = (rb ? 1 : 0) | (r2b ? 2 : 0) | (d.div != 1.0f ? 4 : 0);
    if constexpr (ACT == ST2_ACT_NONE) {
      switch (mode) {
        case 0: return epilogue_as(act_tag, std::integral_constant<int, 0>{}, 0, jn_full);
        case 1: return epilogue_as(act_tag, std::integral_constant<int, 1>{}, 0, jn_full);
        case 2: return epilogue_as(act_tag, std::integral_constant<int, 2>{}, 0, jn_full);
        case 3: return epilogue_as(act_tag, std::integral_constant<int, 3>{}, 0, jn_full);
        case 4: return epilogue_as(act_tag, std::integral_constant<int, 4>{}, 0, jn_full);
        case 5: return epilogue_as(act_tag, std::integral_constant<int, 5>{}, 0, jn_full);
        case 6: return epilogue_as(act_tag, std::integral_constant<int, 6>{}, 0, jn_full);
        default: return epilogue_as(act_tag, std::integral_constant<int, 7>{}, 0, jn_full);
      }
    } else {
      return epilogue_as(act_tag, std::integral_constant<int, 0>{}, 0, jn_full);
    }
  };
  // Incomplete row blocks, unaligned tensors, an activation combined with residual / divide: the generic body for the whole
  // wave tile; otherwise the straight-line build for the complete column blocks and the generic body for the one block that
  // straddles the row end, if any.  ONE call site of the generic body (its 64 predicated element blocks are the bulk of
  // this kernel's code): the range it covers is data.
  int g_lo = 0, g_hi = TN;
  if (rows_ok && vec_ok && !(d.act != ST2_ACT_NONE && (rb || r2b || d.div != 1.0f))) {
    if (jn_full > 0) {
      switch (d.act) {
        case ST2_ACT_GELU:
          epilogue(std::integral_constant<int, ST2_ACT_GELU>{});
          break;
        case ST2_ACT_EXP_SIN:
          epilogue(std::integral_constant<int, ST2_ACT_EXP_SIN>{});
          break;
        case ST2_ACT_TANH:
          epilogue(std::integral_constant<int, ST2_ACT_TANH>{});
          break;
        case ST2_ACT_LEAKY:
          epilogue(std::integral_constant<int, ST2_ACT_LEAKY>{});
          break;
        case ST2_ACT_GELU_TANH:
          epilogue(std::integral_constant<int, ST2_ACT_GELU_TANH>{});
          break;
        default:
          epilogue(std::integral_constant<int, ST2_ACT_NONE>{});
          break;
      }
    }
    g_lo = j_strad >= 0 ? j_strad : 0;
    g_hi = j_strad >= 0 ? j_strad + 1 : 0;
  }
  if (g_hi > g_lo)  // (wave-uniform)
    epilogue_as(std::integral_constant<int, -1>{}, std::integral_constant<int, -1>{}, g_lo, g_hi);
  write_part(g_hi > g_lo ? g_hi - 1 : jn_full - 1);
}

}  // namespace
