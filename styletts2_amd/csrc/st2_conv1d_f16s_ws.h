// Warp-specialised, persistent build of the fused split-f16 conv (same contract, operands and MFMA sequence per accumulator
// as conv1d_f16s_kernel: results are BITWISE those of st2_conv1d_f16s_impl.h, tools/probe_ws.py) for the vocoder's narrow
// AdaIN + Snake convs (C_out <= 64: the HiFi-GAN stages at L = 120 000 / 240 000).
//
// One 512-thread workgroup per CU walks a contiguous range of tiles; its waves have fixed roles, one pair per SIMD:
//   waves 0-3  consumers: LDS fragments + weight fragments -> MFMA (the k loop of conv1d_f16s_kernel, verbatim), epilogue;
//   waves 4-7  producers: global loads two chunks ahead (two named register sets), prologue + hi/lo split one chunk ahead
//              into the other LDS buffer -- across tile boundaries, so a tile's fill happens under its predecessor's k loop
//              and epilogue.
// Phases are separated by one workgroup barrier: a tile is nchunk MFMA phases + one epilogue phase; in every phase the
// producers stage one chunk if a buffer is free (the epilogue phase frees the second one).
//
// What it buys, and what it does not (round 3, profiles/archive/r03/r03s..r03z): on these layers conv1d_f16s_kernel keeps the matrix pipe
// busy 0.99 M cycles per SIMD and the VALU 0.90 M of a 2.56 M-cycle launch (C = 64, k = 11, L = 120 000, B = 32; counters in
// profiles/archive/r03/r03u_*) and removing any one stage changes little (r03s_probe_ws_abl.log).  The phase timeline of this kernel
// (s_memtime stamps of all eight waves, tools/probe_ws_timeline.py, r03y_ws_timeline*.log) shows why specialisation alone
// does not reach max(MFMA, VALU): the consumer's k loop takes 5.0 k cycles per chunk alone and 5.9 k beside a staging
// producer, the producer 4.3-6 k alone and 8-9.7 k beside the k loop -- on one SIMD the two instruction streams nearly ADD
// (s_setprio either way, no SLP-packed f32, eight producer waves, two workgroups per CU of 64-column wave tiles: all within
// 5 %), and the epilogue phase (residual at HBM latency on the consumers, whose weight stream shares the in-order vmcnt) costs
// 11 k cycles per tile.  Measured against the one-role kernel (profiles/archive/r03/r03A_probe_ws.log): x1.12-1.20 at k = 3 for
// C <= 64; x1.01-1.07 at k = 7 / 11 with dilation 3 / 5 but x0.83 at dilation 1; x0.84-0.91 at C = 128 -- hence the k = 3,
// C_out <= 64 rule in st2f16s::ws_eligible.  Two compiler facts the
// structure depends on: (1) the register-set parity of a staging step must be a compile-time constant at every call (a
// run-time `p & 1` made hipcc merge the two sets through copies that wait for loads just issued: +4.6 k cycles per phase);
// (2) no load may sit behind a branch (the per-batch-item parameter loads are issued with every chunk and only the LDS
// write is conditional), or the in-order vmcnt bookkeeping turns conservative.
#pragma once
#include "st2_conv_epilogue.h"
#include <algorithm>

#ifndef ST2_WS_ABLATE
#define ST2_WS_ABLATE 0  // measurement builds only (tools/build_ws_ablate.sh): 1 no prologue math, 2 no MFMA, 3 no activation loads,
#endif                   // 4 no epilogue, 5 no LDS fragment reads, 6 no weight loads; results are then meaningless
namespace st2ws {

#ifdef ST2_WS_TIMELINE  // measurement build: workgroup 0 stamps s_memtime at every phase edge (4 stamps per phase and role)
constexpr int TL_N = 1024;  // stamps per wave; slot 0 = HW_ID of the wave
inline unsigned long long* tl_buffer() {
  static unsigned long long* p = nullptr;
  if (!p && hipMalloc(&p, 12 * TL_N * 8) == hipSuccess) (void)hipMemset(p, 0, 12 * TL_N * 8);
  return p;
}
#define WS_TL_PARAM , unsigned long long* tl
#define WS_TL_INIT(role)                                                                       \
  unsigned long long* tlp = (blockIdx.x == 0 && (threadIdx.x & 63) == 0) ? tl + (threadIdx.x >> 6) * TL_N : nullptr; \
  int tli = 1;                                                                                 \
  if (tlp) tlp[0] = __builtin_amdgcn_s_getreg((31 << 11) | 4);
#define WS_STAMP()                                   \
  do {                                               \
    if (tlp && tli < TL_N) tlp[tli++] = clock64();   \
  } while (0)
// stamps of a phase: start, end of the wave's work, its memory operations drained, after the barrier
#define WS_SYNC()                                           \
  do {                                                      \
    WS_STAMP();                                             \
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");  \
    WS_STAMP();                                             \
    __builtin_amdgcn_s_barrier();                           \
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");  \
    WS_STAMP();                                             \
  } while (0)
#else
#define WS_TL_PARAM
#define WS_TL_INIT(role)
#define WS_STAMP() do {} while (0)
#define WS_SYNC() __syncthreads()
#endif

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef st2_f32x16 f32x16;

constexpr int NCT = 256;        // consumer threads (waves 0-3)
constexpr int NPR = 256;        // producer threads (waves 4-7; eight producer waves: 3-5 % faster, 30-140 spilled VGPRs)
constexpr int NTW = NCT + NPR;  // threads per workgroup

struct ChanPar {  // per input channel, staged per batch item in LDS (32 B); xs = x_scale, 0 for the channel tail of the last chunk
  float mean, rstd, g, beta, alpha, inv_alpha, xs, pad1;
};

// a value every lane of the wave holds, moved to an SGPR
static __device__ __forceinline__ float wave_uniform(float v) {
  return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v)));
}

struct TileGeom {  // tile index -> (batch item, co block, l block); l fastest, so a workgroup's range stays in one batch item
  int tiles_n, tiles_m, ntiles;
};

// PRO: ST2_PRO_ADAIN_SNAKE or ST2_PRO_ADAIN_LEAKY (compile time: the element loop carries no branch)
template <int KS, int CI_T, int WM, int WN, int TN, int PRO>
__global__ __launch_bounds__(NTW, 2) void conv1d_f16s_ws_kernel(const st2_conv_desc d, int* status, const TileGeom tg WS_TL_PARAM) {
  constexpr int BM = 32 * WM;
  constexpr int BN = 32 * TN * WN;
  constexpr int CG = CI_T / 8;     // 8-channel groups per chunk
  constexpr int S16 = CI_T / 16;   // MFMA k-steps per tap per chunk
  constexpr int MAXXW = BN + (KS - 1) * 8;
  constexpr int TPG = NPR / CG;    // staging threads per group
  constexpr int R = (MAXXW + TPG - 1) / TPG;  // staging rounds (positions per producer thread)
  constexpr int SPC = S16 * KS;    // k-steps per chunk
  static_assert(WM * WN == 4, "4 consumer waves");
  static_assert(TPG % 64 == 0, "a producer wave stays inside one channel group");

  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const bool consumer = wave < 4;  // wave-uniform

  const int XW = BN + (KS - 1) * d.dil;  // staged positions
  h8* xs = reinterpret_cast<h8*>(smem_raw);  // [2 buffers][2 planes hi/lo][CG][XW] slots of 16 B, then two parameter tables
  const int plane = CG * XW;
  const int C_pad = (d.C_in + CI_T - 1) / CI_T * CI_T;
  ChanPar* par = reinterpret_cast<ChanPar*>(smem_raw + (size_t)4 * plane * 16);  // [2][C_pad]: slot = batch item & 1
  const int nchunk = C_pad / CI_T;
  // this workgroup's tiles [t0, t1): contiguous, balanced to within one tile
  const int t0 = (int)((int64_t)blockIdx.x * tg.ntiles / gridDim.x);
  const int t1 = (int)((int64_t)(blockIdx.x + 1) * tg.ntiles / gridDim.x);
  if (t0 >= t1) return;  // workgroup-uniform

  if (!consumer) {
    // ================================================ producers =================================================
    const int ptid = tid - NCT;
    const int sg = ptid / TPG;   // this thread's 8-channel group; its positions: sp0 + r * TPG (a flat deal of the CG * XW
    const int sp0 = ptid % TPG;  // slots over the threads balances better but costs 18 registers: the k = 3 tile then spills)
    WS_TL_INIT(1)
    float xq[2][R][8];  // the chunk being activated and the chunk in flight
    int q_lin0[2], q_c0[2], q_b[2];  // what each register set holds: first staged input position, first channel, batch item
    bool sat = false;
    int lt = t0, lc = 0;  // load cursor: next (tile, chunk) to request; parks on the last chunk of the range
    int tab_b = -1;       // batch item of the newest parameter table
    float tp_mean = 0.f, tp_rstd = 1.f, tp_g = 0.f, tp_beta = 0.f, tp_alpha = 1.f;  // table entry in flight (channel ptid)
    int tp_b = -1;
    // A new batch item's table goes to slot b & 1: the other slot serves the chunk being activated in this phase, and the
    // chunk that needs this one is activated one barrier later at the earliest.
    auto table_update = [&]() __attribute__((always_inline)) {
      if (tp_b != tab_b) {  // workgroup-uniform
        ChanPar p = {0.f, 1.f, 1.f, 0.f, 1.f, 1.f, 0.f, 0.f};
        if (ptid < d.C_in) {
          p.mean = tp_mean;
          p.rstd = tp_rstd;
          p.g = 1.0f + tp_g;
          p.beta = tp_beta;
          p.alpha = tp_alpha;
          p.inv_alpha = 1.0f / tp_alpha;
          p.xs = d.x_scale;
        }
        if (ptid < C_pad) par[(size_t)(tp_b & 1) * C_pad + ptid] = p;
        tab_b = tp_b;
      }
    };
    // loads are unconditional on clamped (always valid) addresses; out-of-range values are zeroed at the store
    auto load_set = [&](auto set_tag) __attribute__((always_inline)) {
      constexpr int SET = decltype(set_tag)::value;
      const int nb = lt % tg.tiles_n;
      const int b = lt / (tg.tiles_n * tg.tiles_m);
      const int lin0 = nb * BN - d.pad_left;  // input position held by staged column 0
      const int c0 = lc * CI_T;
      const float* xb = d.x + (int64_t)b * d.x_bs;
      {  // this thread's entry of batch item b's parameter table: requested with every chunk (no branch around loads: the
         // in-order vmcnt bookkeeping stays exact), written by table_update() only when b is new
        const int ci = min(ptid, d.C_in - 1);
        const float* st = d.stats + ((int64_t)b * d.C_in + ci) * 2;
        tp_mean = st[0];
        tp_rstd = st[1];
        tp_g = d.gamma[(int64_t)b * d.gb_bs + ci];
        tp_beta = d.beta[(int64_t)b * d.gb_bs + ci];
        tp_alpha = PRO == ST2_PRO_ADAIN_SNAKE ? d.alpha[ci] : 1.0f;
        tp_b = b;
      }
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const int l = min(max(lin0 + sp0 + r * TPG, 0), d.L_in - 1);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const int ci = min(c0 + sg * 8 + e, d.C_in - 1);
          if constexpr (ST2_WS_ABLATE == 3)
            xq[SET][r][e] = (float)(ci + l) * 1e-6f;
          else
            xq[SET][r][e] = xb[(int64_t)ci * d.x_cs + l];
        }
      }
      q_lin0[SET] = lin0;
      q_c0[SET] = c0;
      q_b[SET] = b;
      if (lc + 1 < nchunk) {
        ++lc;
      } else if (lt + 1 < t1) {
        lc = 0;
        ++lt;
      }
    };
    // prologue + hi/lo split of set SET into buffer `buf`, op for op that of conv1d_f16s_kernel / st2_act_split
    auto activate_set = [&](auto set_tag, int buf) __attribute__((always_inline)) {
      constexpr int SET = decltype(set_tag)::value;
      const int c0 = q_c0[SET];
      const int lin0 = q_lin0[SET];
      const ChanPar* tab = par + (size_t)(q_b[SET] & 1) * C_pad;
      h8* dst = xs + (size_t)buf * 2 * plane + sg * XW;
      // A wave's 64 threads share one channel group (TPG is a multiple of 64), so the group's eight table entries are
      // wave-uniform: read once per chunk (one broadcast LDS read each) and kept in SGPRs.  Read inside the element loop
      // they cost two ds_read_b128 per element, re-issued after every fragment store (same LDS array): eight times the
      // bytes the producers write.
      ChanPar cp[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const ChanPar p = tab[c0 + sg * 8 + e];
        cp[e] = ChanPar{wave_uniform(p.mean), wave_uniform(p.rstd), wave_uniform(p.g), wave_uniform(p.beta),
                        wave_uniform(p.alpha), wave_uniform(p.inv_alpha), wave_uniform(p.xs), 0.f};
      }
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const int pos = sp0 + r * TPG;
        if (pos >= XW) continue;
        const int l = lin0 + pos;
        const bool lok = l >= 0 && l < d.L_in;
        h8 hi, lo;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const ChanPar p = cp[e];
          float v = xq[SET][r][e];
          float u = (v - p.mean) * p.rstd;
          u = p.g * u + p.beta;
          if constexpr (ST2_WS_ABLATE == 1)
            v = v + p.beta;
          else if constexpr (PRO == ST2_PRO_ADAIN_LEAKY)
            v = leaky(u, d.slope);
          else
            v = snake(u, p.alpha, p.inv_alpha);
          // zero padding (and the channel tail, through the table's scale) is applied AFTER the activation, as F.conv1d pads
          // the activated tensor
          v = lok ? v * p.xs : 0.f;
          const float vc = st2_clamp_f16(v);  // saturate instead of inf / NaN, reported via st2_status()
          sat |= vc != v;
          const _Float16 h = (_Float16)vc;
          hi[e] = h;
          lo[e] = (_Float16)(vc - (float)h);
        }
        dst[pos] = hi;
        dst[plane + pos] = lo;
      }
    };
    // stage chunk p (held by set p & 1) into buffer p & 1, after requesting chunk p + 1 into the other set.  The parity is
    // a compile-time constant at every call: with a run-time selection the compiler merges the two register sets through
    // copies and waits for loads it has just issued.
    auto produce = [&](auto parity_tag) __attribute__((always_inline)) {
      constexpr int PAR = decltype(parity_tag)::value;
      load_set(std::integral_constant<int, PAR ^ 1>{});
      activate_set(std::integral_constant<int, PAR>{}, PAR);
      table_update();
    };
    const int total = (t1 - t0) * nchunk;  // chunks of this workgroup
    const int nphase = (t1 - t0) * (nchunk + 1);  // a tile = nchunk MFMA phases + one epilogue phase, one barrier each
    // chunks whose MFMA phase is over when phase ph begins
    auto consumed_at = [&](int ph) { return ph / (nchunk + 1) * nchunk + ph % (nchunk + 1); };
    load_set(std::integral_constant<int, 0>{});
    table_update();
    __syncthreads();  // the first parameter table is visible
    produce(std::integral_constant<int, 0>{});
    __syncthreads();  // buffer 0 holds chunk 0
    int ph = 0;  // current phase
    // chunk p may be staged in a phase that begins with at least p - 1 chunks consumed (two buffers); phases without a
    // free buffer are waited out
    auto wait_for_buffer = [&](int p) __attribute__((always_inline)) {
      while (p - consumed_at(ph) >= 2) {
        WS_STAMP();
        WS_SYNC();
        ++ph;
      }
    };
    for (int p = 1; p < total; p += 2) {
      wait_for_buffer(p);
      WS_STAMP();
      produce(std::integral_constant<int, 1>{});
      WS_SYNC();
      ++ph;
      if (p + 1 < total) {
        wait_for_buffer(p + 1);
        WS_STAMP();
        produce(std::integral_constant<int, 0>{});
        WS_SYNC();
        ++ph;
      }
    }
    for (; ph < nphase; ++ph) {
      WS_STAMP();
      WS_SYNC();
    }
    if (sat) st2_raise_status(status, ST2_STATUS_F16_RANGE);
    return;
  }

  // ================================================== consumers ===================================================
  const int kg = lane >> 5;
  const int l31 = lane & 31;
  const int wm = wave / WN;
  const int wn = wave % WN;
  f32x16 acc[TN];
#pragma unroll
  for (int j = 0; j < TN; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
  // A operand stream: 32 B (hi8|lo8) per lane per k-step, constant stride between steps; restarts at every tile
  const h8* wbase = reinterpret_cast<const h8*>(d.wq) + ((int64_t)kg * d.wq_co_pad + wm * 32 + l31) * 2;
  const int64_t a_step = (int64_t)2 * d.wq_co_pad * 2;  // h8 units per k-step
  auto m0_of = [&](int t) { return ((t / tg.tiles_n) % tg.tiles_m) * BM; };
  const h8* ap = wbase + (int64_t)m0_of(t0) * 2;
  h8 a_hi[2], a_lo[2];  // two named sets indexed by the (compile-time) parity of the k-step inside the chunk
  a_hi[0] = ap[0];
  a_lo[0] = ap[1];
  __syncthreads();  // (the producers' first parameter table)
  __syncthreads();  // buffer 0 holds chunk 0

  WS_TL_INIT(0)
  int g = 0;  // chunks consumed so far: chunk g lives in buffer g & 1
  for (int t = t0; t < t1; ++t) {
    const int nb = t % tg.tiles_n;
    const int m0 = m0_of(t);
    const int b = t / (tg.tiles_n * tg.tiles_m);
    const int n0 = nb * BN;
    // the last prefetch of a tile fetches the next tile's first fragment
    const h8* next_base = wbase + (int64_t)m0_of(min(t + 1, t1 - 1)) * 2;
    __builtin_amdgcn_s_setprio(1);
    for (int c = 0; c < nchunk; ++c, ++g) {
      const int buf = g & 1;
      const bool more = c + 1 < nchunk;
      WS_STAMP();
      const h8* xbuf = xs + (size_t)buf * 2 * plane + kg * XW + wn * (32 * TN) + l31;
#pragma unroll
      for (int s = 0; s < S16; ++s) {
#pragma unroll
        for (int t_ = 0; t_ < KS; ++t_) {
          const int cur = (s * KS + t_) & 1, nxt = cur ^ 1;
          // scalar select, no branch around the loads (see conv1d_f16s_kernel)
          ap = (more || s + 1 < S16 || t_ + 1 < KS) ? ap + a_step : next_base;
          if constexpr (ST2_WS_ABLATE != 6) {
            a_hi[nxt] = ap[0];
            a_lo[nxt] = ap[1];
          } else {
            a_hi[nxt] = a_hi[cur];
            a_lo[nxt] = a_lo[cur];
          }
          __builtin_amdgcn_sched_barrier(0x786);  // neither VMEM nor MFMA crosses: the prefetch stays a full k-step ahead
          const h8 ah = a_hi[cur], al = a_lo[cur];
          const h8* xp = xbuf + (2 * s) * XW + t_ * d.dil;
          h8 bh[TN], bl[TN];
#pragma unroll
          for (int j = 0; j < TN; ++j) {
            if constexpr (ST2_WS_ABLATE == 5) {
              bh[j] = ah;
              bl[j] = al;
            } else {
              bh[j] = xp[j * 32];
              bl[j] = xp[plane + j * 32];
            }
          }
          if constexpr (ST2_WS_ABLATE == 2) {
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[j][(s * KS + t_) & 15] += (float)(bh[j][0] + bl[j][1]) * (float)(ah[2] + al[3]);
          } else {
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh[j], ah, acc[j], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bl[j], ah, acc[j], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh[j], al, acc[j], 0, 0, 0);
          }
        }
      }
      if (SPC & 1) {  // odd step count: next chunk's step 0 reads set 0
        a_hi[0] = a_hi[1];
        a_lo[0] = a_lo[1];
      }
      WS_SYNC();
    }
    __builtin_amdgcn_s_setprio(0);
    WS_STAMP();
    if constexpr (ST2_WS_ABLATE == 4) {
      float sum = 0.f;
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) sum += acc[j][r];
      if (sum == 1234.5f) d.y[0] = sum;
    } else {
      st2_conv_epilogue<TN, WM, WN>(d, acc, b, m0, n0, wm, wn, l31, kg);
    }
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    WS_SYNC();  // end of the epilogue phase
  }
}

template <int KS, int CI_T, int WM, int WN, int TN, int PRO>
int launch_ws(const st2_conv_desc& d, hipStream_t s) {
  constexpr int BM = 32 * WM;
  constexpr int BN = 32 * TN * WN;
  const int XW = BN + (KS - 1) * d.dil;
  const int C_pad = (d.C_in + CI_T - 1) / CI_T * CI_T;
  const size_t smem = (size_t)4 * (CI_T / 8) * XW * 16 + (size_t)2 * C_pad * 32;
  ST2_REQUIRE(smem <= 160 * 1024, "st2_conv1d_f16s: tile needs %zu B of LDS (ks=%d dil=%d C_in=%d)", smem, KS, d.dil, d.C_in);
  ST2_REQUIRE(d.wq_cin_pad == C_pad, "st2_conv1d_f16s: packed weight has %d input channels, kernel needs %d", d.wq_cin_pad,
              C_pad);
  ST2_REQUIRE(d.wq_co_pad % BM == 0 && d.wq_co_pad >= d.C_out, "st2_conv1d_f16s: wq_co_pad=%d must be a multiple of %d "
              "covering C_out=%d", d.wq_co_pad, BM, d.C_out);
  if (d.part) ST2_REQUIRE(d.part_nt >= st2_cdiv(d.L_out, 128), "st2_conv1d_f16s: part_nt=%d < %d tiles", d.part_nt,
                          st2_cdiv(d.L_out, 128));
  ST2_REQUIRE(C_pad <= NPR, "st2_conv1d_f16s (warp-specialised): C_in=%d exceeds %d", d.C_in, NPR);
  // one workgroup per CU: the CU count and the kernel's LDS attribute are set up once per device (a process normally drives
  // one GPU, but nothing here may assume it: cf. the capacity cache of st2_lstm_coop.hip)
  static int num_cu_of[64] = {};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) {
    st2_set_error("st2_conv1d_f16s: cannot query the current device");
    return 1;
  }
  if (!num_cu_of[dev]) {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, dev) != hipSuccess || prop.multiProcessorCount <= 0) {
      st2_set_error("st2_conv1d_f16s: cannot query device %d", dev);
      return 1;
    }
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv1d_f16s_ws_kernel<KS, CI_T, WM, WN, TN, PRO>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    num_cu_of[dev] = prop.multiProcessorCount;
  }
  const int num_cu = num_cu_of[dev];
  TileGeom tg;
  tg.tiles_n = st2_cdiv(d.L_out, BN);
  tg.tiles_m = st2_cdiv(d.C_out, BM);
  const int64_t nt = (int64_t)tg.tiles_n * tg.tiles_m * d.B;
  ST2_REQUIRE(nt < (1ll << 31), "st2_conv1d_f16s: grid too large");
  tg.ntiles = (int)nt;
  const int grid = (int)std::min<int64_t>(nt, num_cu);
  hipLaunchKernelGGL((conv1d_f16s_ws_kernel<KS, CI_T, WM, WN, TN, PRO>), dim3(grid), dim3(NTW), smem, s, d,
                     st2_status_device_ptr(), tg
#ifdef ST2_WS_TIMELINE
                     , tl_buffer()
#endif
  );
  ST2_CHECK_LAUNCH("st2_conv1d_f16s (warp-specialised)");
  return 0;
}

// Same tile-by-C_out rule as st2f16s::launch_by_cout, for C_out <= 64 (st2f16s::ws_eligible).
template <int KS, int CI_T>
int launch_ws_by_cout(const st2_conv_desc& d, hipStream_t s) {
  if (d.C_out > 32) return launch_ws<KS, CI_T, 2, 2, 4, ST2_PRO_ADAIN_SNAKE>(d, s);  // 64 co x 256 l
  return launch_ws<KS, CI_T, 1, 4, 4, ST2_PRO_ADAIN_SNAKE>(d, s);                    // 32 co x 512 l
}

}  // namespace st2ws
