// Bidirectional LSTM recurrence (H = 256), cooperative multi-CU form.
//
// The single-CU kernel (st2_lstm.hip) streams the whole W_hh (1 MB per direction) from L2 on EVERY time step: the
// step is bound by one CU's L2 port (~15 us), 400 steps take 6.5 ms.  Here W_hh is REGISTER RESIDENT: a group of 8
// workgroups (one per CU) serves U utterances of one direction; workgroup `sl` owns hidden units [32 sl, 32 sl + 32),
// i.e. 128 of the 1024 gate rows, and keeps its 128 x 256 fp32 slice in VGPRs (128 per thread) for the whole
// sequence.  Per step a workgroup
//   1. multiplies its slice with the previous hidden state of its U utterances (LDS broadcast reads of h, 128 FMAs
//      per utterance per thread as 64 plain-encoded v_pk_fma_f32; thread = (unit, 32-wide k slice) holds all four gates of
//      its unit),
//   2. reduces the 8 k-slice partials through LDS, applies the gate non-linearities and updates c / h for its
//      32 units x U utterances (one thread each),
//   3. publishes its 32 x U new h values to a double-buffered exchange array in global memory and
//   4. hands them to the other 7 workgroups of the group, then reloads the full h into LDS.  Three hand-off forms
//      (XCH): 2 (default) "the data is the flag" -- every value is ONE 8-byte agent-scope store {step tag, value}
//      and each consumer thread re-reads its U granules until every tag equals the step (cdna_hip_programming.md
//      guideline 16, recipe R2): one fabric round trip per step, no counter, no cache maintenance; 0 = plain stores /
//      loads bracketed by agent-scope release / acquire fences around a monotonic counter (store-ack, counter,
//      reload: three round trips, 8.1 us / step measured); 1 = sc1 atomic stores / loads + the counter (9.2 us).
// The step costs the FMA time of the slice (~1 us for U = 8) plus the hand-off instead of a 1 MB stream.  Every
// spin is bounded: on a time-out the group raises status[0] (and ST2_STATUS_LSTM_TIMEOUT) and every workgroup leaves.
//
// Packed-sequence semantics are those of st2_lstm_bidir (outputs past `length` are zero, the reverse direction
// starts at t = length - 1); arithmetic per gate row is a fixed-order fp32 sum (inside a 32-slice the even and the odd k
// ascending, even + odd, then the 8 slices ascending), bitwise reproducible.
#include "st2_common.h"
#include <algorithm>
#include <atomic>

namespace {

constexpr int H = 256;
constexpr int NSL = 8;         // workgroups (hidden-unit slices) per group
constexpr int UNITS = H / NSL;  // 32 hidden units per workgroup
constexpr int SPIN_LIMIT_DEFAULT = 1 << 22;  // polls before a group gives up (~seconds); st2_lstm_coop_set_spin_limit

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

// SC1 = true: the exchanged hidden state travels as agent-scope (sc1) atomic stores / loads, which bypass the
// non-coherent cache levels, so the per-step hand-off needs no L2 write-back / invalidate fence -- only the counter.
// SC1 = false: plain stores / loads bracketed by agent-scope release / acquire fences (whole-L2 maintenance per step).
typedef unsigned long long gran_t;  // {tag << 32 | float bits}
typedef float f2 __attribute__((ext_vector_type(2)));

template <int U, int XCH>
__global__ __launch_bounds__(256) void lstm_coop_kernel(const float* __restrict__ G, int64_t g_bs, int g_cs,
                                                        const float* __restrict__ whh_t,  // [2][H][4H]
                                                        const int* __restrict__ lengths, int B, int N,
                                                        float* __restrict__ Y, int64_t y_bs, int y_cs,
                                                        int* __restrict__ status,   // [0] error flag
                                                        int* __restrict__ counters,  // [groups]
                                                        float* __restrict__ hx,     // [groups][2][U][H] (granules: x2)
                                                        int* gstatus,                // sticky status word (may be null)
                                                        int spin_limit) {
  constexpr bool SC1 = XCH == 1;
  constexpr bool GRAN = XCH == 2;
  __shared__ __attribute__((aligned(16))) float hs[U][H];
  __shared__ float part[U][4][NSL][UNITS];
  __shared__ int s_fail;

  const int tid = threadIdx.x;
  const int sl = blockIdx.x;                       // slice of hidden units
  const int blk = blockIdx.y;                      // utterance block
  const int dir = blockIdx.z;
  const int group = dir * gridDim.y + blk;
  const int unit = tid & 31;                       // hidden unit inside the slice (matvec role and update role)
  const int kq = tid >> 5;                         // matvec role: k slice [32 kq, 32 kq + 32)
  const int uu = tid >> 5;                         // update role: utterance inside the block (valid if < U)
  const int hu = sl * UNITS + unit;                // global hidden unit

  // ---- weight slice into registers: wg[g][j] = W_hh[g*H + hu][32 kq + 2j .. 2j + 1] -----------------------------------
  // Register pairs run along K, not across gates: the mat-vec below is v_pk_fma_f32 in its PLAIN encoding (weight pair x the
  // pair of consecutive h values a ds_read_b128 delivers in an aligned register pair), two IEEE fmas per instruction at half the
  // VALU time of the scalar form (1024 scalar fmas per thread and step were ~2 us of a 4.8 us step).
  // NOT {gate pair} x {one h broadcast to both halves}, which rounds 1-5 used: broadcasting an ODD element of the LDS vector
  // makes hipcc emit `v_pk_fma_f32 ... op_sel:[0,1,0]`, and on gfx950 a packed-f32 op whose op_sel takes the HIGH dword of
  // src1 for the low result lane returns a wrong low half in lanes 48-63 while another wave of the CU issues MFMAs in certain
  // cadences (any v_mfma_f32_16x16x32_f16 stream, the 32-column conv tile's dependent groups of three v_mfma_f32_32x32x16_f16)
  // -- the "BiLSTM is irreproducible next to narrow-tile convs" of round 5, root-caused in round 6 (tools/simd_hazard_repro.hip,
  // profiles/r06*_hazard.log).  tools/check_isa.py fails the build if such an encoding reappears.
  f2 wg[4][16];
  {
    const float* Wd = whh_t + (int64_t)dir * H * 4 * H;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const float* w0 = Wd + (int64_t)(kq * 32 + 2 * j) * 4 * H + hu;
      const float* w1 = w0 + 4 * H;
#pragma unroll
      for (int g = 0; g < 4; ++g) wg[g][j] = f2{w0[g * H], w1[g * H]};
    }
  }

  // ---- update-role state ------------------------------------------------------------------------------
  const int b = blk * U + uu;
  const bool live = uu < U && b < B;
  const int len = live ? (lengths ? min(lengths[b], N) : N) : 0;
  int maxlen = 0;  // steps this group runs = longest sequence in the block (same value in all 8 workgroups)
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const int bb = blk * U + u;
    if (bb < B) maxlen = max(maxlen, lengths ? min(lengths[bb], N) : N);
  }
  const float* Gb = G + (int64_t)(live ? b : 0) * g_bs + (int64_t)(dir * 4 * H + hu) * g_cs;
  float* Yb = Y + (int64_t)(live ? b : 0) * y_bs + (int64_t)(dir * H + hu) * y_cs;
  if (live)
    for (int t = len; t < N; ++t) Yb[t] = 0.f;  // pad_packed_sequence tail

  for (int e = tid; e < U * H; e += 256) (&hs[0][0])[e] = 0.f;
  if (tid == 0) s_fail = 0;
  float c = 0.f, h = 0.f;
  const int dt = dir == 0 ? 1 : -1;
  int t = dir == 0 ? 0 : len - 1;
  float gi = 0.f, gf = 0.f, gg = 0.f, go = 0.f;
  if (live && len > 0) {
    gi = Gb[(int64_t)(0 * H) * g_cs + t];
    gf = Gb[(int64_t)(1 * H) * g_cs + t];
    gg = Gb[(int64_t)(2 * H) * g_cs + t];
    go = Gb[(int64_t)(3 * H) * g_cs + t];
  }
  int* cnt = counters + group;
  float* hxg = hx + (int64_t)group * 2 * U * H * (GRAN ? 2 : 1);
  gran_t* gxg = reinterpret_cast<gran_t*>(hxg);
  __syncthreads();

  for (int s = 0; s < maxlen; ++s) {
    // prefetch the next step's projected inputs (update role)
    const bool act = live && s < len;
    const int tn = t + dt;
    float ni = 0.f, nf = 0.f, ng = 0.f, no = 0.f;
    if (live && s + 1 < len) {
      ni = Gb[(int64_t)(0 * H) * g_cs + tn];
      nf = Gb[(int64_t)(1 * H) * g_cs + tn];
      ng = Gb[(int64_t)(2 * H) * g_cs + tn];
      no = Gb[(int64_t)(3 * H) * g_cs + tn];
    }
    // 1. partial gate sums of this thread's (unit, k slice) for every utterance of the block: per gate an (even k, odd k)
    //    accumulator pair, added at the end (fixed order: k ascending within each parity, then even + odd)
#pragma unroll
    for (int u = 0; u < U; ++u) {
      f2 a[4] = {f2{0.f, 0.f}, f2{0.f, 0.f}, f2{0.f, 0.f}, f2{0.f, 0.f}};
      const float4* hp = reinterpret_cast<const float4*>(&hs[u][kq * 32]);
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const float4 hv = hp[q];  // the same address in all 32 lanes of a half-wave: LDS broadcast
        const f2 h01 = f2{hv.x, hv.y}, h23 = f2{hv.z, hv.w};
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          a[g] = __builtin_elementwise_fma(wg[g][2 * q + 0], h01, a[g]);
          a[g] = __builtin_elementwise_fma(wg[g][2 * q + 1], h23, a[g]);
        }
      }
#pragma unroll
      for (int g = 0; g < 4; ++g) part[u][g][kq][unit] = a[g].x + a[g].y;
    }
    __syncthreads();
    // 2. gate non-linearities and state update: thread = (unit, utterance uu)
    if (uu < U) {
      if (act) {
        float ai = 0.f, af = 0.f, ag = 0.f, ao = 0.f;
#pragma unroll
        for (int q = 0; q < NSL; ++q) {
          ai += part[uu][0][q][unit];
          af += part[uu][1][q][unit];
          ag += part[uu][2][q][unit];
          ao += part[uu][3][q][unit];
        }
        const float iv = sigmoidf_(gi + ai);
        const float fv = sigmoidf_(gf + af);
        const float gv = tanhf(gg + ag);
        const float ov = sigmoidf_(go + ao);
        c = fv * c + iv * gv;
        h = ov * tanhf(c);
        Yb[t] = h;
      }
      // 3. publish (a finished or absent utterance republishes its last state: nobody consumes it)
      if constexpr (GRAN) {
        const gran_t g = ((gran_t)(unsigned)(s + 1) << 32) | (gran_t)__float_as_uint(h);
        __hip_atomic_store(&gxg[((int64_t)(s & 1) * U + uu) * H + hu], g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      } else {
        float* dst = &hxg[((int64_t)(s & 1) * U + uu) * H + hu];
        if constexpr (SC1)
          __hip_atomic_store(dst, h, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else
          *dst = h;
      }
    }
    if constexpr (GRAN) {
      // 4. the data is the flag: thread = hidden unit index re-reads its U granules of this step's buffer until every
      //    tag carries the step (tags of the buffer's previous use are s - 1; the memset before the launch left 0).
      //    `part` is free to be rewritten only after every thread has read it: the sync below also covers that.
      const gran_t* src = gxg + (int64_t)(s & 1) * U * H + tid;
      float hv[U];
      int spins = 0;
      bool fail = false;
      for (;;) {
        bool ok = true;
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const gran_t g = __hip_atomic_load(src + u * H, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          hv[u] = __uint_as_float((unsigned)g);
          ok &= (unsigned)(g >> 32) == (unsigned)(s + 1);
        }
        if (__all(ok)) break;
        if (++spins > spin_limit) {  // wave-uniform
          fail = true;
          break;
        }
        __builtin_amdgcn_s_sleep(1);
      }
      if (fail) {
        if ((tid & 63) == 0) {
          __hip_atomic_store(status, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          st2_raise_status(gstatus, ST2_STATUS_LSTM_TIMEOUT);
          s_fail = 1;
        }
      } else {
#pragma unroll
        for (int u = 0; u < U; ++u) hs[u][tid] = hv[u];
      }
      __syncthreads();
      if (s_fail) return;
    } else {
    // 4. group hand-off: all stores of this workgroup complete -> release -> count -> wait -> acquire
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
      if constexpr (!SC1) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      __hip_atomic_fetch_add(cnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const int target = NSL * (s + 1);
      int spins = 0;
      while (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
        if (++spins > spin_limit || __hip_atomic_load(status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) {
          __hip_atomic_store(status, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          st2_raise_status(gstatus, ST2_STATUS_LSTM_TIMEOUT);
          s_fail = 1;
          break;
        }
        __builtin_amdgcn_s_sleep(2);
      }
      if constexpr (!SC1) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
    if (s_fail) return;
    // full new hidden state of the block's utterances -> LDS (thread = hidden unit index)
    float* src = hxg + (int64_t)(s & 1) * U * H;
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if constexpr (SC1)
        hs[u][tid] = __hip_atomic_load(&src[u * H + tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      else
        hs[u][tid] = src[u * H + tid];
    }
    }
    gi = ni; gf = nf; gg = ng; go = no;
    if (act) t = tn;
    __syncthreads();
  }
}

__global__ __launch_bounds__(256) void zero16_kernel(uint4* p, size_t n) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) p[i] = make_uint4(0u, 0u, 0u, 0u);
}

int g_spin_limit = SPIN_LIMIT_DEFAULT;
thread_local bool t_report_timeout = true;  // false inside st2_lstm_bidir_coop_recovering: scratch[0] alone carries the flag
int g_xch = 2;  // st2_lstm_coop_set_exchange(): 0 fences + counter (8.1 us/step), 1 sc1 + counter (9.2 us), 2 granules

size_t scratch_head(int groups) { return ((size_t)(1 + groups) * sizeof(int) + 255) / 256 * 256; }
// sized for the granule form (8 bytes per exchanged value) whatever form runs
size_t scratch_need(int groups, int U) { return scratch_head(groups) + (size_t)groups * 2 * U * H * sizeof(gran_t); }

template <int U, int XCH>
int launch_coop_as(const float* G, int64_t g_bs, int g_cs, const float* whh_t, const int* lengths, int B, int N,
                   float* Y, int64_t y_bs, int y_cs, void* scratch, size_t scratch_bytes, hipStream_t s) {
  const int nblk = st2_cdiv(B, U);
  const int groups = 2 * nblk;
  const size_t head = scratch_head(groups);
  const size_t need = scratch_need(groups, U);
  ST2_REQUIRE(scratch_bytes >= need, "st2_lstm_bidir_coop: scratch of %zu B, need %zu B", scratch_bytes, need);
  // Every workgroup of the launch must be resident at once (the groups spin on each other): ask the runtime how many
  // the device holds instead of assuming a CU count.  One query per (U, XCH) and DEVICE (a process may drive several
  // devices of different size); the cache entries are atomics, a racing first query just computes the same value twice.
  constexpr int MAX_DEV = 64;
  static std::atomic<int> capacity_of[MAX_DEV];  // 0 = not queried yet (zero-initialised)
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) {
    (void)hipGetLastError();
    st2_set_error("st2_lstm_bidir_coop: hipGetDevice failed");
    return 1;
  }
  int capacity = dev >= 0 && dev < MAX_DEV ? capacity_of[dev].load(std::memory_order_relaxed) : 0;
  if (capacity <= 0) {
    int per_cu = 0;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, dev) != hipSuccess ||
        hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, lstm_coop_kernel<U, XCH>, 256, 0) != hipSuccess) {
      (void)hipGetLastError();
      st2_set_error("st2_lstm_bidir_coop: occupancy query failed");
      return 1;
    }
    capacity = per_cu * prop.multiProcessorCount;
    if (dev >= 0 && dev < MAX_DEV && capacity > 0) capacity_of[dev].store(capacity, std::memory_order_relaxed);
  }
  // a CU-masked stream (st2_stream_create_cu_mask) owns only its CUs: the co-residency bound is theirs
  if (const int masked = st2_stream_cu_count(s)) {
    int num_cu = 0;
    if (hipDeviceGetAttribute(&num_cu, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && num_cu > 0)
      capacity = (int)((int64_t)capacity * std::min(masked, num_cu) / num_cu);
    else
      (void)hipGetLastError();
  }
  ST2_REQUIRE(groups * NSL <= capacity, "st2_lstm_bidir_coop: %d workgroups cannot be co-resident on this device / stream "
              "(capacity %d): use st2_lstm_bidir", groups * NSL, capacity);
  int* status = reinterpret_cast<int*>(scratch);
  int* counters = status + 1;
  float* hx = reinterpret_cast<float*>(reinterpret_cast<char*>(scratch) + head);
  // status, counters and -- granule form -- every tag start at zero on EVERY call (a tag left by an earlier call would
  // otherwise match).  A KERNEL, not hipMemsetAsync: recorded into a hipGraph, the memset node zeroes on the first replay and
  // writes an 8-byte pointer-like pattern over the head of the buffer on every later one (ROCm 7.2, tools/stress.py lstm_graph,
  // profiles/r05/r05c_lstm_graph.log) -- rounds 1-4 never looked at scratch[0] after a replay; round 5's in-stream recovery does.
  {
    const size_t words = (XCH == 2 ? need : head) / 16;  // both are multiples of 256 bytes; scratch is 256-byte aligned
    hipLaunchKernelGGL(zero16_kernel, dim3((unsigned)((words + 255) / 256)), dim3(256), 0, s, reinterpret_cast<uint4*>(scratch), words);
    ST2_CHECK_LAUNCH("st2_lstm_bidir_coop (scratch clear)");
  }
  hipLaunchKernelGGL((lstm_coop_kernel<U, XCH>), dim3(NSL, nblk, 2), dim3(256), 0, s, G, g_bs, g_cs, whh_t, lengths,
                     B, N, Y, y_bs, y_cs, status, counters, hx, t_report_timeout ? st2_status_device_ptr() : nullptr, g_spin_limit);
  ST2_CHECK_LAUNCH("st2_lstm_bidir_coop");
  return 0;
}

template <int U>
int launch_coop(const float* G, int64_t g_bs, int g_cs, const float* whh_t, const int* lengths, int B, int N, float* Y,
                int64_t y_bs, int y_cs, void* scratch, size_t scratch_bytes, hipStream_t s) {
  switch (g_xch) {
    case 0:
      return launch_coop_as<U, 0>(G, g_bs, g_cs, whh_t, lengths, B, N, Y, y_bs, y_cs, scratch, scratch_bytes, s);
    case 1:
      return launch_coop_as<U, 1>(G, g_bs, g_cs, whh_t, lengths, B, N, Y, y_bs, y_cs, scratch, scratch_bytes, s);
    default:
      return launch_coop_as<U, 2>(G, g_bs, g_cs, whh_t, lengths, B, N, Y, y_bs, y_cs, scratch, scratch_bytes, s);
  }
}

int g_block = 0;  // st2_lstm_coop_set_block(): 0 = by batch size, else 1 / 2 / 4 / 8 (measurement hook)
constexpr int MAX_COOP_WG = 256;  // workgroups of one cooperative launch (the occupancy query at launch has the last word)
// Utterances per cooperative group.  A group's step = its mat-vec (U x 128 fmas per thread) + one cross-CU exchange, so
// fewer utterances per group means a shorter step as long as the groups still fit the chip side by side: at B = 32 blocks
// of 4 (128 workgroups) run 2.9 us / step against 4.35 us for blocks of 8 (64 workgroups) -- profiles/archive/r03/r03h_probe_lstm.log.
int block_size(int B) {
  if (g_block < 0) return 0;  // measurement hook: no cooperative launches at all (callers take st2_lstm_bidir)
  if (g_block == 1 || g_block == 2 || g_block == 4 || g_block == 8)
    return (2 * st2_cdiv(B, g_block) * NSL <= MAX_COOP_WG) ? g_block : 0;
  if (B <= 1) return 1;
  if (2 * st2_cdiv(B, 4) * NSL <= MAX_COOP_WG / 2) return 4;  // up to 32 utterances: 128 workgroups
  return B > 48 ? 0 : 8;
}

}  // namespace

extern "C" int st2_lstm_coop_set_exchange(int mode) {
  if (mode < 0 || mode > 2) {
    st2_set_error("st2_lstm_coop_set_exchange: mode %d (0 fences, 1 sc1, 2 granules)", mode);
    return 1;
  }
  g_xch = mode;
  return 0;
}

extern "C" int st2_lstm_coop_set_block(int utterances) {
  g_block = utterances;
  return 0;
}

extern "C" int st2_lstm_coop_set_spin_limit(int polls) {
  g_spin_limit = polls > 0 ? polls : SPIN_LIMIT_DEFAULT;
  return 0;
}

extern "C" int64_t st2_lstm_coop_scratch_bytes(int32_t B) {
  const int U = block_size(B);
  if (U == 0) return 0;  // batch too large for one co-resident launch: use st2_lstm_bidir
  return (int64_t)scratch_need(2 * st2_cdiv(B, U), U);
}

int st2_lstm_coop_launch(const float* G, int64_t g_bs, int32_t g_cs, const float* whh_t, const int32_t* lengths, int32_t B,
                         int32_t Hn, int32_t N, float* Y, int64_t y_bs, int32_t y_cs, void* scratch, int64_t scratch_bytes,
                         void* stream, bool report_timeout) {
  t_report_timeout = report_timeout;
  const int rc = st2_lstm_bidir_coop(G, g_bs, g_cs, whh_t, lengths, B, Hn, N, Y, y_bs, y_cs, scratch, scratch_bytes, stream);
  t_report_timeout = true;
  return rc;
}

extern "C" int st2_lstm_bidir_coop(const float* G, int64_t g_bs, int32_t g_cs, const float* whh_t,
                                   const int32_t* lengths, int32_t B, int32_t Hn, int32_t N, float* Y, int64_t y_bs,
                                   int32_t y_cs, void* scratch, int64_t scratch_bytes, void* stream) {
  ST2_REQUIRE(G && whh_t && Y && scratch && B > 0 && N > 0, "st2_lstm_bidir_coop: bad arguments");
  ST2_REQUIRE(Hn == H, "st2_lstm_bidir_coop: hidden size %d unsupported (built for 256)", Hn);
  ST2_REQUIRE((reinterpret_cast<uintptr_t>(scratch) & 255) == 0, "st2_lstm_bidir_coop: scratch must be 256-byte aligned");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const int* len = reinterpret_cast<const int*>(lengths);
  switch (block_size(B)) {
    case 8:
      return launch_coop<8>(G, g_bs, g_cs, whh_t, len, B, N, Y, y_bs, y_cs, scratch, (size_t)scratch_bytes, s);
    case 4:
      return launch_coop<4>(G, g_bs, g_cs, whh_t, len, B, N, Y, y_bs, y_cs, scratch, (size_t)scratch_bytes, s);
    case 2:
      return launch_coop<2>(G, g_bs, g_cs, whh_t, len, B, N, Y, y_bs, y_cs, scratch, (size_t)scratch_bytes, s);
    case 1:
      return launch_coop<1>(G, g_bs, g_cs, whh_t, len, B, N, Y, y_bs, y_cs, scratch, (size_t)scratch_bytes, s);
    default:
      st2_set_error("st2_lstm_bidir_coop: batch %d needs more co-resident workgroups than one launch may hold", B);
      return 1;
  }
}
