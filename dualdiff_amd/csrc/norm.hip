// GroupNorm(+SiLU) in NHWC and LayerNorm for gfx950 — HBM-bound kernels:
// 16-byte coalesced channel-contiguous loads, fp32 statistics, wave shuffles + LDS for the
// reductions, one read for statistics + one read/one write for the apply pass (the second
// read is served from L2 / Infinity Cache at these tensor sizes).
#include "dd_common.h"
#include <atomic>

namespace {

constexpr int GN_THREADS = 256;
constexpr int GN_MAX_SPLIT = 64;
constexpr int GN_MAX_C = 4096;

struct GnParams {
  const void* x1; const void* x2; int c1, c2, c;
  const void* gamma; const void* beta; void* y;
  int m, hw, groups, cpg; float eps; int silu;
  float* ws;          // [m][nsplit][groups][2]
  int nsplit, pix_per_split;
  // split-K source (dd_groupnorm_splitk, single-launch form only): the input is NOT a tensor but the fp32 partial slabs
  // of a split-K dd_gemm whose reduce launch this kernel replaces: x[r][c] = T( sum_z part[z][r][c] + bias[c]
  // + rowvec[r / hw][c] + res[r][c] ), the arithmetic (and rounding) of dd_splitk_reduce_kernel's epilogue at alpha = 1.
  const float* sk_part; int sk_nsplit; int64_t sk_rows;
  const void* sk_bias; const void* sk_rowvec; int sk_ld_rowvec;
  const void* sk_res; int64_t sk_ldres;
  void* sk_xout;      // optional: the reduced tensor itself ([m * hw][c]), for consumers besides this GroupNorm
};

// thread -> (pixel lane, channel vector) mapping shared by both passes
struct GnMap { int cv_count, pl_count, cv, pl; bool active; };
__device__ __forceinline__ GnMap gn_map(int c) {
  GnMap mp;
  mp.cv_count = c >> 3;
  if (mp.cv_count >= GN_THREADS) {       // several channel vectors per thread, one pixel lane
    mp.pl_count = 1; mp.cv = threadIdx.x; mp.pl = 0; mp.active = true;
  } else {
    mp.pl_count = GN_THREADS / mp.cv_count;
    mp.cv = threadIdx.x % mp.cv_count;
    mp.pl = threadIdx.x / mp.cv_count;
    mp.active = mp.pl < mp.pl_count;
  }
  return mp;
}

template <typename T>
__device__ __forceinline__ const T* gn_src(const GnParams& p, int64_t row, int ch) {
  return ch < p.c1 ? reinterpret_cast<const T*>(p.x1) + row * p.c1 + ch
                   : reinterpret_cast<const T*>(p.x2) + row * p.c2 + (ch - p.c1);
}

constexpr int GN_UNROLL = 4;

// Variance by E[(x-p)^2] - E[x-p]^2 around a per-(instance, group) PIVOT p = the group's first element
// (pixel 0, first channel): with p within a few standard deviations of the mean the subtraction loses
// nothing, whereas the textbook E[x^2] - mean^2 loses log2(mean^2 / var) bits — real SD checkpoints
// have groups whose |mean| is ~100x their std.  One extra scalar load per group, no extra pass.
template <typename T>
__device__ __forceinline__ float gn_pivot(const GnParams& p, int inst, int g) {
  return (float)*gn_src<T>(p, (int64_t)inst * p.hw, g * p.cpg);
}

// split-K source: 8 consecutive channels of one row, reduced over the slabs in slice order + the GEMM epilogue, rounded
// to T exactly as dd_splitk_reduce_kernel stores them (bias, per-instance vector, residual; alpha = 1, no activation)
template <typename T>
__device__ __forceinline__ u32x4 gn_sk_load8(const GnParams& p, int64_t row, int ch, int inst) {
  float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  for (int z = 0; z < p.sk_nsplit; ++z) {
    const float* src = p.sk_part + ((int64_t)z * p.sk_rows + row) * p.c + ch;
    const f32x4 a = *reinterpret_cast<const f32x4*>(src);
    const f32x4 b = *reinterpret_cast<const f32x4*>(src + 4);
    v[0] += a[0]; v[1] += a[1]; v[2] += a[2]; v[3] += a[3];
    v[4] += b[0]; v[5] += b[1]; v[6] += b[2]; v[7] += b[3];
  }
  float b[8];
  if (p.sk_bias) {
    dd_unpack8<T>(dd_ld16(reinterpret_cast<const T*>(p.sk_bias) + ch), b);
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] += b[i];
  }
  if (p.sk_rowvec) {
    dd_unpack8<T>(dd_ld16(reinterpret_cast<const T*>(p.sk_rowvec) + (int64_t)inst * p.sk_ld_rowvec + ch), b);
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] += b[i];
  }
  if (p.sk_res) {
    dd_unpack8<T>(dd_ld16(reinterpret_cast<const T*>(p.sk_res) + row * p.sk_ldres + ch), b);
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] += b[i];
  }
  return dd_pack8<T>(v);
}

template <typename T>
__global__ __launch_bounds__(GN_THREADS)
void dd_gn_stats_kernel(const GnParams p) {
  // Deterministic (fixed-order) reduction: every thread publishes the partial sums of the (at most
  // two) groups its 8-channel vector touches; one thread per group then adds, in a fixed order, only
  // the threads whose vector can overlap that group.
  __shared__ float s_a0[GN_THREADS], s_b0[GN_THREADS], s_a1[GN_THREADS], s_b1[GN_THREADS];
  __shared__ int s_g0[GN_THREADS], s_g1[GN_THREADS];
  const int split = blockIdx.x, inst = blockIdx.y;
  const GnMap mp = gn_map(p.c);
  const int p0 = split * p.pix_per_split;
  const int p1 = min(p.hw, p0 + p.pix_per_split);
  float gsum = 0.f, gsq = 0.f;          // owned by thread g < groups
  for (int cv0 = 0; cv0 < mp.cv_count; cv0 += GN_THREADS) {
    const int cv = cv0 + mp.cv;
    int g0 = -1, g1 = -1;
    float a0 = 0.f, b0 = 0.f, a1 = 0.f, b1 = 0.f;
    if (mp.active && cv < mp.cv_count) {
      const int ch = cv << 3;
      float s[8], ss[8], pv[8];
      {
        const int ga = ch / p.cpg, gb = min((ch + 7) / p.cpg, p.groups - 1);
        const float pa = gn_pivot<T>(p, inst, ga), pb = gn_pivot<T>(p, inst, gb);
#pragma unroll
        for (int e = 0; e < 8; ++e) { s[e] = 0.f; ss[e] = 0.f; pv[e] = (ch + e) / p.cpg == ga ? pa : pb; }
      }
      int px = p0 + mp.pl;
      for (; px + (GN_UNROLL - 1) * mp.pl_count < p1; px += GN_UNROLL * mp.pl_count) {
        u32x4 v[GN_UNROLL];
#pragma unroll
        for (int u = 0; u < GN_UNROLL; ++u)
          v[u] = dd_ld16(gn_src<T>(p, (int64_t)inst * p.hw + px + u * mp.pl_count, ch));
#pragma unroll
        for (int u = 0; u < GN_UNROLL; ++u) {
          float f[8];
          dd_unpack8<T>(v[u], f);
#pragma unroll
          for (int e = 0; e < 8; ++e) { const float d = f[e] - pv[e]; s[e] += d; ss[e] += d * d; }
        }
      }
      for (; px < p1; px += mp.pl_count) {
        float f[8];
        dd_unpack8<T>(dd_ld16(gn_src<T>(p, (int64_t)inst * p.hw + px, ch)), f);
#pragma unroll
        for (int e = 0; e < 8; ++e) { const float d = f[e] - pv[e]; s[e] += d; ss[e] += d * d; }
      }
      g0 = ch / p.cpg;
      const int gl = (ch + 7) / p.cpg;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const bool first = (ch + e) / p.cpg == g0;
        a0 += first ? s[e] : 0.f;  b0 += first ? ss[e] : 0.f;
        a1 += first ? 0.f : s[e];  b1 += first ? 0.f : ss[e];
      }
      g1 = gl != g0 ? gl : -1;
    }
    s_g0[threadIdx.x] = g0; s_a0[threadIdx.x] = a0; s_b0[threadIdx.x] = b0;
    s_g1[threadIdx.x] = g1; s_a1[threadIdx.x] = a1; s_b1[threadIdx.x] = b1;
    __syncthreads();
    if (threadIdx.x < p.groups) {
      const int g = threadIdx.x;
      // channel vectors that can overlap group g, clipped to this pass
      const int lo = max((g * p.cpg) >> 3, cv0);
      const int hi = min(((g + 1) * p.cpg - 1) >> 3, min(cv0 + GN_THREADS, mp.cv_count) - 1);
      for (int pl = 0; pl < mp.pl_count; ++pl) {
        for (int c = lo; c <= hi; ++c) {
          const int t = mp.cv_count >= GN_THREADS ? (c - cv0) : pl * mp.cv_count + c;
          if (s_g0[t] == g) { gsum += s_a0[t]; gsq += s_b0[t]; }
          if (s_g1[t] == g) { gsum += s_a1[t]; gsq += s_b1[t]; }
        }
      }
    }
    __syncthreads();
  }
  if (threadIdx.x < p.groups) {
    float* dst = p.ws + (((int64_t)inst * p.nsplit + split) * p.groups + threadIdx.x) * 2;
    dst[0] = gsum;
    dst[1] = gsq;
  }
}

template <typename T>
__global__ __launch_bounds__(GN_THREADS)
void dd_gn_apply_kernel(const GnParams p) {
  __shared__ float s_mean[64], s_rstd[64];
  __shared__ float s_ps[GN_THREADS], s_pq[GN_THREADS];
  const int split = blockIdx.x, inst = blockIdx.y;
  const GnMap mp = gn_map(p.c);
  const int p0 = split * p.pix_per_split;
  const int p1 = min(p.hw, p0 + p.pix_per_split);
  // The first batch of pixel vectors does not depend on the statistics: issue its loads BEFORE the
  // partial sums are fetched and combined, so the two memory round trips overlap.
  // this thread's group pivot (threads < groups): loaded first so that it is back long before the combine
  const float my_pivot = (int)threadIdx.x < p.groups ? gn_pivot<T>(p, inst, threadIdx.x) : 0.f;
  const int cv_first = mp.cv;
  const int ch_first = min(cv_first, mp.cv_count - 1) << 3;
  u32x4 pre[GN_UNROLL];
  const bool pre_ok = mp.active && cv_first < mp.cv_count && p0 + mp.pl + (GN_UNROLL - 1) * mp.pl_count < p1;
  if (pre_ok) {
#pragma unroll
    for (int u = 0; u < GN_UNROLL; ++u)
      pre[u] = dd_ld16(gn_src<T>(p, (int64_t)inst * p.hw + p0 + mp.pl + u * mp.pl_count, ch_first));
  }
  {
    // combine the per-split partial sums: all 256 threads fetch in parallel (chunk-strided), then
    // one thread per group adds the chunk sums in a fixed order (bit-reproducible).
    const int g = threadIdx.x % p.groups;
    const int chunk = threadIdx.x / p.groups;
    const int nchunk = GN_THREADS / p.groups;
    float s = 0.f, ss = 0.f;
    if (chunk < nchunk) {
      const float* src = p.ws + ((int64_t)inst * p.nsplit * p.groups + g) * 2;
      for (int i = chunk; i < p.nsplit; i += nchunk) {
        s += src[(int64_t)i * p.groups * 2];
        ss += src[(int64_t)i * p.groups * 2 + 1];
      }
    }
    s_ps[threadIdx.x] = s;
    s_pq[threadIdx.x] = ss;
    __syncthreads();
    if (threadIdx.x < p.groups) {
      float a = 0.f, b = 0.f;
      for (int c = 0; c < nchunk; ++c) { a += s_ps[c * p.groups + threadIdx.x]; b += s_pq[c * p.groups + threadIdx.x]; }
      const float inv_n = 1.0f / ((float)p.hw * (float)p.cpg);
      const float dm = a * inv_n;                        // mean - pivot
      const float var = fmaxf(b * inv_n - dm * dm, 0.f);
      s_mean[threadIdx.x] = my_pivot + dm;
      s_rstd[threadIdx.x] = rsqrtf(var + p.eps);
    }
  }
  __syncthreads();
  if (!mp.active) return;
  for (int cv = mp.cv; cv < mp.cv_count; cv += GN_THREADS) {
    const int ch = cv << 3;
    float ga[8], be[8], sc[8], sh[8];
    dd_unpack8<T>(dd_ld16(reinterpret_cast<const T*>(p.gamma) + ch), ga);
    dd_unpack8<T>(dd_ld16(reinterpret_cast<const T*>(p.beta) + ch), be);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int g = (ch + e) / p.cpg;
      sc[e] = s_rstd[g] * ga[e];
      sh[e] = be[e] - s_mean[g] * sc[e];
    }
    auto apply_store = [&](u32x4 v, int64_t row) {
      float f[8];
      dd_unpack8<T>(v, f);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float y = f[e] * sc[e] + sh[e];
        f[e] = p.silu ? dd_silu_f(y) : y;
      }
      dd_st16(reinterpret_cast<T*>(p.y) + row * p.c + ch, dd_pack8<T>(f));
    };
    int px = p0 + mp.pl;
    if (cv == cv_first && pre_ok) {                    // the prefetched batch
#pragma unroll
      for (int u = 0; u < GN_UNROLL; ++u) apply_store(pre[u], (int64_t)inst * p.hw + px + u * mp.pl_count);
      px += GN_UNROLL * mp.pl_count;
    }
    for (; px + (GN_UNROLL - 1) * mp.pl_count < p1; px += GN_UNROLL * mp.pl_count) {
      u32x4 v[GN_UNROLL];
#pragma unroll
      for (int u = 0; u < GN_UNROLL; ++u)
        v[u] = dd_ld16(gn_src<T>(p, (int64_t)inst * p.hw + px + u * mp.pl_count, ch));
#pragma unroll
      for (int u = 0; u < GN_UNROLL; ++u) apply_store(v[u], (int64_t)inst * p.hw + px + u * mp.pl_count);
    }
    for (; px < p1; px += mp.pl_count)
      apply_store(dd_ld16(gn_src<T>(p, (int64_t)inst * p.hw + px, ch)), (int64_t)inst * p.hw + px);
  }
}

// (A cooperative ONE-launch form for the big images — statistics, grid barrier, apply — was built in round 3 and
//  measured 2 % slower on the step, 0.4 % slower in the single-stream decoder alone (profiles/r03_gn_coop_ab.txt,
//  r04_experiments.txt #4: the barrier costs what the second launch cost); removed in round 5.)

// ---- single-launch GroupNorm for slabs that fit the register file --------------------------------
// One block owns (instance, `cpb` channels = whole groups) and keeps its hw x cpb slab in registers
// (<= NVMAX 16-B vectors per thread): one HBM read, statistics through a fixed-order LDS reduction
// (bit-reproducible), normalise + SiLU + store.  Used from the 14x25 level
// down, where the two-launch path is pure launch + latency; 1024-thread blocks for the larger slabs
// so that the per-thread VALU chain stays short.
template <typename T, int THREADS, int NVMAX, bool SK = false>
__global__ __launch_bounds__(THREADS)
void dd_gn_fused_kernel(const GnParams p, int cpb, int vpp, int plc, int nv, int kred) {
  __shared__ float s_a0[THREADS], s_a1[THREADS], s_q0[THREADS], s_q1[THREADS];
  __shared__ float s_mean[64], s_rstd[64], s_piv[64];
  const int inst = blockIdx.y;
  const int t = threadIdx.x;
  const bool active = t < vpp * plc;
  const int cv = t % vpp, pl = t / vpp;
  const int gpb = cpb / p.cpg;
  const int lch = cv * 8;                               // channel inside the block
  const int g0 = lch / p.cpg;                           // local group of the vector's first channel
  const int g1 = min(g0 + 1, gpb - 1);
  const int nfirst = min(8, (g0 + 1) * p.cpg - lch);    // channels of the vector that belong to g0
  // every lane loads (clamped address) so all nv loads are in flight at once; lanes / pixels
  // outside the slab are masked out of the statistics and never stored
  const int ch = blockIdx.x * cpb + min(cv, vpp - 1) * 8;
  u32x4 raw[NVMAX];
#pragma unroll
  for (int i = 0; i < NVMAX; ++i)
    if (i < nv) {
      const int64_t row = (int64_t)inst * p.hw + min(pl + i * plc, p.hw - 1);
      if constexpr (SK) {
        raw[i] = gn_sk_load8<T>(p, row, ch, inst);
        if (p.sk_xout && active && pl + i * plc < p.hw) dd_st16(reinterpret_cast<T*>(p.sk_xout) + row * p.c + ch, raw[i]);
      } else {
        raw[i] = dd_ld16(gn_src<T>(p, row, ch));
      }
    }
  const u32x4 graw = dd_ld16(reinterpret_cast<const T*>(p.gamma) + ch);
  const u32x4 braw = dd_ld16(reinterpret_cast<const T*>(p.beta) + ch);
  const float inv_n = 1.0f / ((float)p.hw * (float)p.cpg);

  // statistics in one pass (sum, sum of squares — the arithmetic of the two-launch path), reduced in
  // a fixed order (2 barriers, bit-reproducible).
  float a0 = 0.f, a1 = 0.f, q0 = 0.f, q1 = 0.f;
  auto pivot = [&](int g) -> float {
    if constexpr (SK) {        // the group's first element of pixel 0, reduced like any other (8-channel aligned vector)
      const int pc = g * p.cpg;
      float f[8];
      dd_unpack8<T>(gn_sk_load8<T>(p, (int64_t)inst * p.hw, pc & ~7, inst), f);
      float r = f[0];
#pragma unroll
      for (int e = 1; e < 8; ++e) r = (pc & 7) == e ? f[e] : r;
      return r;
    } else {
      return gn_pivot<T>(p, inst, g);
    }
  };
  const float piv0 = pivot(blockIdx.x * gpb + g0);
  const float piv1 = pivot(blockIdx.x * gpb + g1);
  if (t < gpb) s_piv[t] = pivot(blockIdx.x * gpb + t);     // for the combine (no late global load)
#pragma unroll
  for (int i = 0; i < NVMAX; ++i) {
    if (i < nv) {
      float f[8];
      dd_unpack8<T>(raw[i], f);
      const float keep = pl + i * plc < p.hw ? 1.f : 0.f;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float v = keep * (f[e] - (e < nfirst ? piv0 : piv1));
        if (e < nfirst) { a0 += v; q0 += v * v; } else { a1 += v; q1 += v * v; }
      }
    }
  }
  s_a0[t] = active ? a0 : 0.f; s_a1[t] = active ? a1 : 0.f;
  s_q0[t] = active ? q0 : 0.f; s_q1[t] = active ? q1 : 0.f;
  __syncthreads();
  {
    // one wave per group: lanes stride over the (pixel lane, channel vector) shares that can touch
    // the group, then a fixed-order wave reduction
    const int wave = t >> 6, lane = t & 63;
    for (int g = wave; g < gpb; g += THREADS / 64) {
      const int lo = (g * p.cpg) >> 3, hi = ((g + 1) * p.cpg - 1) >> 3;
      const int nvec = hi - lo + 1;
      const int n = plc * nvec;
      float sum = 0.f, sq = 0.f;
      for (int idx = lane; idx < n; idx += 64) {
        const int l = idx / nvec;
        const int c = lo + (idx - l * nvec);
        const int src = l * vpp + c;
        const int cg0 = (c * 8) / p.cpg;
        if (cg0 == g) { sum += s_a0[src]; sq += s_q0[src]; }
        else if (cg0 + 1 == g) { sum += s_a1[src]; sq += s_q1[src]; }
      }
      sum = dd_wave_sum(sum);
      sq = dd_wave_sum(sq);
      if (lane == 0) {
        const float dm = sum * inv_n;                    // mean - pivot
        s_mean[g] = s_piv[g] + dm;
        s_rstd[g] = rsqrtf(fmaxf(sq * inv_n - dm * dm, 0.f) + p.eps);
      }
    }
  }
  __syncthreads();
  const float mean0 = s_mean[g0], mean1 = s_mean[g1];
  const float rstd0 = s_rstd[g0], rstd1 = s_rstd[g1];
  if (!active) return;
  float ga[8], be[8], sc[8], sh[8];
  dd_unpack8<T>(graw, ga);
  dd_unpack8<T>(braw, be);
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const float mu = e < nfirst ? mean0 : mean1, rs = e < nfirst ? rstd0 : rstd1;
    sc[e] = rs * ga[e];
    sh[e] = be[e] - mu * sc[e];
  }
#pragma unroll
  for (int i = 0; i < NVMAX; ++i) {
    const int px = pl + i * plc;
    if (i < nv && px < p.hw) {
      float f[8];
      dd_unpack8<T>(raw[i], f);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float y = f[e] * sc[e] + sh[e];
        f[e] = p.silu ? dd_silu_f(y) : y;
      }
      dd_st16(reinterpret_cast<T*>(p.y) + ((int64_t)inst * p.hw + px) * p.c + ch, dd_pack8<T>(f));
    }
  }
}

constexpr int GNF_NV_SMALL = 8;      // 256-thread blocks
constexpr int GNF_NV_BIG = 8;        // 1024-thread blocks (<= 128 VGPRs): up to 8 vectors -> 28x50 slabs of 40 channels fit

// picks channels-per-block (whole groups, a multiple of 8, >= 40 channels) and the block size so
// that a block's slab fits the register budget; false -> use the two-launch path
bool gn_fused_plan(int hw, int c, int groups, int* cpb, int* vpp, int* plc, int* nv, int* kred, int* threads) {
  const int cpg = c / groups;
  for (int gpb = 1; gpb <= groups && gpb <= 64; ++gpb) {
    if (groups % gpb) continue;
    const int cb = cpg * gpb;
    if (cb % 8 || cb < 40) continue;
    const int v = cb / 8;
    // 256-thread blocks while the slab needs <= 4 vectors per thread, else 1024 threads (short
    // per-thread chain), else 256 threads up to the register budget
    // DD_GN_BIG_CAP: vectors per thread allowed in the 1024-thread form.  4 (default): 28x50 images take the
    // two-launch path; 8: one launch for 28x50 too — measured SLOWER (96 blocks of 1024 threads stream 112 KB
    // each: 19.6 us per GroupNorm against 7.0 + 9.3 us for stats + apply over all CUs), kept for experiments.
    static const int big_cap = getenv("DD_GN_BIG_CAP") ? atoi(getenv("DD_GN_BIG_CAP")) : 4;
    const int try_th[3] = {256, 1024, 256};
    const int try_cap[3] = {4, big_cap < GNF_NV_BIG ? big_cap : GNF_NV_BIG, GNF_NV_SMALL};
    for (int pass = 0; pass < 3; ++pass) {
      const int th = try_th[pass];
      if (v > th) continue;
      int pl = th / v;
      if (pl > hw) pl = hw;
      const int n = (hw + pl - 1) / pl;
      if (n > try_cap[pass]) continue;
      *cpb = cb; *vpp = v; *plc = pl; *nv = n; *kred = 0; *threads = th;
      return true;
    }
    return false;          // the smallest legal channel chunk does not fit: larger ones will not either
  }
  return false;
}

// ---- LayerNorm: one wave per row, row held in registers, two-pass (mean, centred var) ----
constexpr int LN_MAXV = 4;   // up to 4 x (64 lanes x 8) = 2048 channels

template <typename T>
__global__ __launch_bounds__(256)
void dd_layernorm_kernel(const T* __restrict__ x, const T* __restrict__ gamma,
                         const T* __restrict__ beta, T* __restrict__ y,
                         int64_t rows, int c, float eps) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int nv = c >> 3;
  float f[LN_MAXV][8];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < LN_MAXV; ++i) {
    const int v = lane + i * 64;
    if (v < nv) {
      dd_unpack8<T>(dd_ld16(x + row * c + v * 8), f[i]);
#pragma unroll
      for (int e = 0; e < 8; ++e) s += f[i][e];
    }
  }
  const float mean = dd_wave_sum(s) / (float)c;
  float ss = 0.f;
#pragma unroll
  for (int i = 0; i < LN_MAXV; ++i) {
    const int v = lane + i * 64;
    if (v < nv) {
#pragma unroll
      for (int e = 0; e < 8; ++e) { const float d = f[i][e] - mean; ss += d * d; }
    }
  }
  const float rstd = rsqrtf(dd_wave_sum(ss) / (float)c + eps);
#pragma unroll
  for (int i = 0; i < LN_MAXV; ++i) {
    const int v = lane + i * 64;
    if (v < nv) {
      float ga[8], be[8], o[8];
      dd_unpack8<T>(dd_ld16(gamma + v * 8), ga);
      dd_unpack8<T>(dd_ld16(beta + v * 8), be);
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = (f[i][e] - mean) * rstd * ga[e] + be[e];
      dd_st16(y + row * c + v * 8, dd_pack8<T>(o));
    }
  }
}

// LayerNorm for c = 40 * LPR (320 / 640 / 1280, the widths of this network): LPR lanes share a row,
// five 16-B vectors per lane, so a wave normalises 64 / LPR rows with all its loads in flight at
// once and a log2(LPR)-step reduction.  Same two-pass arithmetic as the generic kernel.
template <typename T, int LPR>
__global__ __launch_bounds__(256)
void dd_layernorm_sub_kernel(const T* __restrict__ x, const T* __restrict__ gamma,
                             const T* __restrict__ beta, T* __restrict__ y,
                             int64_t rows, float eps) {
  constexpr int VPL = 5;
  constexpr int C = LPR * VPL * 8;
  constexpr int RPW = 64 / LPR;
  const int lane = threadIdx.x & 63;
  const int sub = lane % LPR;
  const int64_t row = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * RPW + lane / LPR;
  const bool ok = row < rows;
  const int64_t r = ok ? row : rows - 1;             // keep every lane in the shuffles
  u32x4 raw[VPL];
#pragma unroll
  for (int i = 0; i < VPL; ++i) raw[i] = dd_ld16(x + r * C + (sub + i * LPR) * 8);
  float f[VPL][8];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < VPL; ++i) {
    dd_unpack8<T>(raw[i], f[i]);
#pragma unroll
    for (int e = 0; e < 8; ++e) s += f[i][e];
  }
#pragma unroll
  for (int o = LPR / 2; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  const float mean = s * (1.0f / (float)C);
  float ss = 0.f;
#pragma unroll
  for (int i = 0; i < VPL; ++i)
#pragma unroll
    for (int e = 0; e < 8; ++e) { const float d = f[i][e] - mean; ss += d * d; }
#pragma unroll
  for (int o = LPR / 2; o > 0; o >>= 1) ss += __shfl_xor(ss, o, 64);
  const float rstd = rsqrtf(ss * (1.0f / (float)C) + eps);
  if (!ok) return;
#pragma unroll
  for (int i = 0; i < VPL; ++i) {
    const int v = sub + i * LPR;
    float ga[8], be[8], o[8];
    dd_unpack8<T>(dd_ld16(gamma + v * 8), ga);
    dd_unpack8<T>(dd_ld16(beta + v * 8), be);
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = (f[i][e] - mean) * rstd * ga[e] + be[e];
    dd_st16(y + row * C + v * 8, dd_pack8<T>(o));
  }
}

template <typename T>
bool launch_layernorm_sub(const void* x, const void* gamma, const void* beta, void* y, int64_t rows, int c,
                          float eps, hipStream_t s) {
  auto go = [&](auto kern, int lpr) {
    const int rpb = 4 * (64 / lpr);
    hipLaunchKernelGGL(kern, dim3((unsigned)((rows + rpb - 1) / rpb)), dim3(256), 0, s,
                       (const T*)x, (const T*)gamma, (const T*)beta, (T*)y, rows, eps);
  };
  switch (c) {
    case 320: go(dd_layernorm_sub_kernel<T, 8>, 8); return true;
    case 640: go(dd_layernorm_sub_kernel<T, 16>, 16); return true;
    case 1280: go(dd_layernorm_sub_kernel<T, 32>, 32); return true;
    default: return false;
  }
}

int gn_plan(int hw, int c, int* pix_per_split) {
  // enough blocks to fill the chip, but at least a few pixels per pixel-lane
  const int cv = c / 8;
  const int pl = cv >= GN_THREADS ? 1 : GN_THREADS / cv;
  int pps = (hw + GN_MAX_SPLIT - 1) / GN_MAX_SPLIT;
  const int min_pps = pl * GN_UNROLL;
  if (pps < min_pps) pps = min_pps;
  if (pps > hw) pps = hw;
  *pix_per_split = pps;
  return (hw + pps - 1) / pps;
}

}  // namespace

extern "C" int64_t dd_groupnorm_workspace_bytes(int32_t m, int32_t groups) {
  if (m <= 0 || groups <= 0) return 0;
  return (int64_t)m * GN_MAX_SPLIT * groups * 2 * (int64_t)sizeof(float) + 256;     // + 256 B reserved head (ABI 2 layout)
}

extern "C" int dd_groupnorm_nhwc(const void* x1, int32_t c1, const void* x2, int32_t c2,
                                 const void* gamma, const void* beta, void* y,
                                 int32_t m, int32_t hw, int32_t groups, float eps,
                                 int32_t apply_silu, int32_t dtype, void* ws, int64_t ws_bytes,
                                 dd_stream_t stream) {
  if (!x1 || !gamma || !beta || !y || !ws) return DD_ERR_BAD_ARG;
  if (c2 < 0 || (c2 > 0 && !x2)) return DD_ERR_BAD_ARG;
  const int c = c1 + c2;
  if (m <= 0 || hw <= 0 || groups <= 0 || groups > 64 || c1 <= 0) return DD_ERR_BAD_ARG;
  if ((c1 & 7) || (c2 & 7) || c % groups != 0 || c > GN_MAX_C) return DD_ERR_BAD_ARG;
  if (c / groups < 4) return DD_ERR_UNSUPPORTED;   // a 16-B vector may span at most 2 groups (4 channels each at least)
  if (dtype != DD_F16 && dtype != DD_BF16) return DD_ERR_BAD_ARG;
  if (!dd_aligned16(x1) || (x2 && !dd_aligned16(x2)) || !dd_aligned16(y) ||
      !dd_aligned16(gamma) || !dd_aligned16(beta)) return DD_ERR_BAD_ARG;
  if (ws_bytes < dd_groupnorm_workspace_bytes(m, groups)) return DD_ERR_WORKSPACE;
  GnParams p{};
  p.x1 = x1; p.x2 = x2; p.c1 = c1; p.c2 = c2; p.c = c;
  p.gamma = gamma; p.beta = beta; p.y = y;
  p.m = m; p.hw = hw; p.groups = groups; p.cpg = c / groups; p.eps = eps; p.silu = apply_silu;
  p.ws = reinterpret_cast<float*>(ws) + 64;          // the first 256 B stay reserved; the partial sums follow
  p.nsplit = gn_plan(hw, c, &p.pix_per_split);
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  dd_clear_error();
  int cpb, vpp, plc, nv, kred, threads;
  if (gn_fused_plan(hw, c, groups, &cpb, &vpp, &plc, &nv, &kred, &threads)) {
    dim3 fgrid(c / cpb, m);
    if (threads == 256) {
      if (dtype == DD_F16) hipLaunchKernelGGL((dd_gn_fused_kernel<_Float16, 256, GNF_NV_SMALL>), fgrid, dim3(256), 0, s, p, cpb, vpp, plc, nv, kred);
      else hipLaunchKernelGGL((dd_gn_fused_kernel<__bf16, 256, GNF_NV_SMALL>), fgrid, dim3(256), 0, s, p, cpb, vpp, plc, nv, kred);
    } else {
      if (dtype == DD_F16) hipLaunchKernelGGL((dd_gn_fused_kernel<_Float16, 1024, GNF_NV_BIG>), fgrid, dim3(1024), 0, s, p, cpb, vpp, plc, nv, kred);
      else hipLaunchKernelGGL((dd_gn_fused_kernel<__bf16, 1024, GNF_NV_BIG>), fgrid, dim3(1024), 0, s, p, cpb, vpp, plc, nv, kred);
    }
    return dd_check_launch();
  }
  dim3 grid(p.nsplit, m);
  if (dtype == DD_F16) {
    hipLaunchKernelGGL(dd_gn_stats_kernel<_Float16>, grid, dim3(GN_THREADS), 0, s, p);
    hipLaunchKernelGGL(dd_gn_apply_kernel<_Float16>, grid, dim3(GN_THREADS), 0, s, p);
  } else {
    hipLaunchKernelGGL(dd_gn_stats_kernel<__bf16>, grid, dim3(GN_THREADS), 0, s, p);
    hipLaunchKernelGGL(dd_gn_apply_kernel<__bf16>, grid, dim3(GN_THREADS), 0, s, p);
  }
  return dd_check_launch();
}

extern "C" int dd_groupnorm_splitk(const float* partial, int32_t nsplit, const void* bias, const void* rowvec,
                                   int32_t ld_rowvec, const void* res, int64_t ldres, void* x_out,
                                   const void* gamma, const void* beta, void* y, int32_t m, int32_t hw, int32_t c,
                                   int32_t groups, float eps, int32_t apply_silu, int32_t dtype, dd_stream_t stream) {
  if (!partial || !gamma || !beta || !y || nsplit < 2 || nsplit > 64) return DD_ERR_BAD_ARG;
  if (m <= 0 || hw <= 0 || groups <= 0 || groups > 64 || c <= 0 || (c & 7) || c % groups || c > GN_MAX_C) return DD_ERR_BAD_ARG;
  if (c / groups < 4) return DD_ERR_UNSUPPORTED;
  if (dtype != DD_F16 && dtype != DD_BF16) return DD_ERR_BAD_ARG;
  if (!dd_aligned16(partial) || !dd_aligned16(y) || !dd_aligned16(gamma) || !dd_aligned16(beta) ||
      (bias && !dd_aligned16(bias)) || (rowvec && (!dd_aligned16(rowvec) || (ld_rowvec & 7))) ||
      (res && (!dd_aligned16(res) || (ldres & 7))) || (x_out && !dd_aligned16(x_out))) return DD_ERR_BAD_ARG;
  int cpb, vpp, plc, nv, kred, threads;
  if (!gn_fused_plan(hw, c, groups, &cpb, &vpp, &plc, &nv, &kred, &threads)) return DD_ERR_UNSUPPORTED;
  GnParams p{};
  p.x1 = nullptr; p.c1 = c; p.c = c;
  p.gamma = gamma; p.beta = beta; p.y = y;
  p.m = m; p.hw = hw; p.groups = groups; p.cpg = c / groups; p.eps = eps; p.silu = apply_silu;
  p.sk_part = partial; p.sk_nsplit = nsplit; p.sk_rows = (int64_t)m * hw;
  p.sk_bias = bias; p.sk_rowvec = rowvec; p.sk_ld_rowvec = ld_rowvec; p.sk_res = res; p.sk_ldres = ldres; p.sk_xout = x_out;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  dd_clear_error();
  dim3 fgrid(c / cpb, m);
  if (threads == 256) {
    if (dtype == DD_F16) hipLaunchKernelGGL((dd_gn_fused_kernel<_Float16, 256, GNF_NV_SMALL, true>), fgrid, dim3(256), 0, s, p, cpb, vpp, plc, nv, kred);
    else hipLaunchKernelGGL((dd_gn_fused_kernel<__bf16, 256, GNF_NV_SMALL, true>), fgrid, dim3(256), 0, s, p, cpb, vpp, plc, nv, kred);
  } else {
    if (dtype == DD_F16) hipLaunchKernelGGL((dd_gn_fused_kernel<_Float16, 1024, GNF_NV_BIG, true>), fgrid, dim3(1024), 0, s, p, cpb, vpp, plc, nv, kred);
    else hipLaunchKernelGGL((dd_gn_fused_kernel<__bf16, 1024, GNF_NV_BIG, true>), fgrid, dim3(1024), 0, s, p, cpb, vpp, plc, nv, kred);
  }
  return dd_check_launch();
}

extern "C" int dd_groupnorm_is_fused(int32_t hw, int32_t c, int32_t groups) {
  int cpb, vpp, plc, nv, kred, threads;
  if (hw <= 0 || c <= 0 || groups <= 0 || c % groups) return 0;
  return gn_fused_plan(hw, c, groups, &cpb, &vpp, &plc, &nv, &kred, &threads) ? threads : 0;
}

extern "C" int dd_layernorm(const void* x, const void* gamma, const void* beta, void* y,
                            int64_t rows, int32_t c, float eps, int32_t dtype,
                            dd_stream_t stream) {
  if (!x || !gamma || !beta || !y || rows <= 0 || c <= 0) return DD_ERR_BAD_ARG;
  if ((c & 7) || c > LN_MAXV * 512) return DD_ERR_UNSUPPORTED;
  if (dtype != DD_F16 && dtype != DD_BF16) return DD_ERR_BAD_ARG;
  if (!dd_aligned16(x) || !dd_aligned16(y) || !dd_aligned16(gamma) || !dd_aligned16(beta)) return DD_ERR_BAD_ARG;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  dd_clear_error();
  if (dtype == DD_F16 ? launch_layernorm_sub<_Float16>(x, gamma, beta, y, rows, c, eps, s)
                      : launch_layernorm_sub<__bf16>(x, gamma, beta, y, rows, c, eps, s))
    return dd_check_launch();
  const unsigned blocks = (unsigned)((rows + 3) / 4);
  if (dtype == DD_F16) {
    hipLaunchKernelGGL(dd_layernorm_kernel<_Float16>, dim3(blocks), dim3(256), 0, s,
                       (const _Float16*)x, (const _Float16*)gamma, (const _Float16*)beta,
                       (_Float16*)y, rows, c, eps);
  } else {
    hipLaunchKernelGGL(dd_layernorm_kernel<__bf16>, dim3(blocks), dim3(256), 0, s,
                       (const __bf16*)x, (const __bf16*)gamma, (const __bf16*)beta,
                       (__bf16*)y, rows, c, eps);
  }
  return dd_check_launch();
}
