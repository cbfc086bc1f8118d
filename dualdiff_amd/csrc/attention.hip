// Flash-style scaled-dot-product attention for gfx950 (head_dim 40 / 80 / 160, no mask).
//
// Structure (per workgroup = 4 waves, each wave owns QT*16 query rows; K/V tiles of 64 keys
// are staged once per workgroup in LDS and shared by the 4 waves):
//  * "swapped" products so that the softmax row lives on a lane:
//        S^T[key][q] = K . Q^T     (MFMA A = K rows, B = Q rows; both d-contiguous)
//        O^T[d][q]   = V^T . P^T   (MFMA A = V^T via ds_read_b64_tr_b16, B = P^T)
//    The S^T accumulator of a lane is 8 scores of ONE query row per 32-key chunk, and it is
//    already laid out as the B operand of the second product (the k index of both operands
//    is permuted identically) — no LDS round trip and no cross-lane movement for P.
//  * online softmax in fp32 (exp2 with the scale folded into log2e), row max reduced with two
//    xor-shuffles (lane groups 16/32 apart hold the other keys of the same query row).
//  * head_dim is padded with zeros in LDS to a multiple of 32 for QK^T (40->64, 80->96) and
//    to a multiple of 16 for PV (40->48); rows are padded by 16 B against bank conflicts.
//  * K/V tiles staged by buffer loads with the hardware range check (see the kernel's comment).
//  * kv_batch_map + accumulate implement the neighbour-view attention (attn4) as two calls
//    that read the neighbours' K/V in place and sum the normalised outputs.
#include "dd_common.h"
#include "dd_debug.h"      // DD_EXP2, dd_dbg::NOSTAGE: hooks of the diagnostic builds (the product's are v_exp_f32 / false)
// max of 8 scores as three v_max3_f32 + one v_max_f32 (a balanced fmaxf tree compiles to seven v_max_f32: the d = 40 loop is
// VALU-issue-bound and this is 9 instructions of ~70 per 32-key chunk)
__device__ __forceinline__ float dd_max3(float a, float b, float c) { return fmaxf(fmaxf(a, b), c); }
__device__ __forceinline__ float dd_max8(const float (&s)[8]) {
  return fmaxf(dd_max3(dd_max3(s[0], s[1], s[2]), dd_max3(s[3], s[4], s[5]), s[6]), s[7]);
}

namespace {

constexpr float RESCALE_THR = 5.0f;    // log2 units: probabilities stay <= 32

struct AttnParams {
  const void* q; const void* k; const void* v; void* o;
  int64_t ldq, ldk, ldv, ldo;
  int64_t qbs, kbs, vbs, obs;
  int batch, heads, lq, lk;
  float scale_log2;
  const int32_t* kv_map;
  int accumulate;
  int nqb;             // q-blocks per (batch, head), filled by the launcher
  int64_t qhs, khs, vhs;   // head strides (elements)
  int prescaled;           // q carries scale * log2(e)
  const int32_t* kv_map2;  // second K/V batch map: O = Attn(q, kv[map]) + Attn(q, kv[map2]) in ONE launch (v5 family)
  const int32_t* lk_dev;   // key count read from DEVICE memory at kernel start (dd_attn_desc.lk_dev); p.lk is then the capacity
};


// The kernel ("v5"; its register-staged round-1 predecessor dd_attn_kernel and the tuning variants 1-13 of
// dd_attn_desc.variant were removed in round 5: no dispatcher selected them): the products and register layout of the
// file header, with the per-tile overhead taken out of
// the instruction stream (the d = 40 loop is VALU-issue-bound: ~45 % of its slots were bookkeeping):
//  * K/V rows come in through buffer loads with the hardware range check — per-lane byte offsets
//    are computed once, rows past lk read as zeros, no compare / branch per load;
//  * the zero padding of K (d..DQ) and the ones column of V live in LDS columns that the tile
//    stores never touch: written once, not per tile;
//  * head dims with a spare PV column (d = 40): keys past lk need NO score mask — their V row and
//    their ones-column entry are zero, so whatever probability they get contributes nothing to the
//    numerator or the denominator.  Only the running max must not see them, and that is handled
//    inside the (rare) rescale branch; the common path has no tail code at all;
//  * NBUF = 2: two LDS tiles, one barrier per tile instead of two.
// PRE: q arrives pre-multiplied by scale * log2(e) (dd_gemm_desc.hm_scale), so the scores leave the QK^T
// MFMA in log2 units — and with the running max negated in the MFMA's C operand they leave it already
// shifted: p = exp2(acc), no per-score FMA (16 of the ~70 VALU instructions per 32-key chunk).
template <typename T, int D, int QT, int KV_TILE, int NBUF, int WPE, bool PRE, bool PAIR = false>
__global__ __launch_bounds__(256, WPE)
void dd_attn5_kernel(const AttnParams p) {
  using V8 = typename dd_vec<T>::v8;
  using V4 = typename dd_vec<T>::v4;
  constexpr int DQ = (D + 31) / 32 * 32;
  constexpr int KSTEPS = DQ / 32;
  constexpr int DVT = (D + 15) / 16;
  constexpr int KSTR = DQ + 16;
  constexpr int VSTR = DVT * 16 + (D == 160 ? 16 : 0);
  constexpr int KCH = DQ / 8;
  constexpr int VCH = DVT * 2;
  constexpr int DCH = D / 8;
  constexpr bool ONES = (D % 16) != 0;
  constexpr int NCH = KV_TILE * DCH;                 // valid 16-B chunks per K (and per V) tile
  constexpr int PER = (NCH + 255) / 256;
  constexpr bool RAGGED = (NCH % 256) != 0;          // last per-thread slot only partly populated
  constexpr int TILE_ELEMS = KV_TILE * (KSTR + VSTR);

  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  T* lds = reinterpret_cast<T*>(smem);

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int g = lane >> 4;
  const int c = lane & 15;
  // keys per batch entry: a kernel argument, or (lk_dev) one word of device memory read here — a HIP graph recorded
  // at a context CAPACITY then serves every length up to it (model_base.ForwardGraphs).  Uniform: a scalar load.
  const int lk = p.lk_dev ? min(max(__builtin_amdgcn_readfirstlane(*p.lk_dev), 1), p.lk) : p.lk;   // clamped to the capacity

  const int nqb = p.nqb;
  const int nwg = nqb * p.batch * p.heads;
  const int xcd = blockIdx.x & 7;
  const int xq = nwg >> 3, xr = nwg & 7;
  const int item = ((xcd < xr) ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + (blockIdx.x >> 3);
  const int bh = item / nqb;
  const int qb = item - bh * nqb;
  const int b = bh / p.heads;
  const int h = bh - b * p.heads;
  const int q0 = (qb * 4 + wave) * (QT * 16);

  const T* qbase = reinterpret_cast<const T*>(p.q) + (int64_t)b * p.qbs + h * p.qhs;
  const uint32_t k_row_bytes = (uint32_t)p.ldk * sizeof(T), v_row_bytes = (uint32_t)p.ldv * sizeof(T);
  // Neighbour-view PAIR (p.kv_map2, reference blocks.py:203-217: out = Attn(q, kv_left) + Attn(q, kv_right)): two passes
  // of the same loop over two K/V instances with their own softmax state; the first pass's normalised output waits in
  // LDS (fp32, behind the K/V tiles) and the sum is rounded ONCE — one launch, no read-modify-write of O.
  // (PAIR is a template parameter: the single-attention instantiations keep their register allocation — with a run-time
  // pass loop the 48-rows-per-wave d = 40 kernel went from 162 VGPRs / no spill to 168 / 6 spilled)
  constexpr int npass = PAIR ? 2 : 1;
  f32x4* stash = reinterpret_cast<f32x4*>(smem + (size_t)NBUF * TILE_ELEMS * sizeof(T));

  // per-lane chunk tables: global byte offsets inside a tile and LDS element offsets
  uint32_t gk[PER], gv[PER];
  int lk_off[PER], lv_off[PER];
#pragma unroll
  for (int i = 0; i < PER; ++i) {
    const int idx = tid + i * 256;
    const int row = idx / DCH, ch = idx - row * DCH;
    gk[i] = (uint32_t)row * k_row_bytes + ch * 16;
    gv[i] = (uint32_t)row * v_row_bytes + ch * 16;
    lk_off[i] = row * KSTR + ch * 8;
    lv_off[i] = KV_TILE * KSTR + row * VSTR + ch * 8;
  }
  const bool last_slot = !RAGGED || (tid + (PER - 1) * 256 < NCH);

  // ---- one-time LDS columns: K zero padding, V ones column ----------------------------------
  {
    const T one_t = (T)1.0f;
    unsigned short one_u16;
    __builtin_memcpy(&one_u16, &one_t, 2);
    const u32x4 zero = {0u, 0u, 0u, 0u};
    const u32x4 ones = {(unsigned)one_u16, 0u, 0u, 0u};
    for (int idx = tid; idx < NBUF * KV_TILE; idx += 256) {
      const int buf = idx / KV_TILE, row = idx - buf * KV_TILE;
      T* kr = lds + buf * TILE_ELEMS + row * KSTR;
#pragma unroll
      for (int ch = DCH; ch < KCH; ++ch) dd_st16(kr + ch * 8, zero);
      if (ONES) dd_st16(lds + buf * TILE_ELEMS + KV_TILE * KSTR + row * VSTR + DCH * 8, ones);
    }
  }

  V8 qf[QT][KSTEPS];
#pragma unroll
  for (int qt = 0; qt < QT; ++qt) {
    const int qrow = q0 + qt * 16 + c;
#pragma unroll
    for (int ks = 0; ks < KSTEPS; ++ks) {
      const int d0 = ks * 32 + g * 8;
      u32x4 v = {0u, 0u, 0u, 0u};
      if (qrow < p.lq && d0 < D) v = dd_ld16(qbase + (int64_t)qrow * p.ldq + d0);
      qf[qt][ks] = dd_as_v8<T>(v);
    }
  }

  f32x4 oacc[DVT][QT];
  float m_run[QT], l_run[QT];
  f32x4 cinit[QT];                       // PRE: -m_run broadcast, the C operand of the first QK^T MFMA
  const int ntiles = (lk + KV_TILE - 1) / KV_TILE;

  for (int pass = 0; pass < npass; ++pass) {
  const int kb = pass ? p.kv_map2[b] : (p.kv_map ? p.kv_map[b] : b);
  const T* kbase = reinterpret_cast<const T*>(p.k) + (int64_t)kb * p.kbs + h * p.khs;
  const T* vbase = reinterpret_cast<const T*>(p.v) + (int64_t)kb * p.vbs + h * p.vhs;
  const __amdgpu_buffer_rsrc_t rs_k = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<T*>(kbase), 0, (uint32_t)(lk - 1) * k_row_bytes + D * sizeof(T), 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_v = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<T*>(vbase), 0, (uint32_t)(lk - 1) * v_row_bytes + D * sizeof(T), 0x00020000);
#pragma unroll
  for (int i = 0; i < DVT; ++i)
#pragma unroll
    for (int j = 0; j < QT; ++j) oacc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int j = 0; j < QT; ++j) { m_run[j] = PRE ? 0.f : -1e30f; l_run[j] = 0.f; }
#pragma unroll
  for (int j = 0; j < QT; ++j) cinit[j] = f32x4{0.f, 0.f, 0.f, 0.f};
  if (pass) {
    __syncthreads();                     // every wave is done with the first pass's last tile
    if (ONES && (lk % KV_TILE) != 0) { // ... whose ragged rows lost their ones entry: restore the column
      const T one_t = (T)1.0f;
      unsigned short one_u16;
      __builtin_memcpy(&one_u16, &one_t, 2);
      const u32x4 ones = {(unsigned)one_u16, 0u, 0u, 0u};
      for (int idx = tid; idx < NBUF * KV_TILE; idx += 256) {
        const int buf = idx / KV_TILE, row = idx - buf * KV_TILE;
        dd_st16(lds + buf * TILE_ELEMS + KV_TILE * KSTR + row * VSTR + DCH * 8, ones);
      }
    }
  }

  u32x4 kreg[PER], vreg[PER];
  auto load_kv = [&](int tile0) {
    const uint32_t ko = (uint32_t)tile0 * k_row_bytes, vo = (uint32_t)tile0 * v_row_bytes;
    if constexpr (dd_dbg::NOSTAGE) return;
#pragma unroll
    for (int i = 0; i < PER; ++i) kreg[i] = __builtin_amdgcn_raw_buffer_load_b128(rs_k, gk[i] + ko, 0, 0);
#pragma unroll
    for (int i = 0; i < PER; ++i) vreg[i] = __builtin_amdgcn_raw_buffer_load_b128(rs_v, gv[i] + vo, 0, 0);
  };
  auto store_kv = [&](T* tile) {
    if constexpr (dd_dbg::NOSTAGE) return;
#pragma unroll
    for (int i = 0; i < PER; ++i)
      if (i < PER - 1 || last_slot) {
        dd_st16(tile + lk_off[i], kreg[i]);
        dd_st16(tile + lv_off[i], vreg[i]);
      }
  };

  load_kv(0);

  for (int it = 0; it < ntiles; ++it) {
    const int tile0 = it * KV_TILE;
    T* tile = lds + (NBUF == 2 ? (it & 1) * TILE_ELEMS : 0);
    const T* Ks = tile;
    const T* Vs = tile + KV_TILE * KSTR;
    if (NBUF == 1) __syncthreads();       // previous tile fully consumed by every wave
    store_kv(tile);
    if (ONES && tile0 + KV_TILE > lk) { // last, ragged tile: rows past lk lose their ones entry
      const u32x4 zero = {0u, 0u, 0u, 0u};
      if (tid < KV_TILE && tile0 + tid >= lk) dd_st16(tile + KV_TILE * KSTR + tid * VSTR + DCH * 8, zero);
    }
    __syncthreads();                      // tile visible (NBUF == 2: and the other buffer is free)
    if (it + 1 < ntiles) load_kv(tile0 + KV_TILE);

#pragma unroll
    for (int cc = 0; cc < KV_TILE / 32; ++cc) {
      const int key0 = tile0 + cc * 32;
      if (key0 >= lk) break;
      f32x4 sacc[2][QT];
#pragma unroll
      for (int ks = 0; ks < KSTEPS; ++ks) {
        V8 kf[2];
#pragma unroll
        for (int t = 0; t < 2; ++t)
          kf[t] = dd_as_v8<T>(dd_ld16(Ks + (cc * 32 + t * 16 + c) * KSTR + ks * 32 + g * 8));
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int j = 0; j < QT; ++j)
            sacc[t][j] = dd_mfma16(kf[t], qf[j][ks],
                                   ks == 0 ? (PRE ? cinit[j] : f32x4{0.f, 0.f, 0.f, 0.f}) : sacc[t][j]);
      }
      const bool tail = key0 + 32 > lk;             // uniform
      V8 pf[QT];
#pragma unroll
      for (int j = 0; j < QT; ++j) {
        float s[8];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int r = 0; r < 4; ++r) s[t * 4 + r] = sacc[t][j][r];
        auto mask_tail = [&]() {
          int rem = lk - key0 - g * 4;       // pinned here: the compiler otherwise hoists the eight
          asm volatile("" : "+v"(rem));        // compares of this rare path into the common block
#pragma unroll
          for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r)
              if (t * 16 + r >= rem) s[t * 4 + r] = -INFINITY;
        };
        if (!ONES && tail) mask_tail();
        float mx_l = dd_max8(s);
        V8 pv;
        if constexpr (PRE) {
          // s is already (score - m_run) in log2 units.  The very first chunk of a row always takes the
          // branch (m_run = 0 there is not a maximum yet and may be far above the scores).
          const bool first = it == 0 && cc == 0;
          if (first || __any(mx_l > RESCALE_THR)) {
            if (ONES && tail) {
              mask_tail();
              mx_l = dd_max8(s);
            }
            float mx = fmaxf(mx_l, __shfl_xor(mx_l, 16, 64));
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            if (!first) mx = fmaxf(mx, 0.f);              // the running max never decreases
            mx = fmaxf(mx, -1e30f);                        // a fully masked row keeps a finite max
            const float alpha = first ? 1.0f : DD_EXP2(-mx);
            m_run[j] += mx;
            cinit[j] = f32x4{-m_run[j], -m_run[j], -m_run[j], -m_run[j]};
#pragma unroll
            for (int e = 0; e < 8; ++e) s[e] -= mx;
            if (!ONES) l_run[j] *= alpha;
#pragma unroll
            for (int dt = 0; dt < DVT; ++dt) {
              oacc[dt][j][0] *= alpha; oacc[dt][j][1] *= alpha;
              oacc[dt][j][2] *= alpha; oacc[dt][j][3] *= alpha;
            }
          }
          float ls = 0.f;
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const float pe = DD_EXP2(s[e]);
            if (!ONES) ls += pe;
            pv[e] = (T)pe;
          }
          if (!ONES) l_run[j] += ls;
        } else {
          if (__any(mx_l * p.scale_log2 - m_run[j] > RESCALE_THR)) {
            if (ONES && tail) {          // the padded keys scored 0: keep them out of the running max
              mask_tail();
              mx_l = dd_max8(s);
            }
            float mx = fmaxf(mx_l, __shfl_xor(mx_l, 16, 64));
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            const float m_new = fmaxf(m_run[j], mx * p.scale_log2);
            const float alpha = DD_EXP2(m_run[j] - m_new);
            m_run[j] = m_new;
            if (!ONES) l_run[j] *= alpha;
#pragma unroll
            for (int dt = 0; dt < DVT; ++dt) {
              oacc[dt][j][0] *= alpha; oacc[dt][j][1] *= alpha;
              oacc[dt][j][2] *= alpha; oacc[dt][j][3] *= alpha;
            }
          }
          const float m_use = m_run[j];
          float ls = 0.f;
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const float pe = DD_EXP2(fmaf(s[e], p.scale_log2, -m_use));
            if (!ONES) ls += pe;
            pv[e] = (T)pe;
          }
          if (!ONES) l_run[j] += ls;
        }
        pf[j] = pv;
      }
#pragma unroll
      for (int dt = 0; dt < DVT; ++dt) {
        V8 vf;
        const T* a0 = Vs + (cc * 32 + g * 4 + (c >> 2)) * VSTR + dt * 16 + (c & 3) * 4;
        const T* a1 = a0 + 16 * VSTR;
        s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(a0));
        s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(a1));
        __builtin_memcpy(&vf, &lo, 8);
        __builtin_memcpy(reinterpret_cast<char*>(&vf) + 8, &hi, 8);
#pragma unroll
        for (int j = 0; j < QT; ++j) oacc[dt][j] = dd_mfma16(vf, pf[j], oacc[dt][j]);
      }
    }
  }

  // normalise this pass (row sums: the ones column of V, or l_run)
#pragma unroll
  for (int j = 0; j < QT; ++j) {
    float l;
    if (ONES) {
      l = __shfl(oacc[DVT - 1][j][D % 4], ((D % 16) / 4) * 16 + c, 64);
    } else {
      l = l_run[j];
      l += __shfl_xor(l, 16, 64);
      l += __shfl_xor(l, 32, 64);
    }
    const float inv = 1.0f / l;
#pragma unroll
    for (int dt = 0; dt < DVT; ++dt) {
      oacc[dt][j][0] *= inv; oacc[dt][j][1] *= inv; oacc[dt][j][2] *= inv; oacc[dt][j][3] *= inv;
    }
  }
  if (npass == 2) {
    if (pass == 0) {                     // park the first neighbour's result (thread-private slots: no barrier)
#pragma unroll
      for (int dt = 0; dt < DVT; ++dt)
#pragma unroll
        for (int j = 0; j < QT; ++j) stash[(dt * QT + j) * 256 + tid] = oacc[dt][j];
    } else {
#pragma unroll
      for (int dt = 0; dt < DVT; ++dt)
#pragma unroll
        for (int j = 0; j < QT; ++j) {
          const f32x4 o0 = stash[(dt * QT + j) * 256 + tid];
          oacc[dt][j][0] += o0[0]; oacc[dt][j][1] += o0[1]; oacc[dt][j][2] += o0[2]; oacc[dt][j][3] += o0[3];
        }
    }
  }
  }   // pass

  T* obase = reinterpret_cast<T*>(p.o) + (int64_t)b * p.obs + h * D;
#pragma unroll
  for (int j = 0; j < QT; ++j) {
    const int qrow = q0 + j * 16 + c;
    if (qrow >= p.lq) continue;
#pragma unroll
    for (int dt = 0; dt < DVT; ++dt) {
      const int d0 = dt * 16 + g * 4;
      if (d0 >= D) continue;
      T* dst = obase + (int64_t)qrow * p.ldo + d0;
      float o4[4] = {oacc[dt][j][0], oacc[dt][j][1], oacc[dt][j][2], oacc[dt][j][3]};
      if (p.accumulate) {
        V4 prev = *reinterpret_cast<const V4*>(dst);
#pragma unroll
        for (int e = 0; e < 4; ++e) o4[e] += (float)prev[e];
      }
      V4 ov;
#pragma unroll
      for (int e = 0; e < 4; ++e) ov[e] = (T)o4[e];
      *reinterpret_cast<V4*>(dst) = ov;
    }
  }
}

template <typename T, int D, int QT, int KV_TILE, int NBUF, int WPE, bool PRE, bool PAIR = false>
int launch_attn5p(const AttnParams& p, hipStream_t s) {
  constexpr int DQ = (D + 31) / 32 * 32;
  constexpr int DVT = (D + 15) / 16;
  constexpr size_t smem = (size_t)NBUF * KV_TILE * ((DQ + 16) + (DVT * 16 + (D == 160 ? 16 : 0))) * sizeof(T);
  const int qblk = 4 * QT * 16;
  AttnParams pp = p;
  pp.nqb = (p.lq + qblk - 1) / qblk;
  dim3 grid(pp.nqb * p.batch * p.heads);
  auto kern = dd_attn5_kernel<T, D, QT, KV_TILE, NBUF, WPE, PRE, PAIR>;
  constexpr size_t stash = PAIR ? (size_t)DVT * QT * 256 * sizeof(f32x4) : 0;   // neighbour pair: the first pass's output
  static_assert(smem + stash <= 160 * 1024, "LDS");
  if (PAIR != (p.kv_map2 != nullptr)) return DD_ERR_UNSUPPORTED;
  static std::atomic<uint64_t> attr_done{0};       // per instantiation, one bit per device
  dd_ensure_dyn_lds((const void*)kern, smem + stash, attr_done);
  hipLaunchKernelGGL(kern, grid, dim3(256), smem + stash, s, pp);
  return dd_check_launch();
}

template <typename T, int D, int QT, int KV_TILE, int NBUF, int WPE = 1>
int launch_attn5(const AttnParams& p, hipStream_t s) {
  if (p.prescaled) return launch_attn5p<T, D, QT, KV_TILE, NBUF, WPE, true>(p, s);
  return launch_attn5p<T, D, QT, KV_TILE, NBUF, WPE, false>(p, s);
}

// the DEFAULT configurations also exist as neighbour-PAIR kernels (dd_attn_desc.kv_batch_map2)
template <typename T, int D, int QT, int KV_TILE, int NBUF, int WPE = 1>
int launch_attn5d(const AttnParams& p, hipStream_t s) {
  if (p.kv_map2) {
    if (p.prescaled) return launch_attn5p<T, D, QT, KV_TILE, NBUF, WPE, true, true>(p, s);
    return launch_attn5p<T, D, QT, KV_TILE, NBUF, WPE, false, true>(p, s);
  }
  return launch_attn5<T, D, QT, KV_TILE, NBUF, WPE>(p, s);
}

// Which instantiation a call takes — decided on the host from the shape alone (dd_attention_kernel_name reports it without
// launching, so the choice is testable without a GPU): QT 16-row MFMA blocks of queries per wave, KV_TILE keys per LDS
// tile, WPE waves per SIMD the kernel is compiled for.
struct AttnPlan { int qt, kv_tile, wpe; bool ok; };

template <int D>
AttnPlan attn_plan(const AttnParams& p, int variant, int elem_bytes) {
  AttnPlan pl{1, 64, 1, false};
  if (variant != 0) return pl;                       // the tuning variants of rounds 1-3 are gone (ABI 3)
  // K/V staging by buffer loads: 32-bit byte offsets per (batch, head) plane
  if ((int64_t)p.lk * p.ldk * elem_bytes >= (1ll << 31) || (int64_t)p.lk * p.ldv * elem_bytes >= (1ll << 31)) return pl;
  pl.ok = true;
  // 32 query rows per wave when the sequence is long enough to fill the chip, else 16;
  // 128-key tiles (half the barriers) for long key sequences at the small head dims
  const long blocks128 = (long)((p.lq + 127) / 128) * p.batch * p.heads;
  const bool qt2 = p.lq >= 256 && blocks128 >= 512;
  if (D == 40) {
    // 32 or 48 query rows per wave (128 / 192 per workgroup).  The loop is VALU-issue-bound and only
    // reaches that bound with the SIMDs full (4 waves at 32 rows, 3 at 48), so what decides is how
    // evenly the workgroups fill the chip: 12 instances x 8 heads x 1400 rows is 1056 workgroups of
    // 128 rows on 1024 slots (a second, nearly empty generation: 65 us) but 768 of 192 rows on 768
    // slots (56 us).  Take the split with the smaller ceil(generations) / generations.
    if (qt2) {
      const double g2 = (double)((p.lq + 127) / 128) * p.batch * p.heads / 1024.0;
      const double g3 = (double)((p.lq + 191) / 192) * p.batch * p.heads / 768.0;
      auto waste = [](double g) { double c = (double)(long)g; if (c < g) c += 1.0; return c / g; };
      if (waste(g3) < waste(g2) - 0.05) {
        // (the prescaled-q variant of the 128-key tile spills at 48 rows per wave: 64-key tiles there)
        pl.qt = 3; pl.wpe = 3; pl.kv_tile = p.lk >= 512 && !p.prescaled ? 128 : 64;
      } else {
        pl.qt = 2;
      }
    }
  } else if (D == 80) {
    if (qt2) { pl.qt = 2; pl.kv_tile = p.lk >= 512 ? 128 : 64; }
  } else {
    if (qt2) pl.qt = 2;
  }
  return pl;
}

template <typename T, int D>
int launch_attn_d(const AttnParams& p, int variant, hipStream_t s) {
  const AttnPlan pl = attn_plan<D>(p, variant, (int)sizeof(T));
  if (!pl.ok) return DD_ERR_UNSUPPORTED;
  if constexpr (D == 40) {
    if (pl.qt == 3) return pl.kv_tile == 128 ? launch_attn5d<T, D, 3, 128, 1, 3>(p, s) : launch_attn5d<T, D, 3, 64, 1, 3>(p, s);
    return pl.qt == 2 ? launch_attn5d<T, D, 2, 64, 1>(p, s) : launch_attn5d<T, D, 1, 64, 1>(p, s);
  } else if constexpr (D == 80) {
    if (pl.qt == 2) return pl.kv_tile == 128 ? launch_attn5d<T, D, 2, 128, 1>(p, s) : launch_attn5d<T, D, 2, 64, 1>(p, s);
    return launch_attn5d<T, D, 1, 64, 1>(p, s);
  } else {
    return pl.qt == 2 ? launch_attn5d<T, D, 2, 64, 1>(p, s) : launch_attn5d<T, D, 1, 64, 1>(p, s);
  }
}

template <typename T>
int launch_attn_t(const dd_attn_desc* d, const AttnParams& p, hipStream_t s) {
  switch (d->head_dim) {
    case 40: return launch_attn_d<T, 40>(p, d->variant, s);
    case 80: return launch_attn_d<T, 80>(p, d->variant, s);
    case 160: return launch_attn_d<T, 160>(p, d->variant, s);
  }
  return DD_ERR_UNSUPPORTED;
}

// argument checks + parameter block shared by dd_attention and dd_attention_kernel_name
int attn_prepare(const dd_attn_desc* d, AttnParams& p) {
  if (!d || !d->q || !d->k || !d->v || !d->o) return DD_ERR_BAD_ARG;
  if (d->batch <= 0 || d->heads <= 0 || d->lq <= 0 || d->lk <= 0) return DD_ERR_BAD_ARG;
  if (d->dtype != DD_F16 && d->dtype != DD_BF16) return DD_ERR_BAD_ARG;
  if ((d->ldq & 7) || (d->ldk & 7) || (d->ldv & 7) || (d->ldo & 7)) return DD_ERR_BAD_ARG;
  if ((d->q_batch_stride & 7) || (d->k_batch_stride & 7) || (d->v_batch_stride & 7) ||
      (d->o_batch_stride & 7)) return DD_ERR_BAD_ARG;
  if (!dd_aligned16(d->q) || !dd_aligned16(d->k) || !dd_aligned16(d->v) || !dd_aligned16(d->o))
    return DD_ERR_BAD_ARG;
  if (d->head_dim != 40 && d->head_dim != 80 && d->head_dim != 160) return DD_ERR_UNSUPPORTED;
  if ((long)d->batch * d->heads > 65535) return DD_ERR_UNSUPPORTED;
  p.q = d->q; p.k = d->k; p.v = d->v; p.o = d->o;
  p.ldq = d->ldq; p.ldk = d->ldk; p.ldv = d->ldv; p.ldo = d->ldo;
  p.qbs = d->q_batch_stride; p.kbs = d->k_batch_stride; p.vbs = d->v_batch_stride; p.obs = d->o_batch_stride;
  p.batch = d->batch; p.heads = d->heads; p.lq = d->lq; p.lk = d->lk;
  p.scale_log2 = d->scale * 1.44269504088896340736f;
  p.kv_map = d->kv_batch_map; p.accumulate = d->accumulate;
  p.kv_map2 = d->kv_batch_map2;
  if (p.kv_map2 && !p.kv_map) return DD_ERR_BAD_ARG;
  p.qhs = d->q_head_stride ? d->q_head_stride : d->head_dim;
  p.khs = d->k_head_stride ? d->k_head_stride : d->head_dim;
  p.vhs = d->v_head_stride ? d->v_head_stride : d->head_dim;
  if ((p.qhs & 7) || (p.khs & 7) || (p.vhs & 7)) return DD_ERR_BAD_ARG;
  p.prescaled = d->q_prescaled ? 1 : 0;
  if (d->q_prescaled) p.scale_log2 = 1.0f;
  p.lk_dev = d->lk_dev;
  if (d->lk_dev && (reinterpret_cast<uintptr_t>(d->lk_dev) & 3u)) return DD_ERR_BAD_ARG;
  return DD_OK;
}

thread_local char g_attn_name[160];

}  // namespace

extern "C" int dd_attention(const dd_attn_desc* d, dd_stream_t stream) {
  AttnParams p{};
  const int rc = attn_prepare(d, p);
  if (rc != DD_OK) return rc;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  dd_clear_error();
  if (d->dtype == DD_F16) return launch_attn_t<_Float16>(d, p, s);
  return launch_attn_t<__bf16>(d, p, s);
}

extern "C" const char* dd_attention_kernel_name(const dd_attn_desc* d) {
  AttnParams p{};
  const int rc = attn_prepare(d, p);
  if (rc != DD_OK) return rc == DD_ERR_UNSUPPORTED ? "unsupported" : "invalid";
  const AttnPlan pl = d->head_dim == 40 ? attn_plan<40>(p, d->variant, 2)
                    : d->head_dim == 80 ? attn_plan<80>(p, d->variant, 2) : attn_plan<160>(p, d->variant, 2);
  if (!pl.ok) return "unsupported";
  const int nqb = (p.lq + 4 * pl.qt * 16 - 1) / (4 * pl.qt * 16);
  snprintf(g_attn_name, sizeof(g_attn_name), "dd_attn5_kernel<%s, %d, %d, %d, 1, %d, %s, %s> grid=%d",
           d->dtype == DD_F16 ? "_Float16" : "__bf16", d->head_dim, pl.qt, pl.kv_tile, pl.wpe,
           p.prescaled ? "true" : "false", p.kv_map2 ? "true" : "false", nqb * p.batch * p.heads);
  return g_attn_name;
}

