// MFMA GEMM / implicit-GEMM 3x3 convolution with fused epilogue for gfx950.
//
//   out[r, n] (op)= alpha * (sum_k A[r,k] W[n,k] + bias[n] + rowvec[r/rpi, n]) + res[r, n]
//
// Design (MI355X-first, not a CUDA tiling):
//  * wave64, v_mfma_f32_16x16x32_{f16,bf16}; fp32 accumulate.
//  * The MFMA "A" operand is the WEIGHT tile and the "B" operand the ACTIVATION tile, so
//    the accumulator of a lane is a run of consecutive output channels of one row.  The
//    weight rows of a wave are permuted on the global->LDS load so that each lane ends up
//    with 4*TN *consecutive* channels -> 16-byte NHWC stores, 128 B contiguous per row.
//  * Both operands are K-contiguous ([rows][K] activations, [N][K] weights), staged as
//    [row][64] tiles in LDS with a 16-byte-chunk XOR swizzle (conflict-free ds_read_b128).
//  * Register-staged software pipeline: global loads of K-tile t+1 are issued before the
//    MFMAs of tile t and written to the other LDS buffer afterwards (one barrier / K-step).
//  * conv mode gathers the im2col row on the fly (NHWC: one tap = one contiguous Cin run);
//    padding, stride 2 and the nearest-neighbour upsample are folded into the gather.
//  * XCD-aware tile order: consecutive tiles that share an activation panel are mapped to
//    the same XCD (private L2).
//  * split-K for the deep, weight-bound levels (336..1092 rows x K up to 23040).
#include "dd_common.h"
#include <type_traits>

namespace {

constexpr int BK = 64;  // K elements per pipeline step (8 chunks of 16 B per tile row)

struct GemmParams {
  const void* a; const void* a2; int64_t lda, lda2; int k1;
  int rows, n, k;
  const void* w; const void* bias; const void* rowvec; int rows_per_inst, ld_rowvec;
  const void* res; int64_t ldres;
  void* out; int64_t ldc;
  float alpha; int accumulate; int act;
  int hin, win, cin, hv, wv, hout, wout, stride, upsample;
  float scale_h, scale_w;
  int k_per_split;
  float* partial;
  int tiles_m, tiles_n;
  uint32_t a_bytes, a2_bytes, w_bytes;   // buffer extents for the descriptor-based DMA path
  uint32_t out_bytes, res_bytes;         // dd_gemm3_kernel's fast epilogue: extents of out / res (0 = take the general epilogue)
  int g_per_tile, chunks_per_split;      // direct small-image conv (dd_conv3s_kernel)
  int band_rows, bands; float inv_bands; // ... its BAND form: output pixels per band, bands per instance
  const float* ln_colsum; const float* ln_bias; float ln_eps;   // LayerNorm fold (dd_gemm2_kernel, dense)
  int* tile_counters;                    // split-K: per-tile arrival counters (in-kernel ordered reduction) or NULL
  uint32_t partial_bytes;                // extent of the slab region (buffer descriptor of the sc1 slab path)
  int out_f32;                           // store fp32 instead of T
  float* stat_out;                       // [rows][n/32][2] row sum / sum of squares of the stored values, or NULL
  const float* stat_in;                  // LayerNorm fold: [rows][k/32][2] table of the `a` rows, or NULL
  int hm_d, hm_planes; float hm_scale;   // head-major output: plane width D, scaled planes, their factor
  const void* ln_gamma; const void* ln_beta;   // direct LayerNorm prologue of the row-panel family (T [k])
  const float* w_scale;                  // row-panel family, fp8 weights: per-output-channel dequantisation scale (fp32 [n])
  int no_stagger;                        // conv3s A/B switch (DD_STAGGER=0)
  int persist;                           // dd_gemm2_kernel: the grid is smaller than the tile count (see the kernel)
  uint64_t* dbg_stamps;                  // DD_DBG_STAMP builds only
  float inv_hw, inv_wout, inv_rpi;       // 1 / (hout*wout), 1 / wout, 1 / rows_per_inst for dd_fdiv
  void* ln_out; int64_t ld_ln_out;       // LayerNorm EMITTED by the epilogue of the 80x320 tile (second output)
  const void* lno_gamma; const void* lno_beta;
  const void* pf_ptr; uint32_t pf_bytes; int pf_blocks;   // weight prefetch carried by spare workgroups (dd_prefetch_block)
};

// n / d for 0 <= n < 2^22 (host-checked: rows) and the host-side inv = 1.0f / d: (n + 0.5) * inv is never within
// float rounding of an integer boundary there (error <= 2^-23 * (n + 0.5) / d < 0.5 / d), so truncation gives the exact quotient — 3 VALU
// instructions instead of the ~35 of a 32-bit integer division (the table-building prologues divide by the
// image size and width once per tile row: a third of the direct conv kernel's VALU instructions).
__device__ __forceinline__ int dd_fdiv(int n, float inv) { return (int)(((float)n + 0.5f) * inv); }


template <typename T>
__device__ __forceinline__ void store8(const GemmParams& p, int64_t row, int col, float (&v)[8]) {
  if (p.hm_d) {                            // one [rows][D] plane per head; the Q planes carry the softmax scale
    const int plane = col / p.hm_d;
    if (plane < p.hm_planes) {
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] *= p.hm_scale;
    }
    dd_st16(reinterpret_cast<T*>(p.out) + ((int64_t)plane * p.rows + row) * p.hm_d + (col - plane * p.hm_d),
            dd_pack8<T>(v));
    return;
  }
  if (p.out_f32) {
    float* o = reinterpret_cast<float*>(p.out) + row * p.ldc + col;
    *reinterpret_cast<f32x4*>(o) = f32x4{v[0], v[1], v[2], v[3]};
    *reinterpret_cast<f32x4*>(o + 4) = f32x4{v[4], v[5], v[6], v[7]};
  } else {
    dd_st16(reinterpret_cast<T*>(p.out) + row * p.ldc + col, dd_pack8<T>(v));
  }
}

// --- epilogue on 8 consecutive output channels of one row --------------------------------
template <typename T>
__device__ __forceinline__ void epilogue_store8(const GemmParams& p, int row, int col, float (&v)[8]) {
  if (p.bias) {
    float b[8];
    dd_unpack8<T>(dd_ld16(reinterpret_cast<const T*>(p.bias) + col), b);
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] += b[i];
  }
  if (p.rowvec) {
    const int inst = dd_fdiv(row, p.inv_rpi);
    float b[8];
    dd_unpack8<T>(dd_ld16(reinterpret_cast<const T*>(p.rowvec) + (int64_t)inst * p.ld_rowvec + col), b);
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] += b[i];
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) v[i] *= p.alpha;
  if (p.res) {
    float b[8];
    dd_unpack8<T>(dd_ld16(reinterpret_cast<const T*>(p.res) + (int64_t)row * p.ldres + col), b);
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] += b[i];
  }
  if (p.act == DD_EPI_SILU) {
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = dd_silu_f(v[i]);
  }
  if (p.accumulate) {
    float b[8];
    dd_unpack8<T>(dd_ld16(reinterpret_cast<T*>(p.out) + (int64_t)row * p.ldc + col), b);
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] += b[i];
  }
  store8<T>(p, row, col, v);
}

// XCD-aware bijective remap of a 1-D block id (guide T1): blocks b, b+8, ... share an XCD;
// give each XCD a contiguous range of tiles.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
  const int xcd = bid & 7;
  const int q = nwg >> 3, r = nwg & 7;
  const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + (bid >> 3);
}

// WEIGHT PREFETCH BY SPARE WORKGROUPS.  A step streams 3.3 GB of weights through the 256 MiB Infinity Cache, so every
// launch meets its weights cold in HBM (measured: 1092x1280x1280 16.3 us with HBM-cold, 14.2 with Infinity-Cache-resident,
// 12.4 with L2-resident weights) while HBM idles > 95 % of the time.  A launch whose grid leaves workgroup slots empty
// (the few-row levels: 80-240 tiles on 256 CUs) carries `pf_blocks` extra workgroups at the END of its grid that only
// READ the weights of the NEXT weight-bearing launch of the same stream (host: ops._pf_hint) — 16 B per lane, 16 loads
// in flight — so that launch finds them in the memory-side cache.  No graph node, no stream edge (a prefetch stream
// inside the captured step cost 4 ms); reads only, so there is nothing to synchronise.
// MEASURED on the step: not a win at any size (-1.7 % with 32 MB / 96 workgroups, +-0.2 % with 1-2 MB / 8-16); the host
// side does not pass hints by default (ops.PREFETCH).
template <int THREADS>
__device__ __forceinline__ void dd_prefetch_block(const GemmParams& p, int pb) {
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.pf_ptr), 0, p.pf_bytes, 0x00020000);
  const uint32_t per = ((p.pf_bytes / (uint32_t)p.pf_blocks) + 15u) & ~15u;
  const uint32_t beg = (uint32_t)pb * per;
  const uint32_t end = min(p.pf_bytes, beg + per);
  u32x4 acc = {0u, 0u, 0u, 0u};
  for (uint32_t off = beg + threadIdx.x * 16u; off < end; off += THREADS * 16u * 16u) {
    u32x4 v[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) {                 // out-of-range lanes read zeros (buffer bounds) — never past the tensor
      const uint32_t o = off + (uint32_t)j * THREADS * 16u;
      v[j] = __builtin_amdgcn_raw_buffer_load_b128(rs, o < end ? o : 0xFFFFFFF0u, 0, 0);
    }
#pragma unroll
    for (int j = 0; j < 16; ++j) acc |= v[j];
  }
  asm volatile("" ::"v"(acc[0] | acc[1] | acc[2] | acc[3]));     // keeps the loads alive; nothing is stored
}

// ---- accumulator tile -> global (shared by both kernel families) ---------------------------
// acc[tn][tm][reg]: output row = tile row tm*16 + (lane & 15),
//                   output col = q*(4*TN) + tn*4 + reg  (q = lane >> 4)   [non-GEGLU]
// Every global read of the epilogue (bias, time-embedding vector, residual, accumulate target) is
// issued before the stores of its row batch: `out` may alias `res`, so a load placed after a store could
// not be hoisted by the compiler and the tile would pay one memory round trip per 8-column group.
template <typename T, int TM, int TN, bool GEGLU>
__device__ __forceinline__ void store_tile(const GemmParams& p, f32x4 (&acc)[TN][TM], int block_m0,
                                           int block_n0, int wave_m, int wave_n, int lane, int row_end,
                                           const float* ln_mean = nullptr, const float* ln_rstd = nullptr,
                                           int tile_id = 0, int* lds_flag = nullptr) {
  const int q = lane >> 4;
  const int c = lane & 15;
  const int row0 = block_m0 + wave_m * (TM * 16) + c;
  if constexpr (GEGLU) {
    constexpr int TH = TN / 2;
    constexpr int NG = TH / 2;
    const int col0 = block_n0 + wave_n * (TH * 16) + q * (4 * TH);
    u32x4 bh[NG], bg[NG];
    f32x4 lsh[NG][2], lsg[NG][2], lbh[NG][2], lbg[NG][2];     // LayerNorm fold: column sums / folded bias
    if (ln_mean) {
#pragma unroll
      for (int g8 = 0; g8 < NG; ++g8) {
        const int col = min(col0 + g8 * 8, p.n - 8);
#pragma unroll
        for (int h2 = 0; h2 < 2; ++h2) {
          lsh[g8][h2] = *reinterpret_cast<const f32x4*>(p.ln_colsum + col + 4 * h2);
          lsg[g8][h2] = *reinterpret_cast<const f32x4*>(p.ln_colsum + p.n + col + 4 * h2);
          lbh[g8][h2] = *reinterpret_cast<const f32x4*>(p.ln_bias + col + 4 * h2);
          lbg[g8][h2] = *reinterpret_cast<const f32x4*>(p.ln_bias + p.n + col + 4 * h2);
        }
      }
    }
    if (p.bias) {
#pragma unroll
      for (int g8 = 0; g8 < NG; ++g8) {
        const int col = min(col0 + g8 * 8, p.n - 8);
        bh[g8] = dd_ld16(reinterpret_cast<const T*>(p.bias) + col);
        bg[g8] = dd_ld16(reinterpret_cast<const T*>(p.bias) + p.n + col);
      }
    }
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) {
      const int row = row0 + tm * 16;
      if (row >= row_end) continue;
#pragma unroll
      for (int g8 = 0; g8 < NG; ++g8) {
        const int col = col0 + g8 * 8;
        if (col >= p.n) continue;
        float h[8], g[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          h[e] = acc[g8 * 2 + (e >> 2)][tm][e & 3];
          g[e] = acc[TH + g8 * 2 + (e >> 2)][tm][e & 3];
        }
        if (ln_mean) {
          const int lr = wave_m * (TM * 16) + tm * 16 + c;
          const float mu = ln_mean[lr], rs = ln_rstd[lr];
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            h[e] = rs * (h[e] - mu * lsh[g8][e >> 2][e & 3]) + lbh[g8][e >> 2][e & 3];
            g[e] = rs * (g[e] - mu * lsg[g8][e >> 2][e & 3]) + lbg[g8][e >> 2][e & 3];
          }
        }
        if (p.bias) {
          float b[8];
          dd_unpack8<T>(bh[g8], b);
#pragma unroll
          for (int e = 0; e < 8; ++e) h[e] += b[e];
          dd_unpack8<T>(bg[g8], b);
#pragma unroll
          for (int e = 0; e < 8; ++e) g[e] += b[e];
        }
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = h[e] * dd_gelu_erf_f(g[e]);
        dd_st16(reinterpret_cast<T*>(p.out) + (int64_t)row * p.ldc + col, dd_pack8<T>(v));
      }
    }
  } else {
    constexpr int NG = TN / 2;
    const int col0 = block_n0 + wave_n * (TN * 16) + q * (4 * TN);
    if (p.partial) {                       // split-K slab: fp32 stores
      // Two forms.  tile_counters == NULL: plain stores, dd_splitk_reduce_kernel (a second launch) adds the slabs and
      // runs the epilogue.  tile_counters != NULL: IN-LAUNCH ordered reduction by the K-slice that arrives last at
      // the tile's counter — the write-through recipe of the CDNA4 guide (cdna_hip_programming.md "In-launch split-K
      // reduction", MI355X_MICROARCH.md "inter-workgroup visibility"): the slabs are stored sc1 (write-through, no
      // release fence: the L2s of the 8 XCDs are not coherent and a buffer_wbl2 per workgroup costs more than the
      // second launch — the round-1 form with __threadfence() ran 32 -> 66 us on the 4x7 conv), EVERY storing wave
      // drains its stores (s_waitcnt vmcnt(0)), the workgroup barrier, ONE lane's relaxed agent-scope ticket; the
      // workgroup whose add returns split-1 reads all slabs with sc1 loads (L1-bypassing; every load of the handed-off
      // bytes) in slice order — bit-identical to the two-launch form whoever is last — and runs the epilogue.
      const bool ink = p.tile_counters != nullptr;
      const __amdgpu_buffer_rsrc_t rs_p = __builtin_amdgcn_make_buffer_rsrc(p.partial, 0, p.partial_bytes, 0x00020000);
#pragma unroll
      for (int tm = 0; tm < TM; ++tm) {
        const int row = row0 + tm * 16;
        if (row >= row_end) continue;
#pragma unroll
        for (int g8 = 0; g8 < NG; ++g8) {
          const int col = col0 + g8 * 8;
          if (col >= p.n) continue;
          if (ink) {
            const uint32_t off = (uint32_t)((((int64_t)blockIdx.z * p.rows + row) * p.n + col) * 4);
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, acc[g8 * 2][tm]), rs_p, off, 0, 16);
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, acc[g8 * 2 + 1][tm]), rs_p, off + 16, 0, 16);
          } else {
            float* dst = p.partial + ((int64_t)blockIdx.z * p.rows + row) * p.n + col;
            *reinterpret_cast<f32x4*>(dst) = acc[g8 * 2][tm];
            *reinterpret_cast<f32x4*>(dst + 4) = acc[g8 * 2 + 1][tm];
          }
        }
      }
      if (!ink) return;                    // two-launch mode: dd_splitk_reduce_kernel runs the epilogue
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // every storing wave: its write-through stores have landed
      __syncthreads();                                      // ... and every wave is past its last LDS read
      if (threadIdx.x == 0) {
        const int prev = __hip_atomic_fetch_add(p.tile_counters + tile_id, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int last = prev == (int)gridDim.z - 1;
        if (last) __hip_atomic_store(p.tile_counters + tile_id, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // stream-ordered next launch
        *lds_flag = last;
      }
      __syncthreads();
      if (!*lds_flag) return;
#pragma unroll
      for (int tm = 0; tm < TM; ++tm) {
        const int rowc = min(row0 + tm * 16, p.rows - 1);
#pragma unroll
        for (int g8 = 0; g8 < NG; ++g8) {
          const int colc = min(col0 + g8 * 8, p.n - 8);
          f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = {0.f, 0.f, 0.f, 0.f};
          const uint32_t off0 = (uint32_t)(((int64_t)rowc * p.n + colc) * 4);
          const uint32_t zstride = (uint32_t)((int64_t)p.rows * p.n * 4);
          for (int z = 0; z < (int)gridDim.z; ++z) {
            const f32x4 a = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_p, off0 + (uint32_t)z * zstride, 0, 16));
            const f32x4 b = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_p, off0 + (uint32_t)z * zstride + 16, 0, 16));
            s0[0] += a[0]; s0[1] += a[1]; s0[2] += a[2]; s0[3] += a[3];
            s1[0] += b[0]; s1[1] += b[1]; s1[2] += b[2]; s1[3] += b[3];
          }
          acc[g8 * 2][tm] = s0;
          acc[g8 * 2 + 1][tm] = s1;
        }
      }
    }
    // Rows are handled in (at most) two batches: per batch, phase 1 issues ALL its loads (clamped
    // addresses, nothing predicated), phase 2 does the arithmetic and the stores.  One batch would
    // keep TM*TN/2*3 16-B vectors live next to the accumulators (128x128 tile: > 256 VGPRs).
    constexpr int TMB = (TM >= 4 && TM % 2 == 0) ? TM / 2 : TM;      // batches must tile TM exactly
    u32x4 rb[NG];
    int colc[NG];
#pragma unroll
    for (int g8 = 0; g8 < NG; ++g8) colc[g8] = min(col0 + g8 * 8, p.n - 8);
    if (p.bias) {
#pragma unroll
      for (int g8 = 0; g8 < NG; ++g8) rb[g8] = dd_ld16(reinterpret_cast<const T*>(p.bias) + colc[g8]);
    }
    f32x4 lcs[NG][2], lcb[NG][2];                        // LayerNorm fold: column sums / folded bias
    if (ln_mean) {
#pragma unroll
      for (int g8 = 0; g8 < NG; ++g8)
#pragma unroll
        for (int h2 = 0; h2 < 2; ++h2) {
          lcs[g8][h2] = *reinterpret_cast<const f32x4*>(p.ln_colsum + colc[g8] + 4 * h2);
          lcb[g8][h2] = *reinterpret_cast<const f32x4*>(p.ln_bias + colc[g8] + 4 * h2);
        }
    }
#pragma unroll
    for (int tb = 0; tb < TM; tb += TMB) {
      u32x4 rv[TMB][NG], rr[TMB][NG], ra[TMB][NG];
#pragma unroll
      for (int t2 = 0; t2 < TMB; ++t2) {
        const int rowc = min(row0 + (tb + t2) * 16, p.rows - 1);
        if (p.rowvec) {
          const int inst = dd_fdiv(rowc, p.inv_rpi);
#pragma unroll
          for (int g8 = 0; g8 < NG; ++g8)
            rv[t2][g8] = dd_ld16(reinterpret_cast<const T*>(p.rowvec) + (int64_t)inst * p.ld_rowvec + colc[g8]);
        }
        if (p.res) {
#pragma unroll
          for (int g8 = 0; g8 < NG; ++g8)
            rr[t2][g8] = dd_ld16(reinterpret_cast<const T*>(p.res) + (int64_t)rowc * p.ldres + colc[g8]);
        }
        if (p.accumulate) {
#pragma unroll
          for (int g8 = 0; g8 < NG; ++g8)
            ra[t2][g8] = dd_ld16(reinterpret_cast<const T*>(p.out) + (int64_t)rowc * p.ldc + colc[g8]);
        }
      }
      // arithmetic in the reference's order (bias, time vector, alpha, residual, act, accumulate) + stores
#pragma unroll
      for (int t2 = 0; t2 < TMB; ++t2) {
        const int tm = tb + t2;
        const int row = row0 + tm * 16;
        float st_s = 0.f, st_q = 0.f;                      // row statistics of this lane's columns
        if (row < row_end) {
#pragma unroll
        for (int g8 = 0; g8 < NG; ++g8) {
          const int col = col0 + g8 * 8;
          if (col >= p.n) continue;
          float v[8], b[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = acc[g8 * 2 + (e >> 2)][tm][e & 3];
          if (ln_mean) {
            const int lr = wave_m * (TM * 16) + tm * 16 + c;
            const float mu = ln_mean[lr], rs = ln_rstd[lr];
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = rs * (v[e] - mu * lcs[g8][e >> 2][e & 3]) + lcb[g8][e >> 2][e & 3];
          }
          if (p.bias) {
            dd_unpack8<T>(rb[g8], b);
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] += b[e];
          }
          if (p.rowvec) {
            dd_unpack8<T>(rv[t2][g8], b);
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] += b[e];
          }
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] *= p.alpha;
          if (p.res) {
            dd_unpack8<T>(rr[t2][g8], b);
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] += b[e];
          }
          if (p.act == DD_EPI_SILU) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = dd_silu_f(v[e]);
          }
          if (p.accumulate) {
            dd_unpack8<T>(ra[t2][g8], b);
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] += b[e];
          }
          store8<T>(p, row, col, v);
          if (p.stat_out) {
#pragma unroll
            for (int e = 0; e < 8; ++e) { st_s += v[e]; st_q += v[e] * v[e]; }
          }
        }
        }
        if constexpr (TN == 2 || TN == 4) {
          if (p.stat_out) {              // uniform: every lane of the wave takes part in the shuffles
            // a lane holds 4*TN columns of its row; 32-column groups are 4 (TN = 2) or 2 (TN = 4) lanes q
            st_s += __shfl_xor(st_s, 16, 64);  st_q += __shfl_xor(st_q, 16, 64);
            if (TN == 2) { st_s += __shfl_xor(st_s, 32, 64);  st_q += __shfl_xor(st_q, 32, 64); }
            const int gcol = block_n0 + wave_n * (TN * 16) + (TN == 2 ? 0 : (q >> 1) * 32);
            const bool writer = TN == 2 ? q == 0 : (q & 1) == 0;
            if (writer && row < row_end && gcol < p.n) {
              float* dst = p.stat_out + ((int64_t)row * (p.n >> 5) + (gcol >> 5)) * 2;
              dst[0] = st_s;
              dst[1] = st_q;
            }
          }
        }
      }
    }
  }
}

template <typename T, int WAVES_M, int WAVES_N, int TM, int TN, bool CONV, bool GEGLU>
__global__ __launch_bounds__(64 * WAVES_M * WAVES_N)
void dd_gemm_kernel(const GemmParams p) {
  using V8 = typename dd_vec<T>::v8;
  constexpr int NT = 64 * WAVES_M * WAVES_N;
  constexpr int BM = WAVES_M * TM * 16;
  constexpr int BN = WAVES_N * TN * 16;           // weight-tile rows
  constexpr int BN_OUT = GEGLU ? BN / 2 : BN;     // output columns per block
  constexpr int XI = BM * 8 / NT;                 // 16-B chunks per thread, activation tile
  constexpr int WI = BN * 8 / NT;
  static_assert(BM * 8 % NT == 0 && BN * 8 % NT == 0, "tile/threads mismatch");
  static_assert(TN % 2 == 0 && (!GEGLU || TN % 4 == 0), "TN");

  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  T* Xs = reinterpret_cast<T*>(smem);                         // [2][BM][64]
  T* Ws = Xs + 2 * BM * BK;                                   // [2][BN][64]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wave_m = wave / WAVES_N;
  const int wave_n = wave % WAVES_N;

  const int ntiles = p.tiles_m * p.tiles_n;
  const int tile = xcd_remap(blockIdx.x, ntiles);
  const int tile_m = tile / p.tiles_n;
  const int tile_n = tile % p.tiles_n;
  const int block_m0 = tile_m * BM;
  const int block_n0 = tile_n * BN_OUT;

  const int kbeg = blockIdx.z * p.k_per_split;
  const int kend = min(p.k, kbeg + p.k_per_split);
  const int nk = (kend - kbeg + BK - 1) / BK;

  // ---- per-thread loader state --------------------------------------------------------
  const int lchunk = tid & 7;        // which 16-B chunk of the 128-B tile row
  const int lrow0 = tid >> 3;        // first tile row handled by this thread
  constexpr int LROW_STEP = NT / 8;

  // activation rows
  int xm[XI];          // dense: global row (or -1).  conv: instance pixel base (or -1)
  int xiy[XI], xix[XI];
#pragma unroll
  for (int i = 0; i < XI; ++i) {
    const int r = block_m0 + lrow0 + i * LROW_STEP;
    if (r < p.rows) {
      if (CONV) {
        const int hw = p.hout * p.wout;
        const int inst = dd_fdiv(r, p.inv_hw);
        const int rem = r - inst * hw;
        const int oy = dd_fdiv(rem, p.inv_wout);
        const int ox = rem - oy * p.wout;
        xm[i] = inst;
        xiy[i] = oy * p.stride - 1;
        xix[i] = ox * p.stride - 1;
      } else {
        xm[i] = r; xiy[i] = 0; xix[i] = 0;
      }
    } else {
      xm[i] = -1; xiy[i] = 0; xix[i] = 0;
    }
  }
  // weight rows (permuted so each lane owns consecutive output channels)
  int64_t wofs[WI];    // element offset of the weight row, or -1
#pragma unroll
  for (int i = 0; i < WI; ++i) {
    const int R = lrow0 + i * LROW_STEP;           // LDS row in weight tile
    const int wv = R / (TN * 16);
    const int rho = R % (TN * 16);
    const int tn = rho >> 4, r = rho & 15;
    int n_glob;
    if (GEGLU) {
      constexpr int TH = TN / 2;
      const int t = tn % TH;
      const int loc = wv * (TH * 16) + (r >> 2) * (4 * TH) + t * 4 + (r & 3);
      const int col = block_n0 + loc;
      n_glob = (col < p.n) ? col + (tn >= TH ? p.n : 0) : -1;
    } else {
      const int loc = wv * (TN * 16) + (r >> 2) * (4 * TN) + tn * 4 + (r & 3);
      const int col = block_n0 + loc;
      n_glob = (col < p.n) ? col : -1;
    }
    wofs[i] = (n_glob >= 0) ? (int64_t)n_glob * p.k : -1;
  }

  u32x4 xreg[XI], wreg[WI];

  auto load_tiles = [&](int kt) {
    const int k = kbeg + kt * BK + lchunk * 8;
    const bool kok = k < kend;
    // weights
#pragma unroll
    for (int i = 0; i < WI; ++i) {
      u32x4 v = {0u, 0u, 0u, 0u};
      if (kok && wofs[i] >= 0) v = dd_ld16(reinterpret_cast<const T*>(p.w) + wofs[i] + k);
      wreg[i] = v;
    }
    // activations
    if (CONV) {
      const int tap = k / p.cin;
      const int ci = k - tap * p.cin;
      const int ky = tap / 3;
      const int kx = tap - ky * 3;
#pragma unroll
      for (int i = 0; i < XI; ++i) {
        u32x4 v = {0u, 0u, 0u, 0u};
        int iy = xiy[i] + ky, ix = xix[i] + kx;
        if (kok && xm[i] >= 0 && iy >= 0 && iy < p.hv && ix >= 0 && ix < p.wv) {
          if (p.upsample) {
            iy = min((int)floorf(iy * p.scale_h), p.hin - 1);
            ix = min((int)floorf(ix * p.scale_w), p.win - 1);
          }
          const int64_t off = (((int64_t)xm[i] * p.hin + iy) * p.win + ix) * p.cin + ci;
          v = dd_ld16(reinterpret_cast<const T*>(p.a) + off);
        }
        xreg[i] = v;
      }
    } else {
      const bool second = k >= p.k1;
#pragma unroll
      for (int i = 0; i < XI; ++i) {
        u32x4 v = {0u, 0u, 0u, 0u};
        if (kok && xm[i] >= 0) {
          const T* src = second
              ? reinterpret_cast<const T*>(p.a2) + (int64_t)xm[i] * p.lda2 + (k - p.k1)
              : reinterpret_cast<const T*>(p.a) + (int64_t)xm[i] * p.lda + k;
          v = dd_ld16(src);
        }
        xreg[i] = v;
      }
    }
  };

  auto store_tiles = [&](int buf) {
    T* xs = Xs + buf * BM * BK;
    T* ws = Ws + buf * BN * BK;
#pragma unroll
    for (int i = 0; i < XI; ++i) {
      const int R = lrow0 + i * LROW_STEP;
      dd_st16(xs + R * BK + ((lchunk ^ ((R >> 1) & 7)) << 3), xreg[i]);
    }
#pragma unroll
    for (int i = 0; i < WI; ++i) {
      const int R = lrow0 + i * LROW_STEP;
      dd_st16(ws + R * BK + ((lchunk ^ ((R >> 1) & 7)) << 3), wreg[i]);
    }
  };

  f32x4 acc[TN][TM];
#pragma unroll
  for (int i = 0; i < TN; ++i)
#pragma unroll
    for (int j = 0; j < TM; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // fragment addressing: LDS row = base + (lane & 15); chunk = (lane >> 4) + 4*ks, swizzled
  const int frow = lane & 15;
  const int fswz = (lane >> 1) & 7;     // == ((row >> 1) & 7) because tile bases are multiples of 16
  const int fchunk = lane >> 4;

  if (nk > 0) {
    load_tiles(0);
    store_tiles(0);
  }
  __syncthreads();

  int buf = 0;
  for (int kt = 0; kt < nk; ++kt) {
    if (kt + 1 < nk) load_tiles(kt + 1);
    const T* xs = Xs + buf * BM * BK + (wave_m * TM * 16 + frow) * BK;
    const T* ws = Ws + buf * BN * BK + (wave_n * TN * 16 + frow) * BK;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int cofs = ((fchunk + 4 * ks) ^ fswz) << 3;
      V8 wf[TN], xf[TM];
#pragma unroll
      for (int i = 0; i < TN; ++i) wf[i] = dd_as_v8<T>(dd_ld16(ws + i * 16 * BK + cofs));
#pragma unroll
      for (int j = 0; j < TM; ++j) xf[j] = dd_as_v8<T>(dd_ld16(xs + j * 16 * BK + cofs));
#pragma unroll
      for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = 0; j < TM; ++j) acc[i][j] = dd_mfma16(wf[i], xf[j], acc[i][j]);
    }
    if (kt + 1 < nk) store_tiles(buf ^ 1);
    __syncthreads();
    buf ^= 1;
  }

  store_tile<T, TM, TN, GEGLU>(p, acc, block_m0, block_n0, wave_m, wave_n, lane, p.rows, nullptr, nullptr, tile,
                               reinterpret_cast<int*>(smem));
}

// =============================================================================================
// Kernel family 2: LDS-DMA (buffer_load_dwordx4 ... lds, 16 B / lane) multi-stage ring.
//  * no staging registers and no ds_write: tiles land in LDS asynchronously, NSTAGE-1 K-steps ahead;
//  * the XOR swizzle is applied on the per-lane SOURCE offset (the DMA destination is lane-linear);
//  * padding taps / tile tails use an out-of-range lane offset: the descriptor's range check makes
//    the DMA deliver zeros, so nothing is predicated;
//  * counted s_waitcnt vmcnt(N) + raw s_barrier: one barrier per K-step, loads stay in flight
//    across it.
// =============================================================================================
// ---- epilogue of the 80 x 320 tile that ALSO emits LayerNorm(out) ---------------------------------------------
// A workgroup of 10 waves (1 x 10, TM = 5, TN = 2) owns 80 WHOLE rows of a 320-wide output: after bias / alpha /
// residual it rounds the row to T (what the next layer reads), stores it, and normalises it right there — two-pass
// fp32 statistics over the rounded values (the arithmetic of dd_layernorm), partial sums of the 10 waves combined
// through LDS in a fixed order (bit-reproducible) — writing LayerNorm(out) as a second tensor.  The producer of
// the residual stream thereby hands the next sub-layer its normalised input: no LayerNorm launch, no re-read of
// the stream (norm1 / norm2 / norm3 / norm4 of the 28x50 level, blocks.py:150-236).
template <typename T>
__device__ __forceinline__ void store_tile_ln(const GemmParams& p, f32x4 (&acc)[2][5], int block_m0, int wave_n,
                                              int lane, float* scratch) {
  constexpr int TM = 5, NWV = 10, BM = 80, NCOL = 320;
  const int q = lane >> 4, c = lane & 15;
  const int col = wave_n * 32 + q * 8;
  float bias[8], ga[8], be[8];
  if (p.bias) dd_unpack8<T>(dd_ld16(reinterpret_cast<const T*>(p.bias) + col), bias);
  dd_unpack8<T>(dd_ld16(reinterpret_cast<const T*>(p.lno_gamma) + col), ga);
  dd_unpack8<T>(dd_ld16(reinterpret_cast<const T*>(p.lno_beta) + col), be);
  u32x4 rr[TM];
  if (p.res) {
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) {
      const int64_t rowc = min(block_m0 + tm * 16 + c, p.rows - 1);
      rr[tm] = dd_ld16(reinterpret_cast<const T*>(p.res) + rowc * p.ldres + col);
    }
  }
  float v[TM][8], part[TM];
#pragma unroll
  for (int tm = 0; tm < TM; ++tm) {
    const int row = block_m0 + tm * 16 + c;
    float r[8];
    if (p.res) dd_unpack8<T>(rr[tm], r);
    float s = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float x = acc[e >> 2][tm][e & 3];
      if (p.bias) x += bias[e];
      x *= p.alpha;
      if (p.res) x += r[e];
      v[tm][e] = (float)(T)x;                            // the stored (rounded) value is what gets normalised
      s += v[tm][e];
    }
    if (row < p.rows) dd_st16(reinterpret_cast<T*>(p.out) + (int64_t)row * p.ldc + col, dd_pack8<T>(v[tm]));
    s += __shfl_xor(s, 16, 64);
    s += __shfl_xor(s, 32, 64);
    part[tm] = s;
  }
  __syncthreads();                                       // every wave is done with the operand ring: LDS is scratch now
  if (q == 0) {
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) scratch[wave_n * BM + tm * 16 + c] = part[tm];
  }
  __syncthreads();
  float mean[TM];
#pragma unroll
  for (int tm = 0; tm < TM; ++tm) {
    float s = 0.f;
#pragma unroll
    for (int w = 0; w < NWV; ++w) s += scratch[w * BM + tm * 16 + c];
    mean[tm] = s * (1.0f / (float)NCOL);
    float ss = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) { const float d = v[tm][e] - mean[tm]; ss += d * d; }
    ss += __shfl_xor(ss, 16, 64);
    ss += __shfl_xor(ss, 32, 64);
    part[tm] = ss;
  }
  __syncthreads();
  if (q == 0) {
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) scratch[wave_n * BM + tm * 16 + c] = part[tm];
  }
  __syncthreads();
#pragma unroll
  for (int tm = 0; tm < TM; ++tm) {
    const int row = block_m0 + tm * 16 + c;
    float ss = 0.f;
#pragma unroll
    for (int w = 0; w < NWV; ++w) ss += scratch[w * BM + tm * 16 + c];
    const float rstd = rsqrtf(ss * (1.0f / (float)NCOL) + p.ln_eps);
    float o[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = (v[tm][e] - mean[tm]) * rstd * ga[e] + be[e];
    if (row < p.rows) dd_st16(reinterpret_cast<T*>(p.ln_out) + (int64_t)row * p.ld_ln_out + col, dd_pack8<T>(o));
  }
}

template <int N>
__device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// buffer_load_dwordx4 ... offen lds: SGPR descriptor + a 32-bit byte offset per lane + a scalar
// offset.  An offset outside the descriptor's range reads zeros (hardware range check), which is how
// padding taps and tile tails are produced — no 64-bit pointer arithmetic, no select against a zero page.
__device__ __forceinline__ void bdma16(__amdgpu_buffer_rsrc_t rsrc, uint32_t voff, uint32_t soff, void* lds_wave_base) {
#ifndef DD_DBG_NODMA
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)lds_wave_base, 16,
                                           (int)voff, (int)soff, 0, 0);
#endif
}
// every buffer is < 2^31 bytes (checked on the host), so this lane offset is out of range whatever
// scalar offset is added to it
constexpr uint32_t DD_OOB = 0x80000000u;

// DD_DBG_STAMP (diagnostic build only, tools/build_dbg_libs.sh): wave 0 of every workgroup of the direct conv records
// s_memtime at phase boundaries plus s_memrealtime at both ends into the LAST MiB of the workspace (ops.py over-allocates
// it when DD_DBG_STAMP_WS=1); nothing reads them on the device.
#ifdef DD_DBG_STAMP
#define DD_STAMP(i) do { if (threadIdx.x == 0) dbg_t[i] = __builtin_readcyclecounter(); } while (0)
#else
#define DD_STAMP(i) do {} while (0)
#endif

// Occupancy target (round 3): the DENSE four-wave instantiations had grown to 240-272 registers (LayerNorm fold, row
// statistics, head-major planes, persistent walk ... all live in one body), i.e. ONE wave per SIMD and one workgroup per CU
// although their 48-72 KB rings would let two in — the situation in which a latency-bound K loop has nothing to hide
// behind.  Where two rings fit the LDS the compiler is told to fit two workgroups (<= 256 registers per wave).
template <int NW, int TM, int TN, int NSTAGE, bool CONV>
constexpr int gemm2_min_blocks() {
  return (!CONV && NW == 4 && TM * TN <= 8 && NSTAGE <= 3) ? 2 : 1;
}

template <typename T, int WAVES_M, int WAVES_N, int TM, int TN, int NSTAGE, bool CONV, bool GEGLU>
__global__ __launch_bounds__(64 * WAVES_M * WAVES_N, (gemm2_min_blocks<WAVES_M * WAVES_N, TM, TN, NSTAGE, CONV>()))
void dd_gemm2_kernel(const GemmParams p) {
  using V8 = typename dd_vec<T>::v8;
  constexpr int NW = WAVES_M * WAVES_N;
  constexpr int BM = WAVES_M * TM * 16;
  constexpr int BN = WAVES_N * TN * 16;
  constexpr int BN_OUT = GEGLU ? BN / 2 : BN;
  constexpr int XI = BM / 8 / NW;                 // DMA wave-instructions (8 rows x 128 B) per wave
  constexpr int WI = BN / 8 / NW;
  constexpr int LPS = XI + WI;                    // DMA instructions per thread per stage
  constexpr int STAGE = (BM + BN) * BK;           // elements per ring slot
  static_assert(BM % (8 * NW) == 0 && BN % (8 * NW) == 0, "tile/waves mismatch");
  static_assert(NW % 2 == 0, "swizzle must not depend on the instruction index");
  static_assert(TN % 2 == 0 && (!GEGLU || TN % 4 == 0), "TN");
  static_assert(NSTAGE >= 2 && NSTAGE <= 8, "NSTAGE");
  static_assert((NSTAGE - 2) * LPS <= 63, "vmcnt is a 6-bit counter");

#ifdef DD_DBG_STAMP
  uint64_t dbg_t[6];
  const uint64_t dbg_r0 = __builtin_amdgcn_s_memrealtime();
#endif
  DD_STAMP(0);
  if (p.pf_blocks && (int)blockIdx.x >= (int)gridDim.x - p.pf_blocks) {        // spare workgroup: prefetch only
    if (blockIdx.z == 0) dd_prefetch_block<64 * NW>(p, (int)blockIdx.x - ((int)gridDim.x - p.pf_blocks));
    return;
  }
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  T* ring = reinterpret_cast<T*>(smem);

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);    // provably wave-uniform -> SALU address math
  const int wave_m = wave / WAVES_N;
  const int wave_n = wave % WAVES_N;

  // PERSISTENT mode (p.persist: dense, no split-K, more tiles than resident workgroups): a workgroup walks the tiles
  // lin, lin + gridDim.x, ... and the DMA ring runs AHEAD across the tile boundary — the first NSTAGE-1 stages of
  // the next tile are issued during the last K-steps of the current one, so only the very first tile of a workgroup
  // pays the pipeline fill (measured: 25 % of a 5-step tile's life at K = 320, tools/gemm2_stamps.py) and the
  // epilogue's stores overlap the next tile's loads.
  const int ntiles = p.tiles_m * p.tiles_n;
  int lin = blockIdx.x;                              // the tile being multiplied (consumer side)
  int tile = xcd_remap(lin, ntiles);
  int block_m0 = (tile / p.tiles_n) * BM;
  int block_n0 = (tile % p.tiles_n) * BN_OUT;

  const int kbeg = blockIdx.z * p.k_per_split;
  const int kend = min(p.k, kbeg + p.k_per_split);
  const int nk = (kend - kbeg + BK - 1) / BK;

  // DMA mapping: instruction j of this wave fills tile rows (j*NW + wave)*8 .. +7; lane l writes
  // row (l >> 3), chunk position (l & 7).  Logical chunk = position ^ ((row >> 1) & 7), which for an
  // even number of waves does not depend on j.
  const int lrow = lane >> 3;
  const int lc = (lane & 7) ^ ((((wave & 1) << 2) + (lane >> 4)) & 7);
  const uint32_t lcb = (uint32_t)lc * 16u;          // this lane's 16-B chunk inside the 128-B K segment

  // All address state lives in per-lane byte-offset tables that change at most once per conv tap
  // (or at the a/a2 seam); a K-step only moves SCALAR offsets.  K, cin and k1 are multiples of 64
  // here (the host routes other shapes to the register-staged family), so a K-step never straddles
  // a tap or the seam.  Exactly ONE DMA instruction per (operand, j) and stage: the counted vmcnt
  // waits below rely on it.
  uint32_t wv[WI];                                  // weight rows: n * K * 2 + chunk, or out of range
  auto make_wv = [&](const int bn0) __attribute__((always_inline)) {
#pragma unroll
  for (int j = 0; j < WI; ++j) {
    const int R = (j * NW + wave) * 8 + lrow;
    const int wvi = R / (TN * 16);
    const int rho = R % (TN * 16);
    const int tn = rho >> 4, r = rho & 15;
    int n_glob;
    if (GEGLU) {
      constexpr int TH = TN / 2;
      const int t = tn % TH;
      const int loc = wvi * (TH * 16) + (r >> 2) * (4 * TH) + t * 4 + (r & 3);
      const int col = bn0 + loc;
      n_glob = (col < p.n) ? col + (tn >= TH ? p.n : 0) : -1;
    } else {
      const int loc = wvi * (TN * 16) + (r >> 2) * (4 * TN) + tn * 4 + (r & 3);
      const int col = bn0 + loc;
      n_glob = (col < p.n) ? col : -1;
    }
    wv[j] = n_glob >= 0 ? (uint32_t)n_glob * (uint32_t)p.k * 2u + lcb : DD_OOB;
  }
  };
  make_wv(block_n0);

  uint32_t xe[XI];                                  // activation rows: offsets for the current tap / source a
  uint32_t xe2[CONV ? 1 : XI];                      // dense: offsets into a2
  uint32_t syo[CONV ? XI : 1][3], sxo[CONV ? XI : 1][3], xbits[CONV ? XI : 1];   // conv: per-tap source offsets
#pragma unroll
  for (int j = 0; j < XI; ++j) {
    const int r = block_m0 + (j * NW + wave) * 8 + lrow;
    const bool rv = r < p.rows;
    if (CONV) {
      const int hw = p.hout * p.wout;
      const int rr = rv ? r : 0;
      const int inst = dd_fdiv(rr, p.inv_hw);
      const int rem = rr - inst * hw;
      const int oy = dd_fdiv(rem, p.inv_wout);
      const int ox = rem - oy * p.wout;
      const int iy0 = oy * p.stride - 1, ix0 = ox * p.stride - 1;
      uint32_t bits = 0;
#pragma unroll
      for (int t = 0; t < 3; ++t) {
        const int iy = iy0 + t, ix = ix0 + t;
        const bool vy = iy >= 0 && iy < p.hv, vx = ix >= 0 && ix < p.wv;
        int sy = min(max(iy, 0), p.hv - 1), sx = min(max(ix, 0), p.wv - 1);
        if (p.upsample) {                             // torch nearest: min(floor(dst * in/out), in - 1)
          sy = min((int)floorf(sy * p.scale_h), p.hin - 1);
          sx = min((int)floorf(sx * p.scale_w), p.win - 1);
        }
        syo[j][t] = (uint32_t)((inst * p.hin + sy) * p.win) * (uint32_t)p.cin * 2u + lcb;
        sxo[j][t] = (uint32_t)(sx * p.cin) * 2u;
        if (vy) bits |= 1u << t;
        if (vx) bits |= 8u << t;
      }
      uint32_t m9 = 0;                                // bit (ky*3+kx): tap reads a real pixel
#pragma unroll
      for (int t = 0; t < 9; ++t)
        if (rv && ((bits >> (t / 3)) & 1u) && ((bits >> (3 + t % 3)) & 1u)) m9 |= 1u << t;
      xbits[j] = m9;
      xe[j] = DD_OOB;
    } else {
      xe[j] = rv ? (uint32_t)r * (uint32_t)p.lda * 2u + lcb : DD_OOB;
      xe2[j] = rv ? (uint32_t)r * (uint32_t)p.lda2 * 2u + lcb : DD_OOB;
    }
  }
  auto make_xe = [&](const int bm0) __attribute__((always_inline)) {       // dense: tables of another row tile
#pragma unroll
    for (int j = 0; j < XI; ++j) {
      const int r = bm0 + (j * NW + wave) * 8 + lrow;
      const bool rv = r < p.rows;
      xe[j] = rv ? (uint32_t)r * (uint32_t)p.lda * 2u + lcb : DD_OOB;
      xe2[j] = rv ? (uint32_t)r * (uint32_t)p.lda2 * 2u + lcb : DD_OOB;
    }
  };
  // conv: point xe[] at tap `tap` (table select by mask arithmetic: a select of array elements
  // would force the tables to scratch)
  auto set_tap = [&](int tap) __attribute__((always_inline)) {
    if (CONV) {
      const int ky = (tap * 11) >> 5;                 // tap / 3 for tap in [0, 9]
      const int kx = tap - ky * 3;
      const uint32_t y0 = 0u - (uint32_t)(ky == 0), y1 = 0u - (uint32_t)(ky == 1), y2 = 0u - (uint32_t)(ky == 2);
      const uint32_t x0 = 0u - (uint32_t)(kx == 0), x1 = 0u - (uint32_t)(kx == 1), x2 = 0u - (uint32_t)(kx == 2);
#pragma unroll
      for (int j = 0; j < XI; ++j) {
        const uint32_t oy = (syo[j][0] & y0) | (syo[j][1] & y1) | (syo[j][2] & y2);
        const uint32_t ox = (sxo[j][0] & x0) | (sxo[j][1] & x1) | (sxo[j][2] & x2);
        const uint32_t m = 0u - ((xbits[j] >> tap) & 1u);
        xe[j] = ((oy + ox) & m) | (DD_OOB & ~m);
      }
    }
  };
  const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.w), 0, p.w_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.a), 0, p.a_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_a2 = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<void*>(p.a2 ? p.a2 : p.a), 0, p.a2 ? p.a2_bytes : p.a_bytes, 0x00020000);

  // issue cursor (all scalar): next K offset, and for conv its tap / channel split
  int ik0 = kbeg;
  int itap = CONV ? kbeg / p.cin : 0;
  int ici0 = CONV ? kbeg - itap * p.cin : 0;
  set_tap(itap);
  auto issue_next = [&](int slot) __attribute__((always_inline)) {
    T* xs = ring + slot * STAGE;
    T* ws = xs + BM * BK;
#ifdef DD_DBG_SAMEK      // diagnostic: every K-step re-stages the SAME bytes (L1-resident after the first step)
    const uint32_t ksoff = 0u;
#else
    const uint32_t ksoff = (uint32_t)ik0 * 2u;
#endif
#pragma unroll
    for (int j = 0; j < WI; ++j) bdma16(rs_w, wv[j], ksoff, ws + (j * NW + wave) * 8 * BK);
    if (CONV) {
      const uint32_t csoff = (uint32_t)ici0 * 2u;
#pragma unroll
      for (int j = 0; j < XI; ++j) bdma16(rs_a, xe[j], csoff, xs + (j * NW + wave) * 8 * BK);
      ici0 += BK;
      if (ici0 >= p.cin) {                            // scalar branch, no DMA inside
        ici0 = 0;
        ++itap;
        set_tap(itap);
      }
    } else if (ik0 >= p.k1) {                         // scalar; both arms issue XI DMAs
      const uint32_t k2 = (uint32_t)(ik0 - p.k1) * 2u;
#pragma unroll
      for (int j = 0; j < XI; ++j) bdma16(rs_a2, xe2[j], k2, xs + (j * NW + wave) * 8 * BK);
    } else {
#pragma unroll
      for (int j = 0; j < XI; ++j) bdma16(rs_a, xe[j], ksoff, xs + (j * NW + wave) * 8 * BK);
    }
    ik0 += BK;
  };

  f32x4 acc[TN][TM];
#pragma unroll
  for (int i = 0; i < TN; ++i)
#pragma unroll
    for (int j = 0; j < TM; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int frow = lane & 15;
  const int fswz = (lane >> 1) & 7;
  const int fchunk = lane >> 4;

  DD_STAMP(1);
#pragma unroll
  for (int s0 = 0; s0 < NSTAGE - 1; ++s0)
    if (s0 < nk) issue_next(s0);
  DD_STAMP(2);

  // LayerNorm fold: row statistics of the block's A rows (K = 40 * lpr columns: lpr lanes share a
  // row, five 16-B vectors per lane), computed while the first stages are in flight.
  __shared__ float s_ln_mean[BM], s_ln_rstd[BM];
  if (!CONV && p.ln_colsum && p.stat_in) {
    // the producer of `a` left per-row partial sums (one pair per 32 columns): a few loads per row
    const int parts = p.k >> 5;
    const float inv_k = 1.0f / (float)p.k;
    for (int r = tid; r < BM; r += NW * 64) {
      const float* src = p.stat_in + (int64_t)min(block_m0 + r, p.rows - 1) * parts * 2;
      float sum = 0.f, sq = 0.f;
      for (int i = 0; i < parts; i += 2) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(src + i * 2);
        sum += v[0] + v[2];
        sq += v[1] + v[3];
      }
      const float mean = sum * inv_k;
      s_ln_mean[r] = mean;
      s_ln_rstd[r] = rsqrtf(fmaxf(sq * inv_k - mean * mean, 0.f) + p.ln_eps);
    }
    __syncthreads();
  } else if (!CONV && p.ln_colsum) {
    const int lpr = p.k / 40;                           // 8 / 16 / 32 (host-checked)
    const int rpw = 64 / lpr;
    const int sub = lane & (lpr - 1);
    const float inv_k = 1.0f / (float)p.k;
    for (int r0 = wave * rpw; r0 < BM; r0 += NW * rpw) {
      const int r = r0 + lane / lpr;
      const int64_t grow = min(block_m0 + r, p.rows - 1);
      float sum = 0.f, sq = 0.f;
      u32x4 raw[5];
#pragma unroll
      for (int i = 0; i < 5; ++i)
        raw[i] = dd_ld16(reinterpret_cast<const T*>(p.a) + grow * p.lda + (sub + i * lpr) * 8);
#pragma unroll
      for (int i = 0; i < 5; ++i) {
        float f[8];
        dd_unpack8<T>(raw[i], f);
#pragma unroll
        for (int e = 0; e < 8; ++e) { sum += f[e]; sq += f[e] * f[e]; }
      }
      for (int o = lpr >> 1; o > 0; o >>= 1) { sum += __shfl_xor(sum, o, 64); sq += __shfl_xor(sq, o, 64); }
      if (sub == 0) {
        const float mean = sum * inv_k;
        s_ln_mean[r] = mean;
        s_ln_rstd[r] = rsqrtf(fmaxf(sq * inv_k - mean * mean, 0.f) + p.ln_eps);
      }
    }
    __syncthreads();
  }

  // STAGGER (workgroups of >= 8 waves: two or more waves per SIMD behind ONE barrier per K-step would read LDS
  // together and then contend for the matrix pipe together): the second half of the waves executes the MFMAs of
  // K-step kt-1 (operands already in registers) BEFORE the fragment reads of K-step kt, i.e. half a step out of
  // phase with the first half, so one wave's MFMAs run beside its SIMD partner's LDS traffic.  Same arithmetic
  // in the same order per accumulator -> bit-identical results.  Fragments are double-buffered by step parity
  // (compile-time: the loop is unrolled by two).  The direct conv kernel uses this (5.39 -> 4.78 us per 9 steps).
  // MEASURED on this family (tiles 16 / 20 / 26, hot graph chains): 2-9 % SLOWER than the plain schedule (L0 conv
  // 41.4 -> 45.0 us, GEGLU 53.0 -> 55.0 us, 16800x320x1600 27.3 -> 28.7 us) — unlike the direct conv, whose steps
  // carry 12 tap gathers per wave; and the 10-wave 160-wide tiles (168 VGPRs) would spill.  Compiled out.
  constexpr bool STAG = false;
  const bool late = STAG && wave >= NW / 2 && !p.no_stagger;
  V8 wf[STAG ? 2 : 1][2][TN], xf[STAG ? 2 : 1][2][TM];
  auto mfma_step = [&](auto par_c) __attribute__((always_inline)) {
    constexpr int par = decltype(par_c)::value;
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
      for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = 0; j < TM; ++j) acc[i][j] = dd_mfma16(wf[par][ks][i], xf[par][ks][j], acc[i][j]);
    }
    __builtin_amdgcn_s_setprio(0);
  };
  int sbase = 0;                       // ring slot of this tile's stage 0 (persistent: tiles follow each other in the ring)
  bool have_next = false;              // persistent: another tile follows, its first stages are issued from this one
  auto kstep = [&](const int kt, auto par_c) __attribute__((always_inline)) {
    constexpr int par = STAG ? decltype(par_c)::value : 0;
    // stage kt must have landed; up to NSTAGE-2 younger stages may stay in flight
    if (NSTAGE == 2) {
      wait_vmcnt<0>();
    } else {
      const int ahead = have_next ? NSTAGE - 2 : min(nk - 1 - kt, NSTAGE - 2);     // scalar; stages allowed to stay in flight
      if (ahead <= 0) wait_vmcnt<0>();
      else if (ahead == 1 || NSTAGE <= 3) wait_vmcnt<(NSTAGE > 2 ? 1 : 0) * LPS>();
      else if (ahead == 2 || NSTAGE <= 4) wait_vmcnt<(NSTAGE > 3 ? 2 : 0) * LPS>();
      else if (ahead == 3 || NSTAGE <= 5) wait_vmcnt<(NSTAGE > 4 ? 3 : 0) * LPS>();
      else if (ahead == 4 || NSTAGE <= 6) wait_vmcnt<(NSTAGE > 5 ? 4 : 0) * LPS>();
      else if (ahead == 5 || NSTAGE <= 7) wait_vmcnt<(NSTAGE > 6 ? 5 : 0) * LPS>();
      else wait_vmcnt<(NSTAGE > 7 ? 6 : 0) * LPS>();
    }
    __builtin_amdgcn_s_barrier();          // everyone's share of stage kt landed; slot (kt-1) is free
    // (issuing the DMAs after the fragment reads, or between the two MFMA halves, measured the same)
    {
      const int a = kt + NSTAGE - 1;                 // the stage to issue now, counted from this tile's stage 0
      const int islot = (sbase + a) % NSTAGE;
      if (a < nk) {
        issue_next(islot);
      } else if (have_next) {                        // into the next tile (nk >= NSTAGE - 1: host-checked)
        if constexpr (!CONV) {
          if (a == nk) {                             // the issue side crosses the tile boundary: new address tables
            const int nt = xcd_remap(lin + (int)gridDim.x, ntiles);
            make_wv((nt % p.tiles_n) * BN_OUT);
            make_xe((nt / p.tiles_n) * BM);
            ik0 = kbeg;
          }
          issue_next(islot);
        }
      }
    }
    if constexpr (STAG) {
      if (late && kt > 0) mfma_step(std::integral_constant<int, par ^ 1>{});
    }
    const int slot = (sbase + kt) % NSTAGE;
    const T* xs = ring + slot * STAGE + (wave_m * TM * 16 + frow) * BK;
    const T* ws = ring + slot * STAGE + BM * BK + (wave_n * TN * 16 + frow) * BK;
    // all fragment reads of the K-step go out first; the MFMAs of the first half then run while the
    // second half's reads are still landing (counted lgkmcnt waits, reads return in order)
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int cofs = ((fchunk + 4 * ks) ^ fswz) << 3;
#pragma unroll
      for (int i = 0; i < TN; ++i) wf[par][ks][i] = dd_as_v8<T>(dd_ld16(ws + i * 16 * BK + cofs));
#pragma unroll
      for (int j = 0; j < TM; ++j) xf[par][ks][j] = dd_as_v8<T>(dd_ld16(xs + j * 16 * BK + cofs));
    }
    if (!late) {
      __builtin_amdgcn_sched_barrier(0);
      mfma_step(std::integral_constant<int, par>{});
    }
  };
  const bool persist = !CONV && p.persist != 0;
  for (;;) {
  have_next = persist && lin + (int)gridDim.x < ntiles;
  for (int kt = 0; kt < nk; kt += 2) {
    kstep(kt, std::integral_constant<int, 0>{});
#ifdef DD_DBG_STAMP
    if (kt == 0) DD_STAMP(3);                  // after the first K-step
#endif
    if (kt + 1 < nk) kstep(kt + 1, std::integral_constant<int, 1>{});
  }
  DD_STAMP(4);
  if constexpr (STAG) {
    if (late && nk > 0) {                  // the last K-step's MFMAs of the staggered waves
      if ((nk - 1) & 1) mfma_step(std::integral_constant<int, 1>{});
      else mfma_step(std::integral_constant<int, 0>{});
    }
  }
  if constexpr (!CONV && !GEGLU && WAVES_M == 1 && WAVES_N == 10 && TM == 5 && TN == 2) {
    if (p.ln_out) {                       // whole rows in this workgroup: store out AND LayerNorm(out)
      store_tile_ln<T>(p, acc, block_m0, wave_n, lane, reinterpret_cast<float*>(smem));
      return;
    }
  }
  const bool ln = !CONV && p.ln_colsum;
  store_tile<T, TM, TN, GEGLU>(p, acc, block_m0, block_n0, wave_m, wave_n, lane, p.rows,
                               ln ? s_ln_mean : nullptr, ln ? s_ln_rstd : nullptr, tile, reinterpret_cast<int*>(smem));
  if (!have_next) break;
  lin += (int)gridDim.x;                   // next tile of this workgroup; its first stages are already in flight
  tile = xcd_remap(lin, ntiles);
  block_m0 = (tile / p.tiles_n) * BM;
  block_n0 = (tile % p.tiles_n) * BN_OUT;
  sbase = (sbase + nk) % NSTAGE;
#pragma unroll
  for (int i = 0; i < TN; ++i)
#pragma unroll
    for (int j = 0; j < TM; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
#ifdef DD_DBG_STAMP
  DD_STAMP(5);
  if (threadIdx.x == 0 && p.dbg_stamps) {
    uint64_t* o = p.dbg_stamps + ((size_t)blockIdx.z * gridDim.x + blockIdx.x) * 8;
    for (int i = 0; i < 6; ++i) o[i] = dbg_t[i];
    o[6] = dbg_r0;
    o[7] = __builtin_amdgcn_s_memrealtime();
  }
#endif
}

// =============================================================================================
// Kernel family 2p (round 5): the LDS-DMA ring with an UN-SERIALISED K-step.  Dense GEMMs only.
//
// dd_gemm2_kernel runs every K-step as the serial chain  vmcnt wait -> barrier -> DMA issue -> fragment reads ->
// MFMAs: with one wave per SIMD (every dominant shape: <= 256 workgroups) nothing overlaps that chain and the matrix
// pipe is busy 192 of ~800 cycles (profiles/r04_gemm2_timeline.txt).  Here the fragments of K-step
// c+1 are read while the MFMAs of K-step c run, in two HALVES so that no second register set is needed:
//
//   barrier(c)  |  MFMAs on the ks=0 fragments of c   (the DMAs of stage c+D are issued between them)
//               |  ds_reads of the ks=0 fragments of c+1  ||  MFMAs on the ks=1 fragments of c
//               |  ds_reads of the ks=1 fragments of c+1  ||  (next step's wait + barrier + first MFMAs)
//
// The barrier at the top of step c therefore certifies stage c+1 (not c), and the first MFMA after it never waits
// for LDS.  TIGHT (NSTAGE <= 3): the DMA of step c refills the slot of stage c itself, whose last fragment reads
// (ks=1, issued at the end of step c-1) every wave retires with lgkmcnt(0) before the barrier; NSTAGE >= 4: it refills
// the slot of stage c-1, and the only LDS wait of a step is the compiler's counted one in front of the MFMAs.
// Same arithmetic in the same order per accumulator as dd_gemm2_kernel -> bit-identical results.
// No persistent walk (the epilogue's stores would count in the vmcnt window of the next tile's stages), no LayerNorm
// fold, no conv: those stay with dd_gemm2_kernel.
// =============================================================================================
template <int WM, int WN>
constexpr int gemm3_min_waves() {
  // Four-wave workgroups are compiled for TWO waves per SIMD (<= 256 registers) even where only one ring fits the LDS:
  // with the 512-register budget of one wave per SIMD hipcc moves the accumulators to AGPRs and rotates them through
  // v_accvgpr_read / _write / _mov in every K-step of this loop (measured on the 5-slot 96x64 ring: 11.6 us against 9.5).
  return WM * WN == 4 ? 2 : 1;
}

template <typename T, int WAVES_M, int WAVES_N, int TM, int TN, int NSTAGE, bool GEGLU>
__global__ __launch_bounds__(64 * WAVES_M * WAVES_N, (gemm3_min_waves<WAVES_M, WAVES_N>()))
void dd_gemm3_kernel(const GemmParams p) {
  using V8 = typename dd_vec<T>::v8;
  constexpr int NW = WAVES_M * WAVES_N;
  constexpr int BM = WAVES_M * TM * 16;
  constexpr int BN = WAVES_N * TN * 16;
  constexpr int BN_OUT = GEGLU ? BN / 2 : BN;
  constexpr int XI = BM / 8 / NW;                 // DMA wave-instructions (8 rows x 128 B) per wave
  constexpr int WI = BN / 8 / NW;
  constexpr int LPS = XI + WI;                    // DMA instructions per thread per stage
  constexpr int STAGE = (BM + BN) * BK;           // elements per ring slot
  constexpr bool TIGHT = NSTAGE <= 3;
  constexpr int D = TIGHT ? NSTAGE : NSTAGE - 1;  // the DMA of step c carries stage c + D
  // FAST EPILOGUE (plain T output with bias / alpha / residual / SiLU / accumulate): its operands are loaded through
  // buffer descriptors whose extent is ZERO for an absent operand (the range check returns 0.0f: nothing is predicated,
  // no branch per operand) and the loads are issued right behind the LAST DMA of the K loop, D-1 K-steps before the
  // accumulators are complete — so that the epilogue starts with its operands in registers instead of paying a
  // dependent global round trip (measured before: 1.7 us from the last MFMA to the last store of a 96x64 tile).
  // They are ordinary loads counted in the same in-order vmcnt queue as the DMAs and YOUNGER than every DMA, so the
  // remaining stage waits of the drain simply allow EPI more operations in flight.
  // Register budget: 16-byte operands per lane — tiles of more than 6 (and the 10-wave tiles, 168 registers) keep the
  // general epilogue; the accumulate target is preloaded up to 4.
  constexpr int NG = GEGLU ? 1 : TN / 2;                           // 8-column groups per lane
  constexpr bool FASTEPI = !GEGLU && NW <= 8 && TM * NG <= 6;
  constexpr bool PRE_ACC = FASTEPI && TM * NG <= 4;
  constexpr int EPI = FASTEPI ? NG + TM * NG + (PRE_ACC ? TM * NG : 0) : 0;
  static_assert(BM % (8 * NW) == 0 && BN % (8 * NW) == 0, "tile/waves mismatch");
  static_assert(NW % 2 == 0, "swizzle must not depend on the instruction index");
  static_assert(TN % 2 == 0 && (!GEGLU || TN % 4 == 0), "TN");
  static_assert(NSTAGE >= 3 && NSTAGE <= 8 && D >= 2, "NSTAGE");
  static_assert((D - 1) * LPS <= 63 && (D - 2) * LPS + EPI <= 63, "vmcnt is a 6-bit counter");

#ifdef DD_DBG_STAMP
  uint64_t dbg_t[6];
  const uint64_t dbg_r0 = __builtin_amdgcn_s_memrealtime();
#endif
  DD_STAMP(0);
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  T* ring = reinterpret_cast<T*>(smem);

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wave_m = wave / WAVES_N;
  const int wave_n = wave % WAVES_N;

  const int ntiles = p.tiles_m * p.tiles_n;
  const int tile = xcd_remap(blockIdx.x, ntiles);
  const int tm_i = tile / p.tiles_n;
  const int block_m0 = tm_i * BM;
  const int block_n0 = (tile - tm_i * p.tiles_n) * BN_OUT;

  const int kbeg = blockIdx.z * p.k_per_split;
  const int kend = min(p.k, kbeg + p.k_per_split);
  const int nk = (kend - kbeg + BK - 1) / BK;

  // DMA mapping as in dd_gemm2_kernel: instruction j of this wave fills tile rows (j*NW + wave)*8 .. +7; lane l
  // writes row (l >> 3), chunk position (l & 7); logical chunk = position ^ ((row >> 1) & 7)
  const int lrow = lane >> 3;
  const int lc = (lane & 7) ^ ((((wave & 1) << 2) + (lane >> 4)) & 7);
  const uint32_t lcb = (uint32_t)lc * 16u;

  uint32_t wv[WI];
#pragma unroll
  for (int j = 0; j < WI; ++j) {
    const int R = (j * NW + wave) * 8 + lrow;
    const int wvi = R / (TN * 16);
    const int rho = R % (TN * 16);
    const int tn = rho >> 4, r = rho & 15;
    int n_glob;
    if (GEGLU) {
      constexpr int TH = TN / 2;
      const int t = tn % TH;
      const int col = block_n0 + wvi * (TH * 16) + (r >> 2) * (4 * TH) + t * 4 + (r & 3);
      n_glob = (col < p.n) ? col + (tn >= TH ? p.n : 0) : -1;
    } else {
      const int col = block_n0 + wvi * (TN * 16) + (r >> 2) * (4 * TN) + tn * 4 + (r & 3);
      n_glob = (col < p.n) ? col : -1;
    }
    wv[j] = n_glob >= 0 ? (uint32_t)n_glob * (uint32_t)p.k * 2u + lcb : DD_OOB;
  }
  const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.w), 0, p.w_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.a), 0, p.a_bytes, 0x00020000);

  // Issue cursor (all scalar).  The activation source is `a` for K < k1 and `a2` behind it (the up path's concat); the
  // switch is ONE scalar branch without a DMA inside, taken at most once per workgroup and placed behind the K-step's
  // schedule (a branch around the DMAs would cut the step into separate scheduling regions).
  int ik0 = kbeg;
  int islot = 0;                                   // ring slot the next stage goes to
  int kbase = 0;
  int seam_k = p.a2 ? p.k1 : 0x7fffffff;           // first K offset served by a2
  __amdgpu_buffer_rsrc_t rs_x = rs_a;
  uint32_t xe[XI];
  auto make_xe = [&](const int64_t ld) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < XI; ++j) {
      const int r = block_m0 + (j * NW + wave) * 8 + lrow;
      xe[j] = r < p.rows ? (uint32_t)r * (uint32_t)ld * 2u + lcb : DD_OOB;
    }
  };
  auto seam = [&]() __attribute__((always_inline)) {
    if (ik0 >= seam_k) {
      make_xe(p.lda2);
      rs_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.a2), 0, p.a2_bytes, 0x00020000);
      kbase = p.k1;
      seam_k = 0x7fffffff;
    }
  };
  make_xe(p.lda);
  seam();                                          // a split-K slice that starts behind the seam
  auto issue_next = [&]() __attribute__((always_inline)) {
    T* xs = ring + islot * STAGE;
    T* ws = xs + BM * BK;
#pragma unroll
    for (int j = 0; j < WI; ++j) bdma16(rs_w, wv[j], (uint32_t)ik0 * 2u, ws + (j * NW + wave) * 8 * BK);
#pragma unroll
    for (int j = 0; j < XI; ++j) bdma16(rs_x, xe[j], (uint32_t)(ik0 - kbase) * 2u, xs + (j * NW + wave) * 8 * BK);
    ik0 += BK;
    islot = islot + 1 == NSTAGE ? 0 : islot + 1;
  };

  f32x4 acc[TN][TM];
#pragma unroll
  for (int i = 0; i < TN; ++i)
#pragma unroll
    for (int j = 0; j < TM; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int frow = lane & 15;
  const int fswz = (lane >> 1) & 7;
  const int fchunk = lane >> 4;
  const int cofs0 = ((fchunk + 0) ^ fswz) << 3, cofs1 = ((fchunk + 4) ^ fswz) << 3;
  const T* xbase = ring + (wave_m * TM * 16 + frow) * BK;
  const T* wbase = ring + BM * BK + (wave_n * TN * 16 + frow) * BK;

  DD_STAMP(1);
  // prologue: stages 0 and 1 first; the remaining D-2 go out behind the first fragment reads (issuing all D up front
  // kept the wave at the address path for 0.7 us before it even looked at stage 0)
#pragma unroll
  for (int s0 = 0; s0 < 2; ++s0)
    if (s0 < nk) { issue_next(); seam(); }

  V8 wf[2][TN], xf[2][TM];
  int rslot = 0;                                   // ring slot of the stage whose fragments are read next
  auto read_half = [&](auto ks_c) __attribute__((always_inline)) {
    constexpr int ks = decltype(ks_c)::value;
    const int cofs = ks ? cofs1 : cofs0;
    const T* ws = wbase + rslot * STAGE + cofs;
    const T* xs = xbase + rslot * STAGE + cofs;
#pragma unroll
    for (int i = 0; i < TN; ++i) wf[ks][i] = dd_as_v8<T>(dd_ld16(ws + i * 16 * BK));
#pragma unroll
    for (int j = 0; j < TM; ++j) xf[ks][j] = dd_as_v8<T>(dd_ld16(xs + j * 16 * BK));
  };
  auto mfma_half = [&](auto ks_c) __attribute__((always_inline)) {
    constexpr int ks = decltype(ks_c)::value;
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
      for (int j = 0; j < TM; ++j) acc[i][j] = dd_mfma16(wf[ks][i], xf[ks][j], acc[i][j]);
  };
  // all but the `ahead` youngest stages (and the EXTRA operations issued behind them) have landed
  auto wait_stages = [&](const int ahead, auto extra_c) __attribute__((always_inline)) {
    constexpr int X = decltype(extra_c)::value;
    if (ahead <= 0) wait_vmcnt<X>();
    else if (ahead == 1 || D <= 2) wait_vmcnt<(D > 1 ? 1 : 0) * LPS + X>();
    else if (ahead == 2 || D <= 3) wait_vmcnt<(D > 2 ? 2 : 0) * LPS + X>();
    else if (ahead == 3 || D <= 4) wait_vmcnt<(D > 3 ? 3 : 0) * LPS + X>();
    else if (ahead == 4 || D <= 5) wait_vmcnt<(D > 4 ? 4 : 0) * LPS + X>();
    else if (ahead == 5 || D <= 6) wait_vmcnt<(D > 5 ? 5 : 0) * LPS + X>();
    else wait_vmcnt<(D > 6 ? 6 : 0) * LPS + X>();
  };
  using K0 = std::integral_constant<int, 0>;
  using K1 = std::integral_constant<int, 1>;
  constexpr int NMF = TN * TM, NRD = TN + TM;

  // stage 0 -> registers
  wait_stages(min(2, nk) - 1, K0{});
  __builtin_amdgcn_s_barrier();
  read_half(K0{});
  read_half(K1{});
  rslot = 1;
#pragma unroll
  for (int s0 = 2; s0 < D; ++s0)
    if (s0 < nk) { issue_next(); seam(); }
  DD_STAMP(2);

  // One K-step in EXPLICIT issue order (pinned with sched_barrier(0) after every unit: sched_group_barrier patterns were
  // only loosely followed): the DMAs one by one behind the first half's MFMAs, the ks=0 reads behind the second half's,
  // the ks=1 reads last.  MEASURED AND REMOVED: a staggered form in which the upper half of the waves issued its DMAs
  // beside the second half's MFMAs (so that the four waves do not queue at the CU's address path together) — 8.7 us
  // either way on 1092x1280x1280, 0.25-0.26 us per K-step (profiles/r05_experiments.txt).
  auto steady = [&](auto issue_c) __attribute__((always_inline)) {
    constexpr bool ISSUE = decltype(issue_c)::value;
    const T* wp0 = wbase + rslot * STAGE + cofs0;
    const T* xp0 = xbase + rslot * STAGE + cofs0;
    const T* wp1 = wbase + rslot * STAGE + cofs1;
    const T* xp1 = xbase + rslot * STAGE + cofs1;
    T* xs = ring + islot * STAGE;
    T* ws = xs + BM * BK;
    const uint32_t so_w = (uint32_t)ik0 * 2u, so_x = (uint32_t)(ik0 - kbase) * 2u;
    auto dma = [&](const int u) __attribute__((always_inline)) {
      if (u < WI) bdma16(rs_w, wv[u], so_w, ws + (u * NW + wave) * 8 * BK);
      else bdma16(rs_x, xe[u - WI], so_x, xs + ((u - WI) * NW + wave) * 8 * BK);
    };
    auto rd = [&](const int ks, const int u) __attribute__((always_inline)) {
      if (u < TN) wf[ks][u] = dd_as_v8<T>(dd_ld16((ks ? wp1 : wp0) + u * 16 * BK));
      else xf[ks][u - TN] = dd_as_v8<T>(dd_ld16((ks ? xp1 : xp0) + (u - TN) * 16 * BK));
    };
    auto mf = [&](const int ks, const int u) __attribute__((always_inline)) {
      const int i = u / TM, j = u % TM;
      acc[i][j] = dd_mfma16(wf[ks][i], xf[ks][j], acc[i][j]);
    };
    constexpr int NDM = ISSUE ? LPS : 0;
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int u = 0; u < (NMF > NDM ? NMF : NDM); ++u) {
      if (u < NMF) mf(0, u);
      if (u < NDM) dma(u);
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int u = 0; u < (NMF > NRD ? NMF : NRD); ++u) {
      if (u < NRD) rd(0, u);
      if (u < NMF) mf(1, u);
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int u = 0; u < NRD; ++u) rd(1, u);
    __builtin_amdgcn_s_setprio(0);
    if constexpr (ISSUE) {
      ik0 += BK;
      islot = islot + 1 == NSTAGE ? 0 : islot + 1;
    }
    rslot = rslot + 1 == NSTAGE ? 0 : rslot + 1;
  };
  int c = 0;
  for (; c + D < nk; ++c) {                        // steady state: stage c+1 certified, stage c+D issued
    wait_vmcnt<(D - 2) * LPS>();
    if (TIGHT) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    steady(std::true_type{});
    seam();
#ifdef DD_DBG_STAMP
    if (c == 0) DD_STAMP(3);
#endif
  }

  // ---- epilogue operands: issued behind the last DMA -------------------------------------------------------------
  const int q4 = lane >> 4, c16 = lane & 15;
  const int erow0 = block_m0 + wave_m * (TM * 16) + c16;
  const int ecol0 = block_n0 + wave_n * (TN * 16) + q4 * (4 * TN);
  const bool fast = FASTEPI && !p.partial && !p.hm_d && !p.out_f32 && !p.stat_out && !p.rowvec && (PRE_ACC || !p.accumulate) &&
                    p.out_bytes != 0;
  u32x4 pb[NG], pr[TM][NG], pa[PRE_ACC ? TM : 1][NG];
  uint32_t off_o[TM][NG];
  if constexpr (FASTEPI) {
    const uint32_t e_bias = fast && p.bias ? (uint32_t)p.n * 2u : 0u;
    const uint32_t e_res = fast && p.res ? p.res_bytes : 0u;
    const uint32_t e_acc = fast && p.accumulate ? p.out_bytes : 0u;
    const __amdgpu_buffer_rsrc_t rs_b = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.bias), 0, e_bias, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_r = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.res), 0, e_res, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_o = __builtin_amdgcn_make_buffer_rsrc(p.out, 0, e_acc, 0x00020000);
#pragma unroll
    for (int g8 = 0; g8 < NG; ++g8) {
      const int col = ecol0 + g8 * 8;
      pb[g8] = __builtin_amdgcn_raw_buffer_load_b128(rs_b, col < p.n ? (uint32_t)col * 2u : DD_OOB, 0, 0);
    }
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) {
      const int row = erow0 + tm * 16;
#pragma unroll
      for (int g8 = 0; g8 < NG; ++g8) {
        const int col = ecol0 + g8 * 8;
        const bool ok = row < p.rows && col < p.n;
        off_o[tm][g8] = ok ? ((uint32_t)row * (uint32_t)p.ldc + (uint32_t)col) * 2u : DD_OOB;
        pr[tm][g8] = __builtin_amdgcn_raw_buffer_load_b128(rs_r, ok ? ((uint32_t)row * (uint32_t)p.ldres + (uint32_t)col) * 2u : DD_OOB, 0, 0);
        if constexpr (PRE_ACC) pa[tm][g8] = __builtin_amdgcn_raw_buffer_load_b128(rs_o, off_o[tm][g8], 0, 0);
      }
    }
  }
  using EX = std::integral_constant<int, EPI>;

  for (; c + 1 < nk; ++c) {                        // drain: nothing left to issue
    wait_stages(nk - 2 - c, EX{});
    if (TIGHT) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    steady(std::false_type{});
  }
  if (nk > 0) {                                    // last K-step: its fragments are in registers
    __builtin_amdgcn_s_setprio(1);
    mfma_half(K0{});
    mfma_half(K1{});
    __builtin_amdgcn_s_setprio(0);
  }
  DD_STAMP(4);
  bool done = false;
  if constexpr (FASTEPI) {
    if (fast) {
      const __amdgpu_buffer_rsrc_t rs_st = __builtin_amdgcn_make_buffer_rsrc(p.out, 0, p.out_bytes, 0x00020000);
      const bool silu = p.act == DD_EPI_SILU;
#pragma unroll
      for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int g8 = 0; g8 < NG; ++g8) {
          float v[8], b[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = acc[g8 * 2 + (e >> 2)][tm][e & 3];
          dd_unpack8<T>(pb[g8], b);
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = (v[e] + b[e]) * p.alpha;
          dd_unpack8<T>(pr[tm][g8], b);
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] += b[e];
          if (silu) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = dd_silu_f(v[e]);
          }
          if constexpr (PRE_ACC) {
            dd_unpack8<T>(pa[tm][g8], b);
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] += b[e];
          }
          __builtin_amdgcn_raw_buffer_store_b128(dd_pack8<T>(v), rs_st, off_o[tm][g8], 0, 0);
        }
      done = true;
    }
  }
  if (!done) {
    if (p.tile_counters) __syncthreads();          // in-launch split-K: the flag word of store_tile aliases the ring
    store_tile<T, TM, TN, GEGLU>(p, acc, block_m0, block_n0, wave_m, wave_n, lane, p.rows, nullptr, nullptr, tile,
                                 reinterpret_cast<int*>(smem));
  }
#ifdef DD_DBG_STAMP
  DD_STAMP(5);
  if (threadIdx.x == 0 && p.dbg_stamps) {
    uint64_t* o = p.dbg_stamps + ((size_t)blockIdx.z * gridDim.x + blockIdx.x) * 8;
    for (int i = 0; i < 6; ++i) o[i] = dbg_t[i];
    o[6] = dbg_r0;
    o[7] = __builtin_amdgcn_s_memrealtime();
  }
#endif
}

// =============================================================================================
// Kernel family 3: direct 3x3 convolution for SMALL images (14x25 and deeper: H*W <= 384).
// The implicit-GEMM kernels stage the activation tile once per TAP (9 x per 64 input channels); at
// the deep levels (336 / 1092 rows x 1280 channels x 29-59 MB of weights) that makes the kernel
// bytes-in-flight bound.  Here a workgroup owns G whole instances (G*H*W <= BM rows): per 64-channel
// chunk the RAW pixels of its instances are DMA'd into LDS once, and the 9 taps are 9 different
// per-lane LDS row gathers (a padding tap points at a row the range check filled with zeros).
// Staged bytes drop ~5x; the weight matrix is streamed once per row tile through a 3-slot ring.
// Requirements (host-checked): stride 1, no resize, Cin % 64 == 0.  Split-K is over channel chunks.
// =============================================================================================
// BAND = true: images LARGER than the tile (the 28x50 level).  A workgroup owns a band of p.band_rows consecutive output
// pixels (whole image rows) of one instance; its slab holds those pixels plus a halo of W + 1 pixels on either side, so
// the activation is still staged once per 64-channel chunk (the implicit-GEMM kernels stage it once per tap).  LDS rows
// 0..15 are the zero rows, slab pixel s sits in row 16 + s; halo pixels outside the image are out-of-range DMAs = zeros.
template <typename T, int WAVES_M, int WAVES_N, int TM, int TN, int NSW, int GRP = 1, bool BAND = false>
__global__ __launch_bounds__(64 * WAVES_M * WAVES_N)
void dd_conv3s_kernel(const GemmParams p) {
  // GRP = 3: the weight ring is two GROUPS of three taps; a workgroup synchronises (DMA wait + barrier)
  // once per group instead of once per tap — 72 MFMAs per wave between barriers instead of 24 — and the
  // next group's three weight tiles are in flight under them.
  static_assert(GRP == 1 || (GRP == 3 && NSW == 6), "grouped taps: 2 groups of 3 slots");
  using V8 = typename dd_vec<T>::v8;
  constexpr int NW = WAVES_M * WAVES_N;
  constexpr int BM = WAVES_M * TM * 16;
  constexpr int BN = WAVES_N * TN * 16;
  constexpr int AROWS = BAND ? BM + 88 : BM + 64;   // rows >= BM are never valid pixels -> always zeros (BAND: see above)
  constexpr int XA = (AROWS / 8 + NW - 1) / NW;     // activation DMA pieces per wave per chunk
  constexpr int WI = BN / 8 / NW;              // weight DMA pieces per wave per (chunk, tap) step
  // NSW weight ring slots: the weights are cold (HBM, 2-3 us) while a (chunk, tap) step lasts
  // ~0.3 us, so the ring is as deep as LDS allows
  static_assert((BAND ? AROWS % 8 == 0 : AROWS % (8 * NW) == 0) && BN % (8 * NW) == 0 && NW % 2 == 0, "tile/waves mismatch");
  static_assert(TN % 2 == 0, "TN");
  static_assert(NSW >= 3 && NSW <= 10 && (NSW - 2) * WI + XA <= 63, "ring depth / vmcnt");

#ifdef DD_DBG_STAMP
  uint64_t dbg_t[6];
  const uint64_t dbg_r0 = __builtin_amdgcn_s_memrealtime();
#endif
  DD_STAMP(0);
  if (p.pf_blocks && (int)blockIdx.x >= (int)gridDim.x - p.pf_blocks) {        // spare workgroup: prefetch only
    if (blockIdx.z == 0) dd_prefetch_block<64 * NW>(p, (int)blockIdx.x - ((int)gridDim.x - p.pf_blocks));
    return;
  }
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  T* abuf = reinterpret_cast<T*>(smem);                 // [2][AROWS][64]
  T* wring = abuf + 2 * AROWS * BK;                     // [NSW][BN][64]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wave_m = wave / WAVES_N;
  const int wave_n = wave % WAVES_N;

  // row tiles of ONE weight slice are neighbours in the remapped order -> same XCD, same L2: at these levels the
  // weight matrix (29-59 MB) is the big operand and each slice is wanted by every row tile (activations: 1-3 MB)
  const int tile = xcd_remap(blockIdx.x, p.tiles_m * p.tiles_n);
  const int tile_n = p.upsample ? tile % p.tiles_n : tile / p.tiles_m;      // (upsample is unused by this kernel:
  const int tile_m = p.upsample ? tile / p.tiles_n : tile % p.tiles_m;      //  A/B switch DD_CONV3S_ROWMAJOR=1)
  const int hw = p.hout * p.wout;
  const int m_inst = dd_fdiv(p.rows, p.inv_hw);
  int g0_, ng_, vrows_, row0_, band0_ = 0;
  if constexpr (BAND) {
    g0_ = dd_fdiv(tile_m, p.inv_bands);                 // instance
    band0_ = (tile_m - g0_ * p.bands) * p.band_rows;    // first pixel of the band inside the instance
    ng_ = 1;
    vrows_ = min(p.band_rows, hw - band0_);
    row0_ = g0_ * hw + band0_;
  } else {
    g0_ = tile_m * p.g_per_tile;
    ng_ = min(p.g_per_tile, m_inst - g0_);
    vrows_ = ng_ * hw;
    row0_ = g0_ * hw;
  }
  const int g0 = g0_, ng = ng_;
  const int vrows = vrows_;                             // valid rows of this tile
  const int row0 = row0_;                               // first global output row
  const int band0 = band0_;
  (void)ng; (void)g0;
  const int block_n0 = tile_n * BN;

  const bool stagger_off = p.no_stagger != 0;
  const int nchunks = p.cin / BK;
  const int c_beg = blockIdx.z * p.chunks_per_split;
  const int nc = min(nchunks, c_beg + p.chunks_per_split) - c_beg;
  const int nsteps = nc * 9;

  const int lrow = lane >> 3;
  const int lc = (lane & 7) ^ ((((wave & 1) << 2) + (lane >> 4)) & 7);
  const uint32_t lcb = (uint32_t)lc * 16u;

  // The activation slab is swizzled by ROW & 7 (the weight ring by (row >> 1) & 7 like the GEMM family): the tap
  // gathers read 16 consecutive slab rows starting at ANY row (r + dy*W + dx), and ds_read_b128's lane groups
  // ({0-3, 12-15} at chunk c, {4-11} at chunk c+1) are conflict-free for every such window only when the 8
  // rows of a group get 8 different chunk positions whatever the window's parity — (row >> 1) & 7 does that
  // for even shifts only (2-way conflicts on every odd tap: 34-39 % of the LDS cycles measured).
  const uint32_t lcb_a = (uint32_t)((lane & 7) ^ (lane >> 3)) * 16u;
  // ---- DMA tables -----------------------------------------------------------------------
  uint32_t av[XA];                                      // activation rows of the tile (raw pixels)
  int adst[XA];                                         // BAND: LDS row of the piece (surplus pieces rewrite the zero rows)
#pragma unroll
  for (int j = 0; j < XA; ++j) {
    if constexpr (BAND) {
      const int pc = j * NW + wave;                     // 8-row piece of the slab buffer
      const bool real = pc < AROWS / 8;
      const int L = (real ? pc : 0) * 8 + lrow;         // LDS row
      const int sidx = L - 16;                          // slab pixel index
      const int pix = band0 - (p.wout + 1) + sidx;      // pixel inside the instance
      const bool ok = real && sidx >= 0 && sidx < vrows + 2 * (p.wout + 1) && pix >= 0 && pix < hw;
      av[j] = ok ? (uint32_t)(g0 * hw + pix) * (uint32_t)p.cin * 2u + lcb_a : DD_OOB;
      adst[j] = (real ? pc : 0) * 8;
    } else {
      const int r = (j * NW + wave) * 8 + lrow;
      av[j] = r < vrows ? (uint32_t)(row0 + r) * (uint32_t)p.cin * 2u + lcb_a : DD_OOB;
      adst[j] = (j * NW + wave) * 8;
    }
  }
  uint32_t wv[WI];                                      // weight rows, permuted like dd_gemm2_kernel
#pragma unroll
  for (int j = 0; j < WI; ++j) {
    const int R = (j * NW + wave) * 8 + lrow;
    const int wvi = R / (TN * 16);
    const int rho = R % (TN * 16);
    const int tn = rho >> 4, r = rho & 15;
    const int col = block_n0 + wvi * (TN * 16) + (r >> 2) * (4 * TN) + tn * 4 + (r & 3);
    wv[j] = col < p.n ? (uint32_t)col * (uint32_t)p.k * 2u + lcb : DD_OOB;
  }
  const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.w), 0, p.w_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.a), 0, p.a_bytes, 0x00020000);

  auto issue_a = [&](int c) __attribute__((always_inline)) {           // chunk c (local index) -> abuf[c & 1]
    T* dst = abuf + (c & 1) * AROWS * BK;
    const uint32_t so = (uint32_t)((c_beg + c) * BK) * 2u;
#pragma unroll
    for (int j = 0; j < XA; ++j) bdma16(rs_a, av[j], so, dst + adst[j] * BK);
  };
  auto issue_w = [&](int c, int t, int slot) __attribute__((always_inline)) {
    T* dst = wring + slot * BN * BK;
    const uint32_t so = (uint32_t)(t * p.cin + (c_beg + c) * BK) * 2u;
#pragma unroll
    for (int j = 0; j < WI; ++j) bdma16(rs_w, wv[j], so, dst + (j * NW + wave) * 8 * BK);
  };

  f32x4 acc[TN][TM];
#pragma unroll
  for (int i = 0; i < TN; ++i)
#pragma unroll
    for (int j = 0; j < TM; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int frow = lane & 15;
  const int fswz = (lane >> 1) & 7;
  const int fchunk = lane >> 4;

  DD_STAMP(1);
  if (nc > 0) {
    issue_a(0);
#pragma unroll
    for (int s0 = 0; s0 < (GRP == 1 ? NSW - 1 : NSW); ++s0)
      if (s0 < nsteps) issue_w(s0 / 9, s0 % 9, s0);
  }
  DD_STAMP(2);
  // (built AFTER the prologue DMAs are in flight: ~60 entries x ~20 VALU instructions took 3.7 us of a 36 us
  //  kernel in front of the first load; now they run under the 2-3 us the cold weights need to arrive)
  // ---- per-lane tap tables: LDS row of the pixel each tap reads (BM = the zero row), 2 x 16 bit
  uint32_t tab[TM][5];
  // Branch-free (bit selects on 0 / ~0 masks): written with `if`s the compiler emitted 120 exec-mask regions for
  // the 60 entries and the build took 6 200 cycles of a 69 000-cycle kernel (tools/conv3s_stamps.py).
#pragma unroll
  for (int tm = 0; tm < TM; ++tm) {
    const int r = wave_m * (TM * 16) + tm * 16 + (lane & 15);
    const bool rv = r < vrows;
    const int rr = rv ? r : 0;
    const int g = BAND ? 0 : dd_fdiv(rr, p.inv_hw);
    const int rem = BAND ? band0 + rr : rr - g * hw;    // pixel inside its instance
    const int y = dd_fdiv(rem, p.inv_wout);
    const int x = rem - y * p.wout;
    const uint32_t mrv = 0u - (uint32_t)rv;
    const uint32_t my[3] = {mrv & (0u - (uint32_t)(y >= 1)), mrv, mrv & (0u - (uint32_t)(y + 1 < p.hout))};
    const uint32_t mx[3] = {0u - (uint32_t)(x >= 1), ~0u, 0u - (uint32_t)(x + 1 < p.wout)};
#pragma unroll
    for (int t2 = 0; t2 < 5; ++t2) {
      uint32_t packed = 0;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int t = t2 * 2 + h;
        // A valid tap reads slab row r + dy*W + dx (= g*hw + iy*W + ix).  A padding tap reads one of the 16 zero rows
        // BM .. BM+15, the one with the residue mod 16 the real pixel would have had: the 16 lanes of an MFMA row
        // block keep DISTINCT rows mod 16, which is what keeps ds_read_b128 conflict-free under the row & 7
        // swizzle (one shared zero row cost 34-39 % of the LDS cycles in bank conflicts at the 4x7 / 7x13 levels,
        // where a third of all taps are padding)
        // BAND: slab pixel s sits in LDS row 16 + s and output row r is slab pixel r + W + 1; zero rows are 0..15
        const uint32_t lin = (uint32_t)(r + (BAND ? 16 + p.wout + 1 : 0) + (t < 9 ? (t / 3 - 1) * p.wout + (t % 3 - 1) : 0));
        const uint32_t pad = (BAND ? 0u : (uint32_t)BM) | (lin & 15u);   // BM is a multiple of 16
        const uint32_t ok = t < 9 ? (my[t < 9 ? t / 3 : 0] & mx[t < 9 ? t % 3 : 0]) : 0u;
        uint32_t ra = (lin & ok) | (pad & ~ok);
        // the entry is the fragment's LDS address in 16-byte units: row * 8 + swizzled chunk of k-step 0
        // (k-step 1 is the same address with bit 2 of the chunk flipped); AROWS * 8 + 7 < 2^16
        ra = (ra << 3) | ((uint32_t)(lane >> 4) ^ (ra & 7u));
        packed |= ra << (16 * h);
      }
      tab[tm][t2] = packed;
    }
  }
  int wslot = 0;                                        // ring slot of step s (scalar)
  // Gathered activation fragments are double-buffered across steps: while the MFMAs of tap t run,
  // the fragments of tap t+1 (same resident chunk; at t == 8 the next chunk, landed since step NSW-1)
  // are already being read.  The buffer index is a compile-time parity, so the chunk loop is
  // unrolled by two (9 taps per chunk is odd).
  V8 xf[2][2][TM];
  V8 wf[2][2][TN];                                      // weight fragments, by step parity (see STAGGER below)
  const bool late = NW == 8 && GRP == 1 && wave >= 4 && !stagger_off;
  auto gather = [&](const T* ab, auto tap_c, auto par_c) __attribute__((always_inline)) {
    constexpr int t = decltype(tap_c)::value;
    constexpr int par = decltype(par_c)::value;
#pragma unroll
    for (int j = 0; j < TM; ++j) {
      uint32_t ra = (tab[j][t >> 1] >> (16 * (t & 1))) & 0xFFFFu;
      asm volatile("" : "+v"(ra));       // keep the 54 gather addresses out of registers: recompute per step
      xf[par][0][j] = dd_as_v8<T>(dd_ld16(ab + (ra << 3)));
      xf[par][1][j] = dd_as_v8<T>(dd_ld16(ab + ((ra ^ 4u) << 3)));
    }
  };
  auto step = [&](const int c, auto tap_c, auto par_c) __attribute__((always_inline)) {
    constexpr int t = decltype(tap_c)::value;
    constexpr int par = decltype(par_c)::value;
    const bool more_c = c + 1 < nc;
    const int s = c * 9 + t;
    if constexpr (GRP == 1) {
    // W(s) (and with it, in issue order, A(c)) must have landed.  Younger loads that may stay in
    // flight: W(s+1..s+NSW-2), and A(c+1) when it was issued after W(s) (1 <= t <= NSW-2).  The
    // last NSW-2 steps simply drain.
    if (s + NSW - 2 < nsteps) {
      if (t >= 1 && t <= NSW - 2 && more_c) wait_vmcnt<(NSW - 2) * WI + XA>();
      else wait_vmcnt<(NSW - 2) * WI>();
    } else {
      wait_vmcnt<0>();
    }
    __builtin_amdgcn_s_barrier();
    if (t == 0 && more_c) issue_a(c + 1);
    if (s + NSW - 1 < nsteps) {
      int slot = wslot + NSW - 1;
      if (slot >= NSW) slot -= NSW;
      constexpr int ta = (t + NSW - 1) % 9, ca = (t + NSW - 1) / 9;
      issue_w(c + ca, ta, slot);
    }
    } else if constexpr (t % GRP == 0) {
      // group start: this group's taps (issued one group ago; the first two groups in the prologue) must
      // have landed; only at the very first group may the second group still be in flight
      if (s == 0 && GRP < nsteps) wait_vmcnt<GRP * WI>();
      else wait_vmcnt<0>();
      __builtin_amdgcn_s_barrier();              // everyone is done with the previous group's slots
      if (t == 0 && more_c) issue_a(c + 1);
      if (s >= GRP && s + GRP < nsteps) {        // next group into the slots just freed
        int slot = wslot + GRP;
        if (slot >= NSW) slot -= NSW;
        constexpr int t1 = (t + GRP) % 9, c1 = (t + GRP) / 9;
#pragma unroll
        for (int u = 0; u < GRP; ++u) issue_w(c + c1, t1 + u, slot + u);
      }
    }
    const T* ws = wring + wslot * BN * BK + (wave_n * TN * 16 + frow) * BK;
    if (++wslot == NSW) wslot = 0;
    // STAGGER (8-wave tiles): the two waves of a SIMD run the same program behind one barrier per step, so
    // without help they read LDS together and then contend for the matrix pipe together.  Waves 4-7 execute
    // the MFMAs of step s-1 (operands already in registers) BEFORE the reads of step s, i.e. half a step out
    // of phase with waves 0-3: one wave's MFMAs run beside the other's LDS traffic.  Same arithmetic, same
    // order per accumulator -> bit-identical results (MI355X_MICROARCH.md, "Two waves per SIMD", item 9).
    if (late && s > 0) {
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int i = 0; i < TN; ++i)
#pragma unroll
          for (int j = 0; j < TM; ++j) acc[i][j] = dd_mfma16(wf[par ^ 1][ks][i], xf[par ^ 1][ks][j], acc[i][j]);
      __builtin_amdgcn_s_setprio(0);
    }
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int cofs = ((fchunk + 4 * ks) ^ fswz) << 3;
#pragma unroll
      for (int i = 0; i < TN; ++i) wf[par][ks][i] = dd_as_v8<T>(dd_ld16(ws + i * 16 * BK + cofs));
    }
    if (s == 0) gather(abuf, std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});   // first step only
    // next step's activation fragments (other parity)
    if (t < 8) {
      gather(abuf + (c & 1) * AROWS * BK, std::integral_constant<int, (t + 1) % 9>{}, std::integral_constant<int, par ^ 1>{});
    } else if (more_c) {
      gather(abuf + ((c + 1) & 1) * AROWS * BK, std::integral_constant<int, 0>{}, std::integral_constant<int, par ^ 1>{});
    }
    if (!late) {
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int i = 0; i < TN; ++i)
#pragma unroll
          for (int j = 0; j < TM; ++j) acc[i][j] = dd_mfma16(wf[par][ks][i], xf[par][ks][j], acc[i][j]);
      __builtin_amdgcn_s_setprio(0);
    }
  };
  auto chunk = [&](const int c, auto par0) __attribute__((always_inline)) {
    constexpr int p0 = decltype(par0)::value;
    step(c, std::integral_constant<int, 0>{}, std::integral_constant<int, p0>{});
    step(c, std::integral_constant<int, 1>{}, std::integral_constant<int, p0 ^ 1>{});
    step(c, std::integral_constant<int, 2>{}, std::integral_constant<int, p0>{});
    step(c, std::integral_constant<int, 3>{}, std::integral_constant<int, p0 ^ 1>{});
    step(c, std::integral_constant<int, 4>{}, std::integral_constant<int, p0>{});
    step(c, std::integral_constant<int, 5>{}, std::integral_constant<int, p0 ^ 1>{});
    step(c, std::integral_constant<int, 6>{}, std::integral_constant<int, p0>{});
    step(c, std::integral_constant<int, 7>{}, std::integral_constant<int, p0 ^ 1>{});
    step(c, std::integral_constant<int, 8>{}, std::integral_constant<int, p0>{});
  };
  for (int c = 0; c < nc; c += 2) {
    chunk(c, std::integral_constant<int, 0>{});            // even chunk: tap t uses parity t & 1
#ifdef DD_DBG_STAMP
    if (c == 0) DD_STAMP(3);                               // after the first 9 steps
#endif
    if (c + 1 < nc) chunk(c + 1, std::integral_constant<int, 1>{});   // odd chunk: parity (t + 1) & 1
  }
  DD_STAMP(4);
  if (late && nsteps > 0) {                               // staggered waves: the last step's MFMAs are still due
    auto drain = [&](auto par_c) __attribute__((always_inline)) {
      constexpr int par = decltype(par_c)::value;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int i = 0; i < TN; ++i)
#pragma unroll
          for (int j = 0; j < TM; ++j) acc[i][j] = dd_mfma16(wf[par][ks][i], xf[par][ks][j], acc[i][j]);
    };
    if ((nsteps - 1) & 1) drain(std::integral_constant<int, 1>{});
    else drain(std::integral_constant<int, 0>{});
  }
  // rows past the tile's instances are padding
  store_tile<T, TM, TN, false>(p, acc, row0, block_n0, wave_m, wave_n, lane, min(p.rows, row0 + vrows),
                               nullptr, nullptr, tile, reinterpret_cast<int*>(smem));
#ifdef DD_DBG_STAMP
  DD_STAMP(5);
  if (threadIdx.x == 0 && p.dbg_stamps) {
    uint64_t* o = p.dbg_stamps + ((size_t)blockIdx.z * gridDim.x + blockIdx.x) * 8;
    for (int i = 0; i < 6; ++i) o[i] = dbg_t[i];
    o[6] = dbg_r0;
    o[7] = __builtin_amdgcn_s_memrealtime();
  }
#endif
}

// =============================================================================================
// Kernel family 4: ROW-PANEL GEMM for the transformer projections (dense, K = C in {320, 640, 1280}).
//
// The C x C / C x 3C projections of a transformer block (to_q / QKV, to_out, proj_in; 16800 .. 336 rows)
// are far too small for the tiled families: a 64x64 tile re-reads both operands through L2 once per tile
// and a 10-20 step K loop with a barrier per step leaves the kernel latency-bound (~11 us where the bytes
// take 3-6).  Here ONE workgroup per CU owns a column slice of BN = 4 waves x TN x 16 outputs for a whole
// group of rows:
//   * its weight slice lives in REGISTERS for the kernel's lifetime — every wave keeps the MFMA fragments
//     of its TN x 16 weight rows over the full K (TN * K/32 x 4 VGPRs: 200-320 of the 512 a one-wave-per-
//     SIMD kernel has), loaded once by buffer loads straight from global memory;
//   * the rows stream through LDS as PANELS of BM = TM x 16 rows x full K (LDS-DMA ring, the next panel
//     lands while the current one is multiplied); a panel needs no K loop synchronisation at all: one
//     counted vmcnt wait + one barrier, then K/32 MFMA steps whose only LDS traffic is the A fragments;
//   * optional LayerNorm PROLOGUE (`ln_gamma`): the panel holds the un-normalised rows and is normalised in
//     place (two-pass fp32 statistics, result rounded to T — the arithmetic of dd_layernorm_sub_kernel)
//     before the MFMAs, which removes the LayerNorm launch and its HBM round trip in front of every
//     Q / QKV projection (norm1 / norm2 / norm4 of blocks.py:150-222);
//   * 2-D decomposition p row groups x q column slices with p * q <= 256 workgroups: per-CU bytes are
//     |W| / q + |A| / p instead of (rows / 64) x (N / 64) tile pairs.
// Epilogue: bias, alpha, residual, accumulate, head-major planes (+ scale) like the other families.
// Requirements (host-checked): no a2 / conv / GEGLU / split-K / rowvec, N % BN == 0.
// =============================================================================================
template <typename T, int VW>
__device__ __forceinline__ void rp_store_vec(const GemmParams& p, int64_t row, int col, float (&v)[VW]) {
  using VT = typename std::conditional<VW == 8, u32x4, u32x2>::type;
  T* dst;
  if (p.hm_d) {
    const int plane = col / p.hm_d;
    if (plane < p.hm_planes) {
#pragma unroll
      for (int e = 0; e < VW; ++e) v[e] *= p.hm_scale;
    }
    dst = reinterpret_cast<T*>(p.out) + ((int64_t)plane * p.rows + row) * p.hm_d + (col - plane * p.hm_d);
  } else {
    dst = reinterpret_cast<T*>(p.out) + row * p.ldc + col;
  }
  T tmp[VW];
#pragma unroll
  for (int e = 0; e < VW; ++e) tmp[e] = (T)v[e];
  VT pk;
  __builtin_memcpy(&pk, tmp, sizeof(VT));
  *reinterpret_cast<VT*>(dst) = pk;
}

template <typename T, int VW>
__device__ __forceinline__ void rp_load_vec(const T* src, float (&f)[VW]) {
  using VT = typename std::conditional<VW == 8, u32x4, u32x2>::type;
  const VT raw = *reinterpret_cast<const VT*>(src);
  T tmp[VW];
  __builtin_memcpy(tmp, &raw, sizeof(VT));
#pragma unroll
  for (int e = 0; e < VW; ++e) f[e] = (float)tmp[e];
}

// fp8 (OCP e4m3fn) -> T for 8 consecutive weights: one-time conversion when the fragments are loaded
template <typename T>
__device__ __forceinline__ typename dd_vec<T>::v8 rp_dequant8(u32x2 raw) {
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  typename dd_vec<T>::v8 r;
  const f32x2 a = __builtin_amdgcn_cvt_pk_f32_fp8((int)raw[0], false), b = __builtin_amdgcn_cvt_pk_f32_fp8((int)raw[0], true);
  const f32x2 c = __builtin_amdgcn_cvt_pk_f32_fp8((int)raw[1], false), d = __builtin_amdgcn_cvt_pk_f32_fp8((int)raw[1], true);
  r[0] = (T)a[0]; r[1] = (T)a[1]; r[2] = (T)b[0]; r[3] = (T)b[1];
  r[4] = (T)c[0]; r[5] = (T)c[1]; r[6] = (T)d[0]; r[7] = (T)d[1];
  return r;
}

template <typename T, int KS, int TN, int TM, int NBUF, bool LN, bool W8 = false>
__global__ __launch_bounds__(256)
void dd_gemm_rp_kernel(const GemmParams p) {
  using V8 = typename dd_vec<T>::v8;
  constexpr int K = KS * 32;
  constexpr int NSUB = K / 64;                     // 64-column sub-tiles of a panel
  constexpr int BM = TM * 16;
  constexpr int BNW = TN * 16;                     // output columns per wave
  constexpr int BN = 4 * BNW;
  constexpr int PANEL = BM * K;                    // elements per panel buffer
  constexpr int PIECES = (BM / 8) * NSUB;          // 1-KB DMA pieces (8 rows x 128 B) per panel
  constexpr int PPW = (PIECES + 3) / 4;            // per wave; surplus pieces land in a dump slot
  constexpr int VW = (TN % 2 == 0) ? 8 : 4;        // channels per epilogue vector (16 B needs an even TN)
  constexpr int NG = 4 * TN / VW;                  // epilogue vectors per lane and row
  constexpr int NSTORE = TM * NG;                  // store instructions per wave and (full) panel
  static_assert(KS % 2 == 0 && NBUF >= 2, "K must be a multiple of 64");
  static_assert(NSTORE <= 63 && (NBUF - 1) * PPW + NSTORE <= 63, "vmcnt is a 6-bit counter");

  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  T* ring = reinterpret_cast<T*>(smem);                         // [NBUF][NSUB][BM][64], chunk-swizzled
  T* dump = ring + NBUF * PANEL;                                // [4 waves][512]: landing zone of surplus pieces
  T* lnv = dump + 4 * 512;                                      // [2][K]: gamma | beta

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

  // ---- work decomposition: (row group, column slice); slices of one row group sit on one XCD ----
  const int nwg = gridDim.x;
  const int unit = xcd_remap(blockIdx.x, nwg);
  const int qn = p.tiles_n;                                     // column slices
  const int pi = unit / qn, qi = unit - pi * qn;
  const int panels_total = (p.rows + BM - 1) / BM;
  const int ppg = p.k_per_split;                                // panels per row group (host: ceil)
  const int panel0 = pi * ppg;
  const int npan = min(ppg, panels_total - panel0);             // >= 1 by construction
  const int col0 = qi * BN + wave * BNW;

  // ---- weight fragments -> registers (issued first: the longest fetch of the kernel) ------------
  const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.w), 0, p.w_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.a), 0, p.a_bytes, 0x00020000);
  const int fr = lane & 15, fq = lane >> 4;
  V8 wreg[TN][KS];
  {
    // weight row permutation of the other families: lane group q ends up with 4*TN CONSECUTIVE channels
    const int loc0 = (fr >> 2) * (4 * TN) + (fr & 3);
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) {
      const int n = col0 + loc0 + tn * 4;
      if constexpr (W8) {        // fp8 weights [n][K] bytes: 8 B per lane and k-step, dequantised to T once, here
        const uint32_t vo = n < p.n ? (uint32_t)n * (uint32_t)K + (uint32_t)fq * 8u : DD_OOB;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
          wreg[tn][ks] = rp_dequant8<T>(__builtin_amdgcn_raw_buffer_load_b64(rs_w, vo + ks * 32, 0, 0));
      } else {
      const uint32_t vo = n < p.n ? (uint32_t)n * (uint32_t)(K * 2) + (uint32_t)fq * 16u : DD_OOB;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks)
        wreg[tn][ks] = dd_as_v8<T>(__builtin_amdgcn_raw_buffer_load_b128(rs_w, vo + ks * 64, 0, 0));
      }
    }
  }

  // ---- panel DMA tables (constant per lane; a panel moves only the scalar row offset) -----------
  const int lrow8 = lane >> 3;
  uint32_t ptab[PPW];                               // byte offset inside a panel's source rows
  int prow[PPW];                                    // panel row this lane fetches (-1: surplus piece)
  int pdst[PPW];                                    // LDS element offset of the piece inside a buffer
#pragma unroll
  for (int j = 0; j < PPW; ++j) {
    const int pc = j * 4 + wave;
    if (pc < PIECES) {
      const int sub = pc / (BM / 8), rb = pc - sub * (BM / 8);
      const int row = rb * 8 + lrow8;
      const int lc = (lane & 7) ^ ((row >> 1) & 7);
      prow[j] = row;
      ptab[j] = (uint32_t)row * (uint32_t)p.lda * 2u + (uint32_t)sub * 128u + (uint32_t)lc * 16u;
      pdst[j] = (sub * BM + rb * 8) * 64;
    } else {
      prow[j] = -1; ptab[j] = DD_OOB; pdst[j] = -1;
    }
  }
  auto issue_panel = [&](int pl, int buf) __attribute__((always_inline)) {
    const int r0 = (panel0 + pl) * BM;
    const uint32_t so = (uint32_t)r0 * (uint32_t)p.lda * 2u;
#pragma unroll
    for (int j = 0; j < PPW; ++j) {
      const bool ok = prow[j] >= 0 && r0 + prow[j] < p.rows;
      T* dst = pdst[j] >= 0 ? ring + buf * PANEL + pdst[j] : dump + wave * 512;
      bdma16(rs_a, ok ? ptab[j] : DD_OOB, so, dst);
    }
  };
#pragma unroll
  for (int b = 0; b < NBUF - 1; ++b)
    if (b < npan) issue_panel(b, b);

  if (LN) {                                         // gamma | beta -> LDS (read per chunk in the prologue)
    for (int i = tid; i < 2 * K / 8; i += 256) {
      const T* src = i < K / 8 ? reinterpret_cast<const T*>(p.ln_gamma) + i * 8
                               : reinterpret_cast<const T*>(p.ln_beta) + (i - K / 8) * 8;
      dd_st16(lnv + i * 8, dd_ld16(src));
    }
  }
  // epilogue constants
  const int ecol0 = col0 + fq * (4 * TN);
  float ebias[NG][VW], escale[NG][VW];
  if constexpr (W8) {
#pragma unroll
    for (int g = 0; g < NG; ++g)
#pragma unroll
      for (int e = 0; e < VW; ++e) escale[g][e] = p.w_scale[ecol0 + g * VW + e];
  }
  if (p.bias) {
#pragma unroll
    for (int g = 0; g < NG; ++g) rp_load_vec<T, VW>(reinterpret_cast<const T*>(p.bias) + ecol0 + g * VW, ebias[g]);
  }
  const int fswz = (lane >> 1) & 7;

  for (int pl = 0; pl < npan; ++pl) {
    const int buf = pl % NBUF;
    // panel pl must have landed.  Younger operations of this wave: the DMA pieces of up to NBUF-2 later
    // panels and the NSTORE stores of the previous panel's epilogue (full panels only: a partial panel is
    // the last one of the whole problem) — counted, so nothing drains.
    if (pl == 0) {
      wait_vmcnt<0>();                              // first panel + the weight fragments
    } else {
      const int ahead = min(npan - 1 - pl, NBUF - 2);
      if (ahead <= 0) wait_vmcnt<NSTORE>();
      else if (ahead == 1 || NBUF <= 3) wait_vmcnt<NSTORE + (NBUF > 2 ? 1 : 0) * PPW>();
      else wait_vmcnt<NSTORE + (NBUF > 3 ? 2 : 0) * PPW>();
    }
    __builtin_amdgcn_s_barrier();                   // everyone's pieces landed; slot (pl-1) % NBUF is free
    if (pl + NBUF - 1 < npan) issue_panel(pl + NBUF - 1, (pl + NBUF - 1) % NBUF);
    T* ab = ring + buf * PANEL;

    if (LN) {
      // LayerNorm in place: BM/4 rows per wave, LPR lanes per row, 16-B chunks round-robin over the lanes
      constexpr int RPW = BM / 4, LPR = 64 / RPW, NCH = K / 8;
      const int row = wave * RPW + lane / LPR;
      const int sub = lane % LPR;
      const int rsw = (row >> 1) & 7;
      auto chunk_ptr = [&](int ci) __attribute__((always_inline)) {
        return ab + ((ci >> 3) * BM + row) * 64 + (((ci & 7) ^ rsw) << 3);
      };
      float s = 0.f;
      for (int ci = sub; ci < NCH; ci += LPR) {
        float f[8];
        dd_unpack8<T>(dd_ld16(chunk_ptr(ci)), f);
#pragma unroll
        for (int e = 0; e < 8; ++e) s += f[e];
      }
#pragma unroll
      for (int o = LPR / 2; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
      const float mean = s * (1.0f / (float)K);
      float ss = 0.f;
      for (int ci = sub; ci < NCH; ci += LPR) {
        float f[8];
        dd_unpack8<T>(dd_ld16(chunk_ptr(ci)), f);
#pragma unroll
        for (int e = 0; e < 8; ++e) { const float d = f[e] - mean; ss += d * d; }
      }
#pragma unroll
      for (int o = LPR / 2; o > 0; o >>= 1) ss += __shfl_xor(ss, o, 64);
      const float rstd = rsqrtf(ss * (1.0f / (float)K) + p.ln_eps);
      for (int ci = sub; ci < NCH; ci += LPR) {
        float f[8], ga[8], be[8];
        dd_unpack8<T>(dd_ld16(chunk_ptr(ci)), f);
        dd_unpack8<T>(dd_ld16(lnv + ci * 8), ga);
        dd_unpack8<T>(dd_ld16(lnv + K + ci * 8), be);
#pragma unroll
        for (int e = 0; e < 8; ++e) f[e] = (f[e] - mean) * rstd * ga[e] + be[e];
        dd_st16(chunk_ptr(ci), dd_pack8<T>(f));
      }
      __syncthreads();
    }

    // ---- MFMAs over the whole K: A fragments from LDS, weight fragments from registers ----------
    f32x4 acc[TN][TM];
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
      for (int j = 0; j < TM; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const T* xs = ab + fr * 64;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const int cofs = ((fq + 4 * (ks & 1)) ^ fswz) << 3;
      V8 xf[TM];
#pragma unroll
      for (int j = 0; j < TM; ++j) xf[j] = dd_as_v8<T>(dd_ld16(xs + ((ks >> 1) * BM + j * 16) * 64 + cofs));
#pragma unroll
      for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = 0; j < TM; ++j) acc[i][j] = dd_mfma16(wreg[i][ks], xf[j], acc[i][j]);
    }

    // ---- epilogue: every global read before the first store (out may alias res) ------------------
    const int r0 = (panel0 + pl) * BM;
    float eres[TM][NG][VW], eacc[TM][NG][VW];
    if (p.res) {
#pragma unroll
      for (int j = 0; j < TM; ++j) {
        const int64_t rowc = min(r0 + j * 16 + fr, p.rows - 1);
#pragma unroll
        for (int g = 0; g < NG; ++g)
          rp_load_vec<T, VW>(reinterpret_cast<const T*>(p.res) + rowc * p.ldres + ecol0 + g * VW, eres[j][g]);
      }
    }
    if (p.accumulate) {
#pragma unroll
      for (int j = 0; j < TM; ++j) {
        const int64_t rowc = min(r0 + j * 16 + fr, p.rows - 1);
#pragma unroll
        for (int g = 0; g < NG; ++g)
          rp_load_vec<T, VW>(reinterpret_cast<const T*>(p.out) + rowc * p.ldc + ecol0 + g * VW, eacc[j][g]);
      }
    }
#pragma unroll
    for (int j = 0; j < TM; ++j) {
      const int row = r0 + j * 16 + fr;
      if (row < p.rows) {
#pragma unroll
        for (int g = 0; g < NG; ++g) {
          float v[VW];
#pragma unroll
          for (int e = 0; e < VW; ++e) {
            const int c = g * VW + e;                    // channel inside the lane's 4*TN run
            v[e] = acc[c >> 2][j][c & 3];
          }
          if constexpr (W8) {
#pragma unroll
            for (int e = 0; e < VW; ++e) v[e] *= escale[g][e];
          }
          if (p.bias) {
#pragma unroll
            for (int e = 0; e < VW; ++e) v[e] += ebias[g][e];
          }
#pragma unroll
          for (int e = 0; e < VW; ++e) v[e] *= p.alpha;
          if (p.res) {
#pragma unroll
            for (int e = 0; e < VW; ++e) v[e] += eres[j][g][e];
          }
          if (p.accumulate) {
#pragma unroll
            for (int e = 0; e < VW; ++e) v[e] += eacc[j][g][e];
          }
          rp_store_vec<T, VW>(p, row, ecol0 + g * VW, v);
        }
      }
    }
  }
}

// split-K: sum the fp32 partial slabs and run the fused epilogue.
template <typename T>
__global__ __launch_bounds__(256)
void dd_splitk_reduce_kernel(const GemmParams p, int nsplit) {
  const int64_t groups_per_row = p.n / 8;
  const int64_t total = (int64_t)p.rows * groups_per_row;
  for (int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; g < total;
       g += (int64_t)gridDim.x * blockDim.x) {
    const int row = (int)(g / groups_per_row);
    const int col = (int)(g - (int64_t)row * groups_per_row) * 8;
    float v[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int z = 0; z < nsplit; ++z) {
      const float* src = p.partial + ((int64_t)z * p.rows + row) * p.n + col;
      const f32x4 a = *reinterpret_cast<const f32x4*>(src);
      const f32x4 b = *reinterpret_cast<const f32x4*>(src + 4);
      v[0] += a[0]; v[1] += a[1]; v[2] += a[2]; v[3] += a[3];
      v[4] += b[0]; v[5] += b[1]; v[6] += b[2]; v[7] += b[3];
    }
    epilogue_store8<T>(p, row, col, v);
  }
}

// ---- host side --------------------------------------------------------------------------
// stages == 0: register-staged family (dd_gemm_kernel); stages >= 2: LDS-DMA ring (dd_gemm2_kernel)
struct TileCfg { int id, wm, wn, tm, tn, stages; const char* name; };
constexpr TileCfg kTiles[] = {
    {1, 2, 2, 4, 4, 0, "128x128"},
    {2, 2, 2, 4, 2, 0, "128x64"},
    {3, 2, 2, 2, 4, 0, "64x128"},
    {4, 2, 2, 2, 2, 0, "64x64"},
    {5, 4, 2, 4, 4, 0, "256x128"},
    {11, 2, 2, 4, 4, 2, "128x128/dma2"},
    {12, 2, 2, 4, 4, 3, "128x128/dma3"},
    {13, 2, 2, 4, 2, 3, "128x64/dma3"},
    {14, 2, 2, 2, 4, 3, "64x128/dma3"},
    {15, 2, 2, 2, 2, 3, "64x64/dma3"},
    {16, 4, 2, 4, 4, 2, "256x128/dma2"},
    {17, 2, 2, 4, 2, 2, "128x64/dma2"},
    {18, 2, 2, 2, 2, 2, "64x64/dma2"},
    {19, 2, 2, 2, 2, 4, "64x64/dma4"},
    {20, 4, 2, 4, 4, 3, "256x128/dma3"},
    // deep rings for cold-weight streaming (few rows, long K): most of a block's K range in flight
    {21, 2, 2, 2, 2, 6, "64x64/dma6"},
    {22, 2, 2, 2, 2, 8, "64x64/dma8"},
    {23, 2, 2, 4, 2, 4, "128x64/dma4"},
    {24, 2, 2, 2, 4, 4, "64x128/dma4"},
    {25, 2, 2, 4, 4, 4, "128x128/dma4"},
    // tall tiles for the 4x7 / 7x13 levels (336 / 1092 rows x 1280 x up to 23040): all (or a third of)
    // the rows in one tile so the 29-59 MB weight matrix is streamed once, not once per 128 rows
    {26, 4, 2, 6, 2, 2, "384x64/dma2"},
    // 160-wide tiles (10 waves = 2 x 5): every channel count of this network (320, 640, 960, 1280, 1920, 2560) is
    // a multiple of 160, so no column of the tile multiplies padding (a 128-wide tile wastes 1/6 of its MFMAs at
    // N = 320 and 16800 rows / 160 = 105 row tiles x 2 = 210 workgroups fill the chip in ONE generation)
    {27, 2, 5, 5, 2, 2, "160x160/dma2"},
    {28, 2, 5, 5, 2, 3, "160x160/dma3"},
    {29, 2, 5, 5, 4, 2, "160x320/dma2"},            // GEGLU: 160 gated outputs per tile (h | g rows interleaved)
    // 80 WHOLE rows of a 320-wide output per workgroup (1 x 10 waves): the only tile whose epilogue can emit
    // LayerNorm(out) as a second tensor (dd_gemm_desc.ln_out); 16800 rows -> 210 workgroups, one generation
    {40, 1, 10, 5, 2, 2, "80x320/dma2"},
    // 1092 x 1280 outputs over 256 CUs = 5460 per CU: 96x64 -> 12 x 20 = 240 workgroups (one generation, nearly every
    // CU busy) staging 410 KB each where the 64x128 tile stages 491 KB on 180 CUs.  Challenged against the tracked table
    // (bench.py --challenge-tiles 52, cold weights, 3 % to win): takes 28 of the dense shapes per dtype, ~1 us each
    // (1092x1280x1280 15.4 -> 14.4, 336x1280x1280 14.8 -> 13.8 and no split-K, 4200x640x1920 27.1 -> 21.7); 96x128
    // tiles won nothing (profiles/r03_tile_challenge.txt)
    {52, 2, 2, 3, 2, 3, "96x64/dma3"},
    // 32-row tiles for the few-row GEMMs (time / box / text embeddings: 12-240 rows; 336 x 1280 -> 11 x 20 workgroups):
    // 1-2 us each in the same challenge; 96x64 with 2 / 4 slots, 96x128 and 192x64 tiles won nothing and were removed
    {59, 2, 2, 1, 2, 3, "32x64/dma3"},
    {60, 2, 2, 1, 2, 6, "32x64/dma6"},
    // 192 rows: 1092 rows -> 6 row tiles (180 workgroups at N = 3840 where 256x128 has 150): the per-CU staging rate,
    // not the tile's arithmetic intensity, bounds a launch that leaves CUs without a workgroup (1092x3840x1280:
    // 26.5 -> 23.2 us cold, 1092x1280x6400: 41.4 -> 37.6)
    {44, 4, 2, 3, 4, 3, "192x128/dma3"},
    {46, 4, 2, 3, 4, 2, "192x128/dma2"},
    // 256x256 (round 3): the tiled family is bound by L2 -> LDS staging, and staged bytes per flop go with
    // (BM + BN) / (BM * BN): 0.0078 B/flop against 0.0117 for 256x128.  8 waves of 128 x 64 (32 accumulator blocks per
    // wave: one wave per SIMD pair, 2 stages of 64 KB).  Candidates for the wide GEGLU projections and the big convs.
    {50, 2, 4, 8, 4, 2, "256x256/dma2"},
    // stages >= 100: pipelined LDS-DMA family (dd_gemm3_kernel, round 5; dense only), ring depth = stages - 100
    {72, 2, 2, 3, 2, 103, "96x64/p3"},             // 60 KB: two workgroups per CU
    {73, 2, 2, 3, 2, 105, "96x64/p5"},             // deeper rings: one workgroup per CU, cold weights 3-4 K-steps ahead
    {74, 2, 2, 3, 2, 106, "96x64/p6"},
    {75, 4, 2, 3, 4, 103, "192x128/p3"},
    {76, 2, 2, 1, 2, 104, "32x64/p4"},
    {77, 2, 2, 1, 2, 106, "32x64/p6"},
    {78, 2, 5, 5, 2, 103, "160x160/p3"},
    {79, 2, 2, 1, 2, 108, "32x64/p8"},
    // stages < 0: direct small-image conv (dd_conv3s_kernel); conv with stride 1 / no resize /
    // Cin % 64 == 0 / H*W <= rows of the tile only
    {31, 4, 2, 6, 2, -1, "conv3s 384x64"},
    {39, 4, 2, 6, 2, -3, "conv3s band 384x64"},   // stages == -3: BAND form (images larger than the tile: 28x50 level)
    {33, 2, 2, 6, 2, -1, "conv3s 192x64/w8"},
    {34, 2, 2, 6, 2, -1, "conv3s 192x64/w4"},
    {35, 2, 2, 4, 2, -1, "conv3s 128x64/w3"},     // 72 KB of LDS: two workgroups per CU
    {36, 2, 2, 4, 4, -1, "conv3s 128x128/w3"},
    {37, 2, 2, 6, 2, -1, "conv3s 192x64/g3"},     // taps in groups of three: one barrier per 72 MFMAs
    {38, 2, 2, 4, 2, -1, "conv3s 128x64/g3"},
    // stages == -2: row-panel family (dd_gemm_rp_kernel; dense, K in {320, 640, 1280}): tm = 16-row MFMA blocks
    // per panel; wn / tn follow from K (5 x 16 columns per wave at K = 320, 2 x 16 otherwise)
    {41, 1, 4, 1, 0, -2, "rowpanel 16"},
    {42, 1, 4, 2, 0, -2, "rowpanel 32"},
};
constexpr int kNumTiles = sizeof(kTiles) / sizeof(kTiles[0]);

inline int tile_bm(const TileCfg& t) { return t.wm * t.tm * 16; }
inline int tile_bn(const TileCfg& t) { return t.wn * t.tn * 16; }

constexpr int kNumCU = 256;

struct Plan { int tile_idx; int split; int tiles_m, tiles_n; int k_per_split; int g_per_tile, chunks_per_split; bool unsupported; bool persist_ok; int band_rows, bands; };

int ceil_div(int a, int b) { return (a + b - 1) / b; }

// The LDS-DMA family wants a K structure in whole 64-element steps (no step straddles a conv tap or
// the a/a2 seam) and buffers below 2^31 bytes (32-bit lane offsets, DD_OOB out of range for all).
bool dma_ok(const dd_gemm_desc* d) {
  const int64_t lim = (int64_t)1 << 30;              // elements
  const int64_t nw = (d->epilogue == DD_EPI_GEGLU ? 2 : 1) * (int64_t)d->n;
  bool ok = (d->k % BK) == 0 && nw * d->k < lim;
  if (d->conv) {
    ok = ok && (d->cin % BK) == 0 && (int64_t)d->rows / (d->hout * d->wout) * d->hin * d->win * d->cin < lim;
  } else {
    ok = ok && (int64_t)d->rows * d->lda < lim;
    if (d->a2) ok = ok && (d->k1 % BK) == 0 && (int64_t)d->rows * d->lda2 < lim;
  }
  return ok;
}

// row-panel family: 16-column MFMA blocks per wave for a given K (0 = K not covered)
inline int rp_tn(int k) { return k == 320 ? 5 : (k == 640 || k == 1280) ? 2 : 0; }

Plan make_plan(const dd_gemm_desc* d) {
  const bool geglu = d->epilogue == DD_EPI_GEGLU;
  Plan pl{};
  int ti = -1;
  if (d->tile > 0) {
    for (int i = 0; i < kNumTiles; ++i) if (kTiles[i].id == d->tile) ti = i;
  }
  if (ti < 0) {
    // heuristic: biggest tile that still yields >= ~1.5 waves of blocks; else smaller tiles.
    const int order[] = {0, 1, 2, 3};
    ti = 3;
    for (int oi = 0; oi < 4; ++oi) {
      const TileCfg& t = kTiles[order[oi]];
      if (geglu && t.tn % 4 != 0) continue;
      const int bn_out = geglu ? tile_bn(t) / 2 : tile_bn(t);
      const long blocks = (long)ceil_div(d->rows, tile_bm(t)) * ceil_div(d->n, bn_out);
      if (blocks >= (long)kNumCU * 3 / 2) { ti = order[oi]; break; }
      if (oi == 3) ti = geglu ? 2 : 3;
    }
    if (geglu && kTiles[ti].tn % 4 != 0) ti = 2;
  }
  if (d->ln_colsum) {                                // LayerNorm fold lives in the LDS-DMA family only
    if (d->tile <= 0) {                              // heuristic picked a register-staged tile: take its twin
      const int twin[4] = {11, 17, 14, 18};
      for (int i = 0; i < kNumTiles; ++i) if (kTiles[i].id == twin[ti < 4 ? ti : 3]) { ti = i; break; }
    }
    if (kTiles[ti].stages <= 0 || kTiles[ti].stages >= 100 || !dma_ok(d)) { pl.unsupported = true; return pl; }
  }
  if (ti >= 0 && kTiles[ti].stages >= 100 && (d->conv || d->ln_out)) { pl.unsupported = true; return pl; }   // dense only
  if (d->ln_out) {                                   // LayerNorm-emitting epilogue: the 80x320 tile, one column tile
    if (d->tile > 0 && d->tile != 40) { pl.unsupported = true; return pl; }
    for (int i = 0; i < kNumTiles; ++i) if (kTiles[i].id == 40) ti = i;
    if (d->n != 320 || !dma_ok(d)) { pl.unsupported = true; return pl; }
  }
  if ((d->ln_gamma || d->w_scale) && (ti < 0 || kTiles[ti].stages != -2)) {   // LayerNorm prologue / fp8 weights: row-panel family only
    if (d->tile > 0) { pl.unsupported = true; return pl; }
    for (int i = 0; i < kNumTiles; ++i) if (kTiles[i].id == 41) ti = i;
  }
  if (kTiles[ti].stages == -2) {                     // row-panel GEMM
    const TileCfg& t = kTiles[ti];
    const int tn = rp_tn(d->k);
    const int bn = 4 * tn * 16, bm = t.tm * 16;
    const bool ok = tn > 0 && !d->conv && !d->a2 && !geglu && d->epilogue == DD_EPI_NONE && !d->rowvec &&
                    !d->ln_colsum && !d->ln_stats_out && !d->out_f32 && d->split_k <= 1 && (d->n % bn) == 0 &&
                    (t.tm == 1 || d->k <= 640) && (!d->ln_gamma || d->ln_beta) &&
                    (int64_t)d->rows * d->lda < ((int64_t)1 << 30) && (int64_t)d->n * d->k < ((int64_t)1 << 30);
    if (!ok) { pl.unsupported = true; return pl; }
    const int panels = ceil_div(d->rows, bm);
    const int q = d->n / bn;
    int pg = kNumCU / q;
    if (pg < 1) pg = 1;
    if (pg > panels) pg = panels;
    const int ppg = ceil_div(panels, pg);
    pl.tile_idx = ti;
    pl.tiles_m = ceil_div(panels, ppg);
    pl.tiles_n = q;
    pl.split = 1;
    pl.k_per_split = ppg;                            // panels per row group
    return pl;
  }
  if (kTiles[ti].stages == -3) {                     // direct conv on row BANDS with a halo (images larger than the tile)
    const TileCfg& t = kTiles[ti];
    const int hw = d->hout * d->wout, W = d->wout;
    const int arows = tile_bm(t) + 88;               // = the kernel's AROWS
    const int R = W > 0 ? std::min(tile_bm(t), arows - 16 - 2 * (W + 1)) / W : 0;      // whole image rows per band
    const bool ok = d->conv && !geglu && d->stride == 1 && d->hv == d->hin && d->wv == d->win &&
                    d->hout == d->hin && d->wout == d->win && (d->cin % BK) == 0 && hw > tile_bm(t) && R >= 1 &&
                    d->rows % hw == 0 && dma_ok(d);
    if (!ok) { pl.unsupported = true; return pl; }
    const int m_inst = d->rows / hw;
    const int nchunks = d->cin / BK;
    int split = d->split_k > 0 ? d->split_k : 1;
    if (split > nchunks) split = nchunks;
    const int cps = ceil_div(nchunks, split);
    pl.tile_idx = ti;
    pl.band_rows = R * W;
    pl.bands = ceil_div(d->hout, R);
    pl.g_per_tile = 1;
    pl.tiles_m = m_inst * pl.bands;
    pl.tiles_n = ceil_div(d->n, tile_bn(t));
    pl.chunks_per_split = cps;
    pl.split = ceil_div(nchunks, cps);
    pl.k_per_split = cps * BK;
    return pl;
  }
  if (kTiles[ti].stages < 0) {                       // direct small-image conv
    const TileCfg& t = kTiles[ti];
    const int hw = d->hout * d->wout;
    const bool ok = d->conv && !geglu && d->stride == 1 && d->hv == d->hin && d->wv == d->win &&
                    d->hout == d->hin && d->wout == d->win && (d->cin % BK) == 0 && hw > 0 &&
                    hw <= tile_bm(t) && d->rows % hw == 0 && dma_ok(d) && tile_bm(t) < 65535;
    if (!ok) { pl.unsupported = true; return pl; }
    const int m_inst = d->rows / hw;
    int g = tile_bm(t) / hw;
    if (g > m_inst) g = m_inst;
    const int nchunks = d->cin / BK;
    int split = d->split_k > 0 ? d->split_k : 1;
    if (split > nchunks) split = nchunks;
    const int cps = ceil_div(nchunks, split);
    pl.tile_idx = ti;
    pl.g_per_tile = g;
    pl.tiles_m = ceil_div(m_inst, g);
    pl.tiles_n = ceil_div(d->n, tile_bn(t));
    pl.chunks_per_split = cps;
    pl.split = ceil_div(nchunks, cps);
    pl.k_per_split = cps * BK;
    return pl;
  }
  if (kTiles[ti].stages && !dma_ok(d)) {             // same tile shape, register-staged family
    int alt = geglu ? 0 : 3;                          // no twin: 128x128 (GEGLU-capable) / 64x64
    for (int i = 0; i < kNumTiles; ++i)
      if (!kTiles[i].stages && kTiles[i].wm == kTiles[ti].wm && kTiles[i].wn == kTiles[ti].wn &&
          kTiles[i].tm == kTiles[ti].tm && kTiles[i].tn == kTiles[ti].tn) { alt = i; break; }
    ti = alt;
  }
  const TileCfg& t = kTiles[ti];
  const int bn_out = geglu ? tile_bn(t) / 2 : tile_bn(t);
  pl.tile_idx = ti;
  pl.tiles_m = ceil_div(d->rows, tile_bm(t));
  pl.tiles_n = ceil_div(d->n, bn_out);
  int split = d->split_k;
  const int nkt = ceil_div(d->k, BK);
  if (split <= 0) {
    split = 1;
    const long blocks = (long)pl.tiles_m * pl.tiles_n;
    if (!geglu && blocks < kNumCU && nkt >= 16) {
      split = (int)((2L * kNumCU + blocks - 1) / blocks);
      if (split > nkt / 4) split = nkt / 4;
      if (split > 32) split = 32;
      if (split < 1) split = 1;
    }
  }
  if (geglu || d->ln_colsum || d->ln_stats_out || d->out_headmajor_d || d->ln_out) split = 1;
  if (split > nkt) split = nkt;
  int kts = ceil_div(nkt, split);
  split = ceil_div(nkt, kts);
  pl.split = split;
  pl.k_per_split = kts * BK;
  {
    // persistent walk with cross-tile prefetch (dd_gemm2_kernel): dense, one K range per tile, no epilogue that uses
    // LDS or per-tile LDS state, and a K loop at least as long as the ring.  DD_PERSIST=0 is the A/B switch.
    static const bool off = getenv("DD_PERSIST") && atoi(getenv("DD_PERSIST")) == 0;
    pl.persist_ok = !off && !d->conv && t.stages >= 2 && t.stages < 100 && split == 1 && !d->ln_colsum && !d->ln_out && nkt >= t.stages;
  }
  return pl;
}

template <typename T, int WM, int WN, int TM, int TN, bool CONV, bool GEGLU>
int launch_cfg(const GemmParams& p, const Plan& pl, hipStream_t s) {
  constexpr int BM = WM * TM * 16, BN = WN * TN * 16;
  constexpr size_t smem = (size_t)2 * (BM + BN) * BK * sizeof(T);
  auto kern = dd_gemm_kernel<T, WM, WN, TM, TN, CONV, GEGLU>;
  static std::atomic<uint64_t> attr_done{0};
  dd_ensure_dyn_lds(reinterpret_cast<const void*>(kern), smem, attr_done);
  dim3 grid(pl.tiles_m * pl.tiles_n, 1, pl.split);
  hipLaunchKernelGGL(kern, grid, dim3(64 * WM * WN), smem, s, p);
  return dd_check_launch();
}

// Spare workgroups for the weight prefetch: only when every tile of the launch is resident at once and at least 8 slots
// stay empty; ~64 KB in flight per 256-thread workgroup, one workgroup per 128 KB of weights, at most 96.
static int dd_prefetch_blocks(const GemmParams& p, const void* kern, int threads, size_t smem, int blocks,
                              std::atomic<int>& resident) {
  if (!p.pf_ptr || !p.pf_bytes || p.persist) return 0;
  int per_cu = resident.load(std::memory_order_relaxed);
  if (per_cu == 0) {
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, threads, smem) != hipSuccess || per_cu < 1) per_cu = 1;
    resident.store(per_cu, std::memory_order_relaxed);
  }
  const int spare = kNumCU * per_cu - blocks;
  if (spare < 8) return 0;
  const int want = (int)((p.pf_bytes + (128u << 10) - 1) / (128u << 10)) * 256 / threads;
  static const int cap = getenv("DD_PF_MAX_BLOCKS") ? atoi(getenv("DD_PF_MAX_BLOCKS")) : 96;
  return std::max(std::min(8, cap), std::min(std::min(spare, cap), want));
}

template <typename T, int WM, int WN, int TM, int TN, int NSTAGE, bool CONV, bool GEGLU>
int launch_cfg2(const GemmParams& p, const Plan& pl, hipStream_t s) {
  constexpr int BM = WM * TM * 16, BN = WN * TN * 16;
  constexpr size_t smem = (size_t)NSTAGE * (BM + BN) * BK * sizeof(T);
  static_assert(smem <= 160 * 1024, "LDS");
  auto kern = dd_gemm2_kernel<T, WM, WN, TM, TN, NSTAGE, CONV, GEGLU>;
  static std::atomic<uint64_t> attr_done{0};
  dd_ensure_dyn_lds(reinterpret_cast<const void*>(kern), smem, attr_done);
  dim3 grid(pl.tiles_m * pl.tiles_n, 1, pl.split);
  static std::atomic<int> pf_resident{0};
  const int pf = dd_prefetch_blocks(p, reinterpret_cast<const void*>(kern), 64 * WM * WN, smem, (int)grid.x * pl.split, pf_resident);
  if (pf) {                                // spare workgroups at the end of the grid read the next launch's weights
    GemmParams q = p;
    q.pf_blocks = pf;
    grid.x += pf;
    hipLaunchKernelGGL(kern, grid, dim3(64 * WM * WN), smem, s, q);
    return dd_check_launch();
  }
  if constexpr (!CONV) {
    if (pl.persist_ok) {                 // more tiles than resident workgroups: walk them with the ring running ahead
      static std::atomic<int> resident{0};
      int per_cu = resident.load(std::memory_order_relaxed);
      if (per_cu == 0) {
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, 64 * WM * WN, smem) != hipSuccess || per_cu < 1)
          per_cu = 1;
        resident.store(per_cu, std::memory_order_relaxed);
      }
      const int g = kNumCU * per_cu;
      if ((int)grid.x > g) {
        GemmParams q = p;
        q.persist = 1;
        grid.x = g;
        hipLaunchKernelGGL(kern, grid, dim3(64 * WM * WN), smem, s, q);
        return dd_check_launch();
      }
    }
  }
  hipLaunchKernelGGL(kern, grid, dim3(64 * WM * WN), smem, s, p);
  return dd_check_launch();
}

template <typename T, int WM, int WN, int TM, int TN, int NSTAGE, bool GEGLU>
int launch_cfg3(const GemmParams& p, const Plan& pl, hipStream_t s) {
  constexpr int BM = WM * TM * 16, BN = WN * TN * 16;
  constexpr size_t smem = (size_t)NSTAGE * (BM + BN) * BK * sizeof(T);
  static_assert(smem <= 160 * 1024, "LDS");
  auto kern = dd_gemm3_kernel<T, WM, WN, TM, TN, NSTAGE, GEGLU>;
  static std::atomic<uint64_t> attr_done{0};
  dd_ensure_dyn_lds(reinterpret_cast<const void*>(kern), smem, attr_done);
  hipLaunchKernelGGL(kern, dim3(pl.tiles_m * pl.tiles_n, 1, pl.split), dim3(64 * WM * WN), smem, s, p);
  return dd_check_launch();
}

template <typename T, int WM, int WN, int TM, int TN, int NSW, int GRP = 1, bool BAND = false>
int launch_conv3s(const GemmParams& p, const Plan& pl, hipStream_t s) {
  constexpr int BM = WM * TM * 16, BN = WN * TN * 16;
  constexpr size_t smem = (size_t)(2 * (BM + (BAND ? 88 : 64)) + NSW * BN) * BK * sizeof(T);
  static_assert(smem <= 160 * 1024 - 64, "LDS");
  auto kern = dd_conv3s_kernel<T, WM, WN, TM, TN, NSW, GRP, BAND>;
  static std::atomic<uint64_t> attr_done{0};
  dd_ensure_dyn_lds(reinterpret_cast<const void*>(kern), smem, attr_done);
  dim3 grid(pl.tiles_m * pl.tiles_n, 1, pl.split);
  static std::atomic<int> pf_resident{0};
  const int pf = dd_prefetch_blocks(p, reinterpret_cast<const void*>(kern), 64 * WM * WN, smem, (int)grid.x * pl.split, pf_resident);
  if (pf) {
    GemmParams q = p;
    q.pf_blocks = pf;
    grid.x += pf;
    hipLaunchKernelGGL(kern, grid, dim3(64 * WM * WN), smem, s, q);
    return dd_check_launch();
  }
  hipLaunchKernelGGL(kern, grid, dim3(64 * WM * WN), smem, s, p);
  return dd_check_launch();
}

template <typename T, int KS, int TN, int TM, int NBUF>
int launch_rp(const GemmParams& p, const Plan& pl, hipStream_t s) {
  constexpr size_t smem = ((size_t)NBUF * TM * 16 * KS * 32 + 4 * 512 + 2 * KS * 32) * sizeof(T);
  static_assert(smem <= 160 * 1024, "LDS");
  dim3 grid(pl.tiles_m * pl.tiles_n);
  if (p.w_scale) {             // fp8 weights (with or without the LayerNorm prologue)
    if (p.ln_gamma) {
      auto kern = dd_gemm_rp_kernel<T, KS, TN, TM, NBUF, true, true>;
      static std::atomic<uint64_t> attr_done{0};
      dd_ensure_dyn_lds(reinterpret_cast<const void*>(kern), smem, attr_done);
      hipLaunchKernelGGL(kern, grid, dim3(256), smem, s, p);
    } else {
      auto kern = dd_gemm_rp_kernel<T, KS, TN, TM, NBUF, false, true>;
      static std::atomic<uint64_t> attr_done{0};
      dd_ensure_dyn_lds(reinterpret_cast<const void*>(kern), smem, attr_done);
      hipLaunchKernelGGL(kern, grid, dim3(256), smem, s, p);
    }
    return dd_check_launch();
  }
  if (p.ln_gamma) {
    auto kern = dd_gemm_rp_kernel<T, KS, TN, TM, NBUF, true>;
    static std::atomic<uint64_t> attr_done{0};
    dd_ensure_dyn_lds(reinterpret_cast<const void*>(kern), smem, attr_done);
    hipLaunchKernelGGL(kern, grid, dim3(256), smem, s, p);
  } else {
    auto kern = dd_gemm_rp_kernel<T, KS, TN, TM, NBUF, false>;
    static std::atomic<uint64_t> attr_done{0};
    dd_ensure_dyn_lds(reinterpret_cast<const void*>(kern), smem, attr_done);
    hipLaunchKernelGGL(kern, grid, dim3(256), smem, s, p);
  }
  return dd_check_launch();
}

template <typename T>
int launch_rp_k(const GemmParams& p, const Plan& pl, hipStream_t s) {
  const int tm = kTiles[pl.tile_idx].tm;
  switch (p.k) {
    case 320: return tm == 1 ? launch_rp<T, 10, 5, 1, 3>(p, pl, s) : launch_rp<T, 10, 5, 2, 3>(p, pl, s);
    case 640: return tm == 1 ? launch_rp<T, 20, 2, 1, 3>(p, pl, s) : launch_rp<T, 20, 2, 2, 3>(p, pl, s);
    case 1280: if (tm == 1) return launch_rp<T, 40, 2, 1, 3>(p, pl, s); break;
  }
  return DD_ERR_UNSUPPORTED;
}

template <typename T, bool CONV, bool GEGLU>
int launch_tile(const GemmParams& p, const Plan& pl, hipStream_t s) {
#ifndef DD_DBG_ONLY_P        // -DDD_DBG_ONLY_P: a quick-to-compile build with the pipelined family only (reading its ISA)
  if (kTiles[pl.tile_idx].stages == -2) {
    if constexpr (!CONV && !GEGLU) return launch_rp_k<T>(p, pl, s);
    return DD_ERR_UNSUPPORTED;
  }
#endif
  switch (kTiles[pl.tile_idx].id) {
#ifndef DD_DBG_ONLY_P
    case 31: if constexpr (CONV && !GEGLU) return launch_conv3s<T, 4, 2, 6, 2, 5>(p, pl, s); break;
    case 39: if constexpr (CONV && !GEGLU) return launch_conv3s<T, 4, 2, 6, 2, 5, 1, true>(p, pl, s); break;
    case 33: if constexpr (CONV && !GEGLU) return launch_conv3s<T, 2, 2, 6, 2, 8>(p, pl, s); break;
    case 34: if constexpr (CONV && !GEGLU) return launch_conv3s<T, 2, 2, 6, 2, 4>(p, pl, s); break;
    case 35: if constexpr (CONV && !GEGLU) return launch_conv3s<T, 2, 2, 4, 2, 3>(p, pl, s); break;
    case 36: if constexpr (CONV && !GEGLU) return launch_conv3s<T, 2, 2, 4, 4, 3>(p, pl, s); break;
    case 37: if constexpr (CONV && !GEGLU) return launch_conv3s<T, 2, 2, 6, 2, 6, 3>(p, pl, s); break;
    case 38: if constexpr (CONV && !GEGLU) return launch_conv3s<T, 2, 2, 4, 2, 6, 3>(p, pl, s); break;
#endif
    case 72: if constexpr (!GEGLU && !CONV) return launch_cfg3<T, 2, 2, 3, 2, 3, false>(p, pl, s); break;
    case 73: if constexpr (!GEGLU && !CONV) return launch_cfg3<T, 2, 2, 3, 2, 5, false>(p, pl, s); break;
    case 74: if constexpr (!GEGLU && !CONV) return launch_cfg3<T, 2, 2, 3, 2, 6, false>(p, pl, s); break;
    case 75: if constexpr (!CONV) return launch_cfg3<T, 4, 2, 3, 4, 3, GEGLU>(p, pl, s); break;
    case 76: if constexpr (!GEGLU && !CONV) return launch_cfg3<T, 2, 2, 1, 2, 4, false>(p, pl, s); break;
    case 77: if constexpr (!GEGLU && !CONV) return launch_cfg3<T, 2, 2, 1, 2, 6, false>(p, pl, s); break;
    case 78: if constexpr (!GEGLU && !CONV) return launch_cfg3<T, 2, 5, 5, 2, 3, false>(p, pl, s); break;
    case 79: if constexpr (!GEGLU && !CONV) return launch_cfg3<T, 2, 2, 1, 2, 8, false>(p, pl, s); break;
#ifndef DD_DBG_ONLY_P
    case 11: return launch_cfg2<T, 2, 2, 4, 4, 2, CONV, GEGLU>(p, pl, s);
    case 12: return launch_cfg2<T, 2, 2, 4, 4, 3, CONV, GEGLU>(p, pl, s);
    case 14: return launch_cfg2<T, 2, 2, 2, 4, 3, CONV, GEGLU>(p, pl, s);
    case 16: return launch_cfg2<T, 4, 2, 4, 4, 2, CONV, GEGLU>(p, pl, s);
    case 20: return launch_cfg2<T, 4, 2, 4, 4, 3, CONV, GEGLU>(p, pl, s);
    case 13: if constexpr (!GEGLU) return launch_cfg2<T, 2, 2, 4, 2, 3, CONV, false>(p, pl, s); break;
    case 15: if constexpr (!GEGLU) return launch_cfg2<T, 2, 2, 2, 2, 3, CONV, false>(p, pl, s); break;
    case 17: if constexpr (!GEGLU) return launch_cfg2<T, 2, 2, 4, 2, 2, CONV, false>(p, pl, s); break;
    case 18: if constexpr (!GEGLU) return launch_cfg2<T, 2, 2, 2, 2, 2, CONV, false>(p, pl, s); break;
    case 19: if constexpr (!GEGLU) return launch_cfg2<T, 2, 2, 2, 2, 4, CONV, false>(p, pl, s); break;
    case 21: if constexpr (!GEGLU) return launch_cfg2<T, 2, 2, 2, 2, 6, CONV, false>(p, pl, s); break;
    case 22: if constexpr (!GEGLU) return launch_cfg2<T, 2, 2, 2, 2, 8, CONV, false>(p, pl, s); break;
    case 23: if constexpr (!GEGLU) return launch_cfg2<T, 2, 2, 4, 2, 4, CONV, false>(p, pl, s); break;
    case 24: return launch_cfg2<T, 2, 2, 2, 4, 4, CONV, GEGLU>(p, pl, s);
    case 25: return launch_cfg2<T, 2, 2, 4, 4, 4, CONV, GEGLU>(p, pl, s);
    case 26: if constexpr (!GEGLU) return launch_cfg2<T, 4, 2, 6, 2, 2, CONV, false>(p, pl, s); break;
    case 27: if constexpr (!GEGLU) return launch_cfg2<T, 2, 5, 5, 2, 2, CONV, false>(p, pl, s); break;
    case 28: if constexpr (!GEGLU) return launch_cfg2<T, 2, 5, 5, 2, 3, CONV, false>(p, pl, s); break;
    case 40: if constexpr (!GEGLU && !CONV) return launch_cfg2<T, 1, 10, 5, 2, 2, false, false>(p, pl, s); break;
    case 52: if constexpr (!GEGLU) return launch_cfg2<T, 2, 2, 3, 2, 3, CONV, false>(p, pl, s); break;
    case 59: if constexpr (!GEGLU && !CONV) return launch_cfg2<T, 2, 2, 1, 2, 3, false, false>(p, pl, s); break;
    case 60: if constexpr (!GEGLU && !CONV) return launch_cfg2<T, 2, 2, 1, 2, 6, false, false>(p, pl, s); break;
    case 50: return launch_cfg2<T, 2, 4, 8, 4, 2, CONV, GEGLU>(p, pl, s);
    case 44: return launch_cfg2<T, 4, 2, 3, 4, 3, CONV, GEGLU>(p, pl, s);
    case 46: return launch_cfg2<T, 4, 2, 3, 4, 2, CONV, GEGLU>(p, pl, s);
    case 29: if constexpr (GEGLU) return launch_cfg2<T, 2, 5, 5, 4, 2, false, true>(p, pl, s); break;   // 168 VGPRs: only the GEGLU form fits without spills
    case 1: return launch_cfg<T, 2, 2, 4, 4, CONV, GEGLU>(p, pl, s);
    case 3: return launch_cfg<T, 2, 2, 2, 4, CONV, GEGLU>(p, pl, s);
    case 5: return launch_cfg<T, 4, 2, 4, 4, CONV, GEGLU>(p, pl, s);
    case 2: if constexpr (!GEGLU) return launch_cfg<T, 2, 2, 4, 2, CONV, false>(p, pl, s); break;
    case 4: if constexpr (!GEGLU) return launch_cfg<T, 2, 2, 2, 2, CONV, false>(p, pl, s); break;
#endif
  }
  return DD_ERR_UNSUPPORTED;
}

template <typename T>
int launch_dtype(const dd_gemm_desc* d, const GemmParams& p, const Plan& pl, hipStream_t s) {
  int rc = DD_OK;
  if (d->phase == 2) {                      // reduce launch only (per-launch timing of a split-K GEMM)
    if (pl.split <= 1 || p.tile_counters) return DD_OK;
  } else if (d->epilogue == DD_EPI_GEGLU) {
    if (d->conv) return DD_ERR_UNSUPPORTED;
    rc = launch_tile<T, false, true>(p, pl, s);
  } else if (d->conv) {
    rc = launch_tile<T, true, false>(p, pl, s);
  } else {
    rc = launch_tile<T, false, false>(p, pl, s);
  }
  if (rc != DD_OK) return rc;
  if (pl.split > 1 && !p.tile_counters && d->phase != 1) {
    const int64_t total = (int64_t)p.rows * (p.n / 8);
    int blocks = (int)((total + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    GemmParams pr = p;
    hipLaunchKernelGGL(dd_splitk_reduce_kernel<T>, dim3(blocks), dim3(256), 0, s, pr, pl.split);
    rc = dd_check_launch();
  }
  return rc;
}

int validate(const dd_gemm_desc* d) {
  if (!d || !d->a || !d->w || !d->out) return DD_ERR_BAD_ARG;
  if (d->ln_colsum) {                                  // LayerNorm fold
    if (!d->ln_bias || d->conv || d->a2 || d->bias) return DD_ERR_BAD_ARG;
    if (d->k != 320 && d->k != 640 && d->k != 1280) return DD_ERR_UNSUPPORTED;
    if (!dd_aligned16(d->ln_colsum) || !dd_aligned16(d->ln_bias) || (d->lda & 7)) return DD_ERR_BAD_ARG;
  }
  if (d->ln_gamma) {                                   // direct LayerNorm prologue (row-panel family)
    if (!d->ln_beta || d->conv || d->a2 || d->ln_colsum) return DD_ERR_BAD_ARG;
    if (!dd_aligned16(d->ln_gamma) || !dd_aligned16(d->ln_beta) || (d->lda & 7)) return DD_ERR_BAD_ARG;
    if (d->k != 320 && d->k != 640 && d->k != 1280) return DD_ERR_UNSUPPORTED;
  }
  if (d->ln_out) {                                     // LayerNorm emitted by the epilogue (80x320 tile)
    if (!d->lno_gamma || !d->lno_beta || d->conv || d->epilogue != DD_EPI_NONE || d->rowvec || d->accumulate ||
        d->out_f32 || d->out_headmajor_d || d->ln_stats_out || d->ln_colsum || d->ln_gamma || d->w_scale)
      return DD_ERR_UNSUPPORTED;
    if (d->n != 320) return DD_ERR_UNSUPPORTED;
    if (!dd_aligned16(d->ln_out) || !dd_aligned16(d->lno_gamma) || !dd_aligned16(d->lno_beta) || (d->ld_ln_out & 7))
      return DD_ERR_BAD_ARG;
  }
  if (d->w_scale) {                                    // fp8 weights (row-panel family)
    if (d->conv || d->a2 || d->ln_colsum || d->epilogue != DD_EPI_NONE) return DD_ERR_UNSUPPORTED;
    if (d->k != 320 && d->k != 640 && d->k != 1280) return DD_ERR_UNSUPPORTED;
    if (!dd_aligned16(d->w_scale)) return DD_ERR_BAD_ARG;
  }
  if (d->rows <= 0 || d->n <= 0 || d->k <= 0) return DD_ERR_BAD_ARG;
  if (d->rows >= (1 << 22)) return DD_ERR_UNSUPPORTED;         // dd_fdiv's exactness bound (largest real case: 1.08 M)
  if ((d->k & 7) || (d->n & 7) || (d->ldc & 7)) return DD_ERR_BAD_ARG;
  if (d->dtype != DD_F16 && d->dtype != DD_BF16) return DD_ERR_BAD_ARG;
  if (!dd_aligned16(d->a) || !dd_aligned16(d->w) || !dd_aligned16(d->out)) return DD_ERR_BAD_ARG;
  if (d->bias && !dd_aligned16(d->bias)) return DD_ERR_BAD_ARG;
  if (d->res && (!dd_aligned16(d->res) || (d->ldres & 7))) return DD_ERR_BAD_ARG;
  if (d->rowvec && (!dd_aligned16(d->rowvec) || (d->ld_rowvec & 7) || d->rows_per_inst <= 0)) return DD_ERR_BAD_ARG;
  if (d->conv) {
    if (d->a2) return DD_ERR_UNSUPPORTED;
    if (d->cin <= 0 || (d->cin & 7) || d->k != 9 * d->cin) return DD_ERR_BAD_ARG;
    if (d->hin <= 0 || d->win <= 0 || d->hout <= 0 || d->wout <= 0) return DD_ERR_BAD_ARG;
    if (d->stride != 1 && d->stride != 2) return DD_ERR_UNSUPPORTED;
    if (d->hv <= 0 || d->wv <= 0) return DD_ERR_BAD_ARG;
    if (d->rows % (d->hout * d->wout) != 0) return DD_ERR_BAD_ARG;
    if ((d->hv + 2 - 3) / d->stride + 1 != d->hout || (d->wv + 2 - 3) / d->stride + 1 != d->wout) return DD_ERR_BAD_ARG;
  } else {
    if (d->lda & 7) return DD_ERR_BAD_ARG;
    if (d->a2) {
      if (!dd_aligned16(d->a2) || (d->lda2 & 7) || (d->k1 & 7) || d->k1 <= 0 || d->k1 >= d->k) return DD_ERR_BAD_ARG;
    }
  }
  if (d->epilogue != DD_EPI_NONE && d->epilogue != DD_EPI_GEGLU && d->epilogue != DD_EPI_SILU) return DD_ERR_BAD_ARG;
  if (d->out_f32 && (d->epilogue == DD_EPI_GEGLU || d->accumulate)) return DD_ERR_UNSUPPORTED;
  if (d->ln_stats_out && ((d->n & 31) || d->epilogue == DD_EPI_GEGLU || d->out_f32 || !dd_aligned16(d->ln_stats_out)))
    return DD_ERR_UNSUPPORTED;
  if (d->ln_stats_in && (!d->ln_colsum || !dd_aligned16(d->ln_stats_in))) return DD_ERR_BAD_ARG;
  if (d->out_headmajor_d) {
    if (d->out_headmajor_d < 8 || (d->out_headmajor_d & 7) || d->n % d->out_headmajor_d) return DD_ERR_BAD_ARG;
    if (d->conv || d->epilogue == DD_EPI_GEGLU || d->accumulate || d->out_f32 || d->ln_stats_out) return DD_ERR_UNSUPPORTED;
  }
  if (d->epilogue == DD_EPI_GEGLU && (d->res || d->rowvec || d->accumulate || d->alpha != 1.0f)) return DD_ERR_UNSUPPORTED;
  return DD_OK;
}

thread_local char g_kname[160];

}  // namespace

extern "C" int dd_gemm_num_tiles(void) { return kNumTiles; }
extern "C" int dd_gemm_tile_id(int index) { return (index >= 0 && index < kNumTiles) ? kTiles[index].id : -1; }

// Split-K workspace layout: [DD_COUNTER_BYTES of per-tile arrival counters][split fp32 slabs].  The
// counter region must be zero when the workspace is first handed in; every launch leaves it zero.
constexpr int64_t DD_COUNTER_BYTES = 65536;
// In-kernel reduction is OFF by default: the device-scope release / acquire it needs (the L2s of the
// 8 XCDs are not coherent: buffer_wbl2 + buffer_inv per workgroup) costs far more than the second
// launch — 4x7 conv 32 -> 66 us, whole step 73.5 -> 69.1 steps/s.  DD_SPLITK_INKERNEL=1 enables it.
bool inkernel_reduce(const dd_gemm_desc* d, const Plan& pl) {
  // per call: dd_gemm_desc.splitk_inkernel (the tuner times both forms); DD_SPLITK_INKERNEL=1 / 0 forces it on / off
  static const int force = getenv("DD_SPLITK_INKERNEL") ? atoi(getenv("DD_SPLITK_INKERNEL")) : -1;
  const bool want = force >= 0 ? force == 1 : d->splitk_inkernel != 0;
  const int64_t slab_bytes = (int64_t)pl.split * d->rows * d->n * (int64_t)sizeof(float);
  return want && slab_bytes < ((int64_t)1 << 32) &&
         (int64_t)pl.tiles_m * pl.tiles_n * (int64_t)sizeof(int) <= DD_COUNTER_BYTES;
}

extern "C" int64_t dd_gemm_workspace_bytes(const dd_gemm_desc* d) {
  if (validate(d) != DD_OK) return 0;
  const Plan pl = make_plan(d);
  if (pl.unsupported || pl.split <= 1) return 0;
  return DD_COUNTER_BYTES + (int64_t)pl.split * d->rows * d->n * (int64_t)sizeof(float);
}

extern "C" const char* dd_gemm_kernel_name(const dd_gemm_desc* d) {
  if (validate(d) != DD_OK) return "invalid";
  const Plan pl = make_plan(d);
  if (pl.unsupported) return "unsupported";
  const TileCfg& t = kTiles[pl.tile_idx];
  if (t.stages == -2) {
    snprintf(g_kname, sizeof(g_kname), "dd_gemm_rp_kernel<%s, %d, %d, %d, 3, %s> split=1 grid=%dx%d tile=%s",
             d->dtype == DD_F16 ? "_Float16" : "__bf16", d->k / 32, rp_tn(d->k), t.tm, d->ln_gamma ? "true" : "false",
             pl.tiles_m, pl.tiles_n, t.name);
    return g_kname;
  }
  if (t.stages < 0) {
    const bool band = t.stages == -3;
    const int nsw = (t.id == 31 || band) ? 5 : (t.id == 33 ? 8 : (t.id >= 37 ? 6 : (t.id >= 35 ? 3 : 4)));
    snprintf(g_kname, sizeof(g_kname), "dd_conv3s_kernel<%s, %d, %d, %d, %d, %d, %d, %s> split=%d ink=%d grid=%dx%d tile=%s",
             d->dtype == DD_F16 ? "_Float16" : "__bf16", t.wm, t.wn, t.tm, t.tn, nsw, (t.id >= 37 && !band) ? 3 : 1,
             band ? "true" : "false", pl.split, (int)(pl.split > 1 && inkernel_reduce(d, pl)), pl.tiles_m, pl.tiles_n, t.name);
    return g_kname;
  }
  if (t.stages >= 100) {
    snprintf(g_kname, sizeof(g_kname), "dd_gemm3_kernel<%s, %d, %d, %d, %d, %d, %s> split=%d ink=%d grid=%dx%d tile=%s",
             d->dtype == DD_F16 ? "_Float16" : "__bf16", t.wm, t.wn, t.tm, t.tn, t.stages - 100,
             d->epilogue == DD_EPI_GEGLU ? "true" : "false", pl.split, (int)(pl.split > 1 && inkernel_reduce(d, pl)),
             pl.tiles_m, pl.tiles_n, t.name);
    return g_kname;
  }
  // demangled template-argument form, as rocprofv3 prints the kernel symbol
  char stage[16] = "";
  if (t.stages) snprintf(stage, sizeof(stage), " %d,", t.stages);
  snprintf(g_kname, sizeof(g_kname), "dd_gemm%s_kernel<%s, %d, %d, %d, %d,%s %s, %s> split=%d ink=%d grid=%dx%d tile=%s",
           t.stages ? "2" : "", d->dtype == DD_F16 ? "_Float16" : "__bf16", t.wm, t.wn, t.tm, t.tn, stage,
           d->conv ? "true" : "false", d->epilogue == DD_EPI_GEGLU ? "true" : "false",
           pl.split, (int)(pl.split > 1 && inkernel_reduce(d, pl)), pl.tiles_m, pl.tiles_n, t.name);
  return g_kname;
}

extern "C" int dd_gemm(const dd_gemm_desc* d, dd_stream_t stream) {
  const int vc = validate(d);
  if (vc != DD_OK) return vc;
  const Plan pl = make_plan(d);
  if (pl.unsupported) return DD_ERR_UNSUPPORTED;
  GemmParams p{};
  p.g_per_tile = pl.g_per_tile; p.chunks_per_split = pl.chunks_per_split;
  p.ln_colsum = reinterpret_cast<const float*>(d->ln_colsum);
  p.ln_bias = reinterpret_cast<const float*>(d->ln_bias);
  p.ln_eps = d->ln_eps;
  p.ln_gamma = d->ln_gamma; p.ln_beta = d->ln_beta;
  p.w_scale = reinterpret_cast<const float*>(d->w_scale);
  p.ln_out = d->ln_out; p.ld_ln_out = d->ld_ln_out; p.lno_gamma = d->lno_gamma; p.lno_beta = d->lno_beta;
  p.pf_ptr = nullptr; p.pf_bytes = 0; p.pf_blocks = 0;
  if (d->prefetch && d->prefetch_bytes >= (64 << 10) && dd_aligned16(d->prefetch)) {
    p.pf_ptr = d->prefetch;                                   // the launchers decide whether spare workgroups exist
    static const int64_t cap_mb = getenv("DD_PF_MAX_MB") ? atoi(getenv("DD_PF_MAX_MB")) : 32;
    p.pf_bytes = (uint32_t)std::min<int64_t>(d->prefetch_bytes, cap_mb << 20) & ~15u;
  }
  p.a = d->a; p.a2 = d->a2; p.lda = d->lda; p.lda2 = d->lda2;
  p.k1 = d->a2 ? d->k1 : d->k;
  p.rows = d->rows; p.n = d->n; p.k = d->k;
  p.w = d->w; p.bias = d->bias; p.rowvec = d->rowvec;
  p.rows_per_inst = d->rows_per_inst > 0 ? d->rows_per_inst : 1; p.ld_rowvec = d->ld_rowvec;
  p.res = d->res; p.ldres = d->ldres; p.out = d->out; p.ldc = d->ldc;
  p.alpha = d->alpha; p.accumulate = d->accumulate; p.out_f32 = d->out_f32;
  p.stat_out = reinterpret_cast<float*>(d->ln_stats_out);
  p.stat_in = reinterpret_cast<const float*>(d->ln_stats_in);
  p.hm_d = d->out_headmajor_d; p.hm_planes = d->hm_scaled_planes; p.hm_scale = d->hm_scale;
  p.act = d->epilogue == DD_EPI_SILU ? DD_EPI_SILU : DD_EPI_NONE;
  p.hin = d->hin; p.win = d->win; p.cin = d->cin; p.hv = d->hv; p.wv = d->wv;
  p.hout = d->hout; p.wout = d->wout; p.stride = d->stride;
  p.upsample = d->conv && (d->hv != d->hin || d->wv != d->win);
  {
    static const bool nostag = getenv("DD_STAGGER") && atoi(getenv("DD_STAGGER")) == 0;      // A/B switch
    p.no_stagger = nostag ? 1 : 0;
    static const bool rowmajor = getenv("DD_CONV3S_ROWMAJOR") && atoi(getenv("DD_CONV3S_ROWMAJOR")) == 1;
    if (kTiles[pl.tile_idx].stages == -1 && rowmajor) p.upsample = 1;         // conv3s never resizes: flag reused
  }
  // torch nearest: src = min(floor(dst * (in/out)), in-1) with a float scale
  p.scale_h = d->conv ? (float)d->hin / (float)d->hv : 1.f;
  p.scale_w = d->conv ? (float)d->win / (float)d->wv : 1.f;
  p.k_per_split = pl.k_per_split;
  p.tiles_m = pl.tiles_m; p.tiles_n = pl.tiles_n;
  p.band_rows = pl.band_rows; p.bands = pl.bands; p.inv_bands = pl.bands > 0 ? 1.0f / (float)pl.bands : 1.0f;
  p.inv_hw = d->conv ? 1.0f / (float)(d->hout * d->wout) : 1.0f;
  p.inv_wout = d->conv ? 1.0f / (float)d->wout : 1.0f;
  p.inv_rpi = 1.0f / (float)p.rows_per_inst;
  {
    const int64_t nw = (d->epilogue == DD_EPI_GEGLU ? 2 : 1) * (int64_t)d->n;
    p.w_bytes = (uint32_t)(nw * d->k * (d->w_scale ? 1 : 2));
    if (d->conv) {
      p.a_bytes = (uint32_t)((int64_t)d->rows / (d->hout * d->wout) * d->hin * d->win * d->cin * 2);
      p.a2_bytes = 0;
    } else {
      p.a_bytes = (uint32_t)((((int64_t)d->rows - 1) * d->lda + p.k1) * 2);
      p.a2_bytes = d->a2 ? (uint32_t)((((int64_t)d->rows - 1) * d->lda2 + (d->k - d->k1)) * 2) : 0u;
    }
  }
  {
    // extents for the buffer-descriptor epilogue of the pipelined family (T output; 32-bit byte offsets)
    const int64_t ob = (((int64_t)d->rows - 1) * d->ldc + d->n) * 2, rb = (((int64_t)d->rows - 1) * d->ldres + d->n) * 2;
    const bool fits = ob < ((int64_t)1 << 31) && (!d->res || rb < ((int64_t)1 << 31));
    p.out_bytes = fits ? (uint32_t)ob : 0u;
    p.res_bytes = fits && d->res ? (uint32_t)rb : 0u;
  }
  p.persist = 0;
  p.partial = nullptr;
  p.tile_counters = nullptr;
  p.dbg_stamps = nullptr;
#ifdef DD_DBG_STAMP
  if (d->ws && d->ws_bytes >= (4 << 20))
    p.dbg_stamps = reinterpret_cast<uint64_t*>(reinterpret_cast<char*>(d->ws) + d->ws_bytes - (1 << 20));
#endif
  if (pl.split > 1) {
    const int64_t need = DD_COUNTER_BYTES + (int64_t)pl.split * d->rows * d->n * (int64_t)sizeof(float);
    if (!d->ws || d->ws_bytes < need) return DD_ERR_WORKSPACE;
    p.partial = reinterpret_cast<float*>(reinterpret_cast<char*>(d->ws) + DD_COUNTER_BYTES);
    p.partial_bytes = (uint32_t)std::min<int64_t>((int64_t)pl.split * d->rows * d->n * (int64_t)sizeof(float), 0xFFFFFFFFll);
    if (inkernel_reduce(d, pl)) p.tile_counters = reinterpret_cast<int*>(d->ws);
  }
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  dd_clear_error();
  if (d->dtype == DD_F16) return launch_dtype<_Float16>(d, p, pl, s);
  return launch_dtype<__bf16>(d, p, pl, s);
}
