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
#include "dd_debug.h"
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
  int out_f32;                           // store fp32 instead of T
  float* stat_out;                       // [rows][n/32][2] row sum / sum of squares of the stored values, or NULL
  const float* stat_in;                  // LayerNorm fold: [rows][k/32][2] table of the `a` rows, or NULL
  int hm_d, hm_planes; float hm_scale;   // head-major output: plane width D, scaled planes, their factor
  int persist;                           // dd_gemm2_kernel: the grid is smaller than the tile count (see the kernel)
  uint64_t* dbg_stamps;                  // DD_DBG_STAMP builds only
  float inv_hw, inv_wout, inv_rpi;       // 1 / (hout*wout), 1 / wout, 1 / rows_per_inst for dd_fdiv
  void* ln_out; int64_t ld_ln_out;       // LayerNorm EMITTED by the epilogue of the 80x320 tile (second output)
  const void* lno_gamma; const void* lno_beta;
  float inv_tiles_n, inv_hm_d;           // dd_gemm4_kernel: 1 / tiles_n, 1 / hm_d for dd_fdiv
  uint32_t ln_out_bytes;                 // ... extent of ln_out for its buffer stores
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

// ---- accumulator tile -> global (shared by both kernel families) ---------------------------
// acc[tn][tm][reg]: output row = tile row tm*16 + (lane & 15),
//                   output col = q*(4*TN) + tn*4 + reg  (q = lane >> 4)   [non-GEGLU]
// Every global read of the epilogue (bias, time-embedding vector, residual, accumulate target) is
// issued before the stores of its row batch: `out` may alias `res`, so a load placed after a store could
// not be hoisted by the compiler and the tile would pay one memory round trip per 8-column group.
template <typename T, int TM, int TN, bool GEGLU>
__device__ __forceinline__ void store_tile(const GemmParams& p, f32x4 (&acc)[TN][TM], int block_m0,
                                           int block_n0, int wave_m, int wave_n, int lane, int row_end,
                                           const float* ln_mean = nullptr, const float* ln_rstd = nullptr) {
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
        for (int e = 0; e < 8; ++e) v[e] = dd_geglu_f(h[e], g[e]);
        dd_st16(reinterpret_cast<T*>(p.out) + (int64_t)row * p.ldc + col, dd_pack8<T>(v));
      }
    }
  } else {
    constexpr int NG = TN / 2;
    const int col0 = block_n0 + wave_n * (TN * 16) + q * (4 * TN);
    if (p.partial) {                       // split-K slab: fp32 stores; dd_splitk_reduce_kernel (a second launch) adds the
      // slabs and runs the epilogue.  (An IN-LAUNCH ordered reduction by the last-arriving K-slice — write-through slabs,
      // agent-scope ticket, sc1 loads — was built in round 3, bit-identical, 3-80 % slower on the step's 23 split-K
      // shapes, and removed in round 5: profiles/r03_splitk_inkernel_ab.txt.)
#pragma unroll
      for (int tm = 0; tm < TM; ++tm) {
        const int row = row0 + tm * 16;
        if (row >= row_end) continue;
#pragma unroll
        for (int g8 = 0; g8 < NG; ++g8) {
          const int col = col0 + g8 * 8;
          if (col >= p.n) continue;
          float* dst = p.partial + ((int64_t)blockIdx.z * p.rows + row) * p.n + col;
          *reinterpret_cast<f32x4*>(dst) = acc[g8 * 2][tm];
          *reinterpret_cast<f32x4*>(dst + 4) = acc[g8 * 2 + 1][tm];
        }
      }
      return;
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

  store_tile<T, TM, TN, GEGLU>(p, acc, block_m0, block_n0, wave_m, wave_n, lane, p.rows);
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
  if constexpr (!dd_dbg::NODMA)
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)lds_wave_base, 16,
                                             (int)voff, (int)soff, 0, 0);
}
// every buffer is < 2^31 bytes (checked on the host), so this lane offset is out of range whatever
// scalar offset is added to it
constexpr uint32_t DD_OOB = 0x80000000u;

// DD_STAMP*, C3_SEG*, dd_dbg::*: hooks of the diagnostic builds, all empty / false in the product (dd_debug.h).

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

  DD_STAMP_DECL();
  DD_STAMP(0);
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
    const uint32_t ksoff = dd_dbg::SAMEK ? 0u : (uint32_t)ik0 * 2u;
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

  // (A staggered schedule — the second half of the waves half a K-step out of phase, as in the direct conv kernel — was
  //  measured on tiles 16 / 20 / 26 in round 3: 2-9 % SLOWER here (L0 conv 41.4 -> 45.0 us, GEGLU 53.0 -> 55.0 us); its
  //  code was removed in round 5.)
  V8 wf[2][TN], xf[2][TM];
  auto mfma_step = [&]() __attribute__((always_inline)) {
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
      for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = 0; j < TM; ++j) acc[i][j] = dd_mfma16(wf[ks][i], xf[ks][j], acc[i][j]);
    }
    __builtin_amdgcn_s_setprio(0);
  };
  int sbase = 0;                       // ring slot of this tile's stage 0 (persistent: tiles follow each other in the ring)
  bool have_next = false;              // persistent: another tile follows, its first stages are issued from this one
  auto kstep = [&](const int kt) __attribute__((always_inline)) {
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
    const int slot = (sbase + kt) % NSTAGE;
    const T* xs = ring + slot * STAGE + (wave_m * TM * 16 + frow) * BK;
    const T* ws = ring + slot * STAGE + BM * BK + (wave_n * TN * 16 + frow) * BK;
    // all fragment reads of the K-step go out first; the MFMAs of the first half then run while the
    // second half's reads are still landing (counted lgkmcnt waits, reads return in order)
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int cofs = ((fchunk + 4 * ks) ^ fswz) << 3;
#pragma unroll
      for (int i = 0; i < TN; ++i) wf[ks][i] = dd_as_v8<T>(dd_ld16(ws + i * 16 * BK + cofs));
#pragma unroll
      for (int j = 0; j < TM; ++j) xf[ks][j] = dd_as_v8<T>(dd_ld16(xs + j * 16 * BK + cofs));
    }
    __builtin_amdgcn_sched_barrier(0);
    mfma_step();
  };
  const bool persist = !CONV && p.persist != 0;
  for (;;) {
  have_next = persist && lin + (int)gridDim.x < ntiles;
  for (int kt = 0; kt < nk; kt += 2) {
    kstep(kt);
    DD_STAMP_IF(kt == 0, 3);                   // after the first K-step
    if (kt + 1 < nk) kstep(kt + 1);
  }
  DD_STAMP(4);
  if constexpr (!CONV && !GEGLU && WAVES_M == 1 && WAVES_N == 10 && TM == 5 && TN == 2) {
    if (p.ln_out) {                       // whole rows in this workgroup: store out AND LayerNorm(out)
      store_tile_ln<T>(p, acc, block_m0, wave_n, lane, reinterpret_cast<float*>(smem));
      return;
    }
  }
  const bool ln = !CONV && p.ln_colsum;
  store_tile<T, TM, TN, GEGLU>(p, acc, block_m0, block_n0, wave_m, wave_n, lane, p.rows,
                               ln ? s_ln_mean : nullptr, ln ? s_ln_rstd : nullptr);
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
  DD_STAMP_FLUSH(p);
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

  DD_STAMP_DECL();
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
    DD_STAMP_IF(c == 0, 3);
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
  if constexpr (!GEGLU && WAVES_M == 1 && WAVES_N == 10 && TM == 5 && TN == 2) {
    if (p.ln_out) {                       // whole rows in this workgroup: store out AND LayerNorm(out) (store_tile_ln syncs)
      store_tile_ln<T>(p, acc, block_m0, wave_n, lane, reinterpret_cast<float*>(smem));
      done = true;
    }
  }
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
  if (!done) store_tile<T, TM, TN, GEGLU>(p, acc, block_m0, block_n0, wave_m, wave_n, lane, p.rows);
  DD_STAMP_FLUSH(p);
}

// =============================================================================================
// Kernel family 2q (round 6): dd_gemm3's pipelined K-step as a PERSISTENT loop over tiles.
//
// dd_gemm3_kernel runs one tile per workgroup: a launch of G generations of tiles pays, G times over, the ring fill
// (every CU pulls its first stages at once), the drain, the table build and an epilogue whose stores nothing overlaps
// (profiles/r05_gemm3_bound.txt: with BOTH the LDS-DMAs and the MFMAs removed 58-71 % of the K = 320 / 640 launches is still
// there).  Here a workgroup walks the tiles lin, lin + gridDim.x, ... as ONE pipeline of K-steps: the ring never drains
// between tiles — the stages of tile i+1 are issued under the last D K-steps of tile i, its stage-0 fragments are read
// under tile i's last MFMAs — and the epilogue of tile i (its operand loads issued A K-steps ahead, its stores) runs with
// D stages of tile i+1 in flight.  The stated obstacle — loads, stores and LDS-DMAs share ONE in-order vmcnt queue — is
// handled by COUNTING: every epilogue issues a fixed number of vector-memory operations (buffer loads / stores whose
// absent operands and out-of-range rows are descriptor range checks, never predication), and the wait in front of a
// K-step allows, besides the younger stages, exactly those epilogue operations that were issued AFTER the stage it
// certifies (two scalar ages, counted in issued stages).  Same arithmetic in the same order per accumulator as
// dd_gemm3_kernel / dd_gemm2_kernel -> bit-identical results, but for the last rounding of two epilogues, where the compiler
// contracts multiply (+ add) and the conversion to T differently than in store8 / store_tile_ln (one ulp on < 0.01 % of the
// elements: the scaled head-major planes, LayerNorm(out); tests/test_gemm4_gpu.py).
// Dense, no split-K, K >= D steps (host-checked); epilogues: plain (bias, alpha, residual, SiLU, accumulate, head-major
// planes), GEGLU, and the LayerNorm-emitting 80 x 320 tile.
// =============================================================================================
// A 16-byte buffer load the COMPILER DOES NOT TRACK (inline asm): its result is consumed A + 1 K-steps later, behind a
// loop whose LDS-DMAs share the vmcnt queue — for a load it tracks, the compiler's own wait in front of the first use can
// only be vmcnt(0) there (it cannot count the loop's iterations), which would drain the ring once per tile.  The caller
// waits by count (wait_loads) and pins the registers to that wait (dd_pin).
__device__ __forceinline__ u32x4 dd_rsrc_words(const void* base, uint32_t bytes) {
  const uint64_t a = reinterpret_cast<uint64_t>(base);
  return u32x4{(uint32_t)a, (uint32_t)(a >> 32) & 0xffffu, bytes, 0x00020000u};
}
__device__ __forceinline__ u32x4 dd_bload16(u32x4 rsrc, uint32_t voff, uint32_t soff = 0) {
  u32x4 v;
  // s_nop 4: a descriptor word the compiler has just produced with a VALU instruction (v_readlane of a spilled SGPR,
  // v_readfirstlane) needs 5 wait states before a vector-memory instruction may read it, and the hazard recogniser does not
  // look inside inline asm — without it a build whose register allocation spills scalars read garbage descriptors here
  // (round 6: NaNs in the biased epilogues of the 10-wave tiles, profiles/r06_experiments.txt section 8)
  asm volatile("s_nop 4\n\tbuffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(v) : "v"(voff), "s"(rsrc), "s"(soff) : "memory");
  return v;
}
__device__ __forceinline__ void dd_pin(u32x4& v) { asm volatile("" : "+v"(v)); }   // uses of v stay behind this point

// DD_G4_STORE: the epilogue's 16-byte buffer store (dd_debug.h; the diagnostic builds of tools/gemm4_bound.sh drop it).
template <int N>
__device__ __forceinline__ void wait_vmcnt_le() {          // vmcnt(min(N, 63)): waiting for MORE than asked is always safe
  wait_vmcnt<(N > 63 ? 63 : N)>();
}

template <typename T, int WAVES_M, int WAVES_N, int TM, int TN, int NSTAGE, bool GEGLU>
__global__ __launch_bounds__(64 * WAVES_M * WAVES_N, (gemm3_min_waves<WAVES_M, WAVES_N>()))
void dd_gemm4_kernel(const GemmParams p) {
  using V8 = typename dd_vec<T>::v8;
  constexpr int NW = WAVES_M * WAVES_N;
  constexpr int BM = WAVES_M * TM * 16;
  constexpr int BN = WAVES_N * TN * 16;
  constexpr int BN_OUT = GEGLU ? BN / 2 : BN;
  constexpr int XI = BM / 8 / NW;
  constexpr int WI = BN / 8 / NW;
  constexpr int LPS = XI + WI;
  constexpr int STAGE = (BM + BN) * BK;
  constexpr bool TIGHT = NSTAGE <= 3;
  constexpr int D = TIGHT ? NSTAGE : NSTAGE - 1;
  constexpr bool LNOUT = !GEGLU && WAVES_M == 1 && WAVES_N == 10 && TM == 5 && TN == 2;   // tile 74: ALWAYS emits LayerNorm(out)
  constexpr int NG = GEGLU ? TN / 4 : TN / 2;                // 8-column output groups per lane
  constexpr bool PRE_ACC = !GEGLU && !LNOUT && TM * NG <= 4; // accumulate target preloaded (else: host keeps such calls off this kernel)
  // epilogue operand loads / stores per lane and tile — FIXED counts (see the header)
  constexpr int EL = LNOUT ? 3 + TM : GEGLU ? 2 * NG : NG + TM * NG + (PRE_ACC ? TM * NG : 0);
  constexpr int ES = LNOUT ? 2 * TM : TM * NG;
  // the operand loads go out A K-steps before the tile's last one and stay in registers until the epilogue.  LATE (the
  // 10-wave tiles: 168 registers per wave, no room for them beside the accumulators and two fragment sets): the loads go
  // out IN the epilogue and it waits for everything in flight — the stages of the next tile keep landing meanwhile.
  constexpr bool LATE = NW > 8;
  constexpr int A = LATE ? 0 : D - 1;
  constexpr bool SECTOR = !dd_dbg::NOSECTOR && !GEGLU && TN == 4;   // plain 16-column lanes: sector-contiguous stores (see make_wv)
  static_assert(BM % (8 * NW) == 0 && BN % (8 * NW) == 0 && NW % 2 == 0, "tile/waves mismatch");
  static_assert(TN % 2 == 0 && (!GEGLU || TN % 4 == 0), "TN");
  static_assert(NSTAGE >= 3 && NSTAGE <= 8 && D >= 3, "NSTAGE");
  static_assert((D - 2) * LPS + EL + ES <= 63 && (A + 1) * LPS <= 63, "vmcnt is a 6-bit counter");

  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  T* ring = reinterpret_cast<T*>(smem);
  __shared__ float s_ln[LNOUT ? NW * BM : 1];               // LayerNorm partials: NOT in the ring (it is never idle here)

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wave_m = wave / WAVES_N;
  const int wave_n = wave % WAVES_N;

  const int ntiles = p.tiles_m * p.tiles_n;
  const int G = (int)gridDim.x;
  const int my_tiles = (ntiles - (int)blockIdx.x + G - 1) / G;      // >= 1: the grid never exceeds the tile count
  const int nk = p.k / BK;                                           // split == 1, K % 64 == 0 (host-checked)
  const int T_ALL = my_tiles * nk;                                   // K-steps of this workgroup

  const int lrow = lane >> 3;
  const int lc = (lane & 7) ^ ((((wave & 1) << 2) + (lane >> 4)) & 7);
  const uint32_t lcb = (uint32_t)lc * 16u;

  // ---- issue side: the tile whose stages are being issued, its tables, its K cursor ---------------------------------
  int ilin = blockIdx.x;
  uint32_t wv[WI], xe[XI];
  int i_m0 = 0;
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
        const int col = bn0 + wvi * (TH * 16) + (r >> 2) * (4 * TH) + t * 4 + (r & 3);
        n_glob = (col < p.n) ? col + (tn >= TH ? p.n : 0) : -1;
      } else if (SECTOR) {
        // a lane's 16 columns as two 8-column groups 32 columns apart: the four lanes of a row then write 64 CONTIGUOUS
        // bytes per store instruction (whole 32-byte sectors) instead of four 16-byte pieces interleaved with the other
        // group's (every sector written half by one instruction, half by the next)
        const int col = bn0 + wvi * (TN * 16) + (tn >> 1) * 32 + (r >> 2) * 8 + (tn & 1) * 4 + (r & 3);
        n_glob = (col < p.n) ? col : -1;
      } else {
        const int col = bn0 + wvi * (TN * 16) + (r >> 2) * (4 * TN) + tn * 4 + (r & 3);
        n_glob = (col < p.n) ? col : -1;
      }
      wv[j] = n_glob >= 0 ? (uint32_t)n_glob * (uint32_t)p.k * 2u + lcb : DD_OOB;
    }
  };
  auto make_xe = [&](const int64_t ld) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < XI; ++j) {
      const int r = i_m0 + (j * NW + wave) * 8 + lrow;
      xe[j] = r < p.rows ? (uint32_t)r * (uint32_t)ld * 2u + lcb : DD_OOB;
    }
  };
  const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.w), 0, p.w_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.a), 0, p.a_bytes, 0x00020000);
  int ik0 = 0, islot = 0, kbase = 0;
  int seam_k = p.a2 ? p.k1 : 0x7fffffff;
  __amdgpu_buffer_rsrc_t rs_x = rs_a;
  auto issue_tile = [&](const int lin) __attribute__((always_inline)) {     // point the issue side at tile `lin`
    const int t = xcd_remap(lin, ntiles);
    const int tm_i = dd_fdiv(t, p.inv_tiles_n);
    i_m0 = tm_i * BM;
    make_wv((t - tm_i * p.tiles_n) * BN_OUT);
    make_xe(p.lda);
    rs_x = rs_a;
    kbase = 0;
    seam_k = p.a2 ? p.k1 : 0x7fffffff;
    ik0 = 0;
  };
  auto seam = [&]() __attribute__((always_inline)) {
    if (ik0 >= seam_k) {
      make_xe(p.lda2);
      rs_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.a2), 0, p.a2_bytes, 0x00020000);
      kbase = p.k1;
      seam_k = 0x7fffffff;
    }
  };
  auto next_issue_tile = [&]() __attribute__((always_inline)) {
    if (ik0 >= p.k && ilin + G < ntiles) { ilin += G; issue_tile(ilin); }
  };
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
  issue_tile(ilin);

  // ---- compute side ------------------------------------------------------------------------------------------------
  int clin = blockIdx.x;
  int block_m0, block_n0;
  auto compute_tile = [&](const int lin) __attribute__((always_inline)) {
    const int t = xcd_remap(lin, ntiles);
    const int tm_i = dd_fdiv(t, p.inv_tiles_n);
    block_m0 = tm_i * BM;
    block_n0 = (t - tm_i * p.tiles_n) * BN_OUT;
  };
  compute_tile(clin);

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

  // prologue: stages 0 and 1 first, the rest behind the first fragment reads (as dd_gemm3_kernel); T_ALL >= nk >= D
#pragma unroll
  for (int s0 = 0; s0 < 2; ++s0) { issue_next(); seam(); next_issue_tile(); }

  V8 wf[2][TN], xf[2][TM];
  int rslot = 0;
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
  using K0 = std::integral_constant<int, 0>;
  using K1 = std::integral_constant<int, 1>;
  constexpr int NMF = TN * TM, NRD = TN + TM;

  wait_vmcnt<LPS>();                               // stage 0 landed (stage 1 may be in flight)
  __builtin_amdgcn_s_barrier();
  read_half(K0{});
  read_half(K1{});
  rslot = 1;
#pragma unroll
  for (int s0 = 2; s0 < D; ++s0) { issue_next(); seam(); next_issue_tile(); }

  // READ = false: the last K-step of a tile — the next tile's stage-0 fragments are read AFTER the epilogue instead of
  // under these MFMAs (exposed once per tile, ~0.1 us), so that the epilogue does not run with two fragment sets live
  // (with them the 10-wave tiles spilled fragments INSIDE the K loop, and a scratch reload waits vmcnt(0): the whole ring)
  auto steady = [&](auto issue_c, auto read_c) __attribute__((always_inline)) {
    constexpr bool ISSUE = decltype(issue_c)::value;
    constexpr bool READ = decltype(read_c)::value;
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
      if constexpr (dd_dbg::NOLDS) return;
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
      if (READ && u < NRD) rd(0, u);
      if (u < NMF) mf(1, u);
      __builtin_amdgcn_sched_barrier(0);
    }
    if constexpr (READ) {
#pragma unroll
      for (int u = 0; u < NRD; ++u) rd(1, u);
    }
    __builtin_amdgcn_s_setprio(0);
    if constexpr (ISSUE) {
      ik0 += BK;
      islot = islot + 1 == NSTAGE ? 0 : islot + 1;
    }
    if constexpr (READ) rslot = rslot + 1 == NSTAGE ? 0 : rslot + 1;
  };

  // ---- epilogue state: operand registers and store offsets of the tile being multiplied ------------------------------
  const int q4 = lane >> 4, c16 = lane & 15;
  u32x4 pb[LNOUT ? 3 : (GEGLU ? 2 * NG : NG)];               // bias (GEGLU: h then gate; LN: bias, gamma, beta)
  u32x4 pr[(GEGLU ? 1 : TM)][(GEGLU || LNOUT) ? 1 : NG];     // residual
  u32x4 pa[PRE_ACC ? TM : 1][PRE_ACC ? NG : 1];              // accumulate target
  uint32_t off_o[TM][LNOUT ? 1 : NG];                        // byte offset of the 16-byte store, or DD_OOB
  uint32_t off_l[LNOUT ? TM : 1];                            // LN: offset into ln_out
  float hmf[(GEGLU || LNOUT) ? 1 : NG];                      // head-major planes: the Q planes' factor
  const uint32_t e_bias = p.bias ? (uint32_t)(GEGLU ? 2 * p.n : p.n) * 2u : 0u;
  const u32x4 rs_b = dd_rsrc_words(p.bias, e_bias);
  const u32x4 rs_r = dd_rsrc_words(p.res, p.res ? p.res_bytes : 0u);
  const u32x4 rs_ac = dd_rsrc_words(p.out, p.accumulate ? p.out_bytes : 0u);
  const __amdgpu_buffer_rsrc_t rs_st = __builtin_amdgcn_make_buffer_rsrc(p.out, 0, p.out_bytes, 0x00020000);
  auto epi_loads = [&]() __attribute__((always_inline)) {    // EL buffer loads, whatever the operands (absent: extent 0 -> zeros)
    const int erow0 = block_m0 + wave_m * (TM * 16) + c16;
    if constexpr (LNOUT) {
      const int col = wave_n * 32 + q4 * 8;
      const u32x4 rs_g = dd_rsrc_words(p.lno_gamma, 640u);
      const u32x4 rs_be = dd_rsrc_words(p.lno_beta, 640u);
      pb[0] = dd_bload16(rs_b, (uint32_t)col * 2u);
      pb[1] = dd_bload16(rs_g, (uint32_t)col * 2u);
      pb[2] = dd_bload16(rs_be, (uint32_t)col * 2u);
#pragma unroll
      for (int tm = 0; tm < TM; ++tm) {
        const int row = erow0 + tm * 16;
        const bool ok = row < p.rows;
        off_o[tm][0] = ok ? ((uint32_t)row * (uint32_t)p.ldc + (uint32_t)col) * 2u : DD_OOB;
        off_l[tm] = ok ? ((uint32_t)row * (uint32_t)p.ld_ln_out + (uint32_t)col) * 2u : DD_OOB;
        pr[tm][0] = dd_bload16(rs_r, ok ? ((uint32_t)row * (uint32_t)p.ldres + (uint32_t)col) * 2u : DD_OOB);
      }
    } else if constexpr (GEGLU) {
      constexpr int TH = TN / 2;
      const int ecol0 = block_n0 + wave_n * (TH * 16) + q4 * (4 * TH);
#pragma unroll
      for (int g8 = 0; g8 < NG; ++g8) {
        const int col = ecol0 + g8 * 8;
        const uint32_t ob = col < p.n ? (uint32_t)col * 2u : DD_OOB;
        pb[g8] = dd_bload16(rs_b, ob);
        pb[NG + g8] = dd_bload16(rs_b, ob, (uint32_t)p.n * 2u);
#pragma unroll
        for (int tm = 0; tm < TM; ++tm) {
          const int row = erow0 + tm * 16;
          off_o[tm][g8] = (row < p.rows && col < p.n) ? ((uint32_t)row * (uint32_t)p.ldc + (uint32_t)col) * 2u : DD_OOB;
        }
      }
    } else {
      const int ecol0 = block_n0 + wave_n * (TN * 16) + (SECTOR ? q4 * 8 : q4 * (4 * TN));
#pragma unroll
      for (int g8 = 0; g8 < NG; ++g8) {
        const int col = ecol0 + g8 * (SECTOR ? 32 : 8);
        pb[g8] = dd_bload16(rs_b, col < p.n ? (uint32_t)col * 2u : DD_OOB);
        int plane = 0;
        hmf[g8] = 1.0f;
        if (p.hm_d) {                                          // one [rows][D] plane per head; 8 columns never straddle a plane
          plane = dd_fdiv(col, p.inv_hm_d);
          if (plane < p.hm_planes) hmf[g8] = p.hm_scale;
        }
#pragma unroll
        for (int tm = 0; tm < TM; ++tm) {
          const int row = erow0 + tm * 16;
          const bool ok = row < p.rows && col < p.n;
          uint32_t oo = ((uint32_t)row * (uint32_t)p.ldc + (uint32_t)col) * 2u;
          if (p.hm_d) oo = (((uint32_t)plane * (uint32_t)p.rows + (uint32_t)row) * (uint32_t)p.hm_d + (uint32_t)(col - plane * p.hm_d)) * 2u;
          off_o[tm][g8] = ok ? oo : DD_OOB;
          pr[tm][g8] = dd_bload16(rs_r, ok ? ((uint32_t)row * (uint32_t)p.ldres + (uint32_t)col) * 2u : DD_OOB);
          if constexpr (PRE_ACC) pa[tm][g8] = dd_bload16(rs_ac, off_o[tm][g8]);
        }
      }
    }
  };
  auto epi_finish = [&]() __attribute__((always_inline)) {   // operands are in registers: arithmetic + ES buffer stores
    DD_G4_STORE_STATE();
#pragma unroll
    for (auto& v : pb) dd_pin(v);
#pragma unroll
    for (auto& row : pr)
#pragma unroll
      for (auto& v : row) dd_pin(v);
#pragma unroll
    for (auto& row : pa)
#pragma unroll
      for (auto& v : row) dd_pin(v);
    if constexpr (LNOUT) {
      constexpr int NCOL = 320;
      const __amdgpu_buffer_rsrc_t rs_ln = __builtin_amdgcn_make_buffer_rsrc(p.ln_out, 0, p.ln_out_bytes, 0x00020000);
      float bias[8], ga[8], be[8];
      dd_unpack8<T>(pb[0], bias);
      dd_unpack8<T>(pb[1], ga);
      dd_unpack8<T>(pb[2], be);
      float v[TM][8], part[TM];
#pragma unroll
      for (int tm = 0; tm < TM; ++tm) {
        float r[8];
        dd_unpack8<T>(pr[tm][0], r);
        float s = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          float x = (acc[e >> 2][tm][e & 3] + bias[e]) * p.alpha + r[e];
          v[tm][e] = (float)(T)x;                             // the stored (rounded) value is what gets normalised
          s += v[tm][e];
        }
        DD_G4_STORE(dd_pack8<T>(v[tm]), rs_st, off_o[tm][0], 0, 0);
        s += __shfl_xor(s, 16, 64);
        s += __shfl_xor(s, 32, 64);
        part[tm] = s;
      }
      // two-pass statistics over the rounded values, the 10 waves' partial sums combined through LDS in a fixed order
      // (the arithmetic of store_tile_ln); raw barriers: __syncthreads() would drain the LDS-DMAs in flight
      if (q4 == 0) {
#pragma unroll
        for (int tm = 0; tm < TM; ++tm) s_ln[wave_n * BM + tm * 16 + c16] = part[tm];
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      float mean[TM];
#pragma unroll
      for (int tm = 0; tm < TM; ++tm) {
        float s = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) s += s_ln[w * BM + tm * 16 + c16];
        mean[tm] = s * (1.0f / (float)NCOL);
        float ss = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) { const float d = v[tm][e] - mean[tm]; ss += d * d; }
        ss += __shfl_xor(ss, 16, 64);
        ss += __shfl_xor(ss, 32, 64);
        part[tm] = ss;
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      if (q4 == 0) {
#pragma unroll
        for (int tm = 0; tm < TM; ++tm) s_ln[wave_n * BM + tm * 16 + c16] = part[tm];
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
#pragma unroll
      for (int tm = 0; tm < TM; ++tm) {
        float ss = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) ss += s_ln[w * BM + tm * 16 + c16];
        const float rstd = rsqrtf(ss * (1.0f / (float)NCOL) + p.ln_eps);
        float o[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (v[tm][e] - mean[tm]) * rstd * ga[e] + be[e];
        DD_G4_STORE(dd_pack8<T>(o), rs_ln, off_l[tm], 0, 0);
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // the partials are read: the next tile may overwrite them
      __builtin_amdgcn_s_barrier();
    } else if constexpr (GEGLU) {
      constexpr int TH = TN / 2;
#pragma unroll
      for (int g8 = 0; g8 < NG; ++g8) {
        float bh[8], bg[8];
        dd_unpack8<T>(pb[g8], bh);
        dd_unpack8<T>(pb[NG + g8], bg);
#pragma unroll
        for (int tm = 0; tm < TM; ++tm) {
          float v[8];
#pragma unroll
          for (int e = 0; e < 8; ++e)
            v[e] = dd_geglu_f(acc[g8 * 2 + (e >> 2)][tm][e & 3] + bh[e], acc[TH + g8 * 2 + (e >> 2)][tm][e & 3] + bg[e]);
          DD_G4_STORE(dd_pack8<T>(v), rs_st, off_o[tm][g8], 0, 0);
          __builtin_amdgcn_sched_barrier(0);       // one group at a time: interleaved, the groups' temporaries spill
        }
      }
    } else {
      const bool silu = p.act == DD_EPI_SILU;
      // a projection without bias, residual, accumulation or activation (the fused Q|K|V GEMM: a sixth of the dense launches)
      // skips the operand arithmetic — 12 of the ~50 vector instructions per 8 outputs remain (uniform branch; the operand
      // loads were issued all the same: their count is what the waits rely on)
      const bool bare = !p.bias && !p.res && !p.accumulate && !silu && p.alpha == 1.0f;
      if (bare) {
#pragma unroll
        for (int tm = 0; tm < TM; ++tm)
#pragma unroll
          for (int g8 = 0; g8 < NG; ++g8) {
            float v[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = acc[g8 * 2 + (e >> 2)][tm][e & 3] * hmf[g8];
            DD_G4_STORE(dd_pack8<T>(v), rs_st, off_o[tm][g8], 0, 0);
            __builtin_amdgcn_sched_barrier(0);
          }
        return;
      }
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
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] *= hmf[g8];
          DD_G4_STORE(dd_pack8<T>(v), rs_st, off_o[tm][g8], 0, 0);
          __builtin_amdgcn_sched_barrier(0);
        }
    }
  };
  // vmcnt immediates from scalar state: the `ahead` youngest stages, plus the epilogue operations issued after the
  // certified stage; a combination without an instantiation waits for MORE (fewer operations left in flight): safe
  auto wait_step = [&](const int ahead, const bool xl, const bool xs) __attribute__((always_inline)) {
    if (ahead >= D - 2) {
      if (!xl && !xs) wait_vmcnt_le<(D - 2) * LPS>();
      else if (xl && xs) wait_vmcnt_le<(D - 2) * LPS + EL + ES>();
      else if (xl) wait_vmcnt_le<(D - 2) * LPS + EL>();
      else wait_vmcnt_le<(D - 2) * LPS + ES>();
    } else if (D > 3 && ahead == D - 3) {
      if (xl) wait_vmcnt_le<(D > 3 ? D - 3 : 0) * LPS + EL>(); else wait_vmcnt_le<(D > 3 ? D - 3 : 0) * LPS>();
    } else if (D > 4 && ahead == D - 4) {
      if (xl) wait_vmcnt_le<(D > 4 ? D - 4 : 0) * LPS + EL>(); else wait_vmcnt_le<(D > 4 ? D - 4 : 0) * LPS>();
    } else if (ahead >= 1) {
      if (xl) wait_vmcnt_le<LPS + EL>(); else wait_vmcnt_le<LPS>();
    } else {
      if (xl) wait_vmcnt_le<EL>(); else wait_vmcnt_le<0>();
    }
  };
  auto wait_loads = [&](const int since) __attribute__((always_inline)) {   // `since` stages were issued behind the operand loads
    if (since >= A + 1) wait_vmcnt_le<(A + 1) * LPS>();
    else if (A >= 1 && since == A) wait_vmcnt_le<(A >= 1 ? A : 0) * LPS>();
    else if (A >= 2 && since == A - 1) wait_vmcnt_le<(A >= 2 ? A - 1 : 0) * LPS>();
    else if (since >= 1) wait_vmcnt_le<LPS>();
    else wait_vmcnt_le<0>();
  };

  constexpr int OLD = 1 << 20;
  using YES = std::true_type;
  using NO = std::false_type;
  int age_l = OLD, age_s = OLD;                    // stages issued since the operand loads / the stores went out
  // One K-step = top (stage g + 1 certified, slot of stage g - 1 / g free) + body + bookkeeping.  Every loop below has ONE
  // straight-line body: with the body variant chosen by a run-time branch inside one loop the accumulators are no longer
  // updated in place (phis of MFMA results), the kernel needs two accumulator sets and spills fragments INSIDE the K loop
  // — and a scratch reload waits vmcnt(0), i.e. for the whole ring (first form of this kernel: 118-315 spilled registers).
  auto top = [&](const int ahead) __attribute__((always_inline)) {
    wait_step(ahead, age_l <= ahead, age_s <= ahead);
    if (TIGHT) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  };
  auto issued = [&]() __attribute__((always_inline)) { seam(); next_issue_tile(); ++age_l; ++age_s; };
  auto zero_acc = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
      for (int j = 0; j < TM; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  };

  // ---- every tile but the last: each of its K-steps issues a stage (of this tile, then of the next) ------------------
  for (int left = my_tiles; left > 1; --left) {
    if constexpr (!LATE) {
      for (int c = 0; c < nk - 1 - A; ++c) { top(D - 2); steady(YES{}, YES{}); issued(); }
      top(D - 2);
      epi_loads();                                 // A K-steps ahead of the tile's last one
      age_l = 0;
      __builtin_amdgcn_sched_barrier(0);
      steady(YES{}, YES{});
      issued();
      for (int c = nk - A; c < nk - 1; ++c) { top(D - 2); steady(YES{}, YES{}); issued(); }
    } else {
      for (int c = 0; c < nk - 1; ++c) { top(D - 2); steady(YES{}, YES{}); issued(); }
    }
    top(D - 2);
    steady(YES{}, NO{});                           // the tile's last K-step
    issued();
    if constexpr (LATE) { epi_loads(); wait_vmcnt<0>(); } else wait_loads(age_l);
    epi_finish();
    age_l = OLD;
    age_s = 0;
    zero_acc();
    clin += G;
    compute_tile(clin);
    read_half(K0{});                               // stage 0 of the next tile: certified by the last step's barrier
    read_half(K1{});
    rslot = rslot + 1 == NSTAGE ? 0 : rslot + 1;
  }
  // ---- the last tile: dd_gemm3_kernel's flow — issue while stages remain, operand loads behind the last DMA, drain -----
  int c = 0;
  for (; c + D < nk; ++c) { top(D - 2); steady(YES{}, YES{}); issued(); }
  if constexpr (!LATE) { epi_loads(); age_l = 0; __builtin_amdgcn_sched_barrier(0); }
  for (; c + 1 < nk; ++c) { top(min(D - 2, nk - 2 - c)); steady(NO{}, YES{}); }
  __builtin_amdgcn_s_setprio(1);
  mfma_half(K0{});
  mfma_half(K1{});
  __builtin_amdgcn_s_setprio(0);
  if constexpr (LATE) epi_loads();
  wait_vmcnt<0>();
  epi_finish();
}

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
// (C3_MFMA / C3_BARRIER / C3_SEG / dd_dbg::C3_*: hooks of tools/conv3s_bound.sh's diagnostic builds, dd_debug.h)
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
  // LOADER waves: in the staggered 8-wave tiles only the early half (waves 0-3) issues LDS-DMAs — an LDS-DMA blocks the
  // issuing wave for 60-185 cycles while the texture path is busy, and the early waves have that time: they cannot start
  // their MFMAs before the late waves' block has left the matrix pipe.  The late waves never wait on vmcnt; the barrier
  // behind the loaders' counted wait publishes the data.  (Round 5; all waves loading, each blocked ~220 cycles per step
  // at the same time with the matrix pipe idle, cost 18 % of the step: profiles/r05_conv3s_segments.txt.)
  constexpr int NL = (NW == 8 && GRP == 1) ? NW / 2 : NW;
  constexpr int XA = (AROWS / 8 + NL - 1) / NL;     // activation DMA pieces per loader wave per chunk
  constexpr int XPT = (XA + 3) / 4;                 // ... issued over taps 0..3, XPT per tap (GRP == 1)
  constexpr int WI = BN / 8 / NL;              // weight DMA pieces per loader wave per (chunk, tap) step
  // NSW weight ring slots: the weights are cold (HBM, 2-3 us) while a (chunk, tap) step lasts
  // ~0.3 us, so the ring is as deep as LDS allows
  static_assert((BAND ? AROWS % 8 == 0 : AROWS % (8 * NL) == 0) && BN % (8 * NL) == 0 && NW % 2 == 0, "tile/waves mismatch");
  static_assert(TN % 2 == 0, "TN");
  static_assert(NSW >= 3 && NSW <= 10 && (NSW - 2) * WI + XA <= 63 && 9 - (NSW - 1) >= 4, "ring depth / vmcnt");

  DD_STAMP_DECL();
  DD_STAMP(0);
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
  // (the other order — column tiles of one row band as neighbours — was measured in round 5: -0.7 % for all direct convs, neutral
  //  for the band form alone)
  const int tile_n = tile / p.tiles_m;
  const int tile_m = tile % p.tiles_m;
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
      const int pc = j * NL + wave;                     // 8-row piece of the slab buffer (loader waves only)
      const bool real = pc < AROWS / 8;
      const int L = (real ? pc : 0) * 8 + lrow;         // LDS row
      const int sidx = L - 16;                          // slab pixel index
      const int pix = band0 - (p.wout + 1) + sidx;      // pixel inside the instance
      const bool ok = real && sidx >= 0 && sidx < vrows + 2 * (p.wout + 1) && pix >= 0 && pix < hw;
      av[j] = ok ? (uint32_t)(g0 * hw + pix) * (uint32_t)p.cin * 2u + lcb_a : DD_OOB;
      adst[j] = (real ? pc : 0) * 8;
    } else {
      const int r = (j * NL + wave) * 8 + lrow;
      av[j] = r < vrows ? (uint32_t)(row0 + r) * (uint32_t)p.cin * 2u + lcb_a : DD_OOB;
      adst[j] = (j * NL + wave) * 8;
    }
  }
  uint32_t wv[WI];                                      // weight rows, permuted like dd_gemm2_kernel
#pragma unroll
  for (int j = 0; j < WI; ++j) {
    const int R = (j * NL + wave) * 8 + lrow;
    const int wvi = R / (TN * 16);
    const int rho = R % (TN * 16);
    const int tn = rho >> 4, r = rho & 15;
    const int col = block_n0 + wvi * (TN * 16) + (r >> 2) * (4 * TN) + tn * 4 + (r & 3);
    wv[j] = col < p.n ? (uint32_t)col * (uint32_t)p.k * 2u + lcb : DD_OOB;
  }
  const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.w), 0, p.w_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.a), 0, p.a_bytes, 0x00020000);

  auto issue_a = [&](int c, const int j0, const int j1) __attribute__((always_inline)) {   // chunk c (local index) -> abuf[c & 1], pieces [j0, j1)
    T* dst = abuf + (c & 1) * AROWS * BK;
    const uint32_t so = (uint32_t)((c_beg + c) * BK) * 2u;
#pragma unroll
    for (int j = 0; j < XA; ++j)
      if (j >= j0 && j < j1) bdma16(rs_a, av[j], so, dst + adst[j] * BK);
  };
  auto issue_w = [&](int c, int t, int slot) __attribute__((always_inline)) {
    T* dst = wring + slot * BN * BK;
    const uint32_t so = (uint32_t)(t * p.cin + (c_beg + c) * BK) * 2u;
#pragma unroll
    for (int j = 0; j < WI; ++j) bdma16(rs_w, wv[j], so, dst + (j * NL + wave) * 8 * BK);
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
  const bool loader = wave < NL;
  if (nc > 0 && loader) {
    issue_a(0, 0, XA);
#pragma unroll
    for (int s0 = 0; s0 < (GRP == 1 ? NSW - 1 : NSW); ++s0)
      if (s0 < nsteps) issue_w(s0 / 9, s0 % 9, s0);
  }
  DD_STAMP(2);
  // (built AFTER the prologue DMAs are in flight: ~60 entries x ~20 VALU instructions took 3.7 us of a 36 us
  //  kernel in front of the first load; now they run under the 2-3 us the cold weights need to arrive)
  // ---- per-lane tap tables: LDS row of the pixel each tap reads (BM = the zero row), 2 x 16 bit
  uint32_t tab[TM][5];
  // (entries are ABSOLUTE LDS addresses of pixel buffer 0 so that a gather is v_bfe_u32 + ds_read with the buffer as an
  //  immediate offset; the dynamic LDS of this kernel starts at 0, and a build that moved it past the 16 bits traps)
  const uint32_t lds_base = (uint32_t)(uintptr_t)((__attribute__((address_space(3))) unsigned char*)smem);
  if (lds_base + AROWS * BK * sizeof(T) > 65536u) __builtin_trap();
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
        // the entry is the fragment's BYTE offset inside the pixel buffer: (row * 8 + swizzled chunk of k-step 0) * 16
        // (k-step 1 is the same address with bit 2 of the chunk flipped: ^ 64); AROWS * 128 < 2^16, so one v_bfe_u32
        // yields the ds_read address
        ra = (((ra << 3) | ((uint32_t)(lane >> 4) ^ (ra & 7u))) << 4) + lds_base;
        packed |= ra << (16 * h);
      }
      tab[tm][t2] = packed;
    }
  }
  int wslot = 0;                                        // ring slot of step s (scalar)
  C3_SEG_DECL();
  // One (chunk, tap) step of a wave is 24 MFMAs (~410 cycles of matrix pipe), 16 fragment reads and 1.9 LDS-DMA
  // issues (an LDS-DMA blocks the issuing wave for 100-130 cycles).  Rounds 2-4 ran them as three blocks in series per
  // wave and relied on the partner wave of the SIMD to fill the holes: 1235 cycles per step for 768 of MFMA, both waves
  // issuing their DMAs at the same time with the matrix pipe idle (tools/conv3s_stamps.py segment clocks,
  // profiles/r05_conv3s_segments.txt).  Round 5:
  //  * the activation fragments of a wave's NEXT MFMA block are gathered INSIDE the current one, output-row block j at a
  //    time, into the registers the four MFMAs of block j have just read (two ds_read_b128 per MFMA gap are nearly
  //    free: MI355X_MICROARCH.md "Issued between MFMAs"; hence the j-major MFMA order) — ONE fragment buffer, not two;
  //  * STAGGER (8-wave tiles): waves 4-7 ("late") run the MFMAs of step s-1 at the HEAD of step s, waves 0-3 ("early")
  //    those of step s at its tail, so the two waves of a SIMD alternate on the matrix pipe behind one barrier per step
  //    (MI355X_MICROARCH.md, "Two waves per SIMD");  early: weight fragments, DMA, MFMAs + gathers of step s+1;
  //    late: MFMAs + gathers of step s, DMA, weight fragments — the two DMA windows are disjoint and each lies under the
  //    other wave's MFMAs.  The role is a COMPILE-TIME parameter of the loop (two copies of it): as a run-time branch
  //    inside every step it cost in-place accumulation and 800 spilled registers.
  // Same arithmetic in the same order per accumulator as before -> bit-identical results.
  V8 xf[2][TM];                                         // [k half][output-row block]
  V8 wf[2][TN];
  const bool late = NW == 8 && GRP == 1 && wave >= NL;
  static_assert(AROWS * BK * sizeof(T) <= 65535, "16-bit gather addresses / immediate offset of buffer 1");
  auto gather_j = [&](auto buf_c, auto tap_c, const int j) __attribute__((always_inline)) {
    constexpr int t = decltype(tap_c)::value;
    constexpr uint32_t BOFF = decltype(buf_c)::value * (AROWS * BK * sizeof(T));
    using LP = const __attribute__((address_space(3))) u32x4*;
    // (volatile: the extraction stays HERE — hoisted, the 54 gather addresses of a chunk cost more registers than the
    //  kernel has, and the spill reloads wait on vmcnt(0), i.e. on the whole weight ring)
    uint32_t a0;
    asm volatile("v_bfe_u32 %0, %1, %2, 16" : "=v"(a0) : "v"(tab[j][t >> 1]), "n"(16 * (t & 1)));
    xf[0][j] = dd_as_v8<T>(*(LP)(uintptr_t)(a0 + BOFF));
    xf[1][j] = dd_as_v8<T>(*(LP)(uintptr_t)((a0 ^ 64u) + BOFF));
  };
  auto step = [&](const int c, auto tap_c, auto buf_c, auto late_c) __attribute__((always_inline)) {
    constexpr int t = decltype(tap_c)::value;
    constexpr int BUF = decltype(buf_c)::value;         // = c & 1: the pixel buffer of this chunk (chunk loop unrolled by two)
    constexpr bool LATE = decltype(late_c)::value;
    const bool more_c = c + 1 < nc;
    const int s = c * 9 + t;
    // This step's DMAs (GRP == 1, loader waves): at taps 0..3 a quarter of the next chunk's pixels, then W(s + NSW - 1)
    // into the slot step s - 1 read.  Per-wave issue order (A pieces, then W) is what the counted waits below assume.
    int dslot_ = wslot + NSW - 1;
    if (dslot_ >= NSW) dslot_ -= NSW;
    const int dslot = dslot_;
    auto step_dma = [&]() __attribute__((always_inline)) {
      if constexpr (GRP == 1 && !dd_dbg::C3_NODMA) {
        if (t < 4 && more_c) issue_a(c + 1, t * XPT, (t + 1) * XPT);
        if (s + NSW - 1 < nsteps) {
          constexpr int ta = (t + NSW - 1) % 9, ca = (t + NSW - 1) / 9;
          issue_w(c + ca, ta, dslot);
        }
      }
    };
    if constexpr (GRP == 1) {
    // W(s) (and with it, in issue order, everything older) must have landed.  Younger loads that may stay in flight:
    // W(s+1..s+NSW-2) and the pixel pieces issued in the NSW-2 steps before this one (taps 0..3 of THIS chunk only:
    // the previous chunk's last taps issue none).  The last NSW-2 steps simply drain.  A(c+1) is complete at step
    // (c, 8), whose MFMA block gathers from it: its last piece went out at tap 3 <= 8 - (NSW - 1).
    if constexpr (!LATE && !dd_dbg::C3_NOWAIT) {
      constexpr int ta0 = t - (NSW - 2) > 0 ? t - (NSW - 2) : 0, ta1 = t - 1 < 3 ? t - 1 : 3;      // taps [ta0, ta1]
      constexpr int j0 = ta0 * XPT < XA ? ta0 * XPT : XA, j1 = (ta1 + 1) * XPT < XA ? (ta1 + 1) * XPT : XA;
      constexpr int NA = ta1 >= ta0 && j1 > j0 ? j1 - j0 : 0;
      if (s + NSW - 2 < nsteps) {
        if (NA > 0 && more_c) wait_vmcnt<(NSW - 2) * WI + NA>();
        else wait_vmcnt<(NSW - 2) * WI>();
      } else {
        wait_vmcnt<0>();
      }
    }
    C3_SEG(0);
    C3_BARRIER();
    C3_SEG(1);
    } else if constexpr (t % GRP == 0) {
      // group start: this group's taps (issued one group ago; the first two groups in the prologue) must
      // have landed; only at the very first group may the second group still be in flight
      if (s == 0 && GRP < nsteps) wait_vmcnt<GRP * WI>();
      else wait_vmcnt<0>();
      C3_BARRIER();              // everyone is done with the previous group's slots
      if (t == 0 && more_c) issue_a(c + 1, 0, XA);
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
    auto wread = [&]() __attribute__((always_inline)) {
      if (dd_dbg::C3_NOWREAD && s != 0) return;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const int cofs = ((fchunk + 4 * ks) ^ fswz) << 3;
#pragma unroll
        for (int i = 0; i < TN; ++i) wf[ks][i] = dd_as_v8<T>(dd_ld16(ws + i * 16 * BK + cofs));
      }
    };
    auto mfma_j = [&](const int j) __attribute__((always_inline)) {
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int i = 0; i < TN; ++i) acc[i][j] = C3_MFMA(wf[ks][i], xf[ks][j], acc[i][j]);
    };
    if constexpr (LATE) {
      // head: the MFMAs of step s-1; block j's registers are refilled with THIS step's fragments as soon as it is done
      if (s > 0) {
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int j = 0; j < TM; ++j) {
          mfma_j(j);
          if constexpr (!dd_dbg::C3_NOGATHER) gather_j(buf_c, tap_c, j);
          __builtin_amdgcn_sched_barrier(0);
        }
        __builtin_amdgcn_s_setprio(0);
      } else {
#pragma unroll
        for (int j = 0; j < TM; ++j) gather_j(buf_c, tap_c, j);
      }
      C3_SEG(2);
      C3_SEG(3);
      wread();
      // the weight slot and the pixel buffer these reads touch are refilled by the loader waves right after the next barrier
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    } else {
      C3_SEG(2);
      wread();
      if (s == 0) {                                   // first step only: nothing was gathered under a previous block
#pragma unroll
        for (int j = 0; j < TM; ++j) gather_j(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}, j);
      }
      __builtin_amdgcn_sched_barrier(0);
      step_dma();
      C3_SEG(3);
      __builtin_amdgcn_sched_barrier(0);
      // the MFMAs of step s; block j's registers are refilled with the fragments of step s+1 (same resident chunk; at
      // t == 8 the next chunk, landed since step NSW-1)
      const bool have_next = t < 8 || more_c;
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int j = 0; j < TM; ++j) {
        mfma_j(j);
        if constexpr (!dd_dbg::C3_NOGATHER)
          if (have_next) gather_j(std::integral_constant<int, (t < 8 ? BUF : BUF ^ 1)>{}, std::integral_constant<int, (t + 1) % 9>{}, j);
        __builtin_amdgcn_sched_barrier(0);
      }
      __builtin_amdgcn_s_setprio(0);
    }
    C3_SEG(4);
  };
  auto chunk = [&](const int c, auto buf_c, auto late_c) __attribute__((always_inline)) {
    step(c, std::integral_constant<int, 0>{}, buf_c, late_c);
    step(c, std::integral_constant<int, 1>{}, buf_c, late_c);
    step(c, std::integral_constant<int, 2>{}, buf_c, late_c);
    step(c, std::integral_constant<int, 3>{}, buf_c, late_c);
    step(c, std::integral_constant<int, 4>{}, buf_c, late_c);
    step(c, std::integral_constant<int, 5>{}, buf_c, late_c);
    step(c, std::integral_constant<int, 6>{}, buf_c, late_c);
    step(c, std::integral_constant<int, 7>{}, buf_c, late_c);
    step(c, std::integral_constant<int, 8>{}, buf_c, late_c);
  };
  auto main_loop = [&](auto late_c) __attribute__((always_inline)) {
    for (int c = 0; c < nc; c += 2) {
      chunk(c, std::integral_constant<int, 0>{}, late_c);
      DD_STAMP_IF(c == 0, 3);                                        // after the first 9 steps
      if (c + 1 < nc) chunk(c + 1, std::integral_constant<int, 1>{}, late_c);
    }
  };
  if constexpr (NW == 8 && GRP == 1) {
    if (late) main_loop(std::true_type{});
    else main_loop(std::false_type{});
  } else {
    main_loop(std::false_type{});
  }
  DD_STAMP(4);
  if (late && nsteps > 0) {                               // staggered waves: the last step's MFMAs are still due
#pragma unroll
    for (int j = 0; j < TM; ++j)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int i = 0; i < TN; ++i) acc[i][j] = C3_MFMA(wf[ks][i], xf[ks][j], acc[i][j]);
  }
  // rows past the tile's instances are padding
  store_tile<T, TM, TN, false>(p, acc, row0, block_n0, wave_m, wave_n, lane, min(p.rows, row0 + vrows));
  DD_STAMP_FLUSH(p);
  C3_SEG_FLUSH(p, wave, lane, nsteps);
}

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
    {20, 4, 2, 4, 4, 3, "256x128/dma3"},
    {23, 2, 2, 4, 2, 4, "128x64/dma4"},
    {24, 2, 2, 2, 4, 4, "64x128/dma4"},
    // 160-wide tiles (10 waves = 2 x 5): every channel count of this network (320, 640, 960, 1280, 1920, 2560) is
    // a multiple of 160, so no column of the tile multiplies padding (a 128-wide tile wastes 1/6 of its MFMAs at
    // N = 320 and 16800 rows / 160 = 105 row tiles x 2 = 210 workgroups fill the chip in ONE generation)
    {27, 2, 5, 5, 2, 2, "160x160/dma2"},
    {28, 2, 5, 5, 2, 3, "160x160/dma3"},
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
    {75, 4, 2, 3, 4, 103, "192x128/p3"},
    {76, 2, 2, 1, 2, 104, "32x64/p4"},
    {77, 2, 2, 1, 2, 106, "32x64/p6"},
    {78, 2, 5, 5, 2, 103, "160x160/p3"},
    {74, 1, 10, 5, 2, 103, "80x320/p3"},            // the LayerNorm-emitting tile (tile 40) on the pipelined loop
    // stages < 0: direct small-image conv (dd_conv3s_kernel); conv with stride 1 / no resize /
    // Cin % 64 == 0 / H*W <= rows of the tile only
    {31, 4, 2, 6, 2, -1, "conv3s 384x64"},
    {39, 4, 2, 6, 2, -3, "conv3s band 384x64"},   // stages == -3: BAND form (images larger than the tile: 28x50 level)
    // (round 5: the same 384 x 64 tile on FOUR waves of 96 x 64, one per SIMD — 10 fragment reads per 24 MFMAs instead of
    //  8 per 12 — was built, bit-identical, and 12-16 % SLOWER on every level (28x50: 40.5 vs 35.2 us): without a partner
    //  wave the step's chain barrier -> weight reads -> MFMAs is exposed; profiles/r05_conv3s_ab.txt.  Removed.)
    {34, 2, 2, 6, 2, -1, "conv3s 192x64/w4"},
    {35, 2, 2, 4, 2, -1, "conv3s 128x64/w3"},     // 72 KB of LDS: two workgroups per CU
    {37, 2, 2, 6, 2, -1, "conv3s 192x64/g3"},     // taps in groups of three: one barrier per 72 MFMAs
    // (Round 5 removed what no entry of the tracked table used: 64x64 rings of 2 / 4 / 6 / 8 slots, 128x64 / 128x128 with
    //  2 / 4, 384x64, the GEGLU-only 160x320, three direct-conv variants, and the round-2 row-panel family.)
};
constexpr int kNumTiles = sizeof(kTiles) / sizeof(kTiles[0]);

inline int tile_bm(const TileCfg& t) { return t.wm * t.tm * 16; }
inline int tile_bn(const TileCfg& t) { return t.wn * t.tn * 16; }

constexpr int kNumCU = 256;

struct Plan { int tile_idx; int split; int tiles_m, tiles_n; int k_per_split; int g_per_tile, chunks_per_split; bool unsupported; bool persist_ok; int band_rows, bands; bool persist3_ok; };

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
    if (d->tile <= 0) {                              // heuristic picked a register-staged tile: take its LDS-DMA twin
      // (the same wave / block shape on a dd_gemm2 ring, 0 < stages < 100; round 5 removed the 2-slot 128x64 / 64x64
      //  tiles the old table {11, 17, 14, 18} pointed at, so the twin is looked up by shape)
      const TileCfg& h = kTiles[ti < 4 ? ti : 3];
      for (int i = 0; i < kNumTiles; ++i)
        if (kTiles[i].stages > 0 && kTiles[i].stages < 100 && kTiles[i].wm == h.wm && kTiles[i].wn == h.wn &&
            kTiles[i].tm == h.tm && kTiles[i].tn == h.tn) { ti = i; break; }
    }
    if (kTiles[ti].stages <= 0 || kTiles[ti].stages >= 100 || !dma_ok(d)) { pl.unsupported = true; return pl; }
  }
  if (ti >= 0 && kTiles[ti].stages >= 100 && (d->conv || (d->ln_out && kTiles[ti].id != 74))) { pl.unsupported = true; return pl; }   // dense only
  if (d->ln_out) {                                   // LayerNorm-emitting epilogue: the 80x320 tile, one column tile
    if (d->tile > 0 && d->tile != 40 && d->tile != 74) { pl.unsupported = true; return pl; }
    // auto: the pipelined form — one tile per workgroup while its row tiles are one residency generation (153 KB of LDS: one
    // workgroup per CU), the persistent walk of dd_gemm4_kernel beyond (round 6: 67200 x 320 x 320 46.5 us against 56.4 for
    // the dd_gemm2 form's walk, x 1280 112.7 against 143.7: profiles/r06_gemm3_bound.txt); with DD_PERSIST3=0 the round-5
    // rule (beyond one generation the dd_gemm2 form, tile 40)
    static const bool off3 = getenv("DD_PERSIST3") && atoi(getenv("DD_PERSIST3")) == 0;
    const int want = d->tile > 0 ? d->tile : ((!off3 || ceil_div(d->rows, 80) <= kNumCU) ? 74 : 40);
    for (int i = 0; i < kNumTiles; ++i) if (kTiles[i].id == want) ti = i;
    if (d->n != 320 || !dma_ok(d)) { pl.unsupported = true; return pl; }
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
    // persistent walk of the PIPELINED family (dd_gemm4_kernel, round 6): split == 1, K in whole steps and at least as long
    // as the ring, an epilogue with a fixed number of memory operations (plain / head-major / GEGLU / the LayerNorm tile),
    // buffer-descriptor addressing (31-bit extents).  DD_PERSIST3=0 is the A/B switch.
    static const bool off3 = getenv("DD_PERSIST3") && atoi(getenv("DD_PERSIST3")) == 0;
    const int ng = geglu ? t.tn / 4 : t.tn / 2;
    const bool pre_acc = !geglu && t.id != 74 && t.tm * ng <= 4;
    const int64_t ob = (((int64_t)d->rows - 1) * d->ldc + d->n) * 2, rb = (((int64_t)d->rows - 1) * d->ldres + d->n) * 2;
    const int64_t lb = (((int64_t)d->rows - 1) * d->ld_ln_out + d->n) * 2;
    pl.persist3_ok = !off3 && !d->conv && t.stages >= 100 && t.id != 76 && t.id != 77 && split == 1 && (d->k % BK) == 0 && nkt >= t.stages - 100 &&
                     !d->out_f32 && !d->ln_stats_out && !d->rowvec && !d->ln_colsum && (!d->accumulate || pre_acc) &&
                     ((t.id == 74) == (d->ln_out != nullptr)) && ob < ((int64_t)1 << 31) && (!d->res || rb < ((int64_t)1 << 31)) &&
                     (!d->ln_out || lb < ((int64_t)1 << 31));
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

template <typename T, int WM, int WN, int TM, int TN, int NSTAGE, bool CONV, bool GEGLU>
int launch_cfg2(const GemmParams& p, const Plan& pl, hipStream_t s) {
  constexpr int BM = WM * TM * 16, BN = WN * TN * 16;
  constexpr size_t smem = (size_t)NSTAGE * (BM + BN) * BK * sizeof(T);
  static_assert(smem <= 160 * 1024, "LDS");
  auto kern = dd_gemm2_kernel<T, WM, WN, TM, TN, NSTAGE, CONV, GEGLU>;
  static std::atomic<uint64_t> attr_done{0};
  dd_ensure_dyn_lds(reinterpret_cast<const void*>(kern), smem, attr_done);
  dim3 grid(pl.tiles_m * pl.tiles_n, 1, pl.split);
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

// Workgroups of a pipelined tile that are resident per CU — decided from the tile alone (ring bytes and waves), not from an
// occupancy query, so that dd_gemm_kernel_name reports the launcher's choice without a device: two 4-wave workgroups
// where two rings fit the 160 KB of LDS (the 60 KB 96x64 ring; its kernels need <= 128 registers), else one.
inline int gemm4_resident(const TileCfg& t) {
  const int ring = (t.stages - 100) * (tile_bm(t) + tile_bn(t)) * BK * 2;
  return (t.wm * t.wn == 4 && 2 * ring <= 160 * 1024) ? 2 : 1;
}
inline bool gemm4_takes(const Plan& pl) {          // the persistent form: more tiles than one residency generation
  return pl.persist3_ok && pl.tiles_m * pl.tiles_n > kNumCU * gemm4_resident(kTiles[pl.tile_idx]);
}

// the persistent form when the tiles exceed one residency generation, else the one-tile-per-workgroup kernel
template <typename T, int WM, int WN, int TM, int TN, int NSTAGE, bool GEGLU>
int launch_cfg34(const GemmParams& p, const Plan& pl, hipStream_t s) {
  if (pl.persist3_ok) {
    constexpr int BM = WM * TM * 16, BN = WN * TN * 16;
    constexpr size_t smem = (size_t)NSTAGE * (BM + BN) * BK * sizeof(T);
    auto kern = dd_gemm4_kernel<T, WM, WN, TM, TN, NSTAGE, GEGLU>;
    const int g = kNumCU * gemm4_resident(kTiles[pl.tile_idx]);
    if (gemm4_takes(pl)) {
      static std::atomic<uint64_t> attr_done{0};
      dd_ensure_dyn_lds(reinterpret_cast<const void*>(kern), smem, attr_done);
      hipLaunchKernelGGL(kern, dim3(g, 1, 1), dim3(64 * WM * WN), smem, s, p);
      return dd_check_launch();
    }
  }
  return launch_cfg3<T, WM, WN, TM, TN, NSTAGE, GEGLU>(p, pl, s);
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
  hipLaunchKernelGGL(kern, grid, dim3(64 * WM * WN), smem, s, p);
  return dd_check_launch();
}

template <typename T, bool CONV, bool GEGLU>
int launch_tile(const GemmParams& p, const Plan& pl, hipStream_t s) {
  switch (kTiles[pl.tile_idx].id) {
#ifndef DD_DBG_ONLY_P        // -DDD_DBG_ONLY_P: a quick-to-compile build with the pipelined family only (reading its ISA)
    case 31: if constexpr (CONV && !GEGLU) return launch_conv3s<T, 4, 2, 6, 2, 5>(p, pl, s); break;
    case 39: if constexpr (CONV && !GEGLU) return launch_conv3s<T, 4, 2, 6, 2, 5, 1, true>(p, pl, s); break;
    case 34: if constexpr (CONV && !GEGLU) return launch_conv3s<T, 2, 2, 6, 2, 4>(p, pl, s); break;
    case 35: if constexpr (CONV && !GEGLU) return launch_conv3s<T, 2, 2, 4, 2, 3>(p, pl, s); break;
    case 37: if constexpr (CONV && !GEGLU) return launch_conv3s<T, 2, 2, 6, 2, 6, 3>(p, pl, s); break;
#endif
#ifndef DD_DBG_ONLY_C3       // -DDD_DBG_ONLY_C3: the direct-conv family only (tools/conv3s_bound.sh)
    case 72: if constexpr (!GEGLU && !CONV) return launch_cfg34<T, 2, 2, 3, 2, 3, false>(p, pl, s); break;
    case 73: if constexpr (!GEGLU && !CONV) return launch_cfg34<T, 2, 2, 3, 2, 5, false>(p, pl, s); break;
    case 75: if constexpr (!CONV) return launch_cfg34<T, 4, 2, 3, 4, 3, GEGLU>(p, pl, s); break;
    case 76: if constexpr (!GEGLU && !CONV) return launch_cfg3<T, 2, 2, 1, 2, 4, false>(p, pl, s); break;
    case 77: if constexpr (!GEGLU && !CONV) return launch_cfg3<T, 2, 2, 1, 2, 6, false>(p, pl, s); break;
    case 78: if constexpr (!GEGLU && !CONV) return launch_cfg34<T, 2, 5, 5, 2, 3, false>(p, pl, s); break;
    case 74: if constexpr (!GEGLU && !CONV) return launch_cfg34<T, 1, 10, 5, 2, 3, false>(p, pl, s); break;
#endif
#if !defined(DD_DBG_ONLY_P) && !defined(DD_DBG_ONLY_C3)
    case 11: return launch_cfg2<T, 2, 2, 4, 4, 2, CONV, GEGLU>(p, pl, s);
    case 12: return launch_cfg2<T, 2, 2, 4, 4, 3, CONV, GEGLU>(p, pl, s);
    case 14: return launch_cfg2<T, 2, 2, 2, 4, 3, CONV, GEGLU>(p, pl, s);
    case 16: return launch_cfg2<T, 4, 2, 4, 4, 2, CONV, GEGLU>(p, pl, s);
    case 20: return launch_cfg2<T, 4, 2, 4, 4, 3, CONV, GEGLU>(p, pl, s);
    case 13: if constexpr (!GEGLU) return launch_cfg2<T, 2, 2, 4, 2, 3, CONV, false>(p, pl, s); break;
    case 15: if constexpr (!GEGLU) return launch_cfg2<T, 2, 2, 2, 2, 3, CONV, false>(p, pl, s); break;
    case 23: if constexpr (!GEGLU) return launch_cfg2<T, 2, 2, 4, 2, 4, CONV, false>(p, pl, s); break;
    case 24: return launch_cfg2<T, 2, 2, 2, 4, 4, CONV, GEGLU>(p, pl, s);
    case 27: if constexpr (!GEGLU) return launch_cfg2<T, 2, 5, 5, 2, 2, CONV, false>(p, pl, s); break;
    case 28: if constexpr (!GEGLU) return launch_cfg2<T, 2, 5, 5, 2, 3, CONV, false>(p, pl, s); break;
    case 40: if constexpr (!GEGLU && !CONV) return launch_cfg2<T, 1, 10, 5, 2, 2, false, false>(p, pl, s); break;
    case 52: if constexpr (!GEGLU) return launch_cfg2<T, 2, 2, 3, 2, 3, CONV, false>(p, pl, s); break;
    case 59: if constexpr (!GEGLU && !CONV) return launch_cfg2<T, 2, 2, 1, 2, 3, false, false>(p, pl, s); break;
    case 60: if constexpr (!GEGLU && !CONV) return launch_cfg2<T, 2, 2, 1, 2, 6, false, false>(p, pl, s); break;
    case 50: return launch_cfg2<T, 2, 4, 8, 4, 2, CONV, GEGLU>(p, pl, s);
    case 44: return launch_cfg2<T, 4, 2, 3, 4, 3, CONV, GEGLU>(p, pl, s);
    case 46: return launch_cfg2<T, 4, 2, 3, 4, 2, CONV, GEGLU>(p, pl, s);
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
    if (pl.split <= 1) return DD_OK;
  } else if (d->epilogue == DD_EPI_GEGLU) {
    if (d->conv) return DD_ERR_UNSUPPORTED;
    rc = launch_tile<T, false, true>(p, pl, s);
  } else if (d->conv) {
    rc = launch_tile<T, true, false>(p, pl, s);
  } else {
    rc = launch_tile<T, false, false>(p, pl, s);
  }
  if (rc != DD_OK) return rc;
  if (pl.split > 1 && d->phase != 1) {
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
  if (d->ln_out) {                                     // LayerNorm emitted by the epilogue (80x320 tile)
    if (!d->lno_gamma || !d->lno_beta || d->conv || d->epilogue != DD_EPI_NONE || d->rowvec || d->accumulate ||
        d->out_f32 || d->out_headmajor_d || d->ln_stats_out || d->ln_colsum)
      return DD_ERR_UNSUPPORTED;
    if (d->n != 320) return DD_ERR_UNSUPPORTED;
    if (!dd_aligned16(d->ln_out) || !dd_aligned16(d->lno_gamma) || !dd_aligned16(d->lno_beta) || (d->ld_ln_out & 7))
      return DD_ERR_BAD_ARG;
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

// Split-K workspace layout: [DD_COUNTER_BYTES, reserved][split fp32 slabs].  The reserved head held the arrival counters
// of the in-launch reduction (round 3, removed in round 5: slower than the reduce launch on every split-K shape of the
// step); the slab offset is kept because dd_groupnorm_splitk's callers address the slabs behind it.
constexpr int64_t DD_COUNTER_BYTES = 65536;

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
  if (t.stages < 0) {
    const bool band = t.stages == -3;
    const int nsw = (t.id == 31 || band) ? 5 : (t.id >= 37 ? 6 : (t.id >= 35 ? 3 : 4));
    snprintf(g_kname, sizeof(g_kname), "dd_conv3s_kernel<%s, %d, %d, %d, %d, %d, %d, %s> split=%d grid=%dx%d tile=%s",
             d->dtype == DD_F16 ? "_Float16" : "__bf16", t.wm, t.wn, t.tm, t.tn, nsw, (t.id >= 37 && !band) ? 3 : 1,
             band ? "true" : "false", pl.split, pl.tiles_m, pl.tiles_n, t.name);
    return g_kname;
  }
  if (t.stages >= 100) {
    snprintf(g_kname, sizeof(g_kname), "dd_gemm%d_kernel<%s, %d, %d, %d, %d, %d, %s> split=%d grid=%dx%d tile=%s",
             gemm4_takes(pl) ? 4 : 3, d->dtype == DD_F16 ? "_Float16" : "__bf16", t.wm, t.wn, t.tm, t.tn, t.stages - 100,
             d->epilogue == DD_EPI_GEGLU ? "true" : "false", pl.split, pl.tiles_m, pl.tiles_n, t.name);
    return g_kname;
  }
  // demangled template-argument form, as rocprofv3 prints the kernel symbol
  char stage[16] = "";
  if (t.stages) snprintf(stage, sizeof(stage), " %d,", t.stages);
  snprintf(g_kname, sizeof(g_kname), "dd_gemm%s_kernel<%s, %d, %d, %d, %d,%s %s, %s> split=%d grid=%dx%d tile=%s",
           t.stages ? "2" : "", d->dtype == DD_F16 ? "_Float16" : "__bf16", t.wm, t.wn, t.tm, t.tn, stage,
           d->conv ? "true" : "false", d->epilogue == DD_EPI_GEGLU ? "true" : "false",
           pl.split, pl.tiles_m, pl.tiles_n, t.name);
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
  p.ln_out = d->ln_out; p.ld_ln_out = d->ld_ln_out; p.lno_gamma = d->lno_gamma; p.lno_beta = d->lno_beta;
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
    p.w_bytes = (uint32_t)(nw * d->k * 2);
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
  p.inv_tiles_n = 1.0f / (float)(pl.tiles_n > 0 ? pl.tiles_n : 1);
  p.inv_hm_d = d->out_headmajor_d > 0 ? 1.0f / (float)d->out_headmajor_d : 1.0f;
  p.ln_out_bytes = d->ln_out ? (uint32_t)((((int64_t)d->rows - 1) * d->ld_ln_out + d->n) * 2) : 0u;
  p.partial = nullptr;
  p.dbg_stamps = nullptr;
  DD_STAMP_HOST(p, d);
  if (pl.split > 1) {
    const int64_t need = DD_COUNTER_BYTES + (int64_t)pl.split * d->rows * d->n * (int64_t)sizeof(float);
    if (!d->ws || d->ws_bytes < need) return DD_ERR_WORKSPACE;
    p.partial = reinterpret_cast<float*>(reinterpret_cast<char*>(d->ws) + DD_COUNTER_BYTES);
  }
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  dd_clear_error();
  if (d->dtype == DD_F16) return launch_dtype<_Float16>(d, p, pl, s);
  return launch_dtype<__bf16>(d, p, pl, s);
}
