// W8A8 GEMM on the CDNA4 fp8 matrix path (EXTENSION, BASELINE configs[4] "fp8 weights (CDNA4 fp8 MFMA)"; the reference has no
// fp8 code — README.md:47-49 is prose only — so the semantics are this build's, see dd_gemm8 in include/dualdiff_hip.h):
//
//   out[r, n] = a_scale[r] * w_scale[n] * sum_k A8[r, k] * W8[n, k]  (+ bias[n]) (+ res[r, n])
//
// A8 / W8: OCP e4m3fn bytes with a per-row / per-output-channel fp32 scale (symmetric, amax / 448).  The activations are
// quantised by the LayerNorm that produces them (dd_rowquant_fp8: the LayerNorm launch the 16-bit path has anyway).
//
// Why it is faster than the 16-bit projection, not just smaller: this build's tiled GEMMs are bound by L2 -> LDS staging
// bytes and by LDS fragment reads; fp8 halves both per multiply-accumulate, and
// v_mfma_scale_f32_16x16x128_f8f6f4 (unit block scales: a plain fp8 16x16x128) retires 4x the K of the 16-bit 16x16x32 in 2x
// its cycles.  128 x 128 tile, 4 waves of 64 x 64, K steps of 128 BYTES (= 128 fp8 values: the same 128-B LDS rows, 16-B
// XOR swizzle and LDS-DMA staging as csrc/gemm.hip), 2 slots and two workgroups per CU; per K step a wave issues 16 MFMAs
// against 16 ds_read_b128.
// The operand k order inside an MFMA is irrelevant as long as A and B use the same lane -> k map (both are read from LDS
// by the same formula); the C/D layout is dtype-independent on gfx950 (col = lane & 15, row = 4 (lane >> 4) + reg).
//
// Counted vmcnt waits need every DMA instruction to go to memory (round-3 finding, csrc/xattn.hip): tile rows past the
// matrix are CLAMPED to the last row (duplicate loads, never stored) instead of being sent out of range.
#include "dd_common.h"

namespace {

typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

struct Gemm8Params {
  const uint8_t* a; const float* a_scale; int64_t lda;      // [rows][lda] bytes, lda % 128 == 0 (K padded with zeros)
  const uint8_t* w; const float* w_scale; int64_t ldw;      // [n][ldw] bytes
  const void* bias; const void* res; int64_t ldres;
  void* out; int64_t ldc;
  int rows, n, ksteps;                                      // ksteps = padded K / 128
  int tiles_m, tiles_n;
  int hm_d, hm_planes; float hm_scale;                      // head-major output (see dd_gemm_desc.out_headmajor_d)
  int geglu;                                                // W has 2n rows (h | g): out[r, c] = h * gelu_erf(g), c < n
};

constexpr int G8_BM = 128, G8_BN = 128, G8_NST = 2;      // 2 slots = 64 KiB: TWO workgroups per CU hide each other's latencies
constexpr int G8_STAGE = (G8_BM + G8_BN) * 128;             // bytes per ring slot

template <int N>
__device__ __forceinline__ void g8_wait() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

__device__ __forceinline__ void g8_dma(__amdgpu_buffer_rsrc_t rsrc, uint32_t voff, uint32_t soff, void* lds_wave_base) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)lds_wave_base, 16,
                                           (int)voff, (int)soff, 0, 0);
}

template <typename T, bool GEGLU>
__global__ __launch_bounds__(256, 2)
void dd_gemm8_kernel(const Gemm8Params p) {
  using V4 = typename dd_vec<T>::v4;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int c = lane & 15, g = lane >> 4;

  // XCD-aware tile order (csrc/gemm.hip): the tiles of one XCD share activation panels
  const int ntiles = p.tiles_m * p.tiles_n;
  const int xcd = blockIdx.x & 7, xq = ntiles >> 3, xr = ntiles & 7;
  const int tile = ((xcd < xr) ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + (blockIdx.x >> 3);
  const int m0 = (tile / p.tiles_n) * G8_BM;
  constexpr int BN_OUT = GEGLU ? G8_BN / 2 : G8_BN;            // GEGLU: 64 gated channels per tile (h and g rows side by side)
  const int n0 = (tile % p.tiles_n) * BN_OUT;
  const int nw = GEGLU ? 2 * p.n : p.n;                        // weight rows

  const __amdgpu_buffer_rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<uint8_t*>(p.a), 0, (uint32_t)((int64_t)p.rows * p.lda), 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<uint8_t*>(p.w), 0, (uint32_t)((int64_t)nw * p.ldw), 0x00020000);

  // DMA lane tables: instruction j of this wave fills tile rows (j * 4 + wave) * 8 .. + 7 (1 KiB); lane -> row (lane >> 3),
  // position (lane & 7) holding logical chunk pos ^ ((row >> 1) & 7).  Rows past the matrix are clamped (see header).
  uint32_t av[4], wv[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int r = (j * 4 + wave) * 8 + (lane >> 3);
    const uint32_t ch = (uint32_t)(((lane & 7) ^ ((r >> 1) & 7)) << 4);
    av[j] = (uint32_t)(min(m0 + r, p.rows - 1) * (int)p.lda) + ch;
    int wrow;
    if (GEGLU) {     // tile row r = (wave-n block of 64) x (tn 0,1 = h | tn 2,3 = g) x 16: h and g of a channel meet in one lane
      const int wnb = r >> 6, tn = (r >> 4) & 3, rr = r & 15;
      const int chn = min(n0 + wnb * 32 + (tn & 1) * 16 + rr, p.n - 1);
      wrow = (tn >> 1) * p.n + chn;
    } else {
      wrow = min(n0 + r, p.n - 1);
    }
    wv[j] = (uint32_t)(wrow * (int)p.ldw) + ch;
  }
  auto issue = [&](int ks, int slot) {
    unsigned char* base = smem + slot * G8_STAGE;
#pragma unroll
    for (int j = 0; j < 4; ++j) g8_dma(rs_a, av[j], (uint32_t)(ks * 128), base + (j * 4 + wave) * 1024);
#pragma unroll
    for (int j = 0; j < 4; ++j) g8_dma(rs_w, wv[j], (uint32_t)(ks * 128), base + G8_BM * 128 + (j * 4 + wave) * 1024);
  };

  f32x4 acc[4][4];                     // [tn][tm]: lane (c, g): out[row = tm*16 + c][channel = tn*16 + 4g + r]
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nk = p.ksteps;
  issue(0, 0);
  for (int ks = 0; ks < nk; ++ks) {
    g8_wait<0>();                                               // step ks landed (nothing younger is in flight)
    __builtin_amdgcn_s_barrier();                               // ... everyone's share; step ks - 1 is consumed
    asm volatile("" ::: "memory");
    if (ks + 1 < nk) issue(ks + 1, (ks + 1) & 1);               // travels under this step's MFMAs (and the co-resident
                                                                // workgroup's whole step)
    const unsigned char* A = smem + (ks % G8_NST) * G8_STAGE;
    const unsigned char* W = A + G8_BM * 128;
    i32x8 wf[4], xf[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int rw = wn * 64 + i * 16 + c;
      const int ra = wm * 64 + i * 16 + c;
      const i32x4 w0 = *reinterpret_cast<const i32x4*>(W + rw * 128 + (((2 * g) ^ ((rw >> 1) & 7)) << 4));
      const i32x4 w1 = *reinterpret_cast<const i32x4*>(W + rw * 128 + (((2 * g + 1) ^ ((rw >> 1) & 7)) << 4));
      const i32x4 a0 = *reinterpret_cast<const i32x4*>(A + ra * 128 + (((2 * g) ^ ((ra >> 1) & 7)) << 4));
      const i32x4 a1 = *reinterpret_cast<const i32x4*>(A + ra * 128 + (((2 * g + 1) ^ ((ra >> 1) & 7)) << 4));
      wf[i] = i32x8{w0[0], w0[1], w0[2], w0[3], w1[0], w1[1], w1[2], w1[3]};
      xf[i] = i32x8{a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
    }
#pragma unroll
    for (int tn = 0; tn < 4; ++tn)
#pragma unroll
      for (int tm = 0; tm < 4; ++tm)
        acc[tn][tm] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(wf[tn], xf[tm], acc[tn][tm], 0, 0, 0, 127, 0, 127);
  }

  // ---- epilogue: scales, bias, residual / GEGLU gate, 8-byte stores (row-major or one [rows][D] plane per head) ------
#pragma unroll
  for (int tm = 0; tm < 4; ++tm) {
    const int row = m0 + wm * 64 + tm * 16 + c;
    if (row >= p.rows) continue;
    const float as = p.a_scale[row];
    if constexpr (GEGLU) {
#pragma unroll
      for (int t2 = 0; t2 < 2; ++t2) {
        const int ch = n0 + wn * 32 + t2 * 16 + g * 4;
        if (ch >= p.n) continue;
        const f32x4 wsh = *reinterpret_cast<const f32x4*>(p.w_scale + ch);
        const f32x4 wsg = *reinterpret_cast<const f32x4*>(p.w_scale + p.n + ch);
        float h[4], gt[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) { h[r] = acc[t2][tm][r] * as * wsh[r]; gt[r] = acc[t2 + 2][tm][r] * as * wsg[r]; }
        if (p.bias) {
          const V4 bh = *reinterpret_cast<const V4*>(reinterpret_cast<const T*>(p.bias) + ch);
          const V4 bg = *reinterpret_cast<const V4*>(reinterpret_cast<const T*>(p.bias) + p.n + ch);
#pragma unroll
          for (int r = 0; r < 4; ++r) { h[r] += (float)bh[r]; gt[r] += (float)bg[r]; }
        }
        V4 o;
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = (T)dd_geglu_f(h[r], gt[r]);
        *reinterpret_cast<V4*>(reinterpret_cast<T*>(p.out) + (int64_t)row * p.ldc + ch) = o;
      }
    } else {
#pragma unroll
    for (int tn = 0; tn < 4; ++tn) {
      const int ch = n0 + wn * 64 + tn * 16 + g * 4;
      if (ch >= p.n) continue;
      const f32x4 ws = *reinterpret_cast<const f32x4*>(p.w_scale + ch);
      float v[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] = acc[tn][tm][r] * as * ws[r];
      if (p.bias) {
        const V4 b = *reinterpret_cast<const V4*>(reinterpret_cast<const T*>(p.bias) + ch);
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] += (float)b[r];
      }
      if (p.res) {
        const V4 b = *reinterpret_cast<const V4*>(reinterpret_cast<const T*>(p.res) + (int64_t)row * p.ldres + ch);
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] += (float)b[r];
      }
      V4 o;
      if (p.hm_d) {
        const int plane = ch / p.hm_d;
        if (plane < p.hm_planes) {
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] *= p.hm_scale;
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = (T)v[r];
        *reinterpret_cast<V4*>(reinterpret_cast<T*>(p.out) + ((int64_t)plane * p.rows + row) * p.hm_d + (ch - plane * p.hm_d)) = o;
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = (T)v[r];
        *reinterpret_cast<V4*>(reinterpret_cast<T*>(p.out) + (int64_t)row * p.ldc + ch) = o;
      }
    }
    }
  }
}

// ---- row quantisation (optionally behind a LayerNorm): T [rows][c] -> e4m3 [rows][ldq] + fp32 scale[rows] -------------
// One wave per row.  y = LayerNorm(x) (two-pass fp32 statistics, result ROUNDED to T exactly as dd_layernorm stores it) or
// y = x; scale = max|y| / 448 (1 for an all-zero row); q = e4m3(clamp(y / scale, +-448)), round-to-nearest-even; the
// padding columns c .. ldq - 1 are zero.
template <typename T, int NV>
__global__ __launch_bounds__(256)
void dd_rowquant_fp8_kernel(const T* __restrict__ x, const T* __restrict__ gamma, const T* __restrict__ beta,
                            uint8_t* __restrict__ q, float* __restrict__ scale, int64_t rows, int c, int64_t ldq, float eps) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int nvec = c >> 3;
  float v[NV][8];
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int vi = lane + i * 64;
    if (vi < nvec) dd_unpack8<T>(dd_ld16(x + row * c + vi * 8), v[i]);
    else {
#pragma unroll
      for (int e = 0; e < 8; ++e) v[i][e] = 0.f;
    }
  }
  if (gamma) {
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i)
#pragma unroll
      for (int e = 0; e < 8; ++e) s += v[i][e];
    const float mean = dd_wave_sum(s) / (float)c;
    float sq = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const bool on = lane + i * 64 < nvec;
#pragma unroll
      for (int e = 0; e < 8; ++e) { const float d = on ? v[i][e] - mean : 0.f; sq += d * d; }
    }
    const float rstd = rsqrtf(dd_wave_sum(sq) / (float)c + eps);
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int vi = lane + i * 64;
      if (vi < nvec) {
        float ga[8], be[8];
        dd_unpack8<T>(dd_ld16(gamma + vi * 8), ga);
        dd_unpack8<T>(dd_ld16(beta + vi * 8), be);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[i][e] = (float)(T)((v[i][e] - mean) * rstd * ga[e] + be[e]);
      }
    }
  }
  float amax = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i)
#pragma unroll
    for (int e = 0; e < 8; ++e) amax = fmaxf(amax, fabsf(v[i][e]));
  amax = dd_wave_max(amax);
  const float sc = amax > 0.f ? amax / 448.0f : 1.0f;
  const float inv = 1.0f / sc;
  if (lane == 0) scale[row] = sc;
  const int nq = (int)(ldq >> 3);                              // 8-byte groups of the padded row
#pragma unroll
  for (int i = 0; i < NV + 1; ++i) {
    const int vi = lane + i * 64;
    if (vi >= nq) break;
    int lo = 0, hi = 0;
    if (i < NV && vi < nvec) {
      float t[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) t[e] = fminf(fmaxf(v[i < NV ? i : 0][e] * inv, -448.0f), 448.0f);
      lo = __builtin_amdgcn_cvt_pk_fp8_f32(t[0], t[1], lo, false);
      lo = __builtin_amdgcn_cvt_pk_fp8_f32(t[2], t[3], lo, true);
      hi = __builtin_amdgcn_cvt_pk_fp8_f32(t[4], t[5], hi, false);
      hi = __builtin_amdgcn_cvt_pk_fp8_f32(t[6], t[7], hi, true);
    }
    *reinterpret_cast<u32x2*>(q + row * ldq + vi * 8) = u32x2{(uint32_t)lo, (uint32_t)hi};
  }
}

template <typename T>
int launch_gemm8(const Gemm8Params& p, hipStream_t s) {
  constexpr size_t smem = (size_t)G8_NST * G8_STAGE;            // 64 KiB
  if (p.geglu) {
    hipLaunchKernelGGL((dd_gemm8_kernel<T, true>), dim3(p.tiles_m * p.tiles_n), dim3(256), smem, s, p);
  } else {
    hipLaunchKernelGGL((dd_gemm8_kernel<T, false>), dim3(p.tiles_m * p.tiles_n), dim3(256), smem, s, p);
  }
  return dd_check_launch();
}

template <typename T>
int launch_rowquant(const void* x, const void* gamma, const void* beta, uint8_t* q, float* scale, int64_t rows, int c,
                    int64_t ldq, float eps, hipStream_t s) {
  const dim3 grid((unsigned)((rows + 3) / 4));
  const int nv = (c / 8 + 63) / 64;
  auto X = reinterpret_cast<const T*>(x); auto G = reinterpret_cast<const T*>(gamma); auto B = reinterpret_cast<const T*>(beta);
  switch (nv) {
    case 1: hipLaunchKernelGGL((dd_rowquant_fp8_kernel<T, 1>), grid, dim3(256), 0, s, X, G, B, q, scale, rows, c, ldq, eps); break;
    case 2: hipLaunchKernelGGL((dd_rowquant_fp8_kernel<T, 2>), grid, dim3(256), 0, s, X, G, B, q, scale, rows, c, ldq, eps); break;
    case 3: hipLaunchKernelGGL((dd_rowquant_fp8_kernel<T, 3>), grid, dim3(256), 0, s, X, G, B, q, scale, rows, c, ldq, eps); break;
    default: return DD_ERR_UNSUPPORTED;
  }
  return dd_check_launch();
}

}  // namespace

extern "C" int dd_rowquant_fp8(const void* x, const void* gamma, const void* beta, void* q, float* scale, int64_t rows,
                               int32_t c, int64_t ldq, float eps, int32_t dtype, dd_stream_t stream) {
  if (!x || !q || !scale || rows <= 0 || c <= 0) return DD_ERR_BAD_ARG;
  if ((c & 7) || c > 1536 || ldq < c || (ldq & 127) || ldq - c >= 512) return DD_ERR_UNSUPPORTED;
  if ((gamma == nullptr) != (beta == nullptr)) return DD_ERR_BAD_ARG;
  if (dtype != DD_F16 && dtype != DD_BF16) return DD_ERR_BAD_ARG;
  if (!dd_aligned16(x) || !dd_aligned16(q) || (gamma && (!dd_aligned16(gamma) || !dd_aligned16(beta)))) return DD_ERR_BAD_ARG;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  dd_clear_error();
  if (dtype == DD_F16) return launch_rowquant<_Float16>(x, gamma, beta, reinterpret_cast<uint8_t*>(q), scale, rows, c, ldq, eps, s);
  return launch_rowquant<__bf16>(x, gamma, beta, reinterpret_cast<uint8_t*>(q), scale, rows, c, ldq, eps, s);
}

extern "C" int dd_gemm8(const dd_gemm8_desc* d, dd_stream_t stream) {
  if (!d || !d->a || !d->a_scale || !d->w || !d->w_scale || !d->out) return DD_ERR_BAD_ARG;
  if (d->rows <= 0 || d->n <= 0 || d->k_padded <= 0) return DD_ERR_BAD_ARG;
  if ((d->k_padded & 127) || (d->lda & 127) || (d->ldw & 127) || d->lda < d->k_padded || d->ldw < d->k_padded) return DD_ERR_BAD_ARG;
  if ((d->n & 3) || (d->ldc & 3) || (d->res && (d->ldres & 3))) return DD_ERR_BAD_ARG;
  if (d->dtype != DD_F16 && d->dtype != DD_BF16) return DD_ERR_BAD_ARG;
  if (!dd_aligned16(d->a) || !dd_aligned16(d->w) || !dd_aligned16(d->out) || !dd_aligned16(d->w_scale) ||
      (d->bias && !dd_aligned16(d->bias)) || (d->res && !dd_aligned16(d->res))) return DD_ERR_BAD_ARG;
  const int64_t wrows = d->geglu ? 2 * (int64_t)d->n : d->n;
  if ((int64_t)d->rows * d->lda >= (1ll << 31) || wrows * d->ldw >= (1ll << 31)) return DD_ERR_UNSUPPORTED;
  if (d->out_headmajor_d && ((d->out_headmajor_d & 3) || d->n % d->out_headmajor_d || d->res)) return DD_ERR_BAD_ARG;
  if (d->geglu && (d->res || d->out_headmajor_d)) return DD_ERR_UNSUPPORTED;
  Gemm8Params p{};
  p.a = reinterpret_cast<const uint8_t*>(d->a); p.a_scale = d->a_scale; p.lda = d->lda;
  p.w = reinterpret_cast<const uint8_t*>(d->w); p.w_scale = d->w_scale; p.ldw = d->ldw;
  p.bias = d->bias; p.res = d->res; p.ldres = d->ldres; p.out = d->out; p.ldc = d->ldc;
  p.rows = d->rows; p.n = d->n; p.ksteps = d->k_padded / 128;
  p.geglu = d->geglu ? 1 : 0;
  const int bn_out = p.geglu ? G8_BN / 2 : G8_BN;
  p.tiles_m = (d->rows + G8_BM - 1) / G8_BM; p.tiles_n = (d->n + bn_out - 1) / bn_out;
  p.hm_d = d->out_headmajor_d; p.hm_planes = d->hm_scaled_planes; p.hm_scale = d->hm_scale;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  dd_clear_error();
  if (d->dtype == DD_F16) return launch_gemm8<_Float16>(p, s);
  return launch_gemm8<__bf16>(p, s);
}
