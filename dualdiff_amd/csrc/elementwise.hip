// Small HBM-bound kernels of the denoising step: residual adds, layout converts at the
// 4-channel latent boundary, sinusoidal timestep embedding, conv_out (320->4) and the fused
// CFG + DDIM update.  All vectorised to 16 B per lane where the shape allows.
#include "dd_common.h"

namespace {

template <typename T>
__global__ __launch_bounds__(256)
void dd_add_kernel(const T* a, const T* b, const T* c, T* y, int64_t nvec) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nvec;
       i += (int64_t)gridDim.x * blockDim.x) {
    float fa[8], fb[8];
    dd_unpack8<T>(dd_ld16(a + i * 8), fa);
    dd_unpack8<T>(dd_ld16(b + i * 8), fb);
#pragma unroll
    for (int e = 0; e < 8; ++e) fa[e] += fb[e];
    if (c) {
      dd_unpack8<T>(dd_ld16(c + i * 8), fb);
#pragma unroll
      for (int e = 0; e < 8; ++e) fa[e] += fb[e];
    }
    dd_st16(y + i * 8, dd_pack8<T>(fa));
  }
}

template <typename T, int OP>   // OP 0: scale, 1: silu
__global__ __launch_bounds__(256)
void dd_unary_kernel(const T* x, T* y, float s, int64_t nvec) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nvec;
       i += (int64_t)gridDim.x * blockDim.x) {
    float f[8];
    dd_unpack8<T>(dd_ld16(x + i * 8), f);
#pragma unroll
    for (int e = 0; e < 8; ++e) f[e] = OP == 0 ? f[e] * s : dd_silu_f(f[e]);
    dd_st16(y + i * 8, dd_pack8<T>(f));
  }
}

template <typename T>
__global__ __launch_bounds__(256)
void dd_nchw_to_nhwc_kernel(const T* x, T* y, int m, int c, int hw, int c_pad) {
  const int64_t total = (int64_t)m * hw * c_pad;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int ch = (int)(i % c_pad);
    const int64_t row = i / c_pad;
    const int px = (int)(row % hw);
    const int inst = (int)(row / hw);
    y[i] = ch < c ? x[((int64_t)inst * c + ch) * hw + px] : (T)0.f;
  }
}

template <typename T>
__global__ __launch_bounds__(256)
void dd_nhwc_to_nchw_kernel(const T* x, T* y, int m, int c, int hw, int ldx) {
  const int64_t total = (int64_t)m * c * hw;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int px = (int)(i % hw);
    const int64_t r = i / hw;
    const int ch = (int)(r % c);
    const int inst = (int)(r / c);
    y[i] = x[((int64_t)inst * hw + px) * ldx + ch];
  }
}

template <typename T>
__global__ __launch_bounds__(256)
void dd_timestep_embedding_kernel(const float* t, T* out, int n, int dim, int flip, float shift) {
  const int half = dim / 2;
  const int total = n * half;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const int row = i / half, j = i - row * half;
    const float freq = expf(-9.210340371976184f * (float)j / ((float)half - shift));
    const float arg = t[row] * freq;
    const float sv = sinf(arg), cv = cosf(arg);
    T* o = out + (int64_t)row * dim;
    if (flip) { o[j] = (T)cv; o[half + j] = (T)sv; }
    else      { o[j] = (T)sv; o[half + j] = (T)cv; }
  }
}

// ORS ray sampling: one thread per (camera, sample, pixel), pixel fastest (coalesced condition
// writes).  fp32 with the reference's operation order and NO fma contraction: the result is a voxel
// index, a last-bit difference could flip a class.
template <typename T>
__global__ __launch_bounds__(256)
void dd_ors_project_kernel(const uint8_t* __restrict__ occ, const float* __restrict__ origin,
                           const float* __restrict__ dir, uint8_t* __restrict__ labels, T* __restrict__ cond,
                           int n_cam, int hw, int samples, float step, int keep_fg, int keep_bg) {
  const int64_t total = (int64_t)n_cam * samples * hw;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int pix = (int)(i % hw);
    const int64_t r = i / hw;
    const int s = (int)(r % samples);
    const int cam = (int)(r / samples);
    const float t = __fmul_rn((float)s, step);
    const float* o = origin + cam * 3;
    const float* d = dir + ((int64_t)cam * hw + pix) * 3;
    float g[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) g[a] = __fdiv_rn(__fadd_rn(o[a], __fmul_rn(t, d[a])), 40.0f);
    const float gz = __fsub_rn(__fdiv_rn(__fmul_rn(g[2], 40.0f), 3.2f), 0.6875f);
    auto idx = [](float c, float size) -> int {       // grid_sample nearest, align_corners = False
      return (int)rintf(__fdiv_rn(__fsub_rn(__fmul_rn(__fadd_rn(c, 1.0f), size), 1.0f), 2.0f));
    };
    const int ix = idx(g[0], 200.0f), iy = idx(g[1], 200.0f), iz = idx(gz, 16.0f);
    int lab = 17;
    if (ix >= 0 && ix < 200 && iy >= 0 && iy < 200 && iz >= 0 && iz < 16) lab = occ[(ix * 200 + iy) * 16 + iz];
    if (labels) labels[((int64_t)cam * hw + pix) * samples + s] = (uint8_t)lab;
    if (cond) {
      int c = lab;
      if (!keep_fg && c <= 10) c = 17;
      if (!keep_bg && c >= 11) c = 17;
      cond[i] = (T)__fdiv_rn((float)c, 17.0f);
    }
  }
}

// one wave per row; the row is read twice (max+sum pass, write pass) — it sits in L2
template <typename T>
__global__ __launch_bounds__(256)
void dd_softmax_rows_kernel(const float* __restrict__ s, T* __restrict__ p, int64_t rows, int cols,
                            int64_t lds, int64_t ldp) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float* sr = s + row * lds;
  float mx = -INFINITY;
  for (int c = lane; c < cols; c += 64) mx = fmaxf(mx, sr[c]);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
  float sum = 0.f;
  for (int c = lane; c < cols; c += 64) sum += __expf(sr[c] - mx);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
  const float inv = 1.0f / sum;
  T* pr = p + row * ldp;
  for (int c = lane; c < ldp; c += 64) pr[c] = c < cols ? (T)(__expf(sr[c] - mx) * inv) : (T)0.f;
}

struct FourierFreqs { float f[16]; };

template <typename TI, typename TO>
__global__ __launch_bounds__(256)
void dd_fourier_embed_kernel(const TI* x, TO* out, int64_t total, int dims, FourierFreqs fr, int nf, int inc) {
  const int width = dims * (inc + 2 * nf);
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t row = i / dims;
    const int d = (int)(i - row * dims);
    const float v = (float)x[i];
    TO* o = out + row * width + d;
    if (inc) { *o = (TO)v; o += dims; }
    for (int k = 0; k < nf; ++k) {
      const float a = v * fr.f[k];
      o[0] = (TO)sinf(a);
      o[dims] = (TO)cosf(a);
      o += 2 * dims;
    }
  }
}

// conv3x3 (pad 1, stride 1) with a handful of output channels: one wave per output pixel,
// the 9*cin products are spread over the 64 lanes and reduced with shuffles.  Output NCHW.
template <typename T, int MAXCO>
__global__ __launch_bounds__(256)
void dd_conv3x3_small_kernel(const T* x, const T* w, const T* bias, T* y,
                             int m, int h, int wd, int cin, int cout) {
  const int lane = threadIdx.x & 63;
  const int64_t pix = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int64_t npix = (int64_t)m * h * wd;
  if (pix >= npix) return;
  const int hw = h * wd;
  const int inst = (int)(pix / hw);
  const int rem = (int)(pix - (int64_t)inst * hw);
  const int oy = rem / wd, ox = rem - oy * wd;
  const int cv = cin >> 3;
  const int nvec = 9 * cv;
  float acc[MAXCO];
#pragma unroll
  for (int o = 0; o < MAXCO; ++o) acc[o] = 0.f;
  for (int v = lane; v < nvec; v += 64) {
    const int tap = v / cv, cvi = v - tap * cv;
    const int ky = tap / 3, kx = tap - ky * 3;
    const int iy = oy + ky - 1, ix = ox + kx - 1;
    if (iy < 0 || iy >= h || ix < 0 || ix >= wd) continue;
    float f[8];
    dd_unpack8<T>(dd_ld16(x + (((int64_t)inst * h + iy) * wd + ix) * cin + cvi * 8), f);
#pragma unroll
    for (int o = 0; o < MAXCO; ++o) {
      if (o < cout) {
        float g[8];
        dd_unpack8<T>(dd_ld16(w + (int64_t)o * 9 * cin + v * 8), g);
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[o] += f[e] * g[e];
      }
    }
  }
#pragma unroll
  for (int o = 0; o < MAXCO; ++o) {
    const float s = dd_wave_sum(acc[o]);
    if (lane == 0 && o < cout) {
      const float bv = bias ? (float)bias[o] : 0.f;
      y[((int64_t)inst * cout + o) * hw + rem] = (T)(s + bv);
    }
  }
}

// ---- conv3x3 for THIN channel counts on large images (the condition embedder's first layers) ------------------------
// map_embedder.py:79-113: 3 -> 16 -> 16 -> 32 (stride 2) -> 32 channels on 224x400 .. 112x200 images (1.07 M output
// pixels per layer at 12 view-instances).  As an implicit GEMM these are N = 16 / 32, K = 72 .. 288 problems whose every
// input pixel is gathered 9 times through L2 (90 us, 0.76 TB/s of algorithmic bytes for 16 -> 16).  Here a workgroup
// owns a TH x 64 block of output pixels: the input patch (with its halo) is staged ONCE in LDS (zero-filled outside the
// image), the whole weight matrix lives in registers as MFMA A-fragments (<= 18 x 4 VGPRs), and each wave walks 16-pixel
// row segments: one ds_read_b128 per lane and K-step supplies 8 consecutive (tap, channel) values of its pixel.
// HBM-bound by construction: input read once (+ halo), output written once.
template <typename T, int CIN, int COUT, int STRIDE>
__global__ __launch_bounds__(256)
void dd_conv3x3_thin_kernel(const T* __restrict__ x, const T* __restrict__ w, const T* __restrict__ bias,
                            T* __restrict__ y, int hin, int win, int hout, int wout, int silu) {
  using V8 = typename dd_vec<T>::v8;
  using V4 = typename dd_vec<T>::v4;
  constexpr int TW = 64, TH = STRIDE == 1 ? 8 : 4;           // output pixels per workgroup
  constexpr int PW = (TW - 1) * STRIDE + 3, PH = (TH - 1) * STRIDE + 3;
  constexpr int PIX = CIN * 2 + (CIN > 8 ? 16 : 0);           // bytes per patch pixel: +16 keeps 16 pixels' reads apart
  constexpr int TPK = 32 / CIN;                               // taps per 32-wide K-step
  constexpr int KSTEPS = (9 + TPK - 1) / TPK;
  constexpr int NB = COUT / 16;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int px = lane & 15, kc = lane >> 4;
  const int tiles_x = (wout + TW - 1) / TW;
  const int ty = blockIdx.x / tiles_x, tx = blockIdx.x - ty * tiles_x;
  const int inst = blockIdx.y;
  const int oy0 = ty * TH, ox0 = tx * TW;
  const int iy0 = oy0 * STRIDE - 1, ix0 = ox0 * STRIDE - 1;
  const T* xin = x + (int64_t)inst * hin * win * CIN;

  // ---- stage the patch: 16-B vectors, rows of PW pixels x CIN channels ------------------------------------------------
  constexpr int VPP = CIN / 8;                                // 16-B vectors per pixel
  constexpr int NVEC = PH * PW * VPP;
  for (int v = tid; v < NVEC; v += 256) {
    const int pix = v / VPP, cv = v - pix * VPP;
    const int py = pix / PW, pxx = pix - py * PW;
    const int iy = iy0 + py, ix = ix0 + pxx;
    u32x4 val = {0u, 0u, 0u, 0u};
    if (iy >= 0 && iy < hin && ix >= 0 && ix < win) val = dd_ld16(xin + ((int64_t)iy * win + ix) * CIN + cv * 8);
    *reinterpret_cast<u32x4*>(smem + (size_t)pix * PIX + cv * 16) = val;
  }
  // ---- weights -> registers (A operand: row = output channel px, 8 consecutive k of chunk kc) ----------------------------
  V8 wf[NB][KSTEPS];
#pragma unroll
  for (int nb = 0; nb < NB; ++nb)
#pragma unroll
    for (int ks = 0; ks < KSTEPS; ++ks) {
      const int k = ks * 32 + kc * 8;                         // (tap, channel) index; >= 9 * CIN: padding taps
      u32x4 val = {0u, 0u, 0u, 0u};
      if (k < 9 * CIN) val = dd_ld16(w + (int64_t)(nb * 16 + px) * (9 * CIN) + k);
      wf[nb][ks] = dd_as_v8<T>(val);
    }
  // this lane's patch offset per K-step, relative to its output pixel's top-left tap
  int koff[KSTEPS];
#pragma unroll
  for (int ks = 0; ks < KSTEPS; ++ks) {
    const int k = ks * 32 + kc * 8;
    int tap = k / CIN;
    const int ci = k - tap * CIN;
    if (tap > 8) tap = 0;                                     // zero weights: any resident address will do
    koff[ks] = ((tap / 3) * PW + (tap % 3)) * PIX + ci * 2;
  }
  float bv[NB][4];
#pragma unroll
  for (int nb = 0; nb < NB; ++nb)
#pragma unroll
    for (int r = 0; r < 4; ++r) bv[nb][r] = bias ? (float)bias[nb * 16 + kc * 4 + r] : 0.f;
  __syncthreads();

  constexpr int ROWS_PER_WAVE = TH / 4;
  T* yout = y + (int64_t)inst * hout * wout * COUT;
#pragma unroll
  for (int rr = 0; rr < ROWS_PER_WAVE; ++rr) {
    const int oy = oy0 + wave * ROWS_PER_WAVE + rr;
    if (oy >= hout) continue;                                 // wave-uniform
    const int prow = (wave * ROWS_PER_WAVE + rr) * STRIDE;
#pragma unroll
    for (int g = 0; g < TW / 16; ++g) {
      const int ox = ox0 + g * 16 + px;
      if (ox0 + g * 16 >= wout) continue;                     // wave-uniform
      const unsigned char* base = smem + (size_t)(prow * PW + (g * 16 + px) * STRIDE) * PIX;
      f32x4 acc[NB];
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) acc[nb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < KSTEPS; ++ks) {
        const V8 xf = dd_as_v8<T>(*reinterpret_cast<const u32x4*>(base + koff[ks]));
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) acc[nb] = dd_mfma16(wf[nb][ks], xf, acc[nb]);
      }
      if (ox < wout) {
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
          V4 o;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float v = acc[nb][r] + bv[nb][r];
            if (silu) v = dd_silu_f(v);
            o[r] = (T)v;
          }
          *reinterpret_cast<V4*>(yout + ((int64_t)oy * wout + ox) * COUT + nb * 16 + kc * 4) = o;
        }
      }
    }
  }
}

template <typename T>
__global__ __launch_bounds__(256)
void dd_cfg_ddim_kernel(const T* eps, const T* x, T* x_out, T* x_dup, const float* coef,
                        float guidance, int64_t n) {
  const float sa_t = coef[0], s1a_t = coef[1], sa_p = coef[2], s1a_p = coef[3];
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x) {
    const float eu = (float)eps[i], ec = (float)eps[n + i];
    // reference rounds the guided noise to the model dtype before scheduler.step
    const float e = (float)(T)(eu + guidance * (ec - eu));
    const float xv = (float)x[i];
    const float x0 = (xv - s1a_t * e) / sa_t;
    const T r = (T)(sa_p * x0 + s1a_p * e);
    x_out[i] = r;
    if (x_dup) x_dup[i] = r;
  }
}

template <typename T>
__global__ __launch_bounds__(256)
void dd_cfg_unipc_kernel(const T* eps, const T* x, T* x_out, T* x_dup, float* last, float* m1, float* m2,
                         const float* coef, float guidance, int64_t n) {
  const float a_x = coef[0], a_e = coef[1], use_c = coef[2], c_l = coef[3], c_1 = coef[4], c_2 = coef[5],
              c_0 = coef[6], p_x = coef[7], p_0 = coef[8], p_1 = coef[9];
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x) {
    const float eu = (float)eps[i], ec = (float)eps[n + i];
    const float e = (float)(T)(eu + guidance * (ec - eu));     // guided noise rounded to the model dtype
    const float xv = (float)x[i];
    const float x0 = a_x * xv + a_e * e;
    const float o1 = m1[i];
    float xc = xv;
    if (use_c != 0.f) xc = c_l * last[i] + c_1 * o1 + c_2 * m2[i] + c_0 * x0;
    const T r = (T)(p_x * xc + p_0 * x0 + p_1 * o1);
    last[i] = xc;
    m2[i] = o1;
    m1[i] = x0;
    x_out[i] = r;
    if (x_dup) x_dup[i] = r;
  }
}

inline unsigned grid_for(int64_t work, int per_block = 256, unsigned cap = 2048) {
  int64_t b = (work + per_block - 1) / per_block;
  if (b > cap) b = cap;
  if (b < 1) b = 1;
  return (unsigned)b;
}

}  // namespace

extern "C" int dd_add(const void* a, const void* b, const void* c, void* y, int64_t n,
                      int32_t dtype, dd_stream_t stream) {
  if (!a || !b || !y || n <= 0 || (n & 7)) return DD_ERR_BAD_ARG;
  if (dtype != DD_F16 && dtype != DD_BF16) return DD_ERR_BAD_ARG;
  if (!dd_aligned16(a) || !dd_aligned16(b) || !dd_aligned16(y) || (c && !dd_aligned16(c))) return DD_ERR_BAD_ARG;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  dd_clear_error();
  const int64_t nvec = n / 8;
  if (dtype == DD_F16)
    hipLaunchKernelGGL(dd_add_kernel<_Float16>, dim3(grid_for(nvec)), dim3(256), 0, s,
                       (const _Float16*)a, (const _Float16*)b, (const _Float16*)c, (_Float16*)y, nvec);
  else
    hipLaunchKernelGGL(dd_add_kernel<__bf16>, dim3(grid_for(nvec)), dim3(256), 0, s,
                       (const __bf16*)a, (const __bf16*)b, (const __bf16*)c, (__bf16*)y, nvec);
  return dd_check_launch();
}

extern "C" int dd_scale(const void* x, void* y, float sc, int64_t n, int32_t dtype, dd_stream_t stream) {
  if (!x || !y || n <= 0 || (n & 7)) return DD_ERR_BAD_ARG;
  if (dtype != DD_F16 && dtype != DD_BF16) return DD_ERR_BAD_ARG;
  if (!dd_aligned16(x) || !dd_aligned16(y)) return DD_ERR_BAD_ARG;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  dd_clear_error();
  const int64_t nvec = n / 8;
  if (dtype == DD_F16)
    hipLaunchKernelGGL((dd_unary_kernel<_Float16, 0>), dim3(grid_for(nvec)), dim3(256), 0, s,
                       (const _Float16*)x, (_Float16*)y, sc, nvec);
  else
    hipLaunchKernelGGL((dd_unary_kernel<__bf16, 0>), dim3(grid_for(nvec)), dim3(256), 0, s,
                       (const __bf16*)x, (__bf16*)y, sc, nvec);
  return dd_check_launch();
}

extern "C" int dd_silu(const void* x, void* y, int64_t n, int32_t dtype, dd_stream_t stream) {
  if (!x || !y || n <= 0 || (n & 7)) return DD_ERR_BAD_ARG;
  if (dtype != DD_F16 && dtype != DD_BF16) return DD_ERR_BAD_ARG;
  if (!dd_aligned16(x) || !dd_aligned16(y)) return DD_ERR_BAD_ARG;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  dd_clear_error();
  const int64_t nvec = n / 8;
  if (dtype == DD_F16)
    hipLaunchKernelGGL((dd_unary_kernel<_Float16, 1>), dim3(grid_for(nvec)), dim3(256), 0, s,
                       (const _Float16*)x, (_Float16*)y, 1.f, nvec);
  else
    hipLaunchKernelGGL((dd_unary_kernel<__bf16, 1>), dim3(grid_for(nvec)), dim3(256), 0, s,
                       (const __bf16*)x, (__bf16*)y, 1.f, nvec);
  return dd_check_launch();
}

extern "C" int dd_nchw_to_nhwc(const void* x, void* y, int32_t m, int32_t c, int32_t hw,
                               int32_t c_pad, int32_t dtype, dd_stream_t stream) {
  if (!x || !y || m <= 0 || c <= 0 || hw <= 0 || c_pad < c) return DD_ERR_BAD_ARG;
  if (dtype != DD_F16 && dtype != DD_BF16) return DD_ERR_BAD_ARG;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  dd_clear_error();
  const int64_t total = (int64_t)m * hw * c_pad;
  if (dtype == DD_F16)
    hipLaunchKernelGGL(dd_nchw_to_nhwc_kernel<_Float16>, dim3(grid_for(total)), dim3(256), 0, s,
                       (const _Float16*)x, (_Float16*)y, m, c, hw, c_pad);
  else
    hipLaunchKernelGGL(dd_nchw_to_nhwc_kernel<__bf16>, dim3(grid_for(total)), dim3(256), 0, s,
                       (const __bf16*)x, (__bf16*)y, m, c, hw, c_pad);
  return dd_check_launch();
}

extern "C" int dd_nhwc_to_nchw(const void* x, void* y, int32_t m, int32_t c, int32_t hw,
                               int32_t ldx, int32_t dtype, dd_stream_t stream) {
  if (!x || !y || m <= 0 || c <= 0 || hw <= 0 || ldx < c) return DD_ERR_BAD_ARG;
  if (dtype != DD_F16 && dtype != DD_BF16) return DD_ERR_BAD_ARG;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  dd_clear_error();
  const int64_t total = (int64_t)m * hw * c;
  if (dtype == DD_F16)
    hipLaunchKernelGGL(dd_nhwc_to_nchw_kernel<_Float16>, dim3(grid_for(total)), dim3(256), 0, s,
                       (const _Float16*)x, (_Float16*)y, m, c, hw, ldx);
  else
    hipLaunchKernelGGL(dd_nhwc_to_nchw_kernel<__bf16>, dim3(grid_for(total)), dim3(256), 0, s,
                       (const __bf16*)x, (__bf16*)y, m, c, hw, ldx);
  return dd_check_launch();
}

extern "C" int dd_timestep_embedding(const float* t, void* out, int32_t n, int32_t dim,
                                     int32_t flip_sin_to_cos, float freq_shift,
                                     int32_t dtype, dd_stream_t stream) {
  if (!t || !out || n <= 0 || dim <= 0 || (dim & 1)) return DD_ERR_BAD_ARG;
  if (dtype != DD_F16 && dtype != DD_BF16) return DD_ERR_BAD_ARG;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  dd_clear_error();
  const int total = n * (dim / 2);
  if (dtype == DD_F16)
    hipLaunchKernelGGL(dd_timestep_embedding_kernel<_Float16>, dim3(grid_for(total)), dim3(256), 0, s,
                       t, (_Float16*)out, n, dim, flip_sin_to_cos, freq_shift);
  else
    hipLaunchKernelGGL(dd_timestep_embedding_kernel<__bf16>, dim3(grid_for(total)), dim3(256), 0, s,
                       t, (__bf16*)out, n, dim, flip_sin_to_cos, freq_shift);
  return dd_check_launch();
}

extern "C" int dd_ors_project(const uint8_t* occ, const float* origin, const float* dir, uint8_t* labels,
                              void* cond, int32_t n_cam, int32_t hw, int32_t samples, float step,
                              int32_t keep_fg, int32_t keep_bg, int32_t dtype, dd_stream_t stream) {
  if (!occ || !origin || !dir || (!labels && !cond) || n_cam <= 0 || hw <= 0 || samples <= 0) return DD_ERR_BAD_ARG;
  if (cond && dtype != DD_F16 && dtype != DD_BF16) return DD_ERR_BAD_ARG;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  dd_clear_error();
  const int64_t total = (int64_t)n_cam * samples * hw;
  const unsigned g = (unsigned)grid_for(total > (1 << 30) ? (1 << 30) : (int)total);
  if (dtype == DD_F16)
    hipLaunchKernelGGL(dd_ors_project_kernel<_Float16>, dim3(g), dim3(256), 0, s, occ, origin, dir, labels,
                       (_Float16*)cond, n_cam, hw, samples, step, keep_fg, keep_bg);
  else
    hipLaunchKernelGGL(dd_ors_project_kernel<__bf16>, dim3(g), dim3(256), 0, s, occ, origin, dir, labels,
                       (__bf16*)cond, n_cam, hw, samples, step, keep_fg, keep_bg);
  return dd_check_launch();
}

extern "C" int dd_softmax_rows(const float* s, void* p, int64_t rows, int32_t cols, int64_t lds, int64_t ldp,
                               int32_t dtype, dd_stream_t stream) {
  if (!s || !p || rows <= 0 || cols <= 0 || lds < cols || ldp < cols) return DD_ERR_BAD_ARG;
  if (dtype != DD_F16 && dtype != DD_BF16) return DD_ERR_BAD_ARG;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  dd_clear_error();
  const unsigned g = (unsigned)((rows + 3) / 4);
  if (dtype == DD_F16)
    hipLaunchKernelGGL(dd_softmax_rows_kernel<_Float16>, dim3(g), dim3(256), 0, st, s, (_Float16*)p, rows, cols, lds, ldp);
  else
    hipLaunchKernelGGL(dd_softmax_rows_kernel<__bf16>, dim3(g), dim3(256), 0, st, s, (__bf16*)p, rows, cols, lds, ldp);
  return dd_check_launch();
}

template <typename TI>
int launch_fourier(const void* x, void* out, int64_t total, int dims, const FourierFreqs& fr, int nf, int inc,
                   int out_dtype, hipStream_t s) {
  const unsigned g = (unsigned)grid_for(total > (1 << 30) ? (1 << 30) : (int)total);
  if (out_dtype == DD_F16)
    hipLaunchKernelGGL((dd_fourier_embed_kernel<TI, _Float16>), dim3(g), dim3(256), 0, s, (const TI*)x, (_Float16*)out, total, dims, fr, nf, inc);
  else if (out_dtype == DD_BF16)
    hipLaunchKernelGGL((dd_fourier_embed_kernel<TI, __bf16>), dim3(g), dim3(256), 0, s, (const TI*)x, (__bf16*)out, total, dims, fr, nf, inc);
  else
    hipLaunchKernelGGL((dd_fourier_embed_kernel<TI, float>), dim3(g), dim3(256), 0, s, (const TI*)x, (float*)out, total, dims, fr, nf, inc);
  return dd_check_launch();
}

extern "C" int dd_fourier_embed(const void* x, void* out, int64_t rows, int32_t dims, const float* freqs,
                                int32_t num_freqs, int32_t include_input, int32_t in_dtype, int32_t out_dtype,
                                dd_stream_t stream) {
  if (!x || !out || !freqs || rows <= 0 || dims <= 0) return DD_ERR_BAD_ARG;
  if (num_freqs <= 0 || num_freqs > 16) return DD_ERR_UNSUPPORTED;
  if (in_dtype < 0 || in_dtype > DD_F32 || out_dtype < 0 || out_dtype > DD_F32) return DD_ERR_BAD_ARG;
  FourierFreqs fr{};
  for (int i = 0; i < num_freqs; ++i) fr.f[i] = freqs[i];
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  dd_clear_error();
  const int64_t total = rows * dims;
  const int inc = include_input ? 1 : 0;
  if (in_dtype == DD_F16) return launch_fourier<_Float16>(x, out, total, dims, fr, num_freqs, inc, out_dtype, s);
  if (in_dtype == DD_BF16) return launch_fourier<__bf16>(x, out, total, dims, fr, num_freqs, inc, out_dtype, s);
  return launch_fourier<float>(x, out, total, dims, fr, num_freqs, inc, out_dtype, s);
}

extern "C" int dd_conv3x3_small_cout(const void* x, const void* w, const void* bias, void* y_nchw,
                                     int32_t m, int32_t h, int32_t wd, int32_t cin, int32_t cout,
                                     int32_t dtype, dd_stream_t stream) {
  if (!x || !w || !y_nchw || m <= 0 || h <= 0 || wd <= 0 || cin <= 0 || cout <= 0) return DD_ERR_BAD_ARG;
  if ((cin & 7) || cout > 8) return DD_ERR_UNSUPPORTED;
  if (dtype != DD_F16 && dtype != DD_BF16) return DD_ERR_BAD_ARG;
  if (!dd_aligned16(x) || !dd_aligned16(w)) return DD_ERR_BAD_ARG;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  dd_clear_error();
  const int64_t npix = (int64_t)m * h * wd;
  const unsigned blocks = (unsigned)((npix + 3) / 4);
  if (dtype == DD_F16)
    hipLaunchKernelGGL((dd_conv3x3_small_kernel<_Float16, 8>), dim3(blocks), dim3(256), 0, s,
                       (const _Float16*)x, (const _Float16*)w, (const _Float16*)bias,
                       (_Float16*)y_nchw, m, h, wd, cin, cout);
  else
    hipLaunchKernelGGL((dd_conv3x3_small_kernel<__bf16, 8>), dim3(blocks), dim3(256), 0, s,
                       (const __bf16*)x, (const __bf16*)w, (const __bf16*)bias,
                       (__bf16*)y_nchw, m, h, wd, cin, cout);
  return dd_check_launch();
}

template <typename T, int CIN, int COUT, int STRIDE>
static int launch_thin(const void* x, const void* w, const void* bias, void* y, int m, int hin, int win, int hout, int wout,
                       int silu, hipStream_t s) {
  constexpr int TW = 64, TH = STRIDE == 1 ? 8 : 4;
  constexpr int PW = (TW - 1) * STRIDE + 3, PH = (TH - 1) * STRIDE + 3;
  constexpr int PIX = CIN * 2 + (CIN > 8 ? 16 : 0);
  constexpr size_t smem = (size_t)PH * PW * PIX;
  static_assert(smem <= 64 * 1024, "patch fits the default dynamic LDS limit");
  dim3 grid(((wout + TW - 1) / TW) * ((hout + TH - 1) / TH), m);
  hipLaunchKernelGGL((dd_conv3x3_thin_kernel<T, CIN, COUT, STRIDE>), grid, dim3(256), smem, s, (const T*)x, (const T*)w,
                     (const T*)bias, (T*)y, hin, win, hout, wout, silu);
  return dd_check_launch();
}

template <typename T>
static int launch_thin_t(const void* x, const void* w, const void* bias, void* y, int m, int hin, int win, int hout, int wout,
                         int cin, int cout, int stride, int silu, hipStream_t s) {
#define DD_THIN(CI, CO, ST) if (cin == CI && cout == CO && stride == ST) \
    return launch_thin<T, CI, CO, ST>(x, w, bias, y, m, hin, win, hout, wout, silu, s);
  DD_THIN(8, 16, 1) DD_THIN(16, 16, 1) DD_THIN(16, 32, 2) DD_THIN(32, 32, 1)
  DD_THIN(8, 32, 1) DD_THIN(16, 32, 1) DD_THIN(16, 16, 2) DD_THIN(8, 16, 2) DD_THIN(32, 16, 1)
#undef DD_THIN
  return DD_ERR_UNSUPPORTED;
}

extern "C" int dd_conv3x3_thin(const void* x, const void* w, const void* bias, void* y, int32_t m, int32_t hin, int32_t win,
                               int32_t cin, int32_t cout, int32_t stride, int32_t silu, int32_t dtype, dd_stream_t stream) {
  if (!x || !w || !y || m <= 0 || hin <= 0 || win <= 0) return DD_ERR_BAD_ARG;
  if (dtype != DD_F16 && dtype != DD_BF16) return DD_ERR_BAD_ARG;
  if (!dd_aligned16(x) || !dd_aligned16(w) || !dd_aligned16(y) || (reinterpret_cast<uintptr_t>(bias) & 1u)) return DD_ERR_BAD_ARG;
  if (stride != 1 && stride != 2) return DD_ERR_UNSUPPORTED;
  if (m > 65535) return DD_ERR_UNSUPPORTED;
  const int hout = (hin - 1) / stride + 1, wout = (win - 1) / stride + 1;      // kernel 3, pad 1
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  dd_clear_error();
  if (dtype == DD_F16) return launch_thin_t<_Float16>(x, w, bias, y, m, hin, win, hout, wout, cin, cout, stride, silu, s);
  return launch_thin_t<__bf16>(x, w, bias, y, m, hin, win, hout, wout, cin, cout, stride, silu, s);
}

extern "C" int dd_cfg_ddim_step(const void* eps, const void* x, void* x_out, void* x_dup,
                                const float* coef, float guidance, int64_t n,
                                int32_t dtype, dd_stream_t stream) {
  if (!eps || !x || !x_out || !coef || n <= 0) return DD_ERR_BAD_ARG;
  if (dtype != DD_F16 && dtype != DD_BF16) return DD_ERR_BAD_ARG;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  dd_clear_error();
  if (dtype == DD_F16)
    hipLaunchKernelGGL(dd_cfg_ddim_kernel<_Float16>, dim3(grid_for(n)), dim3(256), 0, s,
                       (const _Float16*)eps, (const _Float16*)x, (_Float16*)x_out, (_Float16*)x_dup,
                       coef, guidance, n);
  else
    hipLaunchKernelGGL(dd_cfg_ddim_kernel<__bf16>, dim3(grid_for(n)), dim3(256), 0, s,
                       (const __bf16*)eps, (const __bf16*)x, (__bf16*)x_out, (__bf16*)x_dup,
                       coef, guidance, n);
  return dd_check_launch();
}

extern "C" int dd_cfg_unipc_step(const void* eps, const void* x, void* x_out, void* x_dup, float* last,
                                 float* m1, float* m2, const float* coef, float guidance, int64_t n,
                                 int32_t dtype, dd_stream_t stream) {
  if (!eps || !x || !x_out || !last || !m1 || !m2 || !coef || n <= 0) return DD_ERR_BAD_ARG;
  if (dtype != DD_F16 && dtype != DD_BF16) return DD_ERR_BAD_ARG;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  dd_clear_error();
  if (dtype == DD_F16)
    hipLaunchKernelGGL(dd_cfg_unipc_kernel<_Float16>, dim3(grid_for(n)), dim3(256), 0, s,
                       (const _Float16*)eps, (const _Float16*)x, (_Float16*)x_out, (_Float16*)x_dup,
                       last, m1, m2, coef, guidance, n);
  else
    hipLaunchKernelGGL(dd_cfg_unipc_kernel<__bf16>, dim3(grid_for(n)), dim3(256), 0, s,
                       (const __bf16*)eps, (const __bf16*)x, (__bf16*)x_out, (__bf16*)x_dup,
                       last, m1, m2, coef, guidance, n);
  return dd_check_launch();
}

// Timing probe (bench.py's KernelTimer calibration): ONE wave that waits `ticks` ticks of the constant 100 MHz
// s_memrealtime counter and records its own first / last reading -> a kernel whose device-side duration is known
// independently of events, rocprofv3 and launch gaps.
namespace {
__global__ __launch_bounds__(64) void dd_probe_spin_kernel(uint64_t* stamps, uint32_t ticks) {
  const uint64_t t0 = __builtin_amdgcn_s_memrealtime();
  uint64_t t1 = t0;
  while (t1 - t0 < ticks) {
    __builtin_amdgcn_s_sleep(4);
    t1 = __builtin_amdgcn_s_memrealtime();
  }
  if (threadIdx.x == 0) { stamps[0] = t0; stamps[1] = t1; }
}
}  // namespace

extern "C" int dd_probe_spin(uint64_t* stamps, uint32_t ticks_100mhz, dd_stream_t stream) {
  if (!stamps || ticks_100mhz == 0 || ticks_100mhz > 100000000u) return DD_ERR_BAD_ARG;
  dd_clear_error();
  hipLaunchKernelGGL(dd_probe_spin_kernel, dim3(1), dim3(64), 0, reinterpret_cast<hipStream_t>(stream), stamps, ticks_100mhz);
  return dd_check_launch();
}

extern "C" int dd_abi_version(void) { return DD_ABI_VERSION; }
extern "C" int64_t dd_desc_size(int which) {
  switch (which) {
    case 0: return (int64_t)sizeof(dd_gemm_desc);
    case 1: return (int64_t)sizeof(dd_attn_desc);
    case 2: return (int64_t)sizeof(dd_xattn_desc);
    case 3: return (int64_t)sizeof(dd_gemm8_desc);
    case 4: return (int64_t)sizeof(dd_box_tokens_desc);
  }
  return -1;
}
extern "C" const char* dd_target_arch(void) { return "gfx950"; }
extern "C" const char* dd_error_string(int code) {
  switch (code) {
    case DD_OK: return "ok";
    case DD_ERR_BAD_ARG: return "bad argument (null / non-positive size / misaligned)";
    case DD_ERR_UNSUPPORTED: return "unsupported shape or option";
    case DD_ERR_LAUNCH: return "kernel launch failed";
    case DD_ERR_WORKSPACE: return "workspace too small";
  }
  return "unknown error";
}
