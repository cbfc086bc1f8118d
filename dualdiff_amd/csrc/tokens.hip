// Token / condition preparation of a ControlNet branch as a handful of launches (round 4).
//
// Reference: unet_addon_rawbox.py:308-361 (camera Fourier features -> cam2token, [cam | text] concat), :832-896 and
// :1007 (box tokens appended), bbox_embedder.py:164-203 (Fourier features of the 8 box corners, class-token gather,
// null-masking, concat in front of the MLP), map_embedder.py:116-125 (panorama split into views).  The reference
// recomputes all of it every denoising step (the bench does too, to stay like-for-like), and as torch ops it was ~40 tiny
// launches per step on the sampler's critical stream (profiles/r03_trace_summary.txt "torch/other": 47 launches,
// 0.25 ms).  Here: one layout kernel per image-like input, one kernel that writes BOTH operands of the box MLP, one
// strided Fourier kernel for the camera parameters, one gather-copy that assembles the context.  HBM / launch bound,
// 16-byte vectors wherever the shape allows.  gfx950 only.
#include "dd_common.h"

namespace {

// ---- NCHW (m, c, h, views * w)  ->  NHWC rows of m * views instances (h, w, c_pad), zero channel padding --------------
// LDS-tiled transpose: a block owns (instance, 64 pixels, CT channels).  Reads run along the pixel axis (contiguous in
// NCHW), writes are 16-byte vectors along the channel axis.  CT = 8 for the thin inputs (latents 4 -> 8, panorama
// 3 -> 8), 64 otherwise (the 320-channel ORS-3D condition: 10.7 MB that torch's generic strided copy moved at 0.3 TB/s).
template <typename T, int CT>
__global__ __launch_bounds__(256)
void dd_nchw_to_nhwc_tiled_kernel(const T* __restrict__ x, T* __restrict__ y, int c, int h, int w, int views, int c_pad) {
  __shared__ T tile[64][CT + 2];
  const int hw = h * w;
  const int inst = blockIdx.z;
  const int b = inst / views, v = inst - b * views;
  const int px0 = blockIdx.x * 64, ch0 = blockIdx.y * CT;
  const int wt = views * w;
  const int t = threadIdx.x;
  {
    const int px = px0 + (t & 63);
    const bool pv = px < hw;
    const int yy = pv ? px / w : 0;
    const int64_t src = (int64_t)yy * wt + (int64_t)v * w + (px - yy * w);      // pixel offset inside one channel plane
    const T* plane = x + (int64_t)b * c * h * wt;
#pragma unroll
    for (int i = 0; i < (CT + 3) / 4; ++i) {
      const int cl = (t >> 6) + 4 * i;
      if (cl < CT) {
        const int ch = ch0 + cl;
        tile[t & 63][cl] = (pv && ch < c) ? plane[(int64_t)ch * h * wt + src] : (T)0.f;
      }
    }
  }
  __syncthreads();
  constexpr int VPR = CT / 8;                      // 16-byte vectors per tile row
  for (int i = t; i < 64 * VPR; i += 256) {
    const int pl = i / VPR, cv = (i - pl * VPR) * 8;
    const int px = px0 + pl;
    if (px < hw && ch0 + cv < c_pad) {
      alignas(16) T o[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = tile[pl][cv + e];
      dd_st16(y + ((int64_t)inst * hw + px) * c_pad + ch0 + cv, *reinterpret_cast<const u32x4*>(o));
    }
  }
}

// ---- Fourier features with a strided source and a padded destination -------------------------------------------------
// Row r = outer * inner + j reads its `dims` inputs at x[outer * s_outer + j * s_inner + d * s_dim] and writes its
// `width` features at out[outer * out_ld + j * width ...]; the columns inner * width .. out_ld - 1 of every outer row
// are zeroed (K padding of the Linear that follows).  inner = 1, s_dim = 1, out_ld = width: the plain embedder.
struct FourierFreqs2 { float f[16]; };

template <typename TI, typename TO>
__global__ __launch_bounds__(256)
void dd_fourier_embed2_kernel(const TI* x, TO* out, int64_t rows, int dims, FourierFreqs2 fr, int nf, int inc,
                              int inner, int64_t s_outer, int64_t s_inner, int64_t s_dim, int64_t out_ld) {
  const int width = dims * (inc + 2 * nf);
  const int64_t total = rows * dims;
  const int pad = (int)(out_ld - (int64_t)inner * width);
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t row = i / dims;
    const int d = (int)(i - row * dims);
    const int64_t outer = row / inner;
    const int j = (int)(row - outer * inner);
    // the value passes through the SOURCE dtype exactly as the reference's torch ops leave it there
    const float v = (float)x[outer * s_outer + (int64_t)j * s_inner + (int64_t)d * s_dim];
    TO* o = out + outer * out_ld + (int64_t)j * width + d;
    if (inc) { *o = (TO)(float)(TI)v; o += dims; }
    for (int k = 0; k < nf; ++k) {
      const float a = v * fr.f[k];
      o[0] = (TO)(float)(TI)sinf(a);
      o[dims] = (TO)(float)(TI)cosf(a);
      o += 2 * dims;
    }
    if (pad > 0 && j == inner - 1 && d < pad) out[outer * out_ld + (int64_t)inner * width + d] = (TO)0.f;
  }
}

// ---- both operands of the box MLP in one launch (bbox_embedder.py:164-203) -------------------------------------------
// Per box r (rows = scenes x views x boxes): pos[r] = mask ? Fourier(corners[r]) : null_pos   (T [rows][fdim], the input
// of bbox_proj) and cat[r][cls_off ..] = mask ? class_tokens[classes[r]] : null_class          (the right half of the
// concat that second_linear reads; bbox_proj's SiLU epilogue writes the left half), optionally also cls_out[r].
// `pos * m + null * (1 - m)` with m in {0, 1} IS a select for finite operands, so this is bit-identical to the torch
// chain.  Optional min-max normalisation of the corners (XYZ_MIN / XYZ_RANGE) in the coordinates' own dtype.
template <typename TI, typename T>
__global__ __launch_bounds__(256)
void dd_box_tokens_kernel(const TI* pts, const int64_t* classes, const uint8_t* masks, const T* class_tokens,
                          const T* null_pos, const T* null_cls, T* pos, T* cat, T* cls_out, int rows, int npts,
                          FourierFreqs2 fr, int nf, int inc, int ctd, int64_t ld_cat, int cls_off, int normalize,
                          float mn0, float mn1, float mn2, float rg0, float rg1, float rg2, int n_classes) {
  const int r = blockIdx.x;
  if (r >= rows) return;
  const bool keep = masks ? masks[r] != 0 : true;
  const int width = 3 * (inc + 2 * nf);
  const int fdim = npts * width;
  // Fourier part: one thread per (point, coordinate)
  for (int i = threadIdx.x; i < npts * 3; i += blockDim.x) {
    const int pt = i / 3, d = i - pt * 3;
    T* o = pos + (int64_t)r * fdim + pt * width + d;
    if (keep) {
      float v = (float)pts[(int64_t)r * npts * 3 + i];
      if (normalize) {
        const float mn = d == 0 ? mn0 : (d == 1 ? mn1 : mn2), rg = d == 0 ? rg0 : (d == 1 ? rg1 : rg2);
        // the constants are tensors of the coordinates' dtype in the reference (torch.as_tensor(XYZ_RANGE, dtype=pts.dtype):
        // 650 is not a bf16 number), every intermediate is rounded to it
        v = (float)(TI)((float)(TI)(v - (float)(TI)mn) / (float)(TI)rg);
      }
      if (inc) { *o = (T)(float)(TI)v; o += 3; }
      for (int k = 0; k < nf; ++k) {
        const float a = v * fr.f[k];
        o[0] = (T)(float)(TI)sinf(a);
        o[3] = (T)(float)(TI)cosf(a);
        o += 6;
      }
    } else {
      const T* np_ = null_pos + pt * width + d;
      if (inc) { *o = *np_; o += 3; np_ += 3; }
      for (int k = 0; k < nf; ++k) { o[0] = np_[0]; o[3] = np_[3]; o += 6; np_ += 6; }
    }
  }
  // class-token part: 16-byte vectors
  // class index as torch indexing takes it: negative wraps once; what is still out of range (a padding value on a kept
  // row) must not read foreign memory: its tokens become NaN (torch would raise a device assert)
  int64_t ci = keep ? classes[r] : 0;
  if (n_classes > 0 && ci < 0) ci += n_classes;
  const bool bad = keep && n_classes > 0 && (ci < 0 || ci >= n_classes);
  const T* src = keep && !bad ? class_tokens + ci * ctd : null_cls;
  for (int i = threadIdx.x; i < ctd / 8; i += blockDim.x) {
    const u32x4 v = bad ? u32x4{0x7fff7fffu, 0x7fff7fffu, 0x7fff7fffu, 0x7fff7fffu} : dd_ld16(src + i * 8);
    dd_st16(cat + (int64_t)r * ld_cat + cls_off + i * 8, v);
    if (cls_out) dd_st16(cls_out + (int64_t)r * ctd + i * 8, v);
  }
}

// ---- context assembly: full[i] = [cam_i | text | box tokens], txt[i] = text (unet_addon_rawbox.py:337-361, :1007, :977) -
// instance i = (scene s, view v).  cam: [m][dim]; text: [scenes][lt][dim] (or [m][lt][dim] with text_per_view);
// box: [scenes * box_views][nbox][dim] with box_views in {n_cam, 1}; full: [m][1 + lt + nbox][dim]; txt: [m][lt][dim].
template <typename T>
__global__ __launch_bounds__(256)
void dd_ctx_assemble_kernel(const T* cam, const T* text, const T* box, T* full, T* txt, int m, int n_cam, int lt, int nbox,
                            int dim, int text_per_view, int box_views) {
  const int vpr = dim / 8;
  const int lc = 1 + lt + nbox;
  const int64_t nfull = (int64_t)m * lc * vpr;
  const int64_t total = nfull + (txt ? (int64_t)m * lt * vpr : 0);
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const bool to_txt = i >= nfull;
    const int64_t k = to_txt ? i - nfull : i;
    const int cv = (int)(k % vpr);
    const int64_t row = k / vpr;
    const int tokens = to_txt ? lt : lc;
    const int inst = (int)(row / tokens);
    const int tok = (int)(row - (int64_t)inst * tokens) + (to_txt ? 1 : 0);      // position inside `full`
    const int s = inst / n_cam, v = inst - s * n_cam;
    const T* src;
    if (tok == 0) src = cam + (int64_t)inst * dim;
    else if (tok <= lt) src = text + ((int64_t)(text_per_view ? inst : s) * lt + (tok - 1)) * dim;
    else src = box + (((int64_t)s * box_views + (box_views == 1 ? 0 : v)) * nbox + (tok - 1 - lt)) * dim;
    T* dst = to_txt ? txt + row * dim : full + row * dim;
    dd_st16(dst + cv * 8, dd_ld16(src + cv * 8));
  }
}

inline unsigned grid_for(int64_t n, int threads = 256) {
  int64_t b = (n + threads - 1) / threads;
  return (unsigned)(b < 1 ? 1 : (b > 65535 * 4 ? 65535 * 4 : b));
}

template <typename TI>
int launch_fourier2(const void* x, void* out, int64_t rows, int dims, const FourierFreqs2& fr, int nf, int inc, int inner,
                    int64_t so, int64_t si, int64_t sd, int64_t out_ld, int out_dtype, hipStream_t s) {
  const unsigned g = grid_for(rows * dims);
  if (out_dtype == DD_F16)
    hipLaunchKernelGGL((dd_fourier_embed2_kernel<TI, _Float16>), dim3(g), dim3(256), 0, s, (const TI*)x, (_Float16*)out, rows, dims, fr, nf, inc, inner, so, si, sd, out_ld);
  else if (out_dtype == DD_BF16)
    hipLaunchKernelGGL((dd_fourier_embed2_kernel<TI, __bf16>), dim3(g), dim3(256), 0, s, (const TI*)x, (__bf16*)out, rows, dims, fr, nf, inc, inner, so, si, sd, out_ld);
  else
    hipLaunchKernelGGL((dd_fourier_embed2_kernel<TI, float>), dim3(g), dim3(256), 0, s, (const TI*)x, (float*)out, rows, dims, fr, nf, inc, inner, so, si, sd, out_ld);
  return dd_check_launch();
}

template <typename TI, typename T>
int launch_box(const dd_box_tokens_desc* d, const FourierFreqs2& fr, hipStream_t s) {
  hipLaunchKernelGGL((dd_box_tokens_kernel<TI, T>), dim3(d->rows), dim3(256), 0, s, (const TI*)d->points, d->classes,
                     d->masks, (const T*)d->class_tokens, (const T*)d->null_pos, (const T*)d->null_class, (T*)d->pos,
                     (T*)d->cat, (T*)d->cls_out, d->rows, d->points_per_box, fr, d->num_freqs, d->include_input ? 1 : 0,
                     d->class_token_dim, d->ld_cat, d->cls_offset, d->normalize ? 1 : 0, d->xyz_min[0], d->xyz_min[1],
                     d->xyz_min[2], d->xyz_range[0], d->xyz_range[1], d->xyz_range[2], d->n_classes);
  return dd_check_launch();
}

}  // namespace

extern "C" int dd_nchw_to_nhwc_views(const void* x, void* y, int32_t m, int32_t c, int32_t h, int32_t w, int32_t views,
                                     int32_t c_pad, int32_t dtype, dd_stream_t stream) {
  if (!x || !y || m <= 0 || c <= 0 || h <= 0 || w <= 0 || views <= 0 || c_pad < c || (c_pad & 7)) return DD_ERR_BAD_ARG;
  if (dtype != DD_F16 && dtype != DD_BF16) return DD_ERR_BAD_ARG;
  if (!dd_aligned16(y) || (int64_t)m * views > 65535) return DD_ERR_BAD_ARG;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  dd_clear_error();
  const int hw = h * w;
  if (c_pad <= 8) {
    const dim3 g((hw + 63) / 64, 1, m * views);
    if (dtype == DD_F16) hipLaunchKernelGGL((dd_nchw_to_nhwc_tiled_kernel<_Float16, 8>), g, dim3(256), 0, s, (const _Float16*)x, (_Float16*)y, c, h, w, views, c_pad);
    else hipLaunchKernelGGL((dd_nchw_to_nhwc_tiled_kernel<__bf16, 8>), g, dim3(256), 0, s, (const __bf16*)x, (__bf16*)y, c, h, w, views, c_pad);
  } else {
    const dim3 g((hw + 63) / 64, (c_pad + 63) / 64, m * views);
    if (dtype == DD_F16) hipLaunchKernelGGL((dd_nchw_to_nhwc_tiled_kernel<_Float16, 64>), g, dim3(256), 0, s, (const _Float16*)x, (_Float16*)y, c, h, w, views, c_pad);
    else hipLaunchKernelGGL((dd_nchw_to_nhwc_tiled_kernel<__bf16, 64>), g, dim3(256), 0, s, (const __bf16*)x, (__bf16*)y, c, h, w, views, c_pad);
  }
  return dd_check_launch();
}

extern "C" int dd_fourier_embed_strided(const void* x, void* out, int64_t rows, int32_t dims, const float* freqs,
                                        int32_t num_freqs, int32_t include_input, int32_t in_dtype, int32_t out_dtype,
                                        int32_t inner, int64_t stride_outer, int64_t stride_inner, int64_t stride_dim,
                                        int64_t out_ld, dd_stream_t stream) {
  if (!x || !out || !freqs || rows <= 0 || dims <= 0 || inner <= 0 || rows % inner) return DD_ERR_BAD_ARG;
  if (num_freqs <= 0 || num_freqs > 16) return DD_ERR_UNSUPPORTED;
  if (in_dtype < 0 || in_dtype > DD_F32 || out_dtype < 0 || out_dtype > DD_F32) return DD_ERR_BAD_ARG;
  const int inc = include_input ? 1 : 0;
  const int64_t width = (int64_t)dims * (inc + 2 * num_freqs);
  if (out_ld < inner * width || out_ld - inner * width > dims) return DD_ERR_BAD_ARG;     // at most `dims` pad columns
  FourierFreqs2 fr{};
  for (int i = 0; i < num_freqs; ++i) fr.f[i] = freqs[i];
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  dd_clear_error();
  if (in_dtype == DD_F16) return launch_fourier2<_Float16>(x, out, rows, dims, fr, num_freqs, inc, inner, stride_outer, stride_inner, stride_dim, out_ld, out_dtype, s);
  if (in_dtype == DD_BF16) return launch_fourier2<__bf16>(x, out, rows, dims, fr, num_freqs, inc, inner, stride_outer, stride_inner, stride_dim, out_ld, out_dtype, s);
  return launch_fourier2<float>(x, out, rows, dims, fr, num_freqs, inc, inner, stride_outer, stride_inner, stride_dim, out_ld, out_dtype, s);
}

extern "C" int dd_box_tokens(const dd_box_tokens_desc* d, dd_stream_t stream) {
  if (!d || !d->points || !d->classes || !d->class_tokens || !d->null_pos || !d->null_class || !d->pos || !d->cat)
    return DD_ERR_BAD_ARG;
  if (d->rows <= 0 || d->points_per_box <= 0 || d->num_freqs <= 0 || d->class_token_dim <= 0 || d->n_classes < 0) return DD_ERR_BAD_ARG;
  if (d->num_freqs > 16 || (d->class_token_dim & 7) || (d->cls_offset & 7) || (d->ld_cat & 7)) return DD_ERR_UNSUPPORTED;
  if (d->ld_cat < d->cls_offset + d->class_token_dim) return DD_ERR_BAD_ARG;
  if (d->dtype != DD_F16 && d->dtype != DD_BF16) return DD_ERR_BAD_ARG;
  if (d->points_dtype < 0 || d->points_dtype > DD_F32) return DD_ERR_BAD_ARG;
  if (!dd_aligned16(d->class_tokens) || !dd_aligned16(d->null_class) || !dd_aligned16(d->cat) ||
      (d->cls_out && !dd_aligned16(d->cls_out)))
    return DD_ERR_BAD_ARG;
  FourierFreqs2 fr{};
  for (int i = 0; i < d->num_freqs; ++i) fr.f[i] = d->freqs[i];
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  dd_clear_error();
  const bool h = d->dtype == DD_F16;
  switch (d->points_dtype) {
    case DD_F16: return h ? launch_box<_Float16, _Float16>(d, fr, s) : launch_box<_Float16, __bf16>(d, fr, s);
    case DD_BF16: return h ? launch_box<__bf16, _Float16>(d, fr, s) : launch_box<__bf16, __bf16>(d, fr, s);
    default: return h ? launch_box<float, _Float16>(d, fr, s) : launch_box<float, __bf16>(d, fr, s);
  }
}

extern "C" int dd_ctx_assemble(const void* cam, const void* text, const void* box, void* full, void* txt, int32_t m,
                               int32_t n_cam, int32_t lt, int32_t nbox, int32_t dim, int32_t text_per_view,
                               int32_t box_views, int32_t dtype, dd_stream_t stream) {
  if (!cam || !text || !full || m <= 0 || n_cam <= 0 || lt <= 0 || nbox < 0 || dim <= 0 || (dim & 7) || m % n_cam)
    return DD_ERR_BAD_ARG;
  if (nbox > 0 && (!box || (box_views != 1 && box_views != n_cam))) return DD_ERR_BAD_ARG;
  if (dtype != DD_F16 && dtype != DD_BF16) return DD_ERR_BAD_ARG;
  if (!dd_aligned16(cam) || !dd_aligned16(text) || !dd_aligned16(full) || (box && !dd_aligned16(box)) ||
      (txt && !dd_aligned16(txt)))
    return DD_ERR_BAD_ARG;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  dd_clear_error();
  const int64_t total = (int64_t)m * ((1 + lt + nbox) + (txt ? lt : 0)) * (dim / 8);
  if (dtype == DD_F16)
    hipLaunchKernelGGL(dd_ctx_assemble_kernel<_Float16>, dim3(grid_for(total)), dim3(256), 0, s, (const _Float16*)cam,
                       (const _Float16*)text, (const _Float16*)box, (_Float16*)full, (_Float16*)txt, m, n_cam, lt, nbox,
                       dim, text_per_view ? 1 : 0, box_views);
  else
    hipLaunchKernelGGL(dd_ctx_assemble_kernel<__bf16>, dim3(grid_for(total)), dim3(256), 0, s, (const __bf16*)cam,
                       (const __bf16*)text, (const __bf16*)box, (__bf16*)full, (__bf16*)txt, m, n_cam, lt, nbox, dim,
                       text_per_view ? 1 : 0, box_views);
  return dd_check_launch();
}
