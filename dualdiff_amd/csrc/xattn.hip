// Fused cross-attention for the 320-channel level (gfx950): ONE launch for
//
//     out = ( softmax( scale * (x Wq^T) K^T ) V ) Wo^T + bo + res            [ + LayerNorm(out) ]
//
// with <= 128 context keys per view-instance, 8 heads x 40.  Replaces, per call, the q-projection GEMM, the flash
// attention launch and the out-projection GEMM (+ residual) of
//   * Semantic Fusion Attention, txt_con_XFormersAttn (networks/txt_con_fusion.py:74-181): x = res = the 320-channel ORS
//     condition map (28x50 tokens), K / V = projected text tokens (77 keys);
//   * the text / box cross-attention attn2 of the 28x50 transformer blocks (diffusers BasicTransformerBlock.attn2 via
//     networks/box_adapter.py:102-163 processors; blocks.py:166-187): x = LayerNorm2(h) (emitted by the producer's
//     epilogue), res = h, K / V = column slices of the cross-attention K/V bank (78 + N_box keys).
// Both are HBM-bound passes over 10.75 MB tensors (profiles/r02_sfa_roofline.txt: 7 tensor passes per module, each
// launch 0.33-0.42 of HBM); fused, q and the attention output never leave the CU: 2 passes (x in, out out) + res.
//
// One workgroup = 80 query rows of one view-instance (1400 = 17.5 x 80: 18 workgroups per instance, 216 for the 12
// instances of a scene = one generation on 256 CUs), 10 waves:
//   phase 1  Q = x Wq^T                  80 x 320 x 320, wave = all 80 rows x 32 output channels (5 x 2 MFMA blocks);
//            x sits in LDS for the whole phase, Wq streams through a 2-slot LDS ring in 64-wide K steps;
//            Q (x scale * log2 e, rounded to the storage type like the separate path's head-major planes) overwrites x.
//   phase 2  per head: S^T = K Q^T, softmax over the <= 128 keys in registers (one pass: all keys are resident),
//            O^T = V^T P^T with V^T through ds_read_b64_tr_b16.  Wave = (16-row block, head parity); 4 rounds of two
//            heads; K / V of a round are staged in the (idle) weight ring, the next round's are prefetched in registers.
//            O overwrites Q in place (a wave touches only its own rows and its own head's columns).
//   phase 3  out = O Wo^T + bo + res     as phase 1, epilogue straight to global (8-byte stores).
//            Optional: LayerNorm(out) (two-pass fp32 statistics over the ROUNDED values, as dd_layernorm) as a second
//            output — the whole row lives in this workgroup.
#include "dd_common.h"
#include <type_traits>

namespace {

constexpr int XC = 320;          // channels
constexpr int XH = 8;            // heads
constexpr int XD = 40;           // head dim
constexpr int XR = 80;           // query rows per workgroup
constexpr int XW = 10;           // waves
constexpr int XPITCH = XC + 8;   // LDS row pitch of the activation tile (656 B: conflict-free 16-B fragment reads)
constexpr int XLK = 128;         // max context keys
constexpr size_t kActBytes = 53 * 1024;   // 80 rows x 41 chunk slots (52 whole 1 KiB DMA instructions) + 1 KiB that
                                          // absorbs the padding instructions (they must not touch the tile)

struct XAttnParams {
  const void* x; int64_t ldx;            // [rows][320]: operand of the q projection
  const void* res; int64_t ldres;        // [rows][320] or NULL
  const void* wq; const void* wo;        // PACKED [10 K steps][320 rows][4 chunks of 8] (dd_xattn_pack_weight order)
  const void* bo;                        // [320] or NULL
  const void* k; const void* v;          // context K / V: element (instance i, key j, head h, d) at
  int64_t ldk, ldv;                      //   base + i * k_is + j * ldk + h * k_hs + d   (elements)
  int64_t k_is, k_hs, v_is, v_hs;
  void* out; int64_t ldo;
  void* ln_out; int64_t ld_ln; const void* ln_g; const void* ln_b; float ln_eps;   // optional second output
  int inst, n, lk;                       // view-instances, query rows per instance, keys per instance
  const int32_t* lk_dev;                 // keys per instance from device memory (lk = the capacity), or NULL
  int tiles;                             // ceil(n / 80)
  float qscale;                          // softmax scale * log2(e)
  int dbg;                               // DD_XATTN_DBG (diagnostics): 1 skip phase 2, 2 skip the MFMAs of the products, 4 skip softmax
};

template <int N>
__device__ __forceinline__ void xwait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// buffer_load_dwordx4 ... offen lds: 64 lanes x 16 B land LANE-LINEARLY at `lds_wave_base` (1 KiB per wave instruction);
// the per-lane byte offset picks the source, an offset past the descriptor's range reads zeros.
__device__ __forceinline__ void xdma16(__amdgpu_buffer_rsrc_t rsrc, uint32_t voff, uint32_t soff, void* lds_wave_base) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)lds_wave_base, 16,
                                           (int)voff, (int)soff, 0, 0);
}
constexpr uint32_t X_OOB = 0x80000000u;

template <typename T>
__global__ __launch_bounds__(64 * XW)
void dd_xattn320_kernel(const XAttnParams p) {
  using V8 = typename dd_vec<T>::v8;
  using V4 = typename dd_vec<T>::v4;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  // One workgroup per CU runs its phases in lock step, so every global access that is not issued well ahead of its use
  // is a fully exposed L2 / HBM round trip (the first, register-staged version of this kernel took 37 us per workgroup
  // against ~5 us of MFMA; prefetching through registers spilled: 168 VGPRs at 2.5 waves per SIMD).  Hence EVERY operand
  // travels global -> LDS by DMA (no staging registers), issued one phase / three K steps ahead, with counted vmcnt waits:
  //   act   [80][41 chunks]  x tile, then Q, then the attention output (pitch 656 B: chunk 40 of a row is padding)
  //   ring  80 KiB           4 weight slots [320][32] (XOR-swizzled on the SOURCE side, the DMA destination is linear)
  //                          = 2 K/V buffers, each K [2 heads][128 keys][40] + V likewise (dense rows of 80 B: the 16 lanes
  //                          of a fragment read hit 16 x 4 distinct banks; d = 40..63 of the second QK^T step is supplied as
  //                          zeros by the register side, so no pad columns)
  T* act = reinterpret_cast<T*>(smem);
  T* ring = reinterpret_cast<T*>(smem + kActBytes);
  constexpr int SLOT = XC * 32;                                // elements per weight slot
  constexpr int KVBUF = 2 * XLK * XD * 2;                      // elements per K/V buffer (K then V)
  float* red = reinterpret_cast<float*>(ring);                 // LayerNorm partials (phase 3 epilogue): [80][XW]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c = lane & 15;
  const int g = lane >> 4;
  // keys per instance: a kernel argument, or (lk_dev) a word of device memory read here (dd_xattn_desc.lk_dev): p.lk is
  // then the CAPACITY the instance strides were built for — one captured launch serves every context length up to it
  const int lk = p.lk_dev ? min(max(__builtin_amdgcn_readfirstlane(*p.lk_dev), 1), p.lk) : p.lk;

  const int inst = blockIdx.x / p.tiles;
  const int t = blockIdx.x - inst * p.tiles;
  const int r0 = t * XR;                                       // first query row of this tile inside the instance
  const int nrows = min(XR, p.n - r0);
  const int64_t grow0 = (int64_t)inst * p.n + r0;

  const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<T*>(reinterpret_cast<const T*>(p.x) + grow0 * p.ldx), 0,
      (uint32_t)(((int64_t)(nrows - 1) * p.ldx + XC) * sizeof(T)), 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_wq = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.wq), 0, XC * XC * sizeof(T), 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_wo = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.wo), 0, XC * XC * sizeof(T), 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_k = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<T*>(reinterpret_cast<const T*>(p.k) + (int64_t)inst * p.k_is), 0,
      (uint32_t)(((int64_t)(lk - 1) * p.ldk + (XH - 1) * p.k_hs + XD) * sizeof(T)), 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_v = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<T*>(reinterpret_cast<const T*>(p.v) + (int64_t)inst * p.v_is), 0,
      (uint32_t)(((int64_t)(lk - 1) * p.ldv + (XH - 1) * p.v_hs + XD) * sizeof(T)), 0x00020000);

  // ---- DMA issue helpers (every wave issues the same NUMBER of instructions per call: the counted waits rely on it) ----
  // x tile: 80 rows x 41 chunk slots = 3280 slots -> 52 wave instructions (6 per wave for waves 0-1, 5 for the rest:
  // padded to 6 with out-of-range instructions into the tile's own tail, so that every wave counts 6)
  auto issue_x = [&]() {
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      const int wi = i * XW + wave;                            // wave-instruction index 0..59
      const int q = wi * 64 + lane;                            // chunk slot
      const int row = q / 41, ch = q - row * 41;
      uint32_t voff = (uint32_t)(row * (int)p.ldx * 2 + ch * 16);
      if (ch == 40 || row >= nrows) voff = X_OOB;
      if (wi < 52) xdma16(rs_x, voff, 0, reinterpret_cast<unsigned char*>(act) + wi * 1024);
      else xdma16(rs_x, X_OOB, 0, reinterpret_cast<unsigned char*>(act) + 52 * 1024);   // zeros into the spare KiB
    }
  };
  // weight K step `ks` of the PACKED matrix (one contiguous 20 KiB slab per step, rows already XOR-swizzled: chunk
  // position q % 4 of row n = q / 4 holds logical chunk pos ^ ((n >> 2) & 3)) into ring slot `slot`: a linear copy,
  // 20 wave instructions of 1 KiB, 2 per wave.  (Fetching 64-B pieces of 640-B rows instead — the unpacked torch
  // layout — made every wave instruction 16 half-line requests.)
  auto issue_w = [&](const __amdgpu_buffer_rsrc_t& rs, int ks, int slot) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int wi = i * XW + wave;
      xdma16(rs, (uint32_t)(wi * 1024 + lane * 16), (uint32_t)(ks * (XC * 64)),
             reinterpret_cast<unsigned char*>(ring + slot * SLOT) + wi * 1024);
    }
  };
  // K and V of heads 2*round, 2*round + 1 into K/V buffer `buf`: 2 x 1280 chunks = 40 wave instructions, 4 per wave
  auto issue_kv = [&](int round, int buf) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int wi = i * XW + wave;
      const int q = wi * 64 + lane;                            // (head-in-round, key, chunk) dense
      const int hh = q / (XLK * 5), rem = q - hh * (XLK * 5);
      const int key = rem / 5, ch = rem - key * 5;
      const int head = round * 2 + hh;
      unsigned char* kdst = reinterpret_cast<unsigned char*>(ring + buf * KVBUF) + wi * 1024;
      const bool in = key < lk;                              // keys past lk: zeros (another instance's rows follow)
      xdma16(rs_k, in ? (uint32_t)((key * (int)p.ldk + head * (int)p.k_hs + ch * 8) * 2) : X_OOB, 0, kdst);
      xdma16(rs_v, in ? (uint32_t)((key * (int)p.ldv + head * (int)p.v_hs + ch * 8) * 2) : X_OOB, 0, kdst + XLK * XD * 2 * 2);
    }
  };

  // 80 x 320 x 320 product of the LDS-resident activation tile with a streamed weight matrix, 10 K steps of 32 through
  // the 4 ring slots (step s in slot (s + base) & 3), three steps in flight.  The caller has issued steps 0 .. 2 (and
  // nothing younger); tail_fn = whatever the caller wants in flight under the last step (issued after its barrier, when
  // the slots of steps 6 .. 8 are free).
  // acc[tn][tm]: lane (c, g) holds out[row = tm*16 + c][channel = wave*32 + tn*16 + 4*g + r], r = 0..3
  f32x4 acc[2][5];
  auto gemm320 = [&](const __amdgpu_buffer_rsrc_t& rs, const int base, auto tail_fn) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 5; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < 10; ++ks) {
      // Counted waits are only valid over instructions that all go to memory: an instruction whose lanes are all out
      // of range retires at once, OUT OF ORDER, and a count that includes it can be reached while an older real load is
      // still in flight (found the hard way: K / V prefetch and padding instructions issued at step 8 let the step-9
      // wait pass early under load — wrong bits in 2 of 3 concurrent launches).  So the tail is issued at step 9,
      // behind a full drain, and nothing but real weight slabs is ever counted.
      if (ks <= 7) xwait_vmcnt<4>();                           // steps ks + 1, ks + 2 may still fly (2 instructions each)
      else if (ks == 8) xwait_vmcnt<2>();
      else xwait_vmcnt<0>();
      __builtin_amdgcn_s_barrier();                            // everyone's share of step ks landed, step ks - 1 consumed
      asm volatile("" ::: "memory");
      if (ks + 3 < 10) issue_w(rs, ks + 3, (ks + 3 + base) & 3);
      if (ks == 9) tail_fn();
      const T* slot = ring + ((ks + base) & 3) * SLOT;
      V8 wf[2], xf[5];
#pragma unroll
      for (int tn = 0; tn < 2; ++tn) {
        const int n = wave * 32 + tn * 16 + c;
        wf[tn] = dd_as_v8<T>(dd_ld16(slot + n * 32 + ((g ^ ((n >> 2) & 3)) << 3)));
      }
#pragma unroll
      for (int tm = 0; tm < 5; ++tm)
        xf[tm] = dd_as_v8<T>(dd_ld16(act + (tm * 16 + c) * XPITCH + ks * 32 + g * 8));
      if (!(p.dbg & 2)) {
#pragma unroll
      for (int tn = 0; tn < 2; ++tn)
#pragma unroll
        for (int tm = 0; tm < 5; ++tm) acc[tn][tm] = dd_mfma16(wf[tn], xf[tm], acc[tn][tm]);
      }
    }
  };

  // ---- phase 0 / 1: x tile and the first three Wq steps in flight together; Q = x Wq^T -----------------------------
  issue_x();
#pragma unroll
  for (int i = 0; i < 3; ++i) issue_w(rs_wq, i, i);
  // under step 9 (slots 2, 3 are free by then) the K / V of heads 0, 1 start travelling into buffer 1
  gemm320(rs_wq, 0, [&]() { issue_kv(0, 1); });
  __builtin_amdgcn_s_barrier();                                // every wave is done reading x: Q may overwrite it
  asm volatile("" ::: "memory");
#pragma unroll
  for (int tn = 0; tn < 2; ++tn)
#pragma unroll
    for (int tm = 0; tm < 5; ++tm) {
      V4 q4;
#pragma unroll
      for (int r = 0; r < 4; ++r) q4[r] = (T)(acc[tn][tm][r] * p.qscale);
      *reinterpret_cast<V4*>(act + (tm * 16 + c) * XPITCH + wave * 32 + tn * 16 + g * 4) = q4;
    }

  // ---- phase 2: attention, two heads per round (round r in buffer (r + 1) & 1) ------------------------------------
  const int rb = wave % 5, hp = wave / 5;
  const int nkb = (lk + 15) >> 4;                            // 16-key blocks (<= 8)
  const T* resb = p.res ? reinterpret_cast<const T*>(p.res) + grow0 * p.ldres : nullptr;
  V4 rv[2][5], bv[2];                                          // residual and bias: loaded under the last round
  V4 lng[2], lnb4[2];                                          // LayerNorm gamma / beta of this lane's channels (ln_out), likewise
#pragma unroll 1
  for (int round = 0; round < XH / 2; ++round) {
    xwait_vmcnt<0>();                  // this round's K / V landed (nothing younger is in flight)
    __syncthreads();                   // ... everyone's share; the previous round is consumed; round 0: Q is complete
    if (round + 1 < XH / 2) {
      issue_kv(round + 1, round & 1);
    } else {                           // last round: the residual and the out-projection's first two weight steps
#pragma unroll
      for (int tn = 0; tn < 2; ++tn)
#pragma unroll
        for (int tm = 0; tm < 5; ++tm) {
          const int row = min(tm * 16 + c, nrows - 1);
          V4 z;
#pragma unroll
          for (int r = 0; r < 4; ++r) z[r] = (T)0.f;
          rv[tn][tm] = resb ? *reinterpret_cast<const V4*>(resb + (int64_t)row * p.ldres + wave * 32 + tn * 16 + g * 4) : z;
        }
#pragma unroll
      for (int tn = 0; tn < 2; ++tn)
        bv[tn] = *reinterpret_cast<const V4*>(reinterpret_cast<const T*>(p.bo) + wave * 32 + tn * 16 + g * 4);
      if (p.ln_out) {                  // (round 6) not behind the epilogue's last barrier, where the round trip was exposed
#pragma unroll
        for (int tn = 0; tn < 2; ++tn) {
          lng[tn] = *reinterpret_cast<const V4*>(reinterpret_cast<const T*>(p.ln_g) + wave * 32 + tn * 16 + g * 4);
          lnb4[tn] = *reinterpret_cast<const V4*>(reinterpret_cast<const T*>(p.ln_b) + wave * 32 + tn * 16 + g * 4);
        }
      }
      issue_w(rs_wo, 0, 2);            // slots 2, 3 = buffer 1: consumed by round 2
      issue_w(rs_wo, 1, 3);
    }
    if (p.dbg & 1) continue;
    const int head = round * 2 + hp;
    const T* Ks = ring + ((round + 1) & 1) * KVBUF + hp * XLK * XD;
    const T* Vs = ring + ((round + 1) & 1) * KVBUF + 2 * XLK * XD + hp * XLK * XD;
    // Q fragments of this wave's 16 rows: B operand of S^T = K Q^T; d 40..63 of the second step is zero on both sides
    V8 qf[2];
    const u32x4 zero4 = {0u, 0u, 0u, 0u};
    {
      const T* qrow = act + (rb * 16 + c) * XPITCH + head * XD;
      qf[0] = dd_as_v8<T>(dd_ld16(qrow + g * 8));
      qf[1] = dd_as_v8<T>(g == 0 ? dd_ld16(qrow + 32) : zero4);
    }
    f32x4 s[8];
#pragma unroll
    for (int b = 0; b < 8; ++b) {
      s[b] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (b < nkb) {
        const T* krow = Ks + (b * 16 + c) * XD;
        const V8 k0 = dd_as_v8<T>(dd_ld16(krow + g * 8));
        const V8 k1 = dd_as_v8<T>(g == 0 ? dd_ld16(krow + 32) : zero4);
        s[b] = dd_mfma16(k0, qf[0], s[b]);
        s[b] = dd_mfma16(k1, qf[1], s[b]);
      }
    }
    // lane (c, g) holds, for query row c, the scores of keys b*16 + 4g + r (log2 units: q carries scale * log2 e)
    float mx = -INFINITY;
#pragma unroll
    for (int b = 0; b < 8; ++b) {
      if (b < nkb) {                                           // (uniform) blocks past the last key cost nothing
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          if (b * 16 + g * 4 + r >= lk) s[b][r] = -INFINITY;
          mx = fmaxf(mx, s[b][r]);
        }
      }
    }
    mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    float sum = 0.f;
    V8 pf[4];                                                  // P^T as the B operand of O^T = V^T P^T: 32 keys per step
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      V8 pv;
#pragma unroll
      for (int hb = 0; hb < 2; ++hb) {
        if (j * 2 + hb < nkb) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float pe = __builtin_amdgcn_exp2f(s[j * 2 + hb][r] - mx);
            const T pt = (T)pe;                                // probabilities are an MFMA operand: storage type
            sum += (float)pt;
            pv[hb * 4 + r] = pt;
          }
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r) pv[hb * 4 + r] = (T)0.f;
        }
      }
      pf[j] = pv;
    }
    sum += __shfl_xor(sum, 16, 64);
    sum += __shfl_xor(sum, 32, 64);
    const float inv = 1.0f / sum;
    // V^T fragments by ds_read_b64_tr_b16 in INLINE ASM with their own lgkmcnt wait: through the builtin the compiler
    // puts `s_waitcnt vmcnt(0)` in front of every transposed read while an LDS-DMA is in flight (it cannot tell the
    // read from the DMA's LDS write), which would drain the next round's K / V prefetch 12 times per round.
    f32x4 o[3];
    const uint32_t vaddr = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const void*)(Vs + (g * 4 + (c >> 2)) * XD + (c & 3) * 4);
#pragma unroll
    for (int dt = 0; dt < 3; ++dt) {
      o[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
      u32x2 f[8];                                           // [j][lo / hi]: keys j*32 + {0..15}, {16..31}
      asm volatile(
          "ds_read_b64_tr_b16 %0, %8 offset:%9\n\t"
          "ds_read_b64_tr_b16 %1, %8 offset:%10\n\t"
          "ds_read_b64_tr_b16 %2, %8 offset:%11\n\t"
          "ds_read_b64_tr_b16 %3, %8 offset:%12\n\t"
          "ds_read_b64_tr_b16 %4, %8 offset:%13\n\t"
          "ds_read_b64_tr_b16 %5, %8 offset:%14\n\t"
          "ds_read_b64_tr_b16 %6, %8 offset:%15\n\t"
          "ds_read_b64_tr_b16 %7, %8 offset:%16\n\t"
          "s_waitcnt lgkmcnt(0)"
          : "=&v"(f[0]), "=&v"(f[1]), "=&v"(f[2]), "=&v"(f[3]), "=&v"(f[4]), "=&v"(f[5]), "=&v"(f[6]), "=&v"(f[7])
          : "v"(vaddr + (uint32_t)(dt * 32)),
            "n"((0 * 32) * XD * 2), "n"((0 * 32 + 16) * XD * 2), "n"((1 * 32) * XD * 2), "n"((1 * 32 + 16) * XD * 2),
            "n"((2 * 32) * XD * 2), "n"((2 * 32 + 16) * XD * 2), "n"((3 * 32) * XD * 2), "n"((3 * 32 + 16) * XD * 2)
          : "memory");
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if (j * 2 < nkb) {
          V8 vf;
          __builtin_memcpy(&vf, &f[2 * j], 8);
          __builtin_memcpy(reinterpret_cast<char*>(&vf) + 8, &f[2 * j + 1], 8);
          o[dt] = dd_mfma16(vf, pf[j], o[dt]);
        }
      }
    }
    // O^T: lane (c, g) holds d = dt*16 + 4g + r of query row c -> over the head's Q columns (own rows, own head)
#pragma unroll
    for (int dt = 0; dt < 3; ++dt) {
      const int d0 = dt * 16 + g * 4;
      if (d0 < XD) {
        V4 o4;
#pragma unroll
        for (int r = 0; r < 4; ++r) o4[r] = (T)(o[dt][r] * inv);
        *reinterpret_cast<V4*>(act + (rb * 16 + c) * XPITCH + head * XD + d0) = o4;
      }
    }
  }
  // ---- phase 3: out = O Wo^T + bo + res (steps 0, 1 of Wo are in flight in slots 2, 3; residual in rv) --------------
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");           // this wave's O writes are in LDS
  __builtin_amdgcn_s_barrier();                                // O complete, every K / V buffer consumed (a raw barrier:
  asm volatile("" ::: "memory");                               // __syncthreads() would drain the Wo steps in flight)
  issue_w(rs_wo, 2, 0);
  gemm320(rs_wo, 2, [&]() {});
  T* outb = reinterpret_cast<T*>(p.out) + grow0 * p.ldo;
  V4 yv[2][5];
#pragma unroll
  for (int tn = 0; tn < 2; ++tn) {
    const int ch = wave * 32 + tn * 16 + g * 4;
    float b4[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) b4[r] = (float)bv[tn][r];
#pragma unroll
    for (int tm = 0; tm < 5; ++tm) {
      const int row = tm * 16 + c;
      float y[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) y[r] = acc[tn][tm][r] + b4[r];
#pragma unroll
      for (int r = 0; r < 4; ++r) y[r] += (float)rv[tn][tm][r];
      V4 o4;
#pragma unroll
      for (int r = 0; r < 4; ++r) o4[r] = (T)y[r];
      yv[tn][tm] = o4;
      if (row < nrows) *reinterpret_cast<V4*>(outb + (int64_t)row * p.ldo + ch) = o4;
    }
  }
  if (!p.ln_out) return;
  // ---- LayerNorm(out) over the 320 channels of every row: two-pass statistics over the ROUNDED values ---------
  // per (row, wave) partial = sum over the wave's 32 channels; the ring region is free (gemm320 ended on a barrier).
  // Round 6: the barriers wait for the LDS partials only (a __syncthreads() also drains the stores of `out` issued just
  // above: ~1.5 us each time), the squares go to a second array (one barrier fewer), gamma / beta are already in registers.
  auto lds_barrier = [&]() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  };
  float* red2 = red + XR * XW;
#pragma unroll
  for (int tm = 0; tm < 5; ++tm) {
    float a = 0.f;
#pragma unroll
    for (int tn = 0; tn < 2; ++tn)
#pragma unroll
      for (int r = 0; r < 4; ++r) a += (float)yv[tn][tm][r];
    a += __shfl_xor(a, 16, 64);
    a += __shfl_xor(a, 32, 64);
    if (g == 0) red[(tm * 16 + c) * XW + wave] = a;
  }
  lds_barrier();
  float mean[5];
#pragma unroll
  for (int tm = 0; tm < 5; ++tm) {
    float a = 0.f;
#pragma unroll
    for (int w = 0; w < XW; ++w) a += red[(tm * 16 + c) * XW + w];
    mean[tm] = a * (1.0f / XC);
  }
#pragma unroll
  for (int tm = 0; tm < 5; ++tm) {
    float a = 0.f;
#pragma unroll
    for (int tn = 0; tn < 2; ++tn)
#pragma unroll
      for (int r = 0; r < 4; ++r) { const float d = (float)yv[tn][tm][r] - mean[tm]; a += d * d; }
    a += __shfl_xor(a, 16, 64);
    a += __shfl_xor(a, 32, 64);
    if (g == 0) red2[(tm * 16 + c) * XW + wave] = a;
  }
  lds_barrier();
  T* lnb = reinterpret_cast<T*>(p.ln_out) + grow0 * p.ld_ln;
#pragma unroll
  for (int tm = 0; tm < 5; ++tm) {
    float a = 0.f;
#pragma unroll
    for (int w = 0; w < XW; ++w) a += red2[(tm * 16 + c) * XW + w];
    const float rstd = 1.0f / sqrtf(a * (1.0f / XC) + p.ln_eps);
    const int row = tm * 16 + c;
    if (row >= nrows) continue;
#pragma unroll
    for (int tn = 0; tn < 2; ++tn) {
      const int ch = wave * 32 + tn * 16 + g * 4;
      V4 o4;
#pragma unroll
      for (int r = 0; r < 4; ++r)
        o4[r] = (T)(((float)yv[tn][tm][r] - mean[tm]) * rstd * (float)lng[tn][r] + (float)lnb4[tn][r]);
      *reinterpret_cast<V4*>(lnb + (int64_t)row * p.ld_ln + ch) = o4;
    }
  }
}

constexpr size_t kXSmem = kActBytes + (size_t)4 * XC * 32 * 2;          // 54,272 + 81,920 = 136,192 B

template <typename T>
int launch_xattn(const XAttnParams& p, hipStream_t s) {
  static_assert(2 * (2 * XLK * XD * 2) == 4 * XC * 32, "two K/V buffers = the four weight slots");
  static_assert(XR * 41 * 16 <= 52 * 1024 && 53 * 1024 == (int)kActBytes && XPITCH * 2 == 41 * 16, "activation tile");
  auto kern = dd_xattn320_kernel<T>;
  static std::atomic<uint64_t> attr_done{0};
  dd_ensure_dyn_lds(reinterpret_cast<const void*>(kern), kXSmem, attr_done);
  hipLaunchKernelGGL(kern, dim3(p.inst * p.tiles), dim3(64 * XW), kXSmem, s, p);
  return dd_check_launch();
}

}  // namespace

// Packs a [320][320] torch Linear weight into the order dd_xattn320 streams it in: [K step ks = 0..9][row n][position
// pos = 0..3][8 elements], position pos holding the logical 16-B chunk  pos ^ ((n >> 2) & 3)  of columns ks*32 .. +32.
namespace {
template <typename T>
__global__ void dd_xattn_pack_kernel(const T* __restrict__ w, T* __restrict__ out) {
  const int q = blockIdx.x * blockDim.x + threadIdx.x;         // 16-B chunk of the packed matrix: 10 * 320 * 4
  if (q >= 10 * XC * 4) return;
  const int ks = q / (XC * 4), rem = q - ks * (XC * 4);
  const int n = rem >> 2, pos = rem & 3;
  const int src = ks * 32 + ((pos ^ ((n >> 2) & 3)) << 3);
  dd_st16(out + (int64_t)q * 8, dd_ld16(w + (int64_t)n * XC + src));
}
}  // namespace

extern "C" int dd_xattn_pack_weight(const void* w, void* packed, int32_t dtype, dd_stream_t stream) {
  if (!w || !packed || !dd_aligned16(w) || !dd_aligned16(packed)) return DD_ERR_BAD_ARG;
  if (dtype != DD_F16 && dtype != DD_BF16) return DD_ERR_BAD_ARG;
  dd_clear_error();
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const int chunks = 10 * XC * 4;
  if (dtype == DD_F16)
    hipLaunchKernelGGL(dd_xattn_pack_kernel<_Float16>, dim3((chunks + 255) / 256), dim3(256), 0, s,
                       reinterpret_cast<const _Float16*>(w), reinterpret_cast<_Float16*>(packed));
  else
    hipLaunchKernelGGL(dd_xattn_pack_kernel<__bf16>, dim3((chunks + 255) / 256), dim3(256), 0, s,
                       reinterpret_cast<const __bf16*>(w), reinterpret_cast<__bf16*>(packed));
  return dd_check_launch();
}

extern "C" int dd_xattn320(const dd_xattn_desc* d, dd_stream_t stream) {
  if (!d || !d->x || !d->wq || !d->wo || !d->bo || !d->k || !d->v || !d->out) return DD_ERR_BAD_ARG;
  if (d->instances <= 0 || d->rows_per_inst <= 0 || d->lk <= 0) return DD_ERR_BAD_ARG;
  if (d->channels != XC || d->heads != XH) return DD_ERR_UNSUPPORTED;
  if (d->lk > XLK) return DD_ERR_UNSUPPORTED;
  if (d->dtype != DD_F16 && d->dtype != DD_BF16) return DD_ERR_BAD_ARG;
  if ((d->ldx & 7) || (d->ldk & 7) || (d->ldv & 7) || (d->ldo & 3) || (d->res && (d->ldres & 3))) return DD_ERR_BAD_ARG;
  if (!dd_aligned16(d->x) || !dd_aligned16(d->wq) || !dd_aligned16(d->wo) || !dd_aligned16(d->k) ||
      !dd_aligned16(d->v) || !dd_aligned16(d->out) || (d->res && !dd_aligned16(d->res)) ||
      !dd_aligned16(d->bo)) return DD_ERR_BAD_ARG;
  if (d->ln_out && (!d->ln_gamma || !d->ln_beta || !dd_aligned16(d->ln_out) || (d->ld_ln_out & 3))) return DD_ERR_BAD_ARG;
  if ((int64_t)d->instances * ((d->rows_per_inst + XR - 1) / XR) > 0x7fffffffLL) return DD_ERR_UNSUPPORTED;
  XAttnParams p{};
  p.x = d->x; p.ldx = d->ldx; p.res = d->res; p.ldres = d->ldres;
  p.wq = d->wq; p.wo = d->wo; p.bo = d->bo;
  p.k = d->k; p.v = d->v; p.ldk = d->ldk; p.ldv = d->ldv;
  p.k_is = d->k_inst_stride; p.k_hs = d->k_head_stride; p.v_is = d->v_inst_stride; p.v_hs = d->v_head_stride;
  if ((p.k_is & 7) || (p.k_hs & 7) || (p.v_is & 7) || (p.v_hs & 7)) return DD_ERR_BAD_ARG;
  if (((int64_t)(d->lk - 1) * d->ldk + (XH - 1) * p.k_hs + XD) * 2 >= (1ll << 31) ||
      ((int64_t)(d->lk - 1) * d->ldv + (XH - 1) * p.v_hs + XD) * 2 >= (1ll << 31)) return DD_ERR_UNSUPPORTED;
  p.out = d->out; p.ldo = d->ldo;
  p.ln_out = d->ln_out; p.ld_ln = d->ld_ln_out; p.ln_g = d->ln_gamma; p.ln_b = d->ln_beta; p.ln_eps = d->ln_eps;
  p.inst = d->instances; p.n = d->rows_per_inst; p.lk = d->lk;
  p.lk_dev = d->lk_dev;
  if (d->lk_dev && (reinterpret_cast<uintptr_t>(d->lk_dev) & 3u)) return DD_ERR_BAD_ARG;
  p.tiles = (d->rows_per_inst + XR - 1) / XR;
  p.qscale = d->scale * 1.44269504088896340736f;
  { static const int dbg = getenv("DD_XATTN_DBG") ? atoi(getenv("DD_XATTN_DBG")) : 0; p.dbg = dbg; }
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  dd_clear_error();
  if (d->dtype == DD_F16) return launch_xattn<_Float16>(p, s);
  return launch_xattn<__bf16>(p, s);
}
