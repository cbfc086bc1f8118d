// Diagnostic switches of the kernels — every preprocessor conditional of the hot loops lives HERE.
//
// None of these macros is defined in the product build (dualdiff_amd/_build.py passes no -DDD_DBG_*; bench.py lists every
// DD_* environment variable of a run in its line): each flag below is then `false`, each hook expands to nothing, and the
// kernels compile to the same ISA as before this header existed (checked byte for byte when it was introduced).  The
// diagnostic libraries are built by tools/build_dbg_libs.sh, tools/gemm4_bound.sh and tools/conv3s_bound.sh, loaded with
// DD_HIP_LIB=... (which bench.py refuses without --allow-alt-lib), and exist to answer ONE question each: which side of a
// loop sets its length.  Their results are garbage where noted; they are timing instruments, not code paths.
//
//   switch                 kernel(s)               what it removes / changes
//   DD_DBG_NOMFMA          all GEMM / conv loops   the matrix instructions (operands stay live)              dd_common.h
//   DD_DBG_NODMA           dd_gemm2/3/4, conv3s    the LDS-DMA loads (LDS holds garbage)
//   DD_DBG_SAMEK           dd_gemm2                every K-step re-stages the SAME bytes (L1-resident after the first)
//   DD_DBG_NOSTORE         dd_gemm4                the epilogue's stores
//   DD_DBG_ONESTORE        dd_gemm4                all but the first store of a lane's tile
//   DD_DBG_STORE_AUX=n     dd_gemm4                cache-policy bits of the epilogue stores (2 = nt, 16 = sc1)
//   DD_DBG_NOLDS           dd_gemm4                the fragment reads of the K loop
//   DD_DBG_NOSECTOR        dd_gemm4                sector-contiguous store layout of the 192x128 tile (round-6 A/B)
//   DD_DBG_GEGLU_CHEAP     GEGLU epilogues         the gate becomes h * g                                    dd_common.h
//   DD_DBG_C3_NOMFMA / _NOGATHER / _NOWREAD / _NOBAR / _NOWAIT / _NODMA      dd_conv3s: one side of the (chunk, tap) step
//   DD_DBG_NOEXP / _NOSTAGE dd_attn5               the softmax exponentials become moves / no K, V staging (tiles hold garbage)
//   DD_DBG_STAMP           dd_gemm2/3, conv3s      s_memtime stamps at phase boundaries into the last MiB of the workspace
//   DD_DBG_ONLY_P / _C3    host dispatch           a quick-to-compile library with one kernel family (reading its ISA)
#pragma once
#include <stdint.h>

namespace dd_dbg {
#ifdef DD_DBG_NODMA
constexpr bool NODMA = true;
#else
constexpr bool NODMA = false;
#endif
#ifdef DD_DBG_SAMEK
constexpr bool SAMEK = true;
#else
constexpr bool SAMEK = false;
#endif
#ifdef DD_DBG_NOLDS
constexpr bool NOLDS = true;
#else
constexpr bool NOLDS = false;
#endif
#ifdef DD_DBG_NOSECTOR
constexpr bool NOSECTOR = true;
#else
constexpr bool NOSECTOR = false;
#endif
#ifdef DD_DBG_C3_NODMA
constexpr bool C3_NODMA = true;
#else
constexpr bool C3_NODMA = false;
#endif
#ifdef DD_DBG_C3_NOWAIT
constexpr bool C3_NOWAIT = true;
#else
constexpr bool C3_NOWAIT = false;
#endif
#ifdef DD_DBG_C3_NOWREAD
constexpr bool C3_NOWREAD = true;
#else
constexpr bool C3_NOWREAD = false;
#endif
#ifdef DD_DBG_C3_NOGATHER
constexpr bool C3_NOGATHER = true;
#else
constexpr bool C3_NOGATHER = false;
#endif
#ifdef DD_DBG_NOSTAGE
constexpr bool NOSTAGE = true;
#else
constexpr bool NOSTAGE = false;
#endif
}  // namespace dd_dbg

// ---- dd_attn5_kernel: the softmax exponential -------------------------------------------------------------------------
#ifdef DD_DBG_NOEXP
#define DD_EXP2(x) (x)
#else
#define DD_EXP2(x) __builtin_amdgcn_exp2f(x)
#endif

// ---- dd_gemm4_kernel: the epilogue's 16-byte buffer store ------------------------------------------------------------
#ifndef DD_DBG_STORE_AUX
#define DD_DBG_STORE_AUX 0
#endif
#ifdef DD_DBG_NOSTORE
#define DD_G4_STORE_STATE() ((void)0)
#define DD_G4_STORE(...) ((void)0)
#elif defined(DD_DBG_ONESTORE)
#define DD_G4_STORE_STATE() bool g4_first = true
#define DD_G4_STORE(data, rsrc, voff, soff, aux) do { if (g4_first) __builtin_amdgcn_raw_buffer_store_b128(data, rsrc, voff, soff, 0); g4_first = false; } while (0)
#else
#define DD_G4_STORE_STATE() ((void)0)
#define DD_G4_STORE(data, rsrc, voff, soff, aux) __builtin_amdgcn_raw_buffer_store_b128(data, rsrc, voff, soff, DD_DBG_STORE_AUX)
#endif

// ---- dd_conv3s_kernel: matrix instruction and workgroup barrier ------------------------------------------------------
#ifdef DD_DBG_C3_NOMFMA
#define C3_MFMA(w, x, a) ([&]() { asm volatile("" :: "v"(w), "v"(x)); return a; }())
#else
#define C3_MFMA(w, x, a) dd_mfma16(w, x, a)
#endif
#ifdef DD_DBG_C3_NOBAR
#define C3_BARRIER() ((void)0)
#else
#define C3_BARRIER() __builtin_amdgcn_s_barrier()
#endif

// ---- phase stamps (DD_DBG_STAMP): wave 0 of every workgroup records s_memtime at phase boundaries plus s_memrealtime at
// both ends into the LAST MiB of the workspace (ops.py over-allocates it when DD_DBG_STAMP_WS=1); nothing reads them on the
// device.  dd_conv3s adds per-wave SEGMENT clocks of its steady-state steps (s >= 9): [0] vmcnt wait, [1] barrier, [2] late
// block (MFMAs of step s-1 + DMA), [3] fragment reads issued AND returned (the stamp itself waits lgkmcnt(0)), [4] early
// block (DMA + MFMAs issued); waves 0 (early) and 4 (late) write theirs behind the phase stamps (tools/conv3s_stamps.py).
#ifdef DD_DBG_STAMP
#define DD_STAMP_DECL() uint64_t dbg_t[6]; const uint64_t dbg_r0 = __builtin_amdgcn_s_memrealtime()
#define DD_STAMP(i) do { if (threadIdx.x == 0) dbg_t[i] = __builtin_readcyclecounter(); } while (0)
#define DD_STAMP_IF(cond, i) do { if (cond) DD_STAMP(i); } while (0)
#define DD_STAMP_FLUSH(p)                                                                          \
  do {                                                                                             \
    DD_STAMP(5);                                                                                   \
    if (threadIdx.x == 0 && (p).dbg_stamps) {                                                      \
      uint64_t* o_ = (p).dbg_stamps + ((size_t)blockIdx.z * gridDim.x + blockIdx.x) * 8;          \
      for (int i_ = 0; i_ < 6; ++i_) o_[i_] = dbg_t[i_];                                           \
      o_[6] = dbg_r0;                                                                              \
      o_[7] = __builtin_amdgcn_s_memrealtime();                                                    \
    }                                                                                              \
  } while (0)
#define C3_SEG_DECL() uint64_t seg[5] = {0, 0, 0, 0, 0}; uint64_t seg_prev = 0
#define C3_SEG(k) do { const uint64_t now_ = __builtin_readcyclecounter(); if (s >= 9) seg[k] += now_ - seg_prev; seg_prev = now_; } while (0)
#define C3_SEG_FLUSH(p, wave, lane, nsteps)                                                                          \
  do {                                                                                                               \
    if (((wave) == 0 || (wave) == 4) && (lane) == 0 && (p).dbg_stamps) {                                             \
      uint64_t* o_ = (p).dbg_stamps + 65536 + (((size_t)blockIdx.z * gridDim.x + blockIdx.x) * 2 + ((wave) == 4)) * 8; \
      for (int i_ = 0; i_ < 5; ++i_) o_[i_] = seg[i_];                                                               \
      o_[5] = (nsteps) > 9 ? (nsteps) - 9 : 0;                                                                       \
      o_[6] = (wave);                                                                                                \
      o_[7] = 1;                                                                                                     \
    }                                                                                                                \
  } while (0)
#define DD_STAMP_HOST(p, d)                                                                                          \
  do {                                                                                                               \
    if ((d)->ws && (d)->ws_bytes >= (4 << 20))                                                                       \
      (p).dbg_stamps = reinterpret_cast<uint64_t*>(reinterpret_cast<char*>((d)->ws) + (d)->ws_bytes - (1 << 20));    \
  } while (0)
#else
#define DD_STAMP_DECL() ((void)0)
#define DD_STAMP(i) do {} while (0)
#define DD_STAMP_IF(cond, i) do {} while (0)
#define DD_STAMP_FLUSH(p) do {} while (0)
#define C3_SEG_DECL() ((void)0)
#define C3_SEG(k) do {} while (0)
#define C3_SEG_FLUSH(p, wave, lane, nsteps) do {} while (0)
#define DD_STAMP_HOST(p, d) do {} while (0)
#endif
