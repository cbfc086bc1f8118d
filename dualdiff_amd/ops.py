"""Torch-tensor front end of the HIP C-ABI (device memory + stream plumbing only).

Every function launches hand-written gfx950 kernels from libdualdiff_hip.so on torch's
current stream.  Tensors are token-major / NHWC 2-D views (rows, channels) in fp16 or bf16.
There is no eager fallback: a non-GPU tensor raises.
"""
import ctypes
import threading as _threading

import torch

from . import _native
from ._native import DD_BF16, DD_EPI_GEGLU, DD_EPI_NONE, DD_EPI_SILU, DD_F16, AttnDesc, Gemm8Desc, GemmDesc, XAttnDesc

_WS = {}
_WS_MIN_BYTES = 64 << 20


class KernelTimer:
    """Optional per-launch HIP-event timing of the MFMA kernels (bench.py roofline leg).

    When installed (`ops.set_timer(KernelTimer())`) every gemm / conv3x3 / attention launch is
    bracketed by events recorded on the stream the kernel is launched on; `summary()` groups the
    launches by kernel symbol and returns count, total time and algorithmic FLOPs."""

    def __init__(self, shapes=False):
        self.records = []
        self.shapes = shapes
        self.overhead_ms = 0.0
        self.empty_bracket_ms = 0.0
        self.probe_ms = 0.0

    def calibrate(self, n=48, probe_us=15.0):
        """Event overhead per bracket, MEASURED at run time (not a fitted constant) and subtracted in
        summary() so that the per-kernel averages are comparable with rocprofv3's kernel durations.

        Probe: dd_probe_spin — one wave that busy-waits `probe_us` on the constant 100 MHz s_memrealtime counter and
        stores its own first and last reading, i.e. a kernel whose DEVICE-SIDE duration is known without events or a
        profiler.  n such launches are bracketed like the kernels of the instrumented step (queue kept busy);
        overhead = median(bracket - in-kernel duration): what an event pair adds around a ~15 us kernel (the
        dominant classes are 15-45 us).  A first attempt subtracted a probe's back-to-back time instead; that time
        contains the launch-to-launch gap a profiler does not count and gave 0.  The empty-bracket figure
        (4.6-4.8 us on MI355X) is kept in `empty_bracket_ms` for the JSON line; kernels of ~5 us stay the least
        certain rows of the table."""
        lib = _native.load()
        st = torch.cuda.current_stream()
        dev = torch.device("cuda", torch.cuda.current_device())
        pairs = []
        for _ in range(n):
            e0 = torch.cuda.Event(enable_timing=True)
            e1 = torch.cuda.Event(enable_timing=True)
            e0.record(st)
            e1.record(st)
            pairs.append((e0, e1))
        torch.cuda.synchronize()
        d = sorted(a.elapsed_time(b) for a, b in pairs)
        self.empty_bracket_ms = d[len(d) // 2]
        stamps = torch.zeros((n, 2), dtype=torch.int64, device=dev)
        ticks = int(probe_us * 100)
        backlog = torch.empty(64 << 20, dtype=torch.uint8, device=dev)
        for _ in range(40):                       # keep the GPU behind the host while the brackets are enqueued
            backlog.zero_()
        pairs = []
        for i in range(n):
            a0, a1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a0.record(st)
            _native.check(lib.dd_probe_spin(ctypes.c_void_p(stamps[i].data_ptr()), ticks, _stream()), "probe_spin")
            a1.record(st)
            pairs.append((a0, a1))
        torch.cuda.synchronize()
        inside = (stamps[:, 1] - stamps[:, 0]).double().cpu() * 1e-5          # 100 MHz ticks -> ms
        over = sorted(p.elapsed_time(q) - float(inside[i]) for i, (p, q) in enumerate(pairs))
        self.probe_ms = float(inside.median())
        self.overhead_ms = min(max(over[len(over) // 2], 0.0), self.empty_bracket_ms)
        return self.overhead_ms

    def start(self):
        e = torch.cuda.Event(enable_timing=True)
        e.record(torch.cuda.current_stream())
        return e

    def stop(self, e0, name, flops, nbytes, staged=0.0):
        """staged: bytes the launch moves from L2 into LDS (tiled GEMM / conv families: every tile re-stages its operand
        slabs) — the quantity that actually bounds those kernels, reported beside the contract's roofs."""
        e1 = torch.cuda.Event(enable_timing=True)
        e1.record(torch.cuda.current_stream())
        self.records.append((name, flops, nbytes, e0, e1, staged))

    def summary(self):
        torch.cuda.synchronize()
        out = {}
        for name, flops, nbytes, e0, e1, staged in self.records:
            d = out.setdefault(name, {"count": 0, "ms": 0.0, "flops": 0.0, "bytes": 0.0, "staged": 0.0})
            d["count"] += 1
            d["ms"] += max(e0.elapsed_time(e1) - self.overhead_ms, 0.0)
            d["flops"] += flops
            d["bytes"] += nbytes
            d["staged"] += staged
        return out


_TIMER = None

# ---- run-time tile / split-K selection ---------------------------------------------------------
# The step touches a finite set of GEMM / conv shapes.  On first (eager, non-captured) use of a
# shape every tile config x split-K candidate of dd_gemm is timed with HIP events on a scratch
# output and the fastest is cached; graph capture then records the tuned launches.
# DD_AUTOTUNE=0 falls back to the built-in heuristic of csrc/gemm.hip.
import os as _os

_AUTOTUNE = _os.environ.get("DD_AUTOTUNE", "1") != "0"
_TUNED = {}
_TILES = None           # filled from the library (dd_gemm_tile_id) on first use
_SPLITS = (1, 2, 3, 4, 5, 6, 8, 12, 16)


def _entry(v):
    """(tile, split-K, split-K form): tables written before round 3 carry two values (form 0 = two launches)."""
    return (int(v[0]), int(v[1]), int(v[2]) if len(v) > 2 else 0)


def tuned_table():
    return dict(_TUNED)


def save_tuned(path, merge=True):
    """Persist the tuned (tile, split-K) table so a later process skips the timing sweep.  merge: entries of
    an existing file that this process did not touch (other dtype, other batch) are kept."""
    import ast
    import json
    entries = {}
    if merge and _os.path.exists(path):
        with open(path) as f:
            for k, v in json.load(f).get("entries", []):
                entries[ast.literal_eval(k)] = _entry(v)
    entries.update(_TUNED)
    _os.makedirs(_os.path.dirname(_os.path.abspath(path)), exist_ok=True)
    with open(path, "w") as f:
        json.dump({"arch": "gfx950", "entries": [[repr(k), list(v)] for k, v in sorted(entries.items(), key=repr)]}, f, indent=0)


def load_tuned(path):
    """Load a table written by save_tuned(); entries for shapes already tuned here are kept."""
    import ast
    import json
    with open(path) as f:
        blob = json.load(f)
    if blob.get("arch") != "gfx950":
        raise RuntimeError("tune cache %s is not for gfx950" % path)
    n = 0
    for k, v in blob["entries"]:
        key = ast.literal_eval(k)
        if key not in _TUNED:
            _TUNED[key] = _entry(v)
            n += 1
    return n


_COLD = _os.environ.get("DD_AUTOTUNE_COLD", "1") != "0"
# DD_TUNE_CHALLENGE=52[,..]: tiles added after the tracked table was written are timed against every entry's incumbent
# the first time its shape is met (bench.py --challenge-tiles writes the table back)
CHALLENGE_TILES = tuple(int(t) for t in _os.environ.get("DD_TUNE_CHALLENGE", "").split(",") if t.strip())
_CHALLENGED = set()
_FLUSH = {}

# The tuned table of the shapes the denoising step touches is TRACKED (dualdiff_amd/tuned/gfx950.json,
# written by `bench.py --retune`) and loaded on first use, so every process launches the same kernels
# for the same shapes and the bench's roofline line can be recomputed from profiles/.  Only shapes that
# are not in the table are timed at run time.  DD_TUNE_TABLE=0 ignores the tracked table, DD_TUNE_TABLE=<path>
# loads another one.
TUNE_TABLE_PATH = _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "tuned", "gfx950.json")
_TABLE_LOADED = False


def _load_default_table():
    global _TABLE_LOADED
    if _TABLE_LOADED:
        return
    _TABLE_LOADED = True
    path = _os.environ.get("DD_TUNE_TABLE", TUNE_TABLE_PATH)
    if path != "0" and _os.path.exists(path):
        load_tuned(path)


def forget_tuned():
    """Drop every tuned entry (bench.py --retune) and do not read the tracked table again."""
    global _TABLE_LOADED
    _TUNED.clear()
    _TABLE_LOADED = True


def _flush_and_warm(device, warm):
    """Puts the caches in the state a launch sees inside the step: weights COLD (a step streams
    ~3.3 GB of them, far more than L2 + the 256 MiB Infinity Cache hold), activations just produced
    by the previous kernel and therefore WARM."""
    buf = _FLUSH.get(device)
    if buf is None:
        buf = _FLUSH[device] = torch.empty(320 << 20, dtype=torch.uint8, device=device)
    buf.zero_()
    for t in warm:
        if t is not None:
            t.sum()


def _autotune(lib, d, key, out_shape, dtype, device, warm=()):
    _load_default_table()
    hit = _TUNED.get(key)
    challenge = ()
    if hit is not None:
        if not CHALLENGE_TILES or key in _CHALLENGED or torch.cuda.is_current_stream_capturing():
            return hit
        # a new tile asks for the shapes of the tracked table: time the incumbent against the challengers only
        _CHALLENGED.add(key)
        challenge = tuple((c, sp, 0) for c in CHALLENGE_TILES for sp in sorted({1, max(1, int(hit[1]))}))
    elif not _AUTOTUNE or torch.cuda.is_current_stream_capturing():
        return 0, 0, 0
    saved = (d.out, d.ldc, d.accumulate, d.tile, d.split_k, d.ws, d.ws_bytes)
    scratch = torch.empty(out_shape, dtype=dtype, device=device)
    d.out, d.ldc, d.accumulate = scratch.data_ptr(), scratch.stride(0), 0
    kt = (d.k + 63) // 64
    blocks128 = ((d.rows + 127) // 128) * ((d.n + 127) // 128)
    best, best_t = (0, 0, 0), float("inf")
    stream = _stream()
    global _TILES
    if _TILES is None:
        _TILES = tuple(lib.dd_gemm_tile_id(i) for i in range(lib.dd_gemm_num_tiles()))

    def timed(tile, split, iters, ink=0):          # ink: third entry of a table row (split-K form), always 0 since round 5
        d.tile, d.split_k = tile, split
        need = lib.dd_gemm_workspace_bytes(ctypes.byref(d))
        if need > 0:
            ws = workspace(need, device)
            d.ws, d.ws_bytes = ws.data_ptr(), ws.numel() * 4
        if lib.dd_gemm(ctypes.byref(d), stream) != 0:
            return None
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        if not _COLD:
            e0.record()
            for _ in range(iters):
                lib.dd_gemm(ctypes.byref(d), stream)
            e1.record()
            e1.synchronize()
            return e0.elapsed_time(e1) / iters
        samples = []
        for _ in range(iters):
            _flush_and_warm(device, warm)
            e0.record()
            lib.dd_gemm(ctypes.byref(d), stream)
            e1.record()
            e1.synchronize()
            samples.append(e0.elapsed_time(e1))
        samples.sort()
        return samples[len(samples) // 2]         # median: one slow launch (clock ramp, a neighbour's burst) must not decide

    if challenge:
        t_inc = timed(hit[0], hit[1], 21, hit[2])
        best, best_t = tuple(hit), (t_inc if t_inc is not None else float("inf"))
        for tile, split, ink in challenge:
            if (tile, split, ink) == tuple(hit):
                continue
            t = timed(tile, split, 5, ink)
            if t is None or t > 1.1 * best_t:
                continue
            t = timed(tile, split, 21, ink)
            if t is not None and t < 0.97 * best_t:          # a challenger must win by 3 %: the medians carry ~2 % of noise
                best, best_t = (tile, split, ink), t
        (d.out, d.ldc, d.accumulate, d.tile, d.split_k, d.ws, d.ws_bytes) = saved
        if best != tuple(hit):
            print("[tune] %s: %s -> %s (%.1f -> %.1f us)" % (key, tuple(hit), best, t_inc * 1e3, best_t * 1e3), flush=True)
        _TUNED[key] = best
        return best
    cands = []
    excl = {int(t) for t in _os.environ.get("DD_TUNE_EXCLUDE", "").split(",") if t.strip()}   # A/B experiments
    for tile in _TILES:
        if tile in excl:
            continue
        for split in _SPLITS:
            if split > 1 and (d.epilogue == DD_EPI_GEGLU or kt < 4 * split or blocks128 * split > 4096):
                continue
            for ink in (0,):
                t = timed(tile, split, 3, ink)            # >= 3 samples per candidate, cold or hot
                if t is not None:
                    cands.append((t, tile, split, ink))
    # the coarse pass is noisy: re-time the front-runners with more launches
    cands.sort()
    for t, tile, split, ink in cands[:8]:
        t2 = timed(tile, split, 15 if _COLD else 12, ink)
        if t2 is not None and t2 < best_t:
            best, best_t = (tile, split, ink), t2
    (d.out, d.ldc, d.accumulate, d.tile, d.split_k, d.ws, d.ws_bytes) = saved
    _TUNED[key] = best
    return best


def set_timer(t):
    global _TIMER
    _TIMER = t


def _dt(t):
    if t.dtype == torch.float16:
        return DD_F16
    if t.dtype == torch.bfloat16:
        return DD_BF16
    raise TypeError("dualdiff_amd ops take fp16 / bf16 tensors, got %s" % t.dtype)


def _need_gpu(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise RuntimeError("dualdiff_amd ops run on the GPU only (got a %s tensor); "
                               "there is no CPU fallback" % t.device)


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


_WS_IN_GRAPH = set()      # keys whose current buffer a captured HIP graph holds a raw pointer to
_WS_RETIRED = []          # such buffers after they were outgrown: kept alive for the graphs that use them


# diagnostic builds (-DDD_DBG_STAMP, tools/build_dbg_libs.sh) write phase stamps into the tail of the workspace
_DBG_STAMP_WS = _os.environ.get("DD_DBG_STAMP_WS", "0") == "1"


_WS_OWNER = _threading.local()


class workspace_owner:
    """`with workspace_owner(token):` scratch buffers handed out inside are keyed by `token` instead of the current
    stream's handle.  model_base.ForwardGraphs records each model's graphs under its own token: torch's stream handles
    are pooled and re-used, so two models' capture streams can share a handle — harmless while forward graphs replay one
    after the other, a race once two models' graphs replay concurrently (round 6: sibling overlap)."""

    def __init__(self, token):
        self.token = token

    def __enter__(self):
        self.prev = getattr(_WS_OWNER, "token", None)
        _WS_OWNER.token = self.token

    def __exit__(self, *exc):
        _WS_OWNER.token = self.prev


def workspace(nbytes, device, kind="gemm"):
    """Grow-only fp32 scratch buffer per (device, stream — or workspace_owner token —, kind): kernels on concurrent
    streams must not share scratch, and the split-K buffer (whose leading counter region dd_gemm keeps at zero)
    is never lent to GroupNorm.  Zero-filled on allocation; allocate before graph capture.
    A buffer that was handed out during a capture is never freed (torch's stream handles are pooled and
    re-used, so a later, larger eager workload can outgrow a buffer that a live graph still writes to)."""
    owner = getattr(_WS_OWNER, "token", None)
    key = (device.type, device.index if device.index is not None else torch.cuda.current_device(),
           torch.cuda.current_stream().cuda_stream if owner is None else ("owner", owner), kind)
    ws = _WS.get(key)
    need = max(int(nbytes), _WS_MIN_BYTES)
    capturing = torch.cuda.is_current_stream_capturing()
    if ws is None or ws.numel() * 4 < need:
        if capturing:
            raise RuntimeError("workspace would have to grow during graph capture; run one eager "
                               "warm-up step first")
        if ws is not None and key in _WS_IN_GRAPH:
            _WS_RETIRED.append(ws)
            _WS_IN_GRAPH.discard(key)
        ws = torch.zeros((need + 3) // 4, dtype=torch.float32, device=device)
        _WS[key] = ws
    if capturing:
        _WS_IN_GRAPH.add(key)
    return ws


def _kname(lib, d):
    full = lib.dd_gemm_kernel_name(ctypes.byref(d)).decode()
    name, rest = full.split(" split=")
    return name, int(rest.split(" ")[0])


def _staged_bytes(lib, d):
    """L2 -> LDS bytes of one dd_gemm launch, from its plan (`grid=MxN tile=BMxBN...`): every tile stages a
    (BM + BN) x 64-element slab pair per K-step; the direct small-image conv stages its (BM + 64)-row activation slab
    once per 64-channel chunk and a BN x 64 weight slab per (chunk, tap)."""
    import re
    full = lib.dd_gemm_kernel_name(ctypes.byref(d)).decode()
    m = re.search(r"grid=(\d+)x(\d+) tile=(.*)$", full)
    if m is None:
        return 0.0
    tiles = int(m.group(1)) * int(m.group(2))
    t = re.search(r"(\d+)x(\d+)", m.group(3))
    if t is None:
        return 0.0
    bm, bn = int(t.group(1)), int(t.group(2))
    if "conv3s" in m.group(3):
        chunks = d.cin // 64
        return float(tiles) * chunks * ((bm + (88 if "band" in m.group(3) else 64)) * 128.0 + 9 * bn * 128.0)
    ksteps = (d.k + 63) // 64
    return float(tiles) * ksteps * (bm + bn) * 128.0


def _timed_gemm(lib, d, what, suffix, flops, nbytes, rows, n):
    """dd_gemm under the KernelTimer: a split-K GEMM's two launches are bracketed SEPARATELY (dd_gemm_desc.phase)
    and booked under their own kernel symbols, so that every class of the roofline table is one kernel symbol
    whose average duration can be checked against rocprofv3's."""
    name, split = _kname(lib, d)
    staged = _staged_bytes(lib, d)
    if split <= 1:
        e0 = _TIMER.start()
        _native.check(lib.dd_gemm(ctypes.byref(d), _stream()), what)
        _TIMER.stop(e0, name + suffix, flops, nbytes, staged)
        return
    slab = 4.0 * split * rows * n
    d.phase = 1
    e0 = _TIMER.start()
    _native.check(lib.dd_gemm(ctypes.byref(d), _stream()), what)
    _TIMER.stop(e0, name + suffix, flops, nbytes + slab, staged)     # operands once + the fp32 slabs written
    d.phase = 2
    e0 = _TIMER.start()
    _native.check(lib.dd_gemm(ctypes.byref(d), _stream()), what)
    _TIMER.stop(e0, "dd_splitk_reduce_kernel", 0.0, slab + 2.0 * rows * n)
    d.phase = 0


def _forget_derived(t):
    """A tensor that is about to be (re)written through its data pointer loses whatever an earlier producer
    attached to it (LayerNorm emitted by an epilogue, row statistics): those describe the OLD contents."""
    for a in ("_ln_cache", "_ln_out", "_ln_stats", "_gn_cache", "_unwritten"):
        if hasattr(t, a):
            delattr(t, a)


def _rows2d(t):
    if t.dim() != 2 or t.stride(1) != 1:
        raise ValueError("expected a 2-D tensor with unit inner stride, got shape %s stride %s"
                         % (tuple(t.shape), t.stride()))
    return t


def gemm(a, w, bias=None, *, a2=None, res=None, rowvec=None, rows_per_inst=1, alpha=1.0,
         out=None, accumulate=False, epilogue=DD_EPI_NONE, tile=0, split_k=0, ln=None, out_f32=False,
         ln_stats=False, head_major=None, ln_out=None):
    """out = alpha * (cat(a, a2) @ w.T + bias + rowvec[row // rows_per_inst]) + res  (fused).
    head_major = (D, scaled_planes, scale): the result comes back as (n / D, rows, D) — one contiguous
    [rows][D] plane per head of a fused Q|K|V projection, the first `scaled_planes` planes multiplied by
    `scale` before rounding (what `attention(..., q_prescaled=True)` expects).
    out_f32: the result is stored as fp32 (attention logits that feed a softmax).
    ln_stats: the epilogue also leaves per-row partial sums of the output (n % 32 == 0) as `out._ln_stats`;
    a later gemm(out, ..., ln=...) picks them up instead of recomputing the row statistics.

    ln_out = (gamma, beta, eps) (n == 320 only): the epilogue ALSO writes LayerNorm(out) — returned as the
    attribute `out._ln_out` — from a tile that owns whole rows (tile 40, 80 x 320).

    ln = (colsum_f32, bias_f32, eps): LayerNorm fold — `a` is the UN-normalised input, `w` the
    gamma-scaled weight; the kernel computes the row statistics itself (include/dualdiff_hip.h)."""
    lib = _native.load()
    _need_gpu(a, w, bias, a2, res, rowvec, out)
    stats_in = getattr(a, "_ln_stats", None) if ln is not None else None
    a = _rows2d(a)
    w = _rows2d(w)
    rows = a.shape[0]
    k = a.shape[1] + (a2.shape[1] if a2 is not None else 0)
    n_w = w.shape[0]
    if w.shape[1] != k or not w.is_contiguous():
        raise ValueError("weight must be contiguous [N, K=%d], got %s" % (k, tuple(w.shape)))
    n = n_w // 2 if epilogue == DD_EPI_GEGLU else n_w
    odt = torch.float32 if out_f32 else a.dtype
    hm_out = None
    if head_major is not None:
        hd, hplanes, hscale = head_major
        if out is not None or n % hd or hd % 8 or out_f32 or epilogue != DD_EPI_NONE or accumulate or ln_stats:
            raise ValueError("head_major needs a plain epilogue, its own output and n % D == 0")
        hm_out = torch.empty((n // hd, rows, hd), dtype=a.dtype, device=a.device)
        out = hm_out.view(rows, n)           # same bytes; the kernel ignores ldc in this mode
    if out is None:
        out = torch.empty((rows, n), dtype=odt, device=a.device)
    elif out.dtype != odt:
        raise TypeError("gemm: out must be %s" % odt)
    else:
        _forget_derived(out)
    d = GemmDesc()
    d.out_f32 = int(bool(out_f32))
    if hm_out is not None:
        d.out_headmajor_d, d.hm_scaled_planes, d.hm_scale = int(hd), int(hplanes), float(hscale)
    d.a = a.data_ptr(); d.lda = a.stride(0); d.k1 = a.shape[1]
    if a2 is not None:
        a2 = _rows2d(a2)
        d.a2 = a2.data_ptr(); d.lda2 = a2.stride(0)
    d.rows, d.n, d.k = rows, n, k
    d.w = w.data_ptr()
    d.bias = bias.data_ptr() if bias is not None else None
    if rowvec is not None:
        rowvec = _rows2d(rowvec)
        d.rowvec = rowvec.data_ptr(); d.ld_rowvec = rowvec.stride(0); d.rows_per_inst = rows_per_inst
    if res is not None:
        res = _rows2d(res)
        d.res = res.data_ptr(); d.ldres = res.stride(0)
    out = _rows2d(out)
    d.out = out.data_ptr(); d.ldc = out.stride(0)
    d.alpha = alpha; d.accumulate = int(accumulate); d.epilogue = epilogue
    d.conv = 0
    d.dtype = _dt(a); d.tile = tile; d.split_k = split_k
    if ln is not None:
        if bias is not None or a2 is not None:
            raise ValueError("gemm(ln=...) folds the bias and takes a single source")
        colsum, lnb, eps = ln
        _need_gpu(colsum, lnb)
        if colsum.dtype != torch.float32 or lnb.dtype != torch.float32 or colsum.numel() != n_w or lnb.numel() != n_w:
            raise ValueError("ln fold vectors must be fp32 with %d entries" % n_w)
        d.ln_colsum, d.ln_bias, d.ln_eps = colsum.data_ptr(), lnb.data_ptr(), float(eps)
        if stats_in is not None:
            if tuple(stats_in.shape) != (rows, k // 32, 2) or stats_in.dtype != torch.float32:
                raise ValueError("stale LayerNorm statistics attached to the input")
            d.ln_stats_in = stats_in.data_ptr()
    if w.dtype != a.dtype:
        raise TypeError("gemm: weight dtype %s != activation dtype %s" % (w.dtype, a.dtype))
    ln_second = None
    if ln_out is not None:
        g_, b_, eps_ = ln_out
        _need_gpu(g_, b_)
        if n != 320 or g_.numel() != n or b_.numel() != n or g_.dtype != a.dtype or b_.dtype != a.dtype:
            raise ValueError("ln_out needs n == 320 and %s gamma / beta of 320 entries" % a.dtype)
        ln_second = torch.empty((rows, n), dtype=a.dtype, device=a.device)
        d.ln_out, d.ld_ln_out = ln_second.data_ptr(), n
        d.lno_gamma, d.lno_beta, d.ln_eps = g_.data_ptr(), b_.data_ptr(), float(eps_)
        # the 80 x 320 whole-row tiles: 74 (pipelined K loop, round 5) / 40 (dd_gemm2, persistent walk); 0 = the library picks
        # by the row count (DD_LN_OUT_TILE forces one for A/Bs)
        if tile not in (40, 74):
            tile = LN_OUT_TILE
        d.tile, d.split_k = tile, 1
        tile = -1                                 # (not 0: no tuner lookup for this call)
    stats_out = None
    if ln_stats:
        if n % 32 or epilogue == DD_EPI_GEGLU or out_f32:
            raise ValueError("ln_stats needs a plain epilogue and n % 32 == 0")
        stats_out = torch.empty((rows, n // 32, 2), dtype=torch.float32, device=a.device)
        d.ln_stats_out = stats_out.data_ptr()
    if tile == 0 and split_k == 0:
        d.tile, d.split_k, _ = _autotune(lib, d, ("g", rows, n, k, epilogue, d.dtype, a2 is not None, ln is not None)
                                      + (("f32",) if out_f32 else ()) + (("so",) if ln_stats else ())
                                      + (("si",) if stats_in is not None else ())
                                      + (("hm", head_major[0]) if head_major is not None else ())
                                      + (("res",) if res is not None else ()) + (("acc",) if accumulate else ()),
                                      (rows, n), odt, a.device, warm=(a, a2, res))
    need = lib.dd_gemm_workspace_bytes(ctypes.byref(d))
    if need > 0 or _DBG_STAMP_WS:
        ws = workspace(need, a.device)
        d.ws = ws.data_ptr(); d.ws_bytes = ws.numel() * 4
    if _TIMER is not None:
        _timed_gemm(lib, d, "gemm", " gemm %dx%dx%d" % (rows, n_w, k) if _TIMER.shapes else "", 2.0 * rows * n_w * k,
                    2.0 * (rows * k + n_w * k + rows * n * (1 + (1 if d.res else 0) + (1 if d.accumulate else 0))),
                    rows, n)
    else:
        _native.check(lib.dd_gemm(ctypes.byref(d), _stream()), "gemm")
    if stats_out is not None:
        out._ln_stats = stats_out
    if ln_second is not None:
        out._ln_out = ln_second
    return hm_out if hm_out is not None else out


_THIN_CONV = _os.environ.get("DD_THIN_CONV", "1") != "0"       # A/B switch of dd_conv3x3_thin
# split-K reduce folded into the GroupNorm that consumes the conv's output (dd_groupnorm_splitk): 50 launches fewer per
# step, bit-identical — and 0.5 % SLOWER on the step (84.9 vs 85.4 steps/s over three alternating pairs: the norm then
# reads `split` fp32 slabs with the few workgroups a 28-pixel image gives it, where the separate reduce spreads them over
# the chip), so it is OFF by default; DD_GN_SPLITK=1 turns it on
GN_SPLITK = _os.environ.get("DD_GN_SPLITK", "0") == "1"


LN_OUT_TILE = int(_os.environ.get("DD_LN_OUT_TILE", "0"))


def thin_conv_ok(cin, cout, stride, m):
    return _THIN_CONV and (cin, cout, stride) in ((8, 16, 1), (16, 16, 1), (16, 32, 2), (32, 32, 1), (8, 32, 1), (16, 32, 1),
                                                  (16, 16, 2), (8, 16, 2), (32, 16, 1)) and m <= 65535


def conv3x3(x, w, bias, m, hin, win, *, stride=1, up_size=None, rowvec=None, res=None,
            alpha=1.0, out=None, accumulate=False, epilogue=DD_EPI_NONE, tile=0, split_k=0, gn_next=None):
    """3x3 / pad 1 convolution as an implicit GEMM on an NHWC batch.

    gn_next = (GroupNorm module, silu, want_x): the GroupNorm that reads this conv's output next.  When the conv runs
    split-K (two launches) and the image takes the single-launch GroupNorm, the REDUCE launch is replaced by
    dd_groupnorm_splitk: it adds the slabs, applies the epilogue and normalises in one go; the result is attached as
    `out._gn_cache` (GroupNorm.run picks it up) and `out` itself is written only if want_x.

    x: (m*hin*win, cin); w: (cout, 9*cin) packed [cout][ky][kx][cin]; optional nearest
    upsample of x to `up_size` first; rowvec: (m, cout) per-instance vector (time embedding).
    Returns (m*hout*wout, cout)."""
    lib = _native.load()
    _need_gpu(x, w, bias, res, rowvec, out)
    x = _rows2d(x)
    cin = x.shape[1]
    if not x.is_contiguous() or x.shape[0] != m * hin * win:
        raise ValueError("conv input must be contiguous (m*h*w, cin)")
    hv, wv = (hin, win) if up_size is None else (int(up_size[0]), int(up_size[1]))
    hout = (hv + 2 - 3) // stride + 1
    wout = (wv + 2 - 3) // stride + 1
    cout = w.shape[0]
    if w.shape[1] != 9 * cin or not w.is_contiguous():
        raise ValueError("conv weight must be contiguous [cout, 9*cin]")
    rows = m * hout * wout
    if out is None:
        out = torch.empty((rows, cout), dtype=x.dtype, device=x.device)
    else:
        _forget_derived(out)
    if thin_conv_ok(cin, cout, stride, m) and up_size is None and rowvec is None and res is None and alpha == 1.0 \
            and not accumulate and epilogue in (DD_EPI_NONE, DD_EPI_SILU) and tile == 0 and split_k == 0 \
            and out.is_contiguous():
        # thin channel counts on a large image (the condition embedder's first layers): patch-in-LDS direct conv
        e0 = _TIMER.start() if _TIMER is not None else None
        rc = lib.dd_conv3x3_thin(_ptr(x), _ptr(w), _ptr(bias), _ptr(out), m, hin, win, cin, cout, stride,
                                 int(epilogue == DD_EPI_SILU), _dt(x), _stream())
        _native.check(rc, "conv3x3_thin")
        if _TIMER is not None:
            _TIMER.stop(e0, "dd_conv3x3_thin_kernel" + (" conv %dx%dx%d" % (rows, cout, 9 * cin) if _TIMER.shapes else ""),
                        2.0 * rows * cout * 9 * cin, 2.0 * (x.numel() + w.numel() + rows * cout))
        return out
    d = GemmDesc()
    d.a = x.data_ptr(); d.lda = cin; d.k1 = 9 * cin
    d.rows, d.n, d.k = rows, cout, 9 * cin
    d.w = w.data_ptr()
    d.bias = bias.data_ptr() if bias is not None else None
    if rowvec is not None:
        rowvec = _rows2d(rowvec)
        d.rowvec = rowvec.data_ptr(); d.ld_rowvec = rowvec.stride(0); d.rows_per_inst = hout * wout
    if res is not None:
        res = _rows2d(res)
        d.res = res.data_ptr(); d.ldres = res.stride(0)
    out = _rows2d(out)
    d.out = out.data_ptr(); d.ldc = out.stride(0)
    d.alpha = alpha; d.accumulate = int(accumulate); d.epilogue = epilogue
    d.conv = 1
    d.hin, d.win, d.cin, d.hv, d.wv = hin, win, cin, hv, wv
    d.hout, d.wout, d.stride = hout, wout, stride
    d.dtype = _dt(x); d.tile = tile; d.split_k = split_k
    if tile == 0 and split_k == 0:
        d.tile, d.split_k, _ = _autotune(lib, d, ("c", m, hin, win, cin, cout, stride, hv, wv, d.dtype),
                                      (rows, cout), x.dtype, x.device, warm=(x, res))
    need = lib.dd_gemm_workspace_bytes(ctypes.byref(d))
    if need > 0 or _DBG_STAMP_WS:
        ws = workspace(need, x.device)
        d.ws = ws.data_ptr(); d.ws_bytes = ws.numel() * 4
    if gn_next is not None and GN_SPLITK and need > 0 and alpha == 1.0 and not accumulate and epilogue == DD_EPI_NONE:
        gmod, gsilu, want_x = gn_next
        _, split = _kname(lib, d)
        if split > 1 and lib.dd_groupnorm_is_fused(hout * wout, cout, gmod.num_groups):
            # partial slabs only; dd_groupnorm_splitk is reduce + epilogue + GroupNorm(+SiLU) in one launch
            y = torch.empty((rows, cout), dtype=x.dtype, device=x.device)
            d.phase = 1
            e0 = _TIMER.start() if _TIMER is not None else None
            _native.check(lib.dd_gemm(ctypes.byref(d), _stream()), "conv3x3")
            if e0 is not None:
                name = _kname(lib, d)[0]
                _TIMER.stop(e0, name + (" conv %dx%dx%d" % (rows, cout, 9 * cin) if _TIMER.shapes else ""),
                            2.0 * rows * cout * 9 * cin, 2.0 * (x.numel() + w.numel()) + 4.0 * split * rows * cout,
                            _staged_bytes(lib, d))
                e0 = _TIMER.start()
            rc = lib.dd_groupnorm_splitk(ctypes.c_void_p(ws.data_ptr() + 65536), split, _ptr(bias), _ptr(rowvec),
                                         rowvec.stride(0) if rowvec is not None else 0, _ptr(res),
                                         res.stride(0) if res is not None else 0, _ptr(out) if want_x else None,
                                         _ptr(gmod.weight), _ptr(gmod.bias), _ptr(y), m, hout * wout, cout,
                                         gmod.num_groups, float(gmod.eps), int(gsilu), _dt(x), _stream())
            _native.check(rc, "groupnorm_splitk")
            if e0 is not None:
                _TIMER.stop(e0, "dd_gn_splitk_kernel", 0.0, 4.0 * split * rows * cout + 2.0 * rows * cout * (2 if want_x else 1))
            out._gn_cache = (gmod, bool(gsilu), y)
            if not want_x:
                out._unwritten = True          # nobody but that GroupNorm may read it
            return out
    if _TIMER is not None:
        _timed_gemm(lib, d, "conv3x3", " conv %dx%dx%d" % (rows, cout, 9 * cin) if _TIMER.shapes else "",
                    2.0 * rows * cout * 9 * cin,
                    2.0 * (x.numel() + w.numel() + rows * cout * (1 + (1 if d.res else 0) + (1 if d.accumulate else 0))),
                    rows, cout)
        return out
    _native.check(lib.dd_gemm(ctypes.byref(d), _stream()), "conv3x3")
    return out


def groupnorm(x, gamma, beta, m, hw, groups, eps, silu, x2=None, out=None):
    """GroupNorm (+SiLU) over an NHWC batch; x2 = optional second source concatenated on C."""
    lib = _native.load()
    _need_gpu(x, gamma, beta, x2, out)
    c1 = x.shape[1]
    c2 = x2.shape[1] if x2 is not None else 0
    if not x.is_contiguous() or (x2 is not None and not x2.is_contiguous()):
        raise ValueError("groupnorm inputs must be contiguous")
    if out is None:
        out = torch.empty((m * hw, c1 + c2), dtype=x.dtype, device=x.device)
    need = lib.dd_groupnorm_workspace_bytes(m, groups)
    ws = workspace(need, x.device, "gn")
    e0 = _TIMER.start() if _TIMER is not None else None
    rc = lib.dd_groupnorm_nhwc(_ptr(x), c1, _ptr(x2), c2, _ptr(gamma), _ptr(beta), _ptr(out),
                               m, hw, groups, eps, int(silu), _dt(x), _ptr(ws), ws.numel() * 4,
                               _stream())
    _native.check(rc, "groupnorm")
    if e0 is not None:      # HBM-bound: each element read once and written once
        th = lib.dd_groupnorm_is_fused(hw, c1 + c2, groups)
        _TIMER.stop(e0, "dd_gn_fused_kernel<%s, %d, 8>" % ("f16" if x.dtype == torch.float16 else "bf16", th) if th
                    else "dd_gn_stats_kernel + dd_gn_apply_kernel (2 launches)",
                    0.0, (4.0 if th else 6.0) * out.numel())
    return out


def layernorm(x, gamma, beta, eps=1e-5, out=None):
    lib = _native.load()
    _need_gpu(x, gamma, beta, out)
    x = _rows2d(x)
    if not x.is_contiguous():
        raise ValueError("layernorm input must be contiguous")
    if out is None:
        out = torch.empty_like(x)
    e0 = _TIMER.start() if _TIMER is not None else None
    rc = lib.dd_layernorm(_ptr(x), _ptr(gamma), _ptr(beta), _ptr(out), x.shape[0], x.shape[1],
                          eps, _dt(x), _stream())
    _native.check(rc, "layernorm")
    if e0 is not None:
        _TIMER.stop(e0, "dd_layernorm", 0.0, 4.0 * out.numel())
    return out


def attention(q, k, v, batch, lq, lk, heads, head_dim, scale=None, *, kv_batch_map=None,
              out=None, accumulate=False, variant=0, q_prescaled=False, seq_strides=None,
              out_seq_strides=None, kv_batch_map2=None, kv_seq_strides=None, lk_dev=None):
    """softmax(scale * q k^T) v per (batch, head).

    q / k / v may also be HEAD-MAJOR 3-D tensors (heads, rows, head_dim) — slices of a
    `gemm(..., head_major=...)` result; q_prescaled: q already carries scale * log2(e).

    q: (batch*lq, >= heads*head_dim) row-strided view; k, v: (kv_batches*lk, ...) likewise, so
    slices of a fused QKV projection can be passed without copies.  kv_batch_map (int32 device
    tensor [batch]) redirects batch b to K/V of another batch (neighbour views); with kv_batch_map2 as well the
    result is Attn(q, kv[map]) + Attn(q, kv[map2]) — the neighbour-view PAIR of attn4 in one launch.

    seq_strides = (row_stride, batch_stride) in elements for q, k, v (row-major 2-D views with the same row
    pitch), out_seq_strides likewise for `out` (default: the same): the sequence runs over rows `row_stride`
    apart and consecutive batches start `batch_stride` apart — attention ALONG ANOTHER AXIS of a (frames, tokens, C) activation without a transpose
    (temporal attention: row_stride = tokens_per_frame * C, batch_stride = C).  kv_seq_strides: the same pair for k / v
    when they live in another buffer than q (frame-split temporal attention: local queries, gathered keys).

    lk_dev: int32 device tensor [1] with the number of keys every batch entry REALLY has; `lk` is then the capacity the
    K / V rows were laid out for (dd_attn_desc.lk_dev: the kernel reads the count at its start, so a recorded HIP graph
    serves every context length up to the capacity)."""
    lib = _native.load()
    _need_gpu(q, k, v, out, kv_batch_map, kv_batch_map2, lk_dev)
    if lk_dev is not None and (lk_dev.dtype != torch.int32 or lk_dev.numel() < 1):
        raise TypeError("lk_dev must be an int32 device tensor")
    if kv_batch_map2 is not None and (kv_batch_map is None or seq_strides is not None):
        raise ValueError("kv_batch_map2 needs kv_batch_map and no seq_strides")
    if kv_seq_strides is not None and seq_strides is None:
        raise ValueError("kv_seq_strides needs seq_strides")
    d = AttnDesc()

    def operand(t, l):
        """-> (ptr, ld, batch stride, head stride) for a row-major 2-D view, a head-major 3-D tensor
        (heads, rows, d) or a batch-major 4-D tensor (batches, heads, l, d) with contiguous [l][d] blocks (the
        K/V layout of the view-sharded neighbour attention: whole instances are contiguous messages)."""
        if t.dim() == 4:
            if t.shape[1] != heads or t.shape[2] != l or t.shape[3] != head_dim or t.stride(3) != 1 \
                    or t.stride(2) != head_dim:
                raise ValueError("batch-major operand must be (batches, heads, l, head_dim) with contiguous blocks")
            return t.data_ptr(), head_dim, t.stride(0), t.stride(1)
        if t.dim() == 3:
            if t.shape[0] != heads or t.shape[2] != head_dim or t.stride(2) != 1 or t.stride(1) != head_dim:
                raise ValueError("head-major operand must be (heads, rows, head_dim) with contiguous planes")
            return t.data_ptr(), head_dim, l * head_dim, t.stride(0)
        t = _rows2d(t)
        return t.data_ptr(), t.stride(0), l * t.stride(0), 0
    d.q, d.ldq, d.q_batch_stride, d.q_head_stride = operand(q, lq)
    d.k, d.ldk, d.k_batch_stride, d.k_head_stride = operand(k, lk)
    d.v, d.ldv, d.v_batch_stride, d.v_head_stride = operand(v, lk)
    d.q_prescaled = int(bool(q_prescaled))
    if out is None:
        out = torch.empty((q.shape[0], heads * head_dim) if seq_strides is not None else
                          (batch * lq, heads * head_dim), dtype=q.dtype, device=q.device)
    out = _rows2d(out)
    d.o, d.ldo = out.data_ptr(), out.stride(0)
    d.o_batch_stride = lq * out.stride(0)
    if seq_strides is not None:
        if q.dim() != 2 or k.dim() != 2 or v.dim() != 2 or kv_batch_map is not None:
            raise ValueError("seq_strides takes row-major 2-D q / k / v and no kv_batch_map")
        rs, bs = int(seq_strides[0]), int(seq_strides[1])
        ors, obs = (rs, bs) if out_seq_strides is None else (int(out_seq_strides[0]), int(out_seq_strides[1]))
        krs, kbs = (rs, bs) if kv_seq_strides is None else (int(kv_seq_strides[0]), int(kv_seq_strides[1]))
        d.ldq, d.ldk, d.ldv = rs, krs, krs
        d.q_batch_stride, d.k_batch_stride, d.v_batch_stride = bs, kbs, kbs
        d.ldo, d.o_batch_stride = ors, obs
    d.batch, d.heads, d.head_dim, d.lq, d.lk = batch, heads, head_dim, lq, lk
    d.scale = float(scale) if scale is not None else head_dim ** -0.5
    d.kv_batch_map = kv_batch_map.data_ptr() if kv_batch_map is not None else None
    d.kv_batch_map2 = kv_batch_map2.data_ptr() if kv_batch_map2 is not None else None
    npair = 2 if kv_batch_map2 is not None else 1
    d.accumulate = int(accumulate)
    d.dtype = _dt(q)
    d.variant = variant
    d.lk_dev = lk_dev.data_ptr() if lk_dev is not None else None
    if seq_strides is not None and batch * heads > 65535:
        # the launcher's grid covers at most 65535 (batch, head) pairs: walk the batch in chunks
        es, per = q.element_size(), 65535 // heads
        base = (d.q, d.k, d.v, d.o)
        for b0 in range(0, batch, per):
            d.batch = min(per, batch - b0)
            d.q = base[0] + b0 * d.q_batch_stride * es
            d.k, d.v = base[1] + b0 * d.k_batch_stride * es, base[2] + b0 * d.v_batch_stride * es
            d.o = base[3] + b0 * d.o_batch_stride * es
            _native.check(lib.dd_attention(ctypes.byref(d), _stream()), "attention")
        return out
    if _TIMER is not None:
        e0 = _TIMER.start()
        _native.check(lib.dd_attention(ctypes.byref(d), _stream()), "attention")
        _TIMER.stop(e0, "dd_attn5_kernel<%s,D%d>" % ("f16" if d.dtype == DD_F16 else "bf16", head_dim),
                    4.0 * batch * heads * lq * lk * head_dim * npair,
                    2.0 * heads * head_dim * batch * (2 * lq + 2 * lk * npair))
        return out
    _native.check(lib.dd_attention(ctypes.byref(d), _stream()), "attention")
    return out


_ZERO_BIAS = {}


# dd_xattn320 owns 80-row tiles, one workgroup per CU (135 KB of LDS), and every workgroup re-stages both 205 KB weight
# matrices: it beats the three-launch form while its grid is ONE residency generation (12 instances x 1400 rows = 210
# workgroups on 256 CUs: 40 vs 45 us for the SFA module) and loses beyond it (48 instances = 840 workgroups: 127 vs 114 us,
# profiles/r03_sfa_roofline.txt).  Above this many workgroups the callers take the three launches.
XATTN_MAX_WGS = int(_os.environ.get("DD_XATTN_MAX_WGS", "256"))


def xattn320_ok(c, heads, lk, rows=None):
    """Shapes the fused cross-attention kernel covers (dd_xattn320): the 320-channel level, 8 heads, <= 128 keys — and,
    when the caller states its row count, a grid of at most XATTN_MAX_WGS 80-row workgroups."""
    if rows is not None and (int(rows) + 79) // 80 > XATTN_MAX_WGS:
        return False
    return int(c) == 320 and int(heads) == 8 and 0 < int(lk) <= 128


def xattn_pack_weight(w):
    """[320, 320] Linear weight -> the K-step-major, pre-swizzled 1-D copy dd_xattn320 streams by linear LDS-DMA
    (dd_xattn_pack_weight).  NOT cached here: the owning module keeps the copy as a `_pk_*` entry beside its other packed
    weights (layers.Linear.wx), so it lives exactly as long as the module, is dropped by `_drop_cache` on .to() /
    load_state_dict, and can never be evicted under a captured HIP graph that holds its address (a
    process-global, size-capped cache could free buffers a graph still replays from)."""
    if tuple(w.shape) != (320, 320) or not w.is_contiguous():
        raise ValueError("xattn320 weights must be contiguous (320, 320)")
    _need_gpu(w)
    lib = _native.load()
    packed = torch.empty(320 * 320, dtype=w.dtype, device=w.device)
    _native.check(lib.dd_xattn_pack_weight(_ptr(w), _ptr(packed), _dt(w), _stream()), "xattn_pack_weight")
    return packed


def _xpacked(w):
    """Packed weights are 1-D (102400,); a raw (320, 320) weight is packed on the spot (uncached: a fresh buffer, which
    inside a capture belongs to the graph's own pool)."""
    return w if w.dim() == 1 and w.numel() == 320 * 320 else xattn_pack_weight(w)


def xattn320(x, wq, wo, bo, k, v, instances, rows_per_inst, lk, scale, *, res=None, ln_out=None, out=None, lk_dev=None):
    """Fused q-projection -> attention over the lk context keys of each view-instance -> out-projection + bias +
    residual, one launch (include/dualdiff_hip.h: dd_xattn320).  x: (instances * rows_per_inst, 320); k / v: either
    (instances * lk, >= 320) row-strided 2-D views (column slices of a wider projection) or HEAD-MAJOR 3-D tensors
    (8, instances * lk, 40) — slices of a `gemm(..., head_major=(40, 0, 1.0))` result, the form the kernel streams
    fastest; wq / wo: the packed copies of the (320, 320) Linear weights (xattn_pack_weight; layers.Linear.wx) or the raw weights; ln_out = (gamma, beta, eps):
    LayerNorm(out) comes back as `out._ln_out`.  lk_dev: as in attention() — the real key count in device memory, `lk`
    the capacity of the K / V layout."""
    lib = _native.load()
    _need_gpu(x, wq, wo, bo, k, v, res, out, lk_dev)
    if lk_dev is not None and (lk_dev.dtype != torch.int32 or lk_dev.numel() < 1):
        raise TypeError("lk_dev must be an int32 device tensor")
    x = _rows2d(x)
    rows = instances * rows_per_inst
    if x.shape != (rows, 320):
        raise ValueError("xattn320: x must be (instances * rows_per_inst, 320)")

    def kv_operand(t):
        if t.dim() == 3:                      # head-major planes
            if t.shape[0] != 8 or t.shape[1] != instances * lk or t.shape[2] != 40 or t.stride(2) != 1 or t.stride(1) != 40:
                raise ValueError("head-major K / V must be (8, instances * lk, 40) with contiguous planes")
            return t.data_ptr(), 40, lk * 40, t.stride(0)
        t = _rows2d(t)
        if t.shape[0] != instances * lk or t.shape[1] < 320:
            raise ValueError("row-major K / V must be (instances * lk, >= 320)")
        return t.data_ptr(), t.stride(0), lk * t.stride(0), 40
    if out is None:
        out = torch.empty((rows, 320), dtype=x.dtype, device=x.device)
    else:
        _forget_derived(out)
    out = _rows2d(out)
    wqp, wop = _xpacked(wq), _xpacked(wo)
    d = XAttnDesc()
    d.x, d.ldx = x.data_ptr(), x.stride(0)
    if res is not None:
        res = _rows2d(res)
        d.res, d.ldres = res.data_ptr(), res.stride(0)
    d.wq, d.wo = wqp.data_ptr(), wop.data_ptr()
    if bo is None:                      # the kernel loads the bias unconditionally (a conditional load would be waited for in place)
        key = (x.dtype, x.device)
        if key not in _ZERO_BIAS:
            _ZERO_BIAS[key] = torch.zeros(320, dtype=x.dtype, device=x.device)
        bo = _ZERO_BIAS[key]
    d.bo = bo.data_ptr()
    d.k, d.ldk, d.k_inst_stride, d.k_head_stride = kv_operand(k)
    d.v, d.ldv, d.v_inst_stride, d.v_head_stride = kv_operand(v)
    d.out, d.ldo = out.data_ptr(), out.stride(0)
    d.instances, d.rows_per_inst, d.lk = int(instances), int(rows_per_inst), int(lk)
    d.channels, d.heads, d.scale, d.dtype = 320, 8, float(scale), _dt(x)
    d.lk_dev = lk_dev.data_ptr() if lk_dev is not None else None
    second = None
    if ln_out is not None:
        g_, b_, eps_ = ln_out
        _need_gpu(g_, b_)
        second = torch.empty((rows, 320), dtype=x.dtype, device=x.device)
        d.ln_out, d.ld_ln_out = second.data_ptr(), 320
        d.ln_gamma, d.ln_beta, d.ln_eps = g_.data_ptr(), b_.data_ptr(), float(eps_)
    e0 = _TIMER.start() if _TIMER is not None else None
    _native.check(lib.dd_xattn320(ctypes.byref(d), _stream()), "xattn320")
    if e0 is not None:      # 2 projections + attention; bytes: x in, out out (+ res, + ln_out), weights, K / V
        _TIMER.stop(e0, "dd_xattn320_kernel", 4.0 * rows * 320 * 320 + 4.0 * rows * lk * 320,
                    2.0 * (rows * 320 * (2 + (1 if res is not None else 0) + (1 if second is not None else 0))
                           + 2 * 320 * 320 + 2 * instances * lk * 320))
    if second is not None:
        out._ln_out = second
    return out


def add(a, b, c=None, out=None):
    lib = _native.load()
    _need_gpu(a, b, c, out)
    if out is None:
        out = torch.empty_like(a)
    else:
        _forget_derived(out)
    e0 = _TIMER.start() if _TIMER is not None else None
    rc = lib.dd_add(_ptr(a), _ptr(b), _ptr(c), _ptr(out), a.numel(), _dt(a), _stream())
    _native.check(rc, "add")
    if e0 is not None:
        _TIMER.stop(e0, "dd_add", 0.0, 2.0 * a.numel() * (3 + (1 if c is not None else 0)))
    return out


def scale(x, s, out=None):
    lib = _native.load()
    _need_gpu(x, out)
    if out is None:
        out = torch.empty_like(x)
    _native.check(lib.dd_scale(_ptr(x), _ptr(out), s, x.numel(), _dt(x), _stream()), "scale")
    return out


def silu(x, out=None):
    lib = _native.load()
    _need_gpu(x, out)
    if out is None:
        out = torch.empty_like(x)
    _native.check(lib.dd_silu(_ptr(x), _ptr(out), x.numel(), _dt(x), _stream()), "silu")
    return out


def nchw_to_nhwc(x, c_pad=None, views=1):
    """(m, c, h, views * w) contiguous -> (m * views * h * w, c_pad) NHWC rows with zero-padded channels (default: no
    padding); views > 1 also splits a panorama into its views (map_embedder.py:116-125).  c_pad % 8 == 0: one LDS-tiled
    launch, both sides coalesced (csrc/tokens.hip); other widths (views == 1 only): the element-wise kernel."""
    lib = _native.load()
    _need_gpu(x)
    m, c, h, wt = x.shape
    if wt % views:
        raise ValueError("nchw_to_nhwc: width %d is not %d views" % (wt, views))
    c_pad = c if c_pad is None else c_pad
    x = x.contiguous()
    out = torch.empty((m * views * h * (wt // views), c_pad), dtype=x.dtype, device=x.device)
    if c_pad % 8 == 0:
        rc = lib.dd_nchw_to_nhwc_views(_ptr(x), _ptr(out), m, c, h, wt // views, views, c_pad, _dt(x), _stream())
    elif views == 1:
        rc = lib.dd_nchw_to_nhwc(_ptr(x), _ptr(out), m, c, h * wt, c_pad, _dt(x), _stream())
    else:
        raise ValueError("nchw_to_nhwc: a view split needs c_pad % 8 == 0")
    _native.check(rc, "nchw_to_nhwc")
    return out


def nhwc_to_nchw(x, m, c, h, w):
    """(m*h*w, ld>=c) -> (m, c, h, w) contiguous."""
    lib = _native.load()
    _need_gpu(x)
    x = _rows2d(x)
    out = torch.empty((m, c, h, w), dtype=x.dtype, device=x.device)
    rc = lib.dd_nhwc_to_nchw(_ptr(x), _ptr(out), m, c, h * w, x.stride(0), _dt(x), _stream())
    _native.check(rc, "nhwc_to_nchw")
    return out


def timestep_embedding(t, dim, dtype, flip_sin_to_cos=True, freq_shift=0.0, out=None):
    """t: fp32 device tensor [n] -> (n, dim) sinusoidal embedding in `dtype`."""
    lib = _native.load()
    _need_gpu(t, out)
    if t.dtype != torch.float32:
        raise TypeError("timesteps must be fp32")
    n = t.numel()
    if out is None:
        out = torch.empty((n, dim), dtype=dtype, device=t.device)
    rc = lib.dd_timestep_embedding(_ptr(t), _ptr(out), n, dim, int(flip_sin_to_cos), freq_shift,
                                   _dt(out), _stream())
    _native.check(rc, "timestep_embedding")
    return out


def conv3x3_small_cout(x, w, bias, m, h, wd, out=None):
    """conv_out: x (m*h*wd, cin) NHWC, w (cout<=8, 9*cin) -> (m, cout, h, wd) NCHW."""
    lib = _native.load()
    _need_gpu(x, w, bias, out)
    cin = x.shape[1]
    cout = w.shape[0]
    if out is None:
        out = torch.empty((m, cout, h, wd), dtype=x.dtype, device=x.device)
    rc = lib.dd_conv3x3_small_cout(_ptr(x), _ptr(w), _ptr(bias), _ptr(out), m, h, wd, cin, cout,
                                   _dt(x), _stream())
    _native.check(rc, "conv3x3_small_cout")
    return out


def cfg_ddim_step(eps, x, coef, guidance, x_out=None, x_dup=None):
    """eps: (2, n...) uncond first; x: (n...); coef: fp32 device [4]."""
    lib = _native.load()
    _need_gpu(eps, x, coef, x_out, x_dup)
    if x_out is None:
        x_out = torch.empty_like(x)
    rc = lib.dd_cfg_ddim_step(_ptr(eps), _ptr(x), _ptr(x_out), _ptr(x_dup), _ptr(coef),
                              float(guidance), x.numel(), _dt(x), _stream())
    _native.check(rc, "cfg_ddim_step")
    return x_out


def cfg_unipc_step(eps, x, hist, coef, guidance, x_out=None, x_dup=None):
    """eps: (2, n...) uncond first; x: (n...); hist: fp32 (3, n...) = [last, m1, m2] updated in place;
    coef: fp32 device [10] (dualdiff_amd.pipeline.schedulers.unipc_schedule row)."""
    lib = _native.load()
    _need_gpu(eps, x, hist, coef, x_out, x_dup)
    if hist.dtype != torch.float32 or hist.shape[0] != 3 or hist[0].numel() != x.numel() or not hist.is_contiguous():
        raise ValueError("hist must be a contiguous fp32 (3, n...) tensor")
    if x_out is None:
        x_out = torch.empty_like(x)
    rc = lib.dd_cfg_unipc_step(_ptr(eps), _ptr(x), _ptr(x_out), _ptr(x_dup), _ptr(hist[0]), _ptr(hist[1]),
                               _ptr(hist[2]), _ptr(coef), float(guidance), x.numel(), _dt(x), _stream())
    _native.check(rc, "cfg_unipc_step")
    return x_out


def softmax_rows(s, dtype, pad_to=8):
    """fp32 logits (rows, cols) -> probabilities in `dtype`, (rows, cols rounded up to `pad_to`) with zero
    padding columns (the K dimension of the following P @ V GEMM must be a multiple of 8)."""
    lib = _native.load()
    _need_gpu(s)
    if s.dtype != torch.float32 or s.dim() != 2 or s.stride(1) != 1:
        raise ValueError("softmax_rows takes a row-major fp32 matrix")
    rows, cols = s.shape
    ldp = (cols + pad_to - 1) // pad_to * pad_to
    p = torch.empty((rows, ldp), dtype=dtype, device=s.device)
    code = DD_F16 if dtype == torch.float16 else DD_BF16
    _native.check(lib.dd_softmax_rows(_ptr(s), _ptr(p), rows, cols, s.stride(0), ldp, code, _stream()), "softmax_rows")
    return p


def ors_project(occ, origin, direction, samples, step=0.2, *, want_labels=True, cond_dtype=None,
                keep_fg=True, keep_bg=True):
    """ORS ray sampling (include/dualdiff_hip.h: dd_ors_project).  occ: (200, 200, 16) uint8 on the GPU;
    origin (n, 3), direction (n, hw, 3) fp32.  Returns (labels (n, hw, samples) uint8 | None,
    cond (n, samples, hw) cond_dtype | None)."""
    lib = _native.load()
    _need_gpu(occ, origin, direction)
    if occ.dtype != torch.uint8 or tuple(occ.shape) != (200, 200, 16) or not occ.is_contiguous():
        raise ValueError("occ must be a contiguous (200, 200, 16) uint8 volume")
    origin = origin.to(torch.float32).contiguous()
    direction = direction.to(torch.float32).contiguous()
    n, hw = direction.shape[0], direction.shape[1]
    labels = torch.empty((n, hw, samples), dtype=torch.uint8, device=occ.device) if want_labels else None
    cond = torch.empty((n, samples, hw), dtype=cond_dtype, device=occ.device) if cond_dtype is not None else None
    code = DD_F16 if cond_dtype in (None, torch.float16) else DD_BF16
    if cond_dtype not in (None, torch.float16, torch.bfloat16):
        raise TypeError("cond_dtype must be fp16 / bf16")
    _native.check(lib.dd_ors_project(_ptr(occ), _ptr(origin), _ptr(direction), _ptr(labels), _ptr(cond), n, hw,
                                     int(samples), float(step), int(keep_fg), int(keep_bg), code, _stream()),
                  "ors_project")
    return labels, cond


_FDT = {torch.float16: 0, torch.bfloat16: 1, torch.float32: 2}


def fourier_embed(x, freqs, include_input=True):
    """[x, sin(f0 x), cos(f0 x), ...] on the last dim (networks/embedder.py), one kernel."""
    lib = _native.load()
    _need_gpu(x)
    code = _FDT.get(x.dtype)
    if code is None:
        raise TypeError("fourier_embed takes fp16 / bf16 / fp32 tensors, got %s" % x.dtype)
    x = x.contiguous()
    dims = x.shape[-1]
    rows = x.numel() // dims
    nf = len(freqs)
    out = torch.empty(x.shape[:-1] + (dims * ((1 if include_input else 0) + 2 * nf),), dtype=x.dtype, device=x.device)
    arr = (ctypes.c_float * nf)(*[float(f) for f in freqs])
    _native.check(lib.dd_fourier_embed(_ptr(x), _ptr(out), rows, dims, arr, nf, int(include_input), code, code,
                                       _stream()), "fourier_embed")
    return out


def camera_features(camera_param, freqs, include_input, dtype, k_pad):
    """(b, n, 3, j) camera parameters -> (b * n, k_pad) Fourier features of the j column vectors, back to back, in
    `dtype`, zero-padded from j * 3 * (include_input + 2 F) to k_pad columns — `cam_embedder(camera_param.permute(0, 1,
    3, 2)).reshape(b, n, -1)` of unet_addon_rawbox.py:308-325 plus the K padding of cam2token, in ONE launch (the
    transposed read and the padding are strides of the kernel)."""
    lib = _native.load()
    _need_gpu(camera_param)
    code = _FDT.get(camera_param.dtype)
    if code is None or camera_param.dim() != 4:
        raise TypeError("camera_features takes a (b, n, 3, j) fp16 / bf16 / fp32 tensor")
    x = camera_param.contiguous()
    b, n, dims, j = x.shape
    nf = len(freqs)
    width = dims * ((1 if include_input else 0) + 2 * nf)
    if not j * width <= k_pad <= j * width + dims:
        raise ValueError("camera_features: k_pad %d does not fit %d features" % (k_pad, j * width))
    out = torch.empty((b * n, k_pad), dtype=dtype, device=x.device)
    arr = (ctypes.c_float * nf)(*[float(f) for f in freqs])
    rc = lib.dd_fourier_embed_strided(_ptr(x), _ptr(out), b * n * j, dims, arr, nf, int(include_input), code, _FDT[dtype],
                                      j, dims * j, 1, j, k_pad, _stream())
    _native.check(rc, "camera_features")
    return out


def box_tokens(points, classes, masks, class_tokens, null_pos, null_class, freqs, include_input, pos, cat, cls_offset,
               cls_out=None, normalize=None):
    """Both operands of the box MLP in one launch (include/dualdiff_hip.h: dd_box_tokens; bbox_embedder.py:164-203).
    points (rows, P, 3) fp16 / bf16 / fp32; classes (rows,) int64; masks (rows,) bool or None; pos (rows, P * 3 * (inc + 2
    F)) and cat (rows, >= cls_offset + D) are written in place (dtype of class_tokens); normalize = (xyz_min, xyz_range)."""
    lib = _native.load()
    _need_gpu(points, classes, masks, class_tokens, null_pos, null_class, pos, cat, cls_out)
    rows, npts = points.shape[0], points.shape[1]
    d = _native.BoxTokensDesc()
    points, classes = points.contiguous(), classes.contiguous()
    if classes.dtype != torch.int64:
        classes = classes.long()
    if masks is not None:
        masks = masks.contiguous()
        masks = masks.view(torch.uint8) if masks.dtype == torch.bool else (masks != 0).view(torch.uint8)
    code = _FDT.get(points.dtype)
    if code is None or points.shape[2] != 3 or not pos.is_contiguous() or cat.stride(1) != 1:
        raise TypeError("box_tokens: points (rows, P, 3) in fp16 / bf16 / fp32, pos contiguous, cat row-major")
    nf = len(freqs)
    if tuple(pos.shape) != (rows, npts * 3 * ((1 if include_input else 0) + 2 * nf)) or pos.dtype != class_tokens.dtype \
            or cat.dtype != class_tokens.dtype or cat.shape[0] != rows:
        raise ValueError("box_tokens: operand shapes / dtypes do not match")
    d.points, d.classes, d.masks = points.data_ptr(), classes.data_ptr(), (masks.data_ptr() if masks is not None else None)
    d.class_tokens, d.null_pos, d.null_class = class_tokens.data_ptr(), null_pos.data_ptr(), null_class.data_ptr()
    d.pos, d.cat, d.cls_out = pos.data_ptr(), cat.data_ptr(), (cls_out.data_ptr() if cls_out is not None else None)
    d.rows, d.points_per_box, d.num_freqs, d.include_input = rows, npts, nf, int(include_input)
    d.class_token_dim, d.cls_offset, d.ld_cat = class_tokens.shape[1], int(cls_offset), cat.stride(0)
    d.n_classes = class_tokens.shape[0]          # kept rows with a class outside the table get NaN tokens, never a stray read
    d.points_dtype, d.dtype = code, _dt(class_tokens)
    for i, f in enumerate(freqs):
        d.freqs[i] = float(f)
    if normalize is not None:
        d.normalize = 1
        for i in range(3):
            d.xyz_min[i], d.xyz_range[i] = float(normalize[0][i]), float(normalize[1][i])
    _native.check(lib.dd_box_tokens(ctypes.byref(d), _stream()), "box_tokens")


def ctx_assemble(cam, text, box, n_cam, text_per_view=False, want_txt=True):
    """[cam_i | text | box tokens] per view-instance, and the text tokens alone (unet_addon_rawbox.py:337-361, :1007,
    :977) in one gather-copy.  cam (m, D); text (scenes, L, D) or (m, L, D) with text_per_view; box (scenes * v, N, D)
    with v in {n_cam, 1}, or None.  Returns (full (m, 1 + L + N, D), txt (m, L, D) or None)."""
    lib = _native.load()
    _need_gpu(cam, text, box)
    m, dim = cam.shape
    lt = text.shape[1]
    scenes = m // n_cam
    nbox = 0 if box is None else box.shape[1]
    box_views = n_cam if box is None else box.shape[0] // scenes
    if text.shape[0] != (m if text_per_view else scenes) or (box is not None and box.shape[0] not in (scenes, m)):
        raise ValueError("ctx_assemble: text / box batch does not match %d scenes x %d views" % (scenes, n_cam))
    cam, text = cam.contiguous(), text.contiguous()
    box = None if box is None else box.contiguous()
    full = torch.empty((m, 1 + lt + nbox, dim), dtype=cam.dtype, device=cam.device)
    txt = torch.empty((m, lt, dim), dtype=cam.dtype, device=cam.device) if want_txt else None
    rc = lib.dd_ctx_assemble(_ptr(cam), _ptr(text), _ptr(box), _ptr(full), _ptr(txt), m, n_cam, lt, nbox, dim,
                             int(text_per_view), box_views, _dt(cam), _stream())
    _native.check(rc, "ctx_assemble")
    return full, txt


def quantize_fp8(w):
    """[n, k] weight matrix -> (float8_e4m3fn [n, k], fp32 scale [n]): symmetric per-output-channel quantisation,
    scale = max|w_row| / 448 (the e4m3fn maximum), round-to-nearest (what quantize_fp8_padded() builds on for gemm8)."""
    wf = w.detach().float()
    scale = (wf.abs().amax(dim=1).clamp_min(1e-12) / 448.0).contiguous()
    q = (wf / scale[:, None]).to(torch.float8_e4m3fn).contiguous()
    return q, scale


def _kpad(k):
    return (int(k) + 127) // 128 * 128


def quantize_fp8_padded(w):
    """[n, k] weight -> (float8_e4m3fn [n, k rounded up to 128] with zero padding, fp32 scale [n]): the W8 operand of
    gemm8 (per-output-channel symmetric quantisation, the values quantize_fp8 produces)."""
    q, scale = quantize_fp8(w)
    kp = _kpad(w.shape[1])
    if kp != w.shape[1]:
        qp = torch.zeros((w.shape[0], kp), dtype=torch.uint8, device=w.device)
        qp[:, :w.shape[1]] = q.view(torch.uint8)
        q = qp.view(torch.float8_e4m3fn)
    return q.contiguous(), scale


def rowquant_fp8(x, norm=None):
    """Per-row e4m3fn quantisation of the activations (dd_rowquant_fp8), optionally behind LayerNorm `norm =
    (gamma, beta, eps)` in the same launch -> (float8_e4m3fn [rows, k rounded up to 128], fp32 scale [rows])."""
    lib = _native.load()
    _need_gpu(x)
    x = _rows2d(x)
    if not x.is_contiguous():
        raise ValueError("rowquant_fp8 input must be contiguous")
    rows, c = x.shape
    kp = _kpad(c)
    q = torch.empty((rows, kp), dtype=torch.uint8, device=x.device)
    scale = torch.empty(rows, dtype=torch.float32, device=x.device)
    g_, b_, eps_ = norm if norm is not None else (None, None, 0.0)
    e0 = _TIMER.start() if _TIMER is not None else None
    _native.check(lib.dd_rowquant_fp8(_ptr(x), _ptr(g_), _ptr(b_), _ptr(q), _ptr(scale), rows, c, kp, float(eps_), _dt(x),
                                      _stream()), "rowquant_fp8")
    if e0 is not None:
        _TIMER.stop(e0, "dd_rowquant_fp8_kernel", 0.0, 2.0 * rows * c + rows * kp)
    return q.view(torch.float8_e4m3fn), scale


def gemm8(a8, a_scale, w8, w_scale, bias=None, *, res=None, out=None, head_major=None, dtype=torch.float16, geglu=False):
    """W8A8 projection on the fp8 matrix path (include/dualdiff_hip.h: dd_gemm8): a8 (rows, Kp) / w8 (n, Kp)
    float8_e4m3fn with rows zero-padded to Kp % 128 == 0, fp32 per-row / per-output-channel scales.  head_major as in
    gemm(): the result comes back as (n / D, rows, D) planes, the first `scaled_planes` multiplied by `scale`."""
    lib = _native.load()
    _need_gpu(a8, a_scale, w8, w_scale, bias, res, out)
    if a8.dtype != torch.float8_e4m3fn or w8.dtype != torch.float8_e4m3fn or a8.shape[1] != w8.shape[1] \
            or a8.shape[1] % 128 or not a8.is_contiguous() or not w8.is_contiguous():
        raise ValueError("gemm8 operands must be contiguous float8_e4m3fn with a common K padded to a multiple of 128")
    rows, n = a8.shape[0], w8.shape[0]
    if geglu:
        if n % 2 or res is not None or head_major is not None:
            raise ValueError("geglu: w8 has 2n rows (h | g), no residual, no head-major output")
    if a_scale.numel() != rows or w_scale.numel() != n or a_scale.dtype != torch.float32 or w_scale.dtype != torch.float32:
        raise ValueError("gemm8 scales must be fp32 [rows] / [n]")
    hm_out = None
    d = Gemm8Desc()
    if head_major is not None:
        hd, hplanes, hscale = head_major
        if out is not None or res is not None or n % hd or hd % 4:
            raise ValueError("head_major needs its own output, no residual and n % D == 0")
        hm_out = torch.empty((n // hd, rows, hd), dtype=dtype, device=a8.device)
        out = hm_out.view(rows, n)
        d.out_headmajor_d, d.hm_scaled_planes, d.hm_scale = int(hd), int(hplanes), float(hscale)
    if out is None:
        out = torch.empty((rows, n // 2 if geglu else n), dtype=dtype, device=a8.device)
    else:
        _forget_derived(out)
    out = _rows2d(out)
    d.a, d.a_scale, d.lda = a8.data_ptr(), a_scale.data_ptr(), a8.stride(0)
    d.w, d.w_scale, d.ldw = w8.data_ptr(), w_scale.data_ptr(), w8.stride(0)
    d.bias = bias.data_ptr() if bias is not None else None
    if res is not None:
        res = _rows2d(res)
        d.res, d.ldres = res.data_ptr(), res.stride(0)
    d.out, d.ldc = out.data_ptr(), out.stride(0)
    if geglu:
        n //= 2
        d.geglu = 1
    d.rows, d.n, d.k_padded = rows, n, a8.shape[1]
    d.dtype = _dt(out)
    e0 = _TIMER.start() if _TIMER is not None else None
    _native.check(lib.dd_gemm8(ctypes.byref(d), _stream()), "gemm8")
    if e0 is not None:
        _TIMER.stop(e0, "dd_gemm8_kernel", 2.0 * rows * n * a8.shape[1],
                    1.0 * (rows + n) * a8.shape[1] + 2.0 * rows * n * (1 + (1 if res is not None else 0)))
    return hm_out if hm_out is not None else out


def gemm_kernel_name(rows, n, k, dtype=torch.bfloat16, conv=False, cin=0, hw=(0, 0)):
    """Name/plan string of the kernel dd_gemm picks for a shape (profile matching)."""
    lib = _native.load()
    d = GemmDesc()
    dummy = 16
    d.a = d.w = d.out = dummy
    d.rows, d.n, d.k, d.k1 = rows, n, k, k
    d.lda, d.ldc = k, n
    d.alpha = 1.0
    d.dtype = DD_F16 if dtype == torch.float16 else DD_BF16
    if conv:
        d.conv = 1
        d.cin = cin; d.hin = d.hv = d.hout = hw[0]; d.win = d.wv = d.wout = hw[1]; d.stride = 1
    return lib.dd_gemm_kernel_name(ctypes.byref(d)).decode()
