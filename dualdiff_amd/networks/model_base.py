"""Model-level plumbing shared by the UNet and the ControlNet: the slice of diffusers'
`ModelMixin` / `ConfigMixin` surface that the reference's runner, pipeline and build_pipe touch
(SURVEY.md §8b "B2 — model surface"): `.config`, `.dtype`, `.device`, `from_pretrained` /
`save_pretrained` on diffusers-layout folders, attention-processor registry, and accepted-but-
no-op memory knobs (`enable_xformers_memory_efficient_attention`, gradient checkpointing).
"""
import json
import os

import torch
import torch.nn as nn

import weakref

from . import layers as _layers
from .layers import HIPAttnProcessor, TimeEmbProjBank

# HIP graphs behind the public forward() surfaces (round 5).  DD_GRAPH_FORWARD=0 switches them off process-wide,
# `model.graph_forward = False` per model.
GRAPH_FORWARD = os.environ.get("DD_GRAPH_FORWARD", "1") != "0"
# Context laid out at a bucket capacity, real length in device memory (layers.context_keys); DD_VARLEN_CONTEXT=0: every
# context length is its own shape (and its own forward graph), as in round 5.
VARLEN_CONTEXT = os.environ.get("DD_VARLEN_CONTEXT", "1") != "0"
# data_ptr -> static output tensor of a live forward graph: a caller that hands such a tensor straight to the next
# model (ControlNet residuals / tokens -> UNet, as pipeline_bev_controlnet.py:476-484 does with one branch) is read in
# place instead of through a copy.  Weak: the entries die with the graph that owns the buffers.
_STATIC_OUT = weakref.WeakValueDictionary()
# Sibling overlap (round 6): two models whose forward() calls follow each other on one stream with inputs that were all
# ready before the FIRST call run concurrently (see sibling_overlap below).  DD_SIBLING_OVERLAP=0 switches it off.
SIBLING_OVERLAP = os.environ.get("DD_SIBLING_OVERLAP", "1") != "0"
# storages of forward-graph OUTPUT buffers: a replay rewrites them without any version count (`graph_forward = "alias"` hands
# them to the caller), so they can never be vouched for
_STATIC_STORAGE = set()
_SIB = __import__("threading").local()           # .window: the latest forward() entry on this thread
_SIB_HOLD_BYTES = (64 << 20, 192 << 20)          # per tensor / per entry: larger inputs are not held (and never proven ready)


def sibling_barrier():
    """Forget the latest forward() entry of this thread: called by code of this package that rewrites tensors through raw
    pointers outside the public forward()s (the fused sampler's step), so that nothing it touched can be vouched for."""
    _SIB.window = None


def _flat_tensors(obj, out):
    if torch.is_tensor(obj):
        out.append(obj)
    elif isinstance(obj, dict):
        for v in obj.values():
            _flat_tensors(v, out)
    elif isinstance(obj, (list, tuple)):
        for v in obj:
            _flat_tensors(v, out)
    return out


def _ident(t):
    """(where the bytes are, how they are viewed), version — or None when that cannot be known (inference tensors do not
    count versions; non-CUDA tensors)."""
    try:
        if not t.is_cuda:
            return None
        if t.untyped_storage().data_ptr() in _STATIC_STORAGE:
            return None
        return (t.untyped_storage().data_ptr(), t.storage_offset(), tuple(t.shape), tuple(t.stride()), t.dtype), t._version
    except RuntimeError:
        return None


class _Seen:
    """The tensor arguments of one forward() entry: key -> (the tensor, held so that its memory cannot be re-used for
    something else while this record lives; its version at that moment), the event recorded on the caller's stream at
    that moment and that stream's handle."""

    def __init__(self, tensors, owner, stream):
        self.items = {}
        held = 0
        for t in tensors:
            idv, nbytes = _ident(t), t.numel() * t.element_size()
            if idv is None or nbytes > _SIB_HOLD_BYTES[0] or held + nbytes > _SIB_HOLD_BYTES[1]:
                continue                         # cannot be vouched for later (the ControlNet residuals of a UNet call, say)
            held += nbytes
            self.items[idv[0]] = (t, idv[1])
        self.owner = weakref.ref(owner)
        self.stream = (stream.device_index, stream.cuda_stream)
        self.event = torch.cuda.Event()
        self.event.record(stream)

    def vouches(self, t):
        idv = _ident(t)
        if idv is None:
            return False
        held = self.items.get(idv[0])
        return held is not None and held[1] == idv[1] and held[0]._version == idv[1]


def sibling_overlap(fwd):
    """Decorator of the two public forward()s.  The reference's sampler calls its ControlNet branches one after the other
    and only then the UNet (pipeline_bev_controlnet.py:405-431, 476-484); the branches do not depend on each other, but a
    drop-in forward() has no way to say so — every call is enqueued behind the previous one on the caller's stream (the
    fused sampler of this package overlaps them on three streams; the drop-in loop ran at its one-stream rate, 0.83).
    Here each forward() entry records (event on the caller's stream, its tensor arguments with their versions).  A call
    of ANOTHER model whose every tensor argument is vouched for — the same bytes, view and version as an argument of the
    previous entry on this stream, or of this model's own previous call on it — needs nothing that was enqueued after
    that entry: it runs on the model's side stream behind that entry's event, i.e. concurrently with the previous model,
    and the caller's stream waits for it before forward() returns (outputs are recorded on the caller's stream for the
    allocator).  The version count is the proof that nothing rewrote an argument since: memory that is rewritten WITHOUT
    one — the output buffers of a forward graph handed out by `graph_forward = "alias"` (_STATIC_STORAGE); a foreign
    extension writing through raw pointers, which this cannot see — must not be relied on; the former is refused.
    Anything else — a new tensor (say, one computed from the sibling's outputs), an in-place update since (version),
    tensors too large to hold, inference tensors, alias outputs, eager or sharded models, a capture in progress — takes
    the ordinary path.  Results are bit-identical either way (tests/test_forward_graphs_gpu.py)."""
    import functools

    @functools.wraps(fwd)
    def wrapper(self, *args, **kwargs):
        if not (SIBLING_OVERLAP and GRAPH_FORWARD and self.graph_forward is True) or not torch.cuda.is_available() \
                or torch.cuda.is_current_stream_capturing() or self._graphs() is None:
            _SIB.window = None
            return fwd(self, *args, **kwargs)
        # (host tensors — a CPU timestep, say — are read at call time: nothing to prove)
        tensors = [t for t in _flat_tensors((args, kwargs), []) if t.is_cuda]
        cur = torch.cuda.current_stream()
        entry = _Seen(tensors, self, cur)
        here = (cur.device_index, cur.cuda_stream)
        window, mine = getattr(_SIB, "window", None), self.__dict__.get("_sib_prev")
        _SIB.window = entry
        self.__dict__["_sib_prev"] = entry
        ok = window is not None and window.stream == here and window.owner() is not self \
            and window.owner() is not None and len(tensors) > 0
        if ok:
            mine_ok = mine is not None and mine.stream == here
            ok = all(window.vouches(t) or (mine_ok and mine.vouches(t)) for t in tensors)
        if not ok:
            return fwd(self, *args, **kwargs)
        side = self.__dict__.get("_sib_stream")
        if side is None:
            side = self.__dict__["_sib_stream"] = torch.cuda.Stream()
        side.wait_event(window.event)
        from .. import ops as _ops
        with torch.cuda.stream(side), _ops.workspace_owner(("sibling", id(self))):   # its eager launches: own scratch too
            out = fwd(self, *args, **kwargs)
        cur.wait_stream(side)
        res = out.to_tuple() if hasattr(out, "to_tuple") else (out.sample if hasattr(out, "sample") else out)
        for t in _flat_tensors(res, []):
            t.record_stream(cur)
        self.__dict__["_sib_overlapped"] = self.__dict__.get("_sib_overlapped", 0) + 1
        return out

    return wrapper


class ForwardGraphs:
    """One HIP graph per (input shapes / dtypes / strides, scalar arguments, attribute flags) of a model's public
    forward().  The reference's callers (`pipeline_bev_controlnet.py:405-446,476-484`, `val_set_gen.py`, the runner's
    validation loop) call `controlnet(...)` / `unet(...)` through `forward()`; launched eagerly that is ~270 (ControlNet)
    / ~290 (UNet) kernel launches from Python per call.  Here the first call with a new key runs eagerly; the second one
    runs once more eagerly on the capture stream (tile tuning, weight packing, workspace sizing), records the same code
    into a graph on STATIC input buffers, and every later call copies its inputs into those buffers (skipped when the caller hands in the static
    output of another forward graph) and replays.

    OUTPUTS (round 6): by default the caller gets its OWN copies (one multi-tensor copy per call), like the
    fresh tensors the reference returns — a caller that keeps the results of two same-shaped calls (separate uncond /
    cond passes, two conditions through one net, a step-to-step comparison) must not find call 1's tensors holding call
    2's values.  `model.graph_forward = "alias"` hands out views of the graph-owned buffers instead (valid until the
    next call of that model with the same key): what a sampler loop that consumes the residuals within the step can opt
    into — the next model then reads them in place.

    CONTEXT LENGTH (round 6): the shapes in the key are those of the CAPACITY layout (layers.box_capacity: box counts in
    buckets of 32), and the real length travels as one int32 input tensor — one graph per bucket serves every box count
    in it (dataset/utils.py:165-244 pads the boxes to each batch's maximum, so the count changes from sample to sample).

    Invalidation: `load_state_dict` / `.to()` / `set_attn_processor` on the model (`_invalidate`), any packed-weight
    drop anywhere below it (layers.CACHE_EPOCH), attribute pokes such as `use_txt_con_fusion` and nulled sub-modules such as
    `controlnet_cond_embedding = None` (the flag snapshot is part of the key, misc/test_utils.py:123-136).  Weights
    rewritten through `.data` must be followed by `model._invalidate()`, as for the packed-weight caches.
    A capture that fails (a non-capturable op in a user-modified block, no memory for the private pool) marks its key
    eager-only: the call and every later one with that key run eagerly, with one warning.
    Scratch buffers (ops.workspace) are per ForwardGraphs (ops.workspace_owner), not per capture stream: the graphs of two
    MODELS may replay concurrently (sibling_overlap); those of one model share its buffers and are always replayed one
    after the other — do not replay forward graphs of one model concurrently from several streams."""

    # A key is recorded the SECOND time it is seen (the first call runs eagerly): a caller that uses a shape once
    # should not pay an eager run AND a capture for it.  At most MAX_ENTRIES graphs per model stay alive (least
    # recently used goes first): each owns a private memory pool the size of the forward's activations.
    MAX_ENTRIES = int(os.environ.get("DD_GRAPH_FORWARD_MAX", "6"))

    def __init__(self):
        self.entries = {}                      # insertion order = recency (re-inserted on every hit)
        self.no_alias = set()
        self.seen = set()
        self.eager_only = set()                # keys whose capture failed
        self.captures = 0                      # graphs recorded over the life of this cache (bench: dropin_varlen)

    def clear(self):
        self.entries.clear()
        self.seen.clear()
        self.eager_only.clear()

    @staticmethod
    def flags(module):
        """Everything outside the tensors that changes what a forward launches: the scalar attributes of the model (the
        poke protocol), WHICH registered sub-modules are None (`controlnet_cond_embedding`, `txt_con_fusionp`, `adm_proj`
        are nulled by the reference's loader, misc/test_utils.py:123-136 — they live in `_modules`, not in vars()),
        box_adapter.SPLIT_SIZE (box_adapter.py:11, chunked attention calls) and this package's own module-level
        switches.  `use_aug_text` is an argument of every call (and written back by prepare_tokens): not a flag."""
        from . import box_adapter, unet_addon_rawbox
        return tuple(sorted((k, v) for k, v in vars(module).items()
                            if not k.startswith("_") and k not in ("training", "use_aug_text", "graph_forward")
                            and isinstance(v, (bool, int, float, str, type(None))))) \
            + tuple(sorted((k, v is None) for k, v in module._modules.items())) \
            + (("SPLIT_SIZE", box_adapter.SPLIT_SIZE), ("XATTN_FUSED", _layers.XATTN_FUSED),
               ("LN_PRODUCER", _layers.LN_PRODUCER), ("FUSED_TOKENS", unet_addon_rawbox.FUSED_TOKENS))

    def call(self, key, tensors, impl, alias_out=False):
        """tensors: flat list of tensors / None; impl(list) -> flat list of output tensors.  alias_out: return views of
        the graph's static output buffers instead of copies."""
        key = (key, tuple(None if t is None else (tuple(t.shape), t.dtype, tuple(t.stride())) for t in tensors))
        if key in self.eager_only:
            return impl(list(tensors))
        e = self.entries.get(key)
        if e is not None and e["epoch"] != _layers.CACHE_EPOCH[0]:
            e = None
        if e is not None:
            dst, src = [], []
            for i, (st, t) in enumerate(zip(e["static"], tensors)):
                if t is None or st.data_ptr() == t.data_ptr():
                    continue
                if e["alias"][i]:                 # the caller no longer passes that graph's output: own the buffer
                    self.no_alias.add(key)
                    e = None
                    break
                dst.append(st)
                src.append(t)
            if e is not None and dst:             # one multi-tensor launch per dtype group instead of one copy per input
                torch._foreach_copy_(dst, src, non_blocking=True)
        if e is None:
            if key not in self.seen and key not in self.no_alias:
                if len(self.seen) > 64:
                    self.seen.clear()
                self.seen.add(key)
                return impl(list(tensors))        # first sight: eager, on the caller's tensors
            self.entries.pop(key, None)
            while len(self.entries) >= max(1, self.MAX_ENTRIES):
                self.entries.pop(next(iter(self.entries)))
            try:
                e = self._capture(key, tensors, impl)
            except Exception as exc:              # the eager path still works: remember, warn once, run it
                self.eager_only.add(key)
                import warnings
                warnings.warn("dualdiff_amd: HIP-graph capture of a forward() failed (%s: %s); this call shape runs "
                              "eagerly from now on" % (type(exc).__name__, str(exc)[:200]))
                return impl(list(tensors))
            self.captures += 1
        else:
            self.entries.pop(key)
        self.entries[key] = e                     # most recently used last
        e["graph"].replay()
        if alias_out:
            return e["out"]
        outs = e["out"]
        own = [torch.empty_like(o) if torch.is_tensor(o) else o for o in outs]      # preserve_format: NHWC views stay NHWC
        torch._foreach_copy_([o for o in own if torch.is_tensor(o)], [o for o in outs if torch.is_tensor(o)], non_blocking=True)
        return own

    def _capture(self, key, tensors, impl):
        static, alias = [], []
        for t in tensors:
            own = t is None or key in self.no_alias or _STATIC_OUT.get(t.data_ptr()) is None
            alias.append(not own)
            static.append(None if t is None else (t.clone(memory_format=torch.preserve_format) if own else t))
        cur = torch.cuda.current_stream()
        s = torch.cuda.Stream()
        s.wait_stream(cur)
        g = torch.cuda.CUDAGraph()
        from .. import ops as _ops
        try:
            # scratch buffers of this model's graphs are its own (ops.workspace_owner): graphs of two models may replay
            # concurrently (sibling_overlap), and pooled stream handles are no identity
            with torch.cuda.stream(s), _ops.workspace_owner(id(self)):
                impl(static)                          # eager on the capture stream: tuning, packing, workspaces of this stream
                torch.cuda.synchronize()
                with torch.cuda.graph(g, stream=s):
                    out = impl(static)
        finally:
            cur.wait_stream(s)
        for o in out:
            if torch.is_tensor(o):
                _STATIC_OUT[o.data_ptr()] = o
                _STATIC_STORAGE.add(o.untyped_storage().data_ptr())
        return {"graph": g, "static": static, "alias": alias, "out": out, "epoch": _layers.CACHE_EPOCH[0]}


class Config(dict):
    """`model.config`: attribute and mapping access (`unet.config.in_channels`, `**unet.config`)."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)


class ModelBase(nn.Module):
    config_name = "config.json"
    weights_name = "diffusion_pytorch_model.bin"
    safetensors_name = "diffusion_pytorch_model.safetensors"
    _supports_gradient_checkpointing = True

    def _register_config(self, **kw):
        cfg = Config(kw)
        cfg["_class_name"] = type(self).__name__
        cfg["_diffusers_version"] = "0.17.1"
        self.__dict__["_config"] = cfg
        self.__dict__["_temb_bank"] = None

    @property
    def config(self):
        return self.__dict__["_config"]

    @property
    def dtype(self):
        return next(self.parameters()).dtype

    @property
    def device(self):
        return next(self.parameters()).device

    # -- caches ------------------------------------------------------------------------------
    @property
    def temb_bank(self):
        if self.__dict__.get("_temb_bank") is None:
            self.__dict__["_temb_bank"] = TimeEmbProjBank(self)
        return self.__dict__["_temb_bank"]

    def _invalidate(self):
        if self.__dict__.get("_temb_bank") is not None:
            self.__dict__["_temb_bank"].invalidate()
        if self.__dict__.get("_fwd_graphs") is not None:
            self.__dict__["_fwd_graphs"].clear()

    # -- HIP graphs behind forward() -----------------------------------------------------------
    graph_forward = True        # True: graphs, outputs copied out (safe default); "alias": graphs, outputs are views of
                                # graph-owned buffers (valid until the next same-key call); False: eager launches

    def _builtin_processors(self):
        """True when every attention layer runs one of this package's own processors (which honour the capacity layout of
        the context, layers.context_keys); a foreign callable may read all keys it is handed, so such a model gets exact
        context lengths and eager launches."""
        ok = self.__dict__.get("_fwd_builtin_procs")
        if ok is None or ok[1] != _layers.CACHE_EPOCH[0]:    # Attention.set_processor bumps the epoch
            from .box_adapter import Adapter_XFormersAttnProcessor, XFormersAttnProcessor
            builtin = (HIPAttnProcessor, Adapter_XFormersAttnProcessor, XFormersAttnProcessor)
            procs = list(self.attn_processors.values())
            ok = (all(type(p) in builtin for p in procs), _layers.CACHE_EPOCH[0],
                  not any(type(p) is Adapter_XFormersAttnProcessor for p in procs))
            self.__dict__["_fwd_builtin_procs"] = ok
        return ok

    def _varlen_ok(self):
        """The context may be laid out at a bucket capacity with its real length in device memory: built-in processors
        only, no box / class adapter (its context is [text | box | class] split by token counts), no sharding."""
        if not VARLEN_CONTEXT:
            return False
        if getattr(self, "view_shard", None) is not None or getattr(self, "frame_shard", None) is not None:
            return False
        ok = self._builtin_processors()
        return ok[0] and ok[2] and not getattr(self, "use_box_adapter", False)

    def _graphs(self):
        """The forward-graph cache, or None when this call must run eagerly: switched off, already inside a capture
        (BEVDenoiser records the whole step itself), a sharded model (its exchanges cannot live in a private graph), or
        a foreign attention processor (an arbitrary callable may do anything, e.g. read host state)."""
        if not (GRAPH_FORWARD and self.graph_forward) or torch.cuda.is_current_stream_capturing():
            return None
        if getattr(self, "view_shard", None) is not None or getattr(self, "frame_shard", None) is not None:
            return None
        if not self._builtin_processors()[0]:
            return None
        if self.__dict__.get("_fwd_graphs") is None:
            self.__dict__["_fwd_graphs"] = ForwardGraphs()
        return self.__dict__["_fwd_graphs"]

    def _apply(self, fn, *a, **k):
        self._invalidate()
        return super()._apply(fn, *a, **k)

    def load_state_dict(self, *a, **k):
        self._invalidate()
        return super().load_state_dict(*a, **k)

    # -- attention-processor registry (unet_addon_rawbox.py:523-593) --------------------------
    @property
    def attn_processors(self):
        return {name + ".processor": m.processor for name, m in self.named_modules()
                if hasattr(m, "set_processor")}

    def set_attn_processor(self, processor):
        count = len(self.attn_processors)
        if isinstance(processor, dict) and len(processor) != count:
            raise ValueError(
                f"A dict of processors was passed, but the number of processors {len(processor)} does not "
                f"match the number of attention layers: {count}.")
        for name, m in [(n, mm) for n, mm in self.named_modules() if hasattr(mm, "set_processor")]:
            m.set_processor(processor.pop(name + ".processor") if isinstance(processor, dict) else processor)
        self.__dict__["_fwd_builtin_procs"] = None
        self._invalidate()

    def set_default_attn_processor(self):
        self.set_attn_processor(HIPAttnProcessor())

    # -- knobs the reference flips; nothing to do on this implementation ------------------------
    def enable_xformers_memory_efficient_attention(self, attention_op=None):
        return None          # attention is always the fused HIP kernel

    def disable_xformers_memory_efficient_attention(self):
        return None

    def enable_gradient_checkpointing(self, flag=None):
        return None          # inference-only path

    def set_attention_slice(self, slice_size):
        return None

    # -- diffusers-layout checkpoints -----------------------------------------------------------
    def save_pretrained(self, save_directory, safe_serialization=False, **unused):
        os.makedirs(save_directory, exist_ok=True)
        cfg = {k: (list(v) if isinstance(v, tuple) else v) for k, v in self.config.items()}
        with open(os.path.join(save_directory, self.config_name), "w") as f:
            json.dump(cfg, f, indent=2, sort_keys=True)
        sd = {k: v.detach().cpu().contiguous() for k, v in self.state_dict().items()}
        if safe_serialization:
            from safetensors.torch import save_file
            save_file(sd, os.path.join(save_directory, self.safetensors_name))
        else:
            torch.save(sd, os.path.join(save_directory, self.weights_name))

    @classmethod
    def from_pretrained(cls, pretrained_model_name_or_path, torch_dtype=None, subfolder=None,
                        ignore_mismatched_sizes=False, **unused):
        """Loads `config.json` + `diffusion_pytorch_model.{safetensors,bin}` (diffusers key names).
        Like diffusers, returns the model in eval mode; extra kwargs (low_cpu_mem_usage,
        device_map, ...) are accepted and ignored (misc/test_utils.py:111-113)."""
        path = pretrained_model_name_or_path
        if subfolder:
            path = os.path.join(path, subfolder)
        with open(os.path.join(path, cls.config_name)) as f:
            cfg = {k: v for k, v in json.load(f).items() if not k.startswith("_")}
        model = cls(**cfg)
        st = os.path.join(path, cls.safetensors_name)
        if os.path.exists(st):
            from safetensors.torch import load_file
            sd = load_file(st)
        else:
            sd = torch.load(os.path.join(path, cls.weights_name), map_location="cpu")
        own = model.state_dict()
        if ignore_mismatched_sizes:
            sd = {k: v for k, v in sd.items() if k not in own or own[k].shape == v.shape}
        missing, unexpected = model.load_state_dict(sd, strict=False)
        allowed = tuple(getattr(cls, "_keys_to_ignore_on_load_missing", ()))
        hard_missing = [k for k in missing if not k.startswith(allowed)] if allowed else list(missing)
        if hard_missing and not ignore_mismatched_sizes:
            raise RuntimeError("checkpoint %s is missing keys: %s" % (path, hard_missing[:8]))
        if torch_dtype is not None:
            model = model.to(torch_dtype)
        return model.eval()
