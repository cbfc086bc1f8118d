"""Multi-GPU sharding of the denoising workload (one process per GPU, torch.distributed / RCCL).

The unit of work is a scene (6 views x CFG = 12 view-instances).  Scenes are independent — the
reference itself only ever shards its val set over ranks (`tools/downstream_v3_batched.py:120,157`,
`val_set_gen.py:121`) — so ranks take disjoint scene slices and there is NO data-path collective.
Collectives are used only for bookkeeping: max-over-ranks timing and gathering result tensors.

For single-scene LATENCY the classifier-free-guidance halves can additionally be split over a pair of
GPUs (SURVEY.md §8e "CFG split"): each rank runs the 6 view-instances of its half and the two noise
predictions (67 KB each in bf16) are exchanged once per step — `cfg_all_gather`, the one real
exchange step on that path (pipeline_bev_controlnet.py:487-490 combines the halves).
"""
import torch
import torch.distributed as dist


def shard_scenes(n_scenes, rank, world):
    """Contiguous, balanced slice [lo, hi) of scene indices for `rank` (first ranks take the remainder)."""
    if not (0 <= rank < world):
        raise ValueError("rank %d outside world %d" % (rank, world))
    base, rem = divmod(n_scenes, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def max_over_ranks(seconds, device=None):
    """Wall time of the slowest rank (the throughput denominator of bench.py)."""
    if not (dist.is_available() and dist.is_initialized()):
        return float(seconds)
    t = torch.tensor([float(seconds)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_scenes(local, n_scenes):
    """All-gather per-rank result tensors (scene-major, dim 0) back into scene order on every rank.
    Ragged shards (n_scenes % world != 0) are padded to the largest shard for the collective."""
    if not (dist.is_available() and dist.is_initialized()):
        return local
    world, rank = dist.get_world_size(), dist.get_rank()
    sizes = [shard_scenes(n_scenes, r, world) for r in range(world)]
    mx = max(hi - lo for lo, hi in sizes)
    pad = torch.zeros((mx,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[: local.shape[0]] = local
    out = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(out, pad)
    return torch.cat([o[: hi - lo] for o, (lo, hi) in zip(out, sizes)], dim=0)


def cfg_pair_groups(world):
    """Process groups [2g, 2g+1] (rank 2g = unconditional half, 2g+1 = conditional half).  Every rank
    must call this (new_group is collective); returns the list of groups."""
    if world % 2:
        raise ValueError("CFG split needs an even number of ranks, got %d" % world)
    return [dist.new_group([2 * g, 2 * g + 1]) for g in range(world // 2)]


def cfg_all_gather(eps_half, group=None):
    """(b*n, 4, h, w) noise prediction of this rank's CFG half -> (2, b*n, 4, h, w) in
    [unconditional, conditional] order = rank order inside the pair group."""
    if not (dist.is_available() and dist.is_initialized()):
        raise RuntimeError("cfg_all_gather needs an initialised process group")
    if dist.get_world_size(group) != 2:
        raise ValueError("a CFG pair group has exactly 2 ranks")
    eps_half = eps_half.contiguous()
    out = [torch.empty_like(eps_half), torch.empty_like(eps_half)]
    dist.all_gather(out, eps_half, group=group)
    return torch.stack(out)


# ------------------------------------------------------------------------------------------------
# View split: the 6 views of ONE scene (per CFG half) over several GPUs.
#
# Everything in the denoising step is per view-instance except
#   (i)  attn4 of `BasicMultiviewTransformerBlock` (reference blocks.py:106-142,190-222): view v attends to
#        the K/V of its neighbours pair[v] — in each of the 16 UNet transformer blocks; and
#   (ii) the CFG combine (pipeline_bev_controlnet.py:487-490), once per step.
# Ranks are laid out as  rank = shard * halves + half : with `cfg_split` (even world) the two ranks
# {2s, 2s+1} hold the unconditional / conditional half of the SAME views (the pair groups of the CFG
# split exchange the noise prediction), and ranks of equal parity form a half group inside which the
# neighbour K/V travel.  Without it (odd world) every rank holds both halves of its views.
#
# What travels is the PROJECTED K/V of whole view-instances, head-major ([2*heads][l][d] per instance and
# CFG half — contiguous, so a message is one zero-copy slice of the block's K/V buffer), straight to the
# ranks that need them: point-to-point isend / irecv, never a ring collective.  Per view and block:
# 2 * l * C * 2 B = 1.8 MB (28x50, C = 320), 0.9 MB (14x25, 640), 0.47 MB (7x13, 1280), 0.14 MB (4x7).
# ------------------------------------------------------------------------------------------------
def split_views(n_cam, shards):
    """Contiguous balanced view ranges: [[views of shard 0], ...] (first shards take the remainder)."""
    if not (1 <= shards <= n_cam):
        raise ValueError("cannot split %d views over %d shards" % (n_cam, shards))
    return [list(range(*shard_scenes(n_cam, s, shards))) for s in range(shards)]


class ViewSplitPlan:
    """Static plan of one rank: which views it owns, which neighbour views it needs from whom, which of its
    views others need.  Pure Python / deterministic on every rank (no communication)."""

    def __init__(self, world, rank, view_pair, cfg_split=None):
        self.world, self.rank = world, rank
        self.pair = {int(k): [int(x) for x in v] for k, v in view_pair.items()}
        self.n_cam = len(self.pair)
        if cfg_split is None:
            cfg_split = world % 2 == 0
        if cfg_split and world % 2:
            raise ValueError("cfg_split needs an even world size")
        self.cfg_split = bool(cfg_split)
        self.halves = 2 if self.cfg_split else 1          # CFG halves spread over ranks
        self.shards = world // self.halves
        self.shard, self.half = rank // self.halves, (rank % self.halves if self.cfg_split else None)
        self.views_of = split_views(self.n_cam, self.shards)
        self.local = self.views_of[self.shard]
        self.owner = {v: s for s, vs in enumerate(self.views_of) for v in vs}
        self.remote = self._needs(self.shard)
        # shard -> views (send: mine that they need; recv: theirs that I need), each in sorted view order
        self.send = {s: [v for v in self._needs(s) if self.owner[v] == self.shard]
                     for s in range(self.shards) if s != self.shard}
        self.recv = {s: [v for v in self.remote if self.owner[v] == s] for s in range(self.shards) if s != self.shard}
        self.send = {s: v for s, v in self.send.items() if v}
        self.recv = {s: v for s, v in self.recv.items() if v}

    def _needs(self, shard):
        mine = set(self.views_of[shard])
        return sorted({u for v in self.views_of[shard] for u in self.pair[v]} - mine)

    def rank_of(self, shard):
        """Global rank of `shard` inside this rank's half group."""
        return shard * self.halves + (self.half or 0)

    @property
    def n_slots(self):
        return len(self.local) + len(self.remote)

    def slot(self, view):
        """Slot of `view` in this rank's K/V buffer: local views first (in order), then the remote ones."""
        if view in self.local:
            return self.local.index(view)
        return len(self.local) + self.remote.index(view)

    def kv_maps(self, nb):
        """One int list per neighbour position j: local instance (bi, vi) [index bi * n_local + vi] -> batch
        index of the K/V of pair[view][j] in the slot-major buffer (slot * nb + bi)."""
        depth = max(len(v) for v in self.pair.values())
        nloc = len(self.local)
        return [[self.slot(self.pair[self.local[i % nloc]][j]) * nb + i // nloc for i in range(nb * nloc)]
                for j in range(depth)]

    def cfg_partner_group_ranks(self):
        """Ranks [uncond, cond] that hold the two CFG halves of this rank's views (cfg_split only)."""
        return [self.shard * 2, self.shard * 2 + 1] if self.cfg_split else None

    def message_bytes(self, n_tokens, channels, nb=1, elem=2):
        """Bytes this rank sends / receives per transformer block at a level (documentation)."""
        one = 2 * n_tokens * channels * elem * nb
        return sum(len(v) for v in self.send.values()) * one, len(self.remote) * one


class HaloExchange:
    """Moves neighbour-view K/V between the ranks of a half group with point-to-point messages.
    `kv` is the slot-major buffer (n_slots, nb, 2*heads, l, d): slots [0, n_local) hold this rank's views
    (already written), the exchange fills the remote slots.  RCCL backend: device tensors go out as they
    are (batch_isend_irecv on the backend's stream, the current stream waits — no host sync).  gloo backend
    (CPU tests / the shared-GPU plumbing mode): device tensors are staged through host memory."""

    def __init__(self, plan, group=None):
        self.plan, self.group = plan, group

    def __call__(self, kv):
        pl = self.plan
        if not pl.send and not pl.recv:
            return kv
        if not (dist.is_available() and dist.is_initialized()):
            raise RuntimeError("HaloExchange needs an initialised process group")
        staged = kv.is_cuda and dist.get_backend(self.group) == "gloo"
        ops, holds, landing = [], [], []
        for s in sorted(set(pl.send) | set(pl.recv)):
            peer = pl.rank_of(s)
            for v in pl.send.get(s, ()):
                t = kv[pl.slot(v)]
                t = t.cpu() if staged else t
                holds.append(t)
                ops.append(dist.P2POp(dist.isend, t, peer, self.group))
            for v in pl.recv.get(s, ()):
                dst = kv[pl.slot(v)]
                t = torch.empty(dst.shape, dtype=dst.dtype) if staged else dst
                landing.append((dst, t))
                ops.append(dist.P2POp(dist.irecv, t, peer, self.group))
        for req in dist.batch_isend_irecv(ops):
            req.wait()
        if staged:
            for dst, t in landing:
                dst.copy_(t)
        return kv


def view_split_groups(world, cfg_split=True):
    """(half groups, pair groups): process groups of ranks with equal parity (neighbour K/V exchange) and of
    the rank pairs {2s, 2s+1} (CFG exchange).  Collective: EVERY rank must call it, in the same order.
    Without cfg_split there is one half group (all ranks) and no pair group."""
    if not cfg_split:
        return [dist.new_group(list(range(world)))], []
    halves = [dist.new_group(list(range(h, world, 2))) for h in (0, 1)]
    return halves, cfg_pair_groups(world)


class SegmentedGraph:
    """A step recorded as a CHAIN of HIP graphs with eager hand-offs between them (round 4).

    The view-split step needs the neighbour views' K/V from other ranks in each of the 16 UNet transformer blocks.  The
    exchange cannot live inside a captured graph on every backend (gloo stages through the host; RCCL point-to-point
    inside a capture has never run on >= 2 GPUs in this build's reach), and launching the ~800 kernels of the step
    eagerly costs more host time than a 1-3-view shard's device time.  So the step is captured in SEGMENTS: every
    exchange ends the current capture (`cut`), is stored as a callable, runs once (all ranks record in lock step), and
    the next capture begins in the same memory pool — tensors keep their addresses, a replay is graph, exchange, graph,
    ... in recording order: 17 graph launches + 16 exchanges instead of ~800 kernel launches per step.
    Side streams forked inside the step must be joined before a cut (the pipeline joins the ControlNet branches
    before the UNet encoder in this mode)."""

    def __init__(self):
        self.graphs, self.between = [], []
        self.recording = False
        self._pool = None

    def _begin(self):
        g = torch.cuda.CUDAGraph()
        g.capture_begin(pool=self._pool)
        self.graphs.append(g)

    def record(self, fn):
        """Records fn() on the CURRENT (non-default) stream; `cut` calls inside fn split the recording.
        If fn(), an exchange or a capture_end raises, the open capture is ENDED before the error propagates (a raw
        capture_begin has no context manager to do it: the stream would stay in capture mode and every later launch,
        allocation or synchronize would fail with 'operation not permitted when stream is capturing', hiding the real
        error) and the half-recorded chain is dropped, so an eager step can follow."""
        self._pool = torch.cuda.graph_pool_handle()
        torch.cuda.synchronize()
        self.recording = True
        try:
            self._begin()
            fn()
            self.graphs[-1].capture_end()
        except BaseException:
            self._abort()
            raise
        finally:
            self.recording = False
        return self

    def _abort(self):
        try:
            if self.graphs and torch.cuda.is_current_stream_capturing():
                self.graphs[-1].capture_end()
        except Exception:
            pass                                  # the capture was already invalidated by the failing call
        self.graphs, self.between = [], []

    def cut(self, fn):
        self.graphs[-1].capture_end()
        self.between.append(fn)
        fn()
        self._begin()

    def replay(self):
        for i, g in enumerate(self.graphs):
            g.replay()
            if i < len(self.between):
                self.between[i]()

    @property
    def segments(self):
        return len(self.graphs)


class ViewShard:
    """What a view-sharded model needs at run time: the static plan, the exchange callable (HaloExchange,
    or an in-process stand-in for tests) and cached device-side kv_batch_maps.  Installed on every
    `BasicMultiviewTransformerBlock` by `UNet2DConditionModelMultiview.set_view_shard`."""

    def __init__(self, plan, exchange=None):
        self.plan = plan
        self._exchange = exchange if exchange is not None else HaloExchange(plan)
        self._maps = {}
        self.segmenter = None        # a SegmentedGraph while the step is being recorded / replayed in segments

    def exchange(self, kv):
        """Fills the remote slots of `kv`.  While a SegmentedGraph records the step, the exchange is a CUT: the current
        HIP-graph segment ends, the exchange runs eagerly (now, and between the segments at every replay, on the same
        `kv` buffer — graph memory keeps its address), the next segment begins."""
        seg = self.segmenter
        if seg is not None and seg.recording:
            seg.cut(lambda: self._exchange(kv))
        else:
            self._exchange(kv)
        return kv

    @property
    def n_local(self):
        return len(self.plan.local)

    def maps(self, nb, device):
        key = (nb, str(device))
        if key not in self._maps:
            self._maps[key] = [torch.tensor(m, dtype=torch.int32, device=device) for m in self.plan.kv_maps(nb)]
        return self._maps[key]

    # ---- input slicing (sampler level) -------------------------------------------------------
    def take_views(self, t, dim, n_cam=None):
        """Keeps this rank's views along `dim` (size n_cam); tensors whose `dim` is 1 are shared by all views."""
        n_cam = n_cam or self.plan.n_cam
        if t.shape[dim] == 1:
            return t
        if t.shape[dim] != n_cam:
            raise ValueError("dim %d of %s is neither 1 nor n_cam = %d" % (dim, tuple(t.shape), n_cam))
        lo, hi = self.plan.local[0], self.plan.local[-1] + 1
        return t.narrow(dim, lo, hi - lo)

    def take_panorama(self, cond):
        """(b, c, h, n_cam * w) panorama (views side by side, map_embedder.py:116-125) -> this rank's columns."""
        w = cond.shape[-1] // self.plan.n_cam
        lo, hi = self.plan.local[0], self.plan.local[-1] + 1
        return cond[..., lo * w:hi * w]

    def take_instances(self, t, nb):
        """(nb * n_cam, ...) view-instances, scene-major -> (nb * n_local, ...) of this rank's views."""
        n = self.plan.n_cam
        lo, hi = self.plan.local[0], self.plan.local[-1] + 1
        return t.reshape(nb, n, *t.shape[1:])[:, lo:hi].reshape(nb * (hi - lo), *t.shape[1:])


# ------------------------------------------------------------------------------------------------
# Frame split (EXTENSION — BASELINE configs[3]/[4]): the T frames of a scene over several GPUs,
# all 6 views of a frame local.  Instances stay ordered scene-major, then (local) frame, then view, so every
# per-instance layer, the ControlNets, cross-attention AND the cross-view attention (attn4: neighbours of a
# view are views of the SAME frame) run unchanged on fewer instances.  What crosses ranks, per video block
# (networks/video_blocks.py):
#   ST-Attn   every frame attends to [first frame ; previous frame]: the owner of frame 0 sends its K/V to every
#             other rank, and each rank sends the K/V of its LAST frame to the next rank — point to point;
#   temporal  attention over the T frames of every token position: queries stay local, the K|V rows of all
#             frames are all-gathered (frame-major, so a rank's contribution is one contiguous block).
# ------------------------------------------------------------------------------------------------
class FrameSplitPlan:
    """Static plan of one rank: contiguous, balanced frame ranges in rank order (first ranks take the remainder)."""

    def __init__(self, world, rank, n_frames):
        if not (1 <= world <= n_frames):
            raise ValueError("cannot split %d frames over %d ranks" % (n_frames, world))
        self.world, self.rank, self.n_frames = world, rank, int(n_frames)
        self.ranges = [shard_scenes(self.n_frames, r, world) for r in range(world)]
        self.lo, self.hi = self.ranges[rank]
        self.local = list(range(self.lo, self.hi))

    @property
    def n_local(self):
        return self.hi - self.lo

    def counts(self):
        return [hi - lo for lo, hi in self.ranges]

    def message_bytes(self, n_tokens, channels, n_views=6, nb=1, elem=2):
        """(ST-Attn bytes sent, temporal all-gather bytes received) by this rank per video block at a level."""
        frame_kv = 2 * n_views * n_tokens * channels * elem * nb
        st = (self.world - 1) * frame_kv if self.rank == 0 else 0
        st += frame_kv if self.rank + 1 < self.world else 0
        return st, (self.n_frames - self.n_local) * frame_kv


class FrameExchange:
    """The two exchanges of a frame-sharded video block.  RCCL: device tensors travel as they are; gloo (CPU tests,
    shared-GPU plumbing mode): device tensors are staged through host memory, as in HaloExchange."""

    def __init__(self, plan, group=None):
        self.plan, self.group = plan, group

    def _staged(self, t):
        return t.is_cuda and dist.get_backend(self.group) == "gloo"

    def _rank(self, r):
        return r if self.group is None else dist.get_global_rank(self.group, r)

    def st_sources(self, first_local, last_local):
        """K/V of this rank's first / last local frame (contiguous, equal shapes) -> (K/V of frame 0, K/V of the
        frame before this rank's first one).  Rank 0 gets its own first frame twice (frame 0's previous frame is
        frame 0, video_blocks.py)."""
        pl = self.plan
        if pl.world == 1:
            return first_local, first_local
        if not (dist.is_available() and dist.is_initialized()):
            raise RuntimeError("FrameExchange needs an initialised process group")
        staged = self._staged(first_local)
        ops, holds = [], []

        def out(t):
            t = t.cpu() if staged else t
            holds.append(t)
            return t

        def landing():
            return torch.empty(first_local.shape, dtype=first_local.dtype, device="cpu" if staged else first_local.device)
        first = prev = None
        # posting order is the same on both ends of every pair: [frame-0 message, last-frame message]
        if pl.rank == 0:
            for r in range(1, pl.world):
                ops.append(dist.P2POp(dist.isend, out(first_local), self._rank(r), self.group))
        else:
            first = landing()
            ops.append(dist.P2POp(dist.irecv, first, self._rank(0), self.group))
        if pl.rank + 1 < pl.world:
            ops.append(dist.P2POp(dist.isend, out(last_local), self._rank(pl.rank + 1), self.group))
        if pl.rank > 0:
            prev = landing()
            ops.append(dist.P2POp(dist.irecv, prev, self._rank(pl.rank - 1), self.group))
        for req in dist.batch_isend_irecv(ops):
            req.wait()
        if pl.rank == 0:
            return first_local, first_local
        if staged:
            first, prev = first.to(first_local.device), prev.to(first_local.device)
        return first, prev

    def gather_frames(self, local):
        """(n_local frames, ...) contiguous -> (n_frames, ...) in frame order on every rank."""
        pl = self.plan
        if pl.world == 1:
            return local
        if not (dist.is_available() and dist.is_initialized()):
            raise RuntimeError("FrameExchange needs an initialised process group")
        staged = self._staged(local)
        src = local.cpu() if staged else local.contiguous()
        counts, mx = pl.counts(), max(pl.counts())
        if mx != src.shape[0]:                                       # ragged split: pad to the largest shard
            pad = torch.zeros((mx,) + tuple(src.shape[1:]), dtype=src.dtype, device=src.device)
            pad[: src.shape[0]] = src
            src = pad
        if len(set(counts)) == 1:
            out = torch.empty((pl.n_frames,) + tuple(src.shape[1:]), dtype=src.dtype, device=src.device)
            dist.all_gather(list(out.split(mx)), src, group=self.group)
        else:
            parts = [torch.empty_like(src) for _ in counts]
            dist.all_gather(parts, src, group=self.group)
            out = torch.cat([p[:c] for p, c in zip(parts, counts)], dim=0)
        return out.to(local.device) if staged else out


class FrameShard:
    """Run-time handle of a frame-sharded video UNet (`UNet2DConditionModelMultiviewVideo.set_frame_shard`): the
    plan plus the exchange (FrameExchange, or an in-process stand-in with the same two methods)."""

    def __init__(self, plan, exchange=None):
        self.plan = plan
        self.exchange = exchange if exchange is not None else FrameExchange(plan)

    @property
    def n_local(self):
        return self.plan.n_local

    def take_frames(self, t, nb, per_frame):
        """(nb * T * per_frame, ...) instances ordered scene, frame, view -> this rank's frames of every scene."""
        pl = self.plan
        if t.shape[0] != nb * pl.n_frames * per_frame:
            raise ValueError("dim 0 of %s is not %d scenes x %d frames x %d" % (tuple(t.shape), nb, pl.n_frames, per_frame))
        return t.reshape(nb, pl.n_frames, per_frame, *t.shape[1:])[:, pl.lo:pl.hi] \
            .reshape(nb * pl.n_local * per_frame, *t.shape[1:])
