"""Denoising loop of the DualDiff sampler on the HIP path.

Reproduces the loop body of
/root/reference/MD_txt_con_fusion/magicdrive/pipeline/pipeline_bev_controlnet.py:381-504
(CFG doubling :384-386, per-branch ControlNet call and residual sum :405-431, UNet call
:476-484, guidance :487-492, scheduler step :497-499) for classifier-free guidance with
guess_mode off, with diffusers' DDIMScheduler (eta = 0, SD-v1.5 schedule) as BASELINE.json's
metric names.  (The reference's own default sampler is UniPC-20, misc/test_utils.py:162.)

MI355X-first structure:
  * the whole step (ControlNet branches + UNet + CFG + DDIM + latent re-layout) is recorded once
    into a HIP graph and replayed per step — ~1.6k kernel launches leave the host's critical path;
  * branch 1's zero convs accumulate straight into branch 0's residual buffers (GEMM epilogue);
  * CFG combine + DDIM update + the CFG duplicate of the latents are one kernel whose schedule
    coefficients live in a 4-float device buffer refreshed between replays;
  * optionally (`hoist_invariant=True`) the step-invariant conditioning (tokens, ORS embedder, SFA)
    is evaluated once per sample instead of once per step; the default
    keeps the reference's per-step recomputation so step timings are like-for-like.
"""
import math
from typing import List, Optional

import torch

from .. import ops as O
from .schedulers import unipc_schedule


def ddim_schedule(num_inference_steps, num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012,
                  steps_offset=1, set_alpha_to_one=False):
    """SD-v1.5 DDIM tables: timesteps (descending) and per-step
    {sqrt(a_t), sqrt(1-a_t), sqrt(a_prev), sqrt(1-a_prev)} (DDIMScheduler.set_timesteps / .step)."""
    betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps, dtype=torch.float64) ** 2
    acp = torch.cumprod(1.0 - betas, dim=0)
    ratio = num_train_timesteps // num_inference_steps
    ts = (torch.arange(0, num_inference_steps) * ratio).flip(0) + steps_offset
    coefs = []
    for t in ts.tolist():
        a_t = acp[t]
        prev = t - ratio
        a_p = acp[prev] if prev >= 0 else (torch.tensor(1.0, dtype=torch.float64) if set_alpha_to_one else acp[0])
        coefs.append([a_t.sqrt(), (1 - a_t).sqrt(), a_p.sqrt(), (1 - a_p).sqrt()])
    return ts, torch.tensor(coefs, dtype=torch.float32)


class BEVDenoiser:
    """One scene batch (b scenes x n_cam views) of CFG denoising with 1 or 2 ControlNet branches."""

    def __init__(self, unet, controlnets: List, guidance_scale=2.0, num_inference_steps=50,
                 conditioning_scale=1.0, hoist_invariant=False, use_graph=True, use_aug_text=False,
                 parallel_branches=True, cfg_half: Optional[int] = None, cfg_exchange=None,
                 sampler="ddim", view_shard=None, frame_shard=None, segmented_graph=True):
        self.unet = unet
        self.controlnets = list(controlnets)
        self.guidance_scale = float(guidance_scale)
        self.conditioning_scale = float(conditioning_scale)
        self.hoist_invariant = hoist_invariant
        self.use_graph = use_graph
        self.use_aug_text = use_aug_text
        self.num_inference_steps = num_inference_steps
        # "ddim": BASELINE.json's metric; "unipc": the reference test pipeline's scheduler
        # (misc/test_utils.py:161-162).  Either way the update is one fused kernel fed from a coefficient row.
        if sampler not in ("ddim", "unipc"):
            raise ValueError("sampler is 'ddim' or 'unipc'")
        self.sampler = sampler
        if sampler == "ddim":
            self.timesteps, self.coef_table = ddim_schedule(num_inference_steps)
        else:
            self.timesteps, self.coef_table = unipc_schedule(num_inference_steps)
            self.num_inference_steps = len(self.timesteps)
        self._graph = None
        self._prepared = None
        self._segmented = False
        # view split + use_graph: record the step as graph SEGMENTS with the exchanges between them (default) instead of
        # one graph that would have to contain the point-to-point operations
        self.segmented_graph = bool(segmented_graph)
        # The ControlNet branches and the UNet encoder (conv_in + down + mid) are mutually
        # independent until the residual add: run them on separate HIP streams (fork / join inside
        # the captured graph) so the small deep-level kernels of one fill the CUs the others leave idle.
        self.parallel_branches = parallel_branches
        self._side = None
        # CFG split: this denoiser runs only the unconditional (0) or conditional (1) half
        # of every scene; after the model part of a step `cfg_exchange(eps_half)` must return both
        # halves as (2, b*n, 4, h, w) in [uncond, cond] order (dualdiff_amd.parallel.cfg_all_gather).
        if cfg_half is not None and cfg_half not in (0, 1):
            raise ValueError("cfg_half is 0 (unconditional), 1 (conditional) or None")
        if cfg_half is not None and cfg_exchange is None:
            raise ValueError("cfg_half needs a cfg_exchange callable")
        self.cfg_half, self.cfg_exchange = cfg_half, cfg_exchange
        self._eps_half = None
        # View split: this denoiser holds only `view_shard.plan.local` of the n_cam views of every
        # scene; the ControlNet branches and everything per-instance run on those, attn4 fetches the
        # neighbour views' K/V from the other ranks (dualdiff_amd.parallel.ViewShard).  set_inputs() takes the
        # FULL n_cam-view inputs and keeps this rank's slice.  Composes with cfg_half.
        # A shard installed directly on the UNet (INTEGRATION.md "drop-in level": unet.set_view_shard(...)) is
        # ADOPTED, never silently removed, when the argument is left out.
        if view_shard is not None:
            unet.set_view_shard(view_shard)
        self.view_shard = view_shard if view_shard is not None else getattr(unet, "view_shard", None)
        # Frame split (extension): a video UNet whose T frames are spread over ranks
        # (dualdiff_amd.parallel.FrameShard): set_inputs() takes the inputs of ALL frames — batch entries ordered
        # scene-major, then frame — and keeps this rank's frame range; adopted from the UNet like the view shard.
        if frame_shard is not None:
            unet.set_frame_shard(frame_shard)
        self.frame_shard = frame_shard if frame_shard is not None else getattr(unet, "frame_shard", None)

    # ---------------------------------------------------------------------------- inputs ----
    def set_inputs(self, latents, prompt_embeds, camera_param, bboxes_list, conds):
        """latents (b, n, 4, h, w); prompt_embeds (2b, L, 768), camera_param (2b, n, 3, 7),
        bboxes_list[i] dict of (2b, n|1, N, ...), conds[i] (2b, 3, 224, 2400) or (2b*n, 320, h, w):
        all with the unconditional half FIRST (pipeline_bev_controlnet.py:349-375)."""
        dev = latents.device
        if not latents.is_cuda:
            raise RuntimeError("BEVDenoiser runs on the GPU only")
        dt = self.unet.dtype
        fs = self.frame_shard
        if fs is not None:                                              # keep this rank's frames of every input
            t_all = fs.plan.n_frames
            if latents.shape[0] % t_all:
                raise ValueError("%d batch entries are not scenes x %d frames" % (latents.shape[0], t_all))
            sc, n_all = latents.shape[0] // t_all, latents.shape[1]
            latents = fs.take_frames(latents, sc, 1)
            prompt_embeds = fs.take_frames(prompt_embeds, 2 * sc, 1)    # [uncond ; cond] x scenes x frames
            camera_param = fs.take_frames(camera_param, 2 * sc, 1)
            bboxes_list = [None if d is None else {k: fs.take_frames(v, 2 * sc, 1) for k, v in d.items()}
                           for d in bboxes_list]
            conds = [fs.take_frames(cd, 2 * sc, 1 if cd.shape[0] == 2 * sc * t_all else n_all) for cd in conds]
        vs = self.view_shard
        if vs is not None:                                              # keep this rank's views of every input
            n_all = latents.shape[1]
            latents = vs.take_views(latents, 1, n_all)
            camera_param = vs.take_views(camera_param, 1, n_all)
            bboxes_list = [None if d is None else {k: vs.take_views(v, 1, n_all) for k, v in d.items()}
                           for d in bboxes_list]
            nb2 = prompt_embeds.shape[0]
            conds = [vs.take_panorama(cd) if cd.shape[0] == nb2 else vs.take_instances(cd, nb2) for cd in conds]
        b, n, c, h, w = latents.shape
        self.b, self.n, self.h, self.w = b, n, h, w
        self.m = 2 * b * n
        flat = latents.reshape(b * n, c, h, w).to(dt)
        self.lat2 = torch.stack([flat, flat]).contiguous()              # (2, b*n, 4, h, w): CFG duplicate
        if self.cfg_half is not None:                                   # keep this rank's half of every input
            hf = self.cfg_half
            self.m = b * n

            def half(x, per):
                return x[hf * per:(hf + 1) * per]
            prompt_embeds = half(prompt_embeds, b)
            camera_param = half(camera_param, b)
            bboxes_list = [None if d is None else {k: half(v, b) for k, v in d.items()} for d in bboxes_list]
            conds = [half(cd, b if cd.shape[0] == 2 * b else b * n) for cd in conds]
        self.prompt_embeds = prompt_embeds.to(dt)
        self.camera_param = camera_param
        self.bboxes_list = bboxes_list
        self.conds = conds
        self.t_table = self.timesteps.to(dev, torch.float32)[:, None].expand(-1, self.m).contiguous()
        self.coef_dev = self.coef_table.to(dev)
        self.t_dev = torch.empty(self.m, dtype=torch.float32, device=dev)
        self.coef = torch.empty(self.coef_table.shape[1], dtype=torch.float32, device=dev)
        self.hist = torch.zeros((3, b * n, c, h, w), dtype=torch.float32, device=dev) if self.sampler == "unipc" else None
        self._graph = None
        self._prepared = None
        if self.hoist_invariant:
            self._prepared = self._prepare()
        self._set_step(0)

    def _prepare(self):
        return [cn.prepare_condition(self.camera_param, self.bboxes_list[i], self.prompt_embeds,
                                     self.conds[i], self.use_aug_text)
                for i, cn in enumerate(self.controlnets)]

    def _set_step(self, i):
        self.t_dev.copy_(self.t_table[i], non_blocking=True)
        self.coef.copy_(self.coef_dev[i], non_blocking=True)

    # ------------------------------------------------------------------------------ step ----
    def _step_body(self):
        m, h, w = self.m, self.h, self.w
        lat_in = self.lat2[0] if self.cfg_half is not None else self.lat2.reshape(m, 4, h, w)
        x8 = O.nchw_to_nhwc(lat_in, 8)                                  # latent_model_input, NHWC pad 8
        n = len(self.controlnets)
        if not self.parallel_branches:
            prep = self._prepared if self._prepared is not None else self._prepare()
            res = None
            for i, cn in enumerate(self.controlnets):                    # :405-431, sum in the epilogue
                res = cn.forward_nhwc(x8, m, h, w, self.t_dev, prep[i], self.conditioning_scale,
                                      out=res, accumulate=i > 0)
            ctx = prep[0]                                                # tokens from branch 0 (:430-431)
            eps = self.unet.forward_nhwc(x8, m, h, w, self.t_dev, ctx["ctx2d"], ctx["lc"],
                                         [r[0] for r in res[:-1]], res[-1][0])      # :476-484
        else:
            main = torch.cuda.current_stream()
            if self._side is None:
                self._side = [torch.cuda.Stream() for _ in range(n)]
            # branch 0's TOKENS feed the UNet: only they are prepared on the main stream; the condition
            # image embedding + SFA of every branch run on the branch's own stream.  Branches that do not
            # feed the UNet fork first (they need nothing from the main stream but the latents).
            results = [None] * n

            def run_branch(i, p_tok=None):
                cn, s = self.controlnets[i], self._side[i]
                s.wait_stream(main)
                with torch.cuda.stream(s):
                    if self._prepared is not None:
                        p_i = self._prepared[i]
                    elif p_tok is not None:
                        p_i = cn.prepare_cond(p_tok, self.conds[i])
                    else:
                        p_i = cn.prepare_condition(self.camera_param, self.bboxes_list[i], self.prompt_embeds,
                                                   self.conds[i], self.use_aug_text)
                    results[i] = cn.forward_nhwc(x8, m, h, w, self.t_dev, p_i, self.conditioning_scale)

            for i in range(1, n):                                        # fork the independent branches
                run_branch(i)
            if self._prepared is not None:
                tok0 = self._prepared[0]
            else:
                tok0 = self.controlnets[0].prepare_tokens(self.camera_param, self.bboxes_list[0],
                                                          self.prompt_embeds, self.use_aug_text)
            run_branch(0, tok0)                                          # fork branch 0 once its tokens exist
            prep0 = tok0
            if self._segmented:                                          # segments cannot end with forked work in flight:
                for s in self._side:                                     # the branches run beside each other, the UNet after
                    main.wait_stream(s)
            state = self.unet.encode_nhwc(x8, m, h, w, self.t_dev, prep0["ctx2d"], prep0["lc"])
            if not self._segmented:
                for s in self._side:                                     # join
                    main.wait_stream(s)
            if n == 1:
                down = [r[0] for r in results[0][:-1]]
                mid = results[0][-1][0]
            else:                                                        # branch sum (:421-429) in the add
                down = [tuple(results[i][j][0] for i in range(n)) for j in range(len(results[0]) - 1)]
                mid = tuple(results[i][-1][0] for i in range(n))
            eps = self.unet.decode_nhwc(state, down, mid)
        if self.cfg_half is not None:                                   # combine happens after the exchange
            self._eps_half = eps
            return eps
        self._scheduler_step(eps)                                        # :487-499
        return eps

    def _scheduler_step(self, eps2):
        if self.sampler == "ddim":
            O.cfg_ddim_step(eps2, self.lat2[0], self.coef, self.guidance_scale,
                            x_out=self.lat2[0], x_dup=self.lat2[1])
        else:
            O.cfg_unipc_step(eps2, self.lat2[0], self.hist, self.coef, self.guidance_scale,
                             x_out=self.lat2[0], x_dup=self.lat2[1])

    def _combine_halves(self):
        """CFG split: exchange the two halves' noise predictions, then guidance + DDIM on both ranks
        (each keeps the full latents; the update is deterministic and identical)."""
        self._scheduler_step(self.cfg_exchange(self._eps_half))

    def capture(self):
        """Eager warm-up (packs weights, sizes workspaces) then records the step into a HIP graph."""
        saved = self.lat2.clone()
        saved_hist = None if self.hist is None else self.hist.clone()

        def restore():
            self.lat2.copy_(saved)
            if saved_hist is not None:
                self.hist.copy_(saved_hist)
        self._step_body()
        torch.cuda.synchronize()
        restore()
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        if self.view_shard is not None and self.segmented_graph:
            # view split: a chain of graph segments with the neighbour K/V exchanges between them (parallel.SegmentedGraph)
            from ..parallel import SegmentedGraph
            self._segmented = True
            g = SegmentedGraph()
            with torch.cuda.stream(s):
                self._step_body()                                        # warm-up in the segmented stream layout
                torch.cuda.synchronize()
                restore()
                self.view_shard.segmenter = g
                try:
                    g.record(self._step_body)
                except BaseException:
                    self._segmented = False                          # the eager step that may follow keeps the 3-stream layout
                    raise
                finally:
                    self.view_shard.segmenter = None
        else:
            g = torch.cuda.CUDAGraph()
            with torch.cuda.stream(s):
                self._step_body()                                        # warm-up on the capture stream
                torch.cuda.synchronize()
                restore()
                with torch.cuda.graph(g, stream=s):
                    self._step_body()
        torch.cuda.current_stream().wait_stream(s)
        restore()
        self._graph = g

    def step(self, i):
        """Denoising step i (0-based) on the current latents."""
        from ..networks.model_base import sibling_barrier
        sibling_barrier()                                # this step rewrites its buffers in place (no version counts)
        self._set_step(i)
        if self.use_graph:
            if self._graph is None:
                self.capture()
                self._set_step(i)
            self._graph.replay()
        else:
            self._step_body()
        if self.cfg_half is not None:
            self._combine_halves()

    def run(self, steps: Optional[int] = None):
        for i in range(steps if steps is not None else self.num_inference_steps):
            self.step(i)
        return self.latents

    @property
    def latents(self):
        return self.lat2[0].reshape(self.b, self.n, 4, self.h, self.w)
