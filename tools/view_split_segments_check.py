"""Segmented-graph replay of the view-split step == its eager form, bit for bit, on every rank.

Run under torchrun with all ranks on ONE GPU and gloo as the transport (host-staged exchanges):
    DD_BENCH_SHARE_GPU=1 python -m torch.distributed.run --nproc-per-node 3 --master-addr 127.0.0.1 --master-port 29655 \
        tools/view_split_segments_check.py
Every rank builds the same models, takes its view shard (world 3: three shards of two views holding both CFG halves;
world 4: two CFG halves x two shards of three views), runs STEPS denoising steps eagerly and again through
parallel.SegmentedGraph (17 graph segments, 16 exchanges between them per step), and compares the latents bitwise.
Rank 0 prints one JSON line: {"world", "segments", "exchanges", "bitwise_equal_all_ranks", "ms_eager", "ms_segments"}."""
import json, os, sys, time
import torch
import torch.distributed as dist
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from dualdiff_amd.parallel import HaloExchange, ViewShard, ViewSplitPlan, cfg_all_gather, view_split_groups
from dualdiff_amd.pipeline.pipeline_bev_controlnet import BEVDenoiser

STEPS = 3
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dev = torch.device("cuda", 0 if os.environ.get("DD_BENCH_SHARE_GPU") == "1" else int(os.environ.get("LOCAL_RANK", "0")))
torch.cuda.set_device(dev)
dist.init_process_group(os.environ.get("DD_BENCH_BACKEND", "gloo"))
dt = torch.float16
unet, cns = bench.build_models(dt, dev)
plan = ViewSplitPlan(world, rank, bench.PAIR)
halves, pair_groups = view_split_groups(world, plan.cfg_split)


def make(graph):
    shard = ViewShard(plan, HaloExchange(plan, halves[plan.half or 0]))
    kw = {"view_shard": shard}
    if plan.cfg_split:
        grp = pair_groups[plan.shard]
        kw.update({"cfg_half": plan.half, "cfg_exchange": lambda e: cfg_all_gather(e, grp)})
    den = BEVDenoiser(unet, cns, guidance_scale=2.0, num_inference_steps=50, use_graph=graph, **kw)
    den.set_inputs(*bench.synthetic_inputs(1, dt, dev, seed=1234))
    return den


def run(den):
    torch.cuda.synchronize(); dist.barrier()
    t0 = time.perf_counter()
    for i in range(STEPS):
        den.step(i)
    torch.cuda.synchronize(); dist.barrier()
    return (time.perf_counter() - t0) / STEPS * 1e3, den.latents.clone()


with torch.no_grad():
    eager = make(False)
    eager.step(0)                                  # tunes this shard's shapes
    eager.set_inputs(*bench.synthetic_inputs(1, dt, dev, seed=1234))
    ms_e, lat_e = run(eager)
    seg = make(True)
    seg.capture()
    # capture() leaves the latents at their initial values (it restores what its warm-up passes changed)
    ms_s, lat_s = run(seg)
g = seg._graph
same = torch.tensor([int(torch.equal(lat_e, lat_s) and bool(torch.isfinite(lat_s.float()).all()))])
dist.all_reduce(same, op=dist.ReduceOp.MIN)
if rank == 0:
    print(json.dumps({"world": world, "views_per_rank": [len(v) for v in plan.views_of], "cfg_split": plan.cfg_split,
                      "segments": g.segments, "exchanges": len(g.between), "bitwise_equal_all_ranks": bool(same.item()),
                      "ms_eager": round(ms_e, 2), "ms_segments": round(ms_s, 2),
                      "note": "all ranks time-slice ONE GPU over gloo: the times mean nothing beyond eager vs segments"}))
dist.destroy_process_group()
