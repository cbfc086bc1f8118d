#!/usr/bin/env python3
"""2-rank RCCL check of the view split: every rank runs the multiview UNet UNSHARDED on all six
views of a scene and then SHARDED over the two ranks (three views each; neighbour-view K/V by point-to-point
`batch_isend_irecv` on device tensors inside every transformer block, dualdiff_amd/parallel.py:HaloExchange) and
compares its local views' noise prediction.  Needs >= 2 GPUs:

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 tools/view_split_rccl.py

Prints `VIEW_SPLIT_RCCL rel_l2=<max over ranks> ...` on rank 0 (tests/test_parity_r03_gpu.py parses it).  Never run
on hardware by this build (1-GPU boxes only); bench.py labels view-split numbers accordingly.
"""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

PAIR = {0: [5, 1], 1: [0, 2], 2: [1, 3], 3: [2, 4], 4: [3, 5], 5: [4, 0]}


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    local = int(os.environ.get("LOCAL_RANK", rank))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist.init_process_group("nccl", device_id=dev)
    from dualdiff_amd.networks.layers import device_init_
    from dualdiff_amd.networks.unet_2d_condition_multiview import UNet2DConditionModelMultiview
    from dualdiff_amd.parallel import HaloExchange, ViewShard, ViewSplitPlan
    dt = torch.float16
    with torch.device(dev):
        unet = UNet2DConditionModelMultiview(cross_attention_dim=768, neighboring_view_pair=PAIR).to(dt)
    device_init_(unet, 1)                                  # same seed -> same weights on every rank
    unet.eval()
    g = torch.Generator(device=dev).manual_seed(5)
    nb = 2                                                  # both CFG halves
    x = torch.randn((nb * 6, 4, 28, 50), generator=g, device=dev).to(dt)
    ctx = torch.randn((nb * 6, 98, 768), generator=g, device=dev).to(dt)
    with torch.no_grad():
        full = unet(x, 481, encoder_hidden_states=ctx).sample.float()
        plan = ViewSplitPlan(world, rank, PAIR, cfg_split=False)
        shard = ViewShard(plan, HaloExchange(plan, None))
        unet.set_view_shard(shard)
        xs = shard.take_instances(x, nb).contiguous()
        cs = shard.take_instances(ctx, nb).contiguous()
        part = unet(xs, 481, encoder_hidden_states=cs).sample.float()
        again = unet(xs, 481, encoder_hidden_states=cs).sample.float()   # a second pass: stream ordering of the exchange
    want = shard.take_instances(full, nb)
    e = ((part - want).norm() / want.norm()).item()
    rep = float((part - again).abs().max().item())
    t = torch.tensor([e, rep], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    torch.cuda.synchronize()
    if rank == 0:
        print("VIEW_SPLIT_RCCL rel_l2=%.4e repeat_maxabs=%.3e world=%d views/rank=%s" % (t[0].item(), t[1].item(), world, plan.local))
    dist.destroy_process_group()
    if t[0].item() > 2e-3 or t[1].item() != 0.0:
        sys.exit(1)


if __name__ == "__main__":
    main()
