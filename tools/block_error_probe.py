"""Where does the multiview block's error come from?  Runs the standalone L1 block (tests/test_model_gpu.py
case) on the HIP path with individual fusions switched off and after each sub-layer, against the fp32 oracle
and the storage-dtype floor.  python tools/block_error_probe.py [bf16|fp16]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import dualdiff_restated as R                      # noqa: E402
from oracle.init_utils import seeded_state_dict, seeded_tensor  # noqa: E402
from oracle.numerics import storage_emulation                  # noqa: E402

PAIR = {0: [5, 1], 1: [0, 2], 2: [1, 3], 3: [2, 4], 4: [3, 5], 5: [4, 0]}
dtype = torch.bfloat16 if (len(sys.argv) < 2 or sys.argv[1] == "bf16") else torch.float16


def r16(t):
    return t.to(torch.bfloat16).float()


def rel(y, ref):
    return ((y.float().cpu() - ref).norm() / ref.norm()).item()


ora = R.BasicMultiviewTransformerBlock(640, 8, 80, cross_attention_dim=768, neighboring_view_pair=PAIR).eval()
sd = {k: r16(v) for k, v in seeded_state_dict(ora, 5).items()}
ora.load_state_dict(sd)
x = r16(seeded_tensor((6, 350, 640), 1))
ctx = r16(seeded_tensor((6, 30, 768), 2))


def stages(block, emulate):
    """oracle outputs after attn1, attn2, attn4(+connector), ff"""
    import contextlib
    cm = storage_emulation(block, dtype) if emulate else contextlib.nullcontext()
    out = []
    with torch.no_grad(), cm:
        h = x
        h = block.attn1(block.norm1(h)) + h; out.append(h)
        h = block.attn2(block.norm2(h), encoder_hidden_states=ctx) + h; out.append(h)
        n_cam = 6
        xx = block.norm4(h)
        xv = xx.reshape(-1, n_cam, xx.shape[1], xx.shape[2])
        a = block.attn4
        q, k, v = a.to_q(xv), a.to_k(xv), a.to_v(xv)
        o = torch.zeros_like(xv)
        for view, nbs in block.neighboring_view_pair.items():
            for u in nbs:
                o[:, view] += a.to_out[0](R.sdpa(q[:, view], k[:, u], v[:, u], a.heads, a.scale))
        h = block.connector(o.reshape_as(xx)) + h; out.append(h)
        h = block.ff(block.norm3(h)) + h; out.append(h)
    return out


ref = stages(ora, False)
flo = stages(ora, True)
print("floor      : " + "  ".join("%.3e" % rel(f, r) for f, r in zip(flo, ref)))

from dualdiff_amd import ops as O                              # noqa: E402
from dualdiff_amd.networks import blocks as B, layers as Ly     # noqa: E402


def hip(**sw):
    Ly.HEAD_MAJOR = sw.get("head_major", True)
    blk = B.BasicMultiviewTransformerBlock(640, 8, 80, cross_attention_dim=768, neighboring_view_pair=PAIR)
    blk.load_state_dict(sd)
    blk = blk.to("cuda", dtype)
    blk.fold_connector = sw.get("fold", True)
    out = []
    with torch.no_grad():
        h = x.cuda().to(dtype).reshape(-1, 640)
        c2 = ctx.cuda().to(dtype).reshape(-1, 768)
        h = blk._attn(blk.attn1, blk.norm1, h, 6, 350); out.append(h)
        h = blk._attn(blk.attn2, blk.norm2, h, 6, 350, c2, 30); out.append(h)
        h = blk._cross_view(h, 6, 350); out.append(h)
        h = blk.ff.run(h, res=h, norm=blk.norm3); out.append(h)
    return [rel(o.reshape(6, 350, 640), r) for o, r in zip(out, ref)]


for name, sw in (("default", {}), ("no connector fold", {"fold": False}),
                 ("row-major qkv", {"head_major": False}),
                 ("all off", {"fold": False, "head_major": False})):
    print("%-18s: " % name + "  ".join("%.3e" % e for e in hip(**sw)))
