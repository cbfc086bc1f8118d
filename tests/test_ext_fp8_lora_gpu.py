"""EXTENSION (BASELINE configs[4], no reference semantics): LoRA deltas folded into the attention projections and
fp8 (e4m3fn) weights for them.  Oracle = the CPU multiview block with (a) the LoRA path evaluated explicitly
(`W x + scale * up(down x)`), (b) the projection weights replaced by their dequantised fp8 values."""
import os

import pytest
import torch

from oracle import dualdiff_restated as R
from oracle.init_utils import seeded_state_dict, seeded_tensor
from oracle.numerics import storage_emulation
from tests.golden import cases as C
from tests.parity_util import rel_l2, report

pytestmark = pytest.mark.gpu
PAIR = C.VIEW_PAIR
torch.set_num_threads(min(32, os.cpu_count() or 1))


def _block_case(dim=640, n=350):
    ora = R.BasicMultiviewTransformerBlock(dim, 8, dim // 8, cross_attention_dim=768, neighboring_view_pair=PAIR).eval()
    sd = {k: C.bf16_round(v) for k, v in seeded_state_dict(ora, 5).items()}
    ora.load_state_dict(sd)
    x = C.bf16_round(seeded_tensor((6, n, dim), 1))
    ctx = C.bf16_round(seeded_tensor((6, 30, 768), 2))
    return ora, sd, x, ctx


def _hip_block(sd, dtype, dim=640):
    from dualdiff_amd.networks.blocks import BasicMultiviewTransformerBlock
    blk = BasicMultiviewTransformerBlock(dim, 8, dim // 8, cross_attention_dim=768, neighboring_view_pair=PAIR)
    blk.load_state_dict(sd)
    return blk.to("cuda", dtype)


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_lora_fold_matches_explicit_low_rank_path(gpu, dtype):
    from dualdiff_amd.lora import fold_lora_, lora_keys
    ora, sd, x, ctx = _block_case()
    blk = _hip_block(sd, dtype)
    rank, scale = 4, 0.8
    lora = {k: C.bf16_round(seeded_tensor(shape, 900 + i, 0.05)) for i, (k, shape) in enumerate(sorted(lora_keys(blk, rank).items()))}
    assert len(lora) == 3 * 4 * 2                                     # attn1, attn2, attn4 x q, k, v, out x down, up
    # oracle: explicit low-rank branch on every projection (diffusers LoRAAttnProcessor semantics)
    hooks = []
    for name, mod in ora.named_modules():
        for proj in ("to_q", "to_k", "to_v", "to_out"):
            key = "%s.processor.%s_lora.down.weight" % (name, proj)
            if key in lora:
                lin = mod.to_out[0] if proj == "to_out" else getattr(mod, proj)
                dn, up = lora[key], lora[key.replace(".down.", ".up.")]
                hooks.append(lin.register_forward_hook(
                    lambda m, a, out, dn=dn, up=up: out + scale * (a[0] @ dn.t()) @ up.t()))
    with torch.no_grad():
        ref = ora(x, encoder_hidden_states=ctx)
    for h in hooks:
        h.remove()
    assert fold_lora_(blk, lora, scale) == 12
    with torch.no_grad():
        y = blk.run(x.cuda().to(dtype).reshape(-1, 640), 6, 350, ctx.cuda().to(dtype).reshape(-1, 768), 30)
        base = ora(x, encoder_hidden_states=ctx)
    e = rel_l2(y.reshape(6, 350, 640), ref)
    live = rel_l2(base, ref)
    print("LoRA-folded block vs explicit low-rank oracle: rel-L2 %.3e (the adapter moves the output by %.3e)" % (e, live))
    assert e <= (2e-3 if dtype == torch.float16 else 8e-3) and live > 5 * e


