"""Mints tests/golden/trajectory_ddim50.npz: the CPU oracle's latents along a 50-step DDIM sample of the
full-width dual-branch step (tests/golden/cases.py: step_inputs / step_latents), at the checkpoints
cases.TRAJ_CHECKPOINTS — once in exact fp32 (`ref_<k>`) and once with every inter-module tensor rounded to
fp16 / bf16 (`floor_f16_<k>`, `floor_bf16_<k>`: oracle/numerics.py, the reference dtype's own rounding noise, the
yardstick of the drift curve).  TEST INFRASTRUCTURE; only the oracle is executed (no reference code).

    python tests/golden/mint_trajectory.py ref|floor_f16|floor_bf16 [threads] [steps]

Each mode writes its own part file after every checkpoint (a 50-step oracle run takes ~10 minutes on 16+ cores and
hours on a few — the tracked fixture was minted on the 64-core host of a GPU box, CPU only); `merge` joins the part files into the tracked fixture.
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.environ.get("DD_MINT_OUT", HERE)      # part files (the merged fixture always lands next to this script)

from oracle import dualdiff_restated as R                      # noqa: E402
from oracle.init_utils import seeded_state_dict                # noqa: E402
from oracle.numerics import storage_emulation                  # noqa: E402
from tests.golden import cases as C                            # noqa: E402


def build():
    unet = R.UNet2DConditionModelMultiview(cross_attention_dim=768, neighboring_view_pair=C.VIEW_PAIR).eval()
    unet.load_state_dict({k: C.bf16_round(v) for k, v in seeded_state_dict(unet, C.SEED_STEP_UNET).items()})
    cns = []
    for occ3d, seed in ((False, C.SEED_STEP_CNET_BG), (True, C.SEED_STEP_CNET_FG)):
        cn = R.BEVControlNetModel(use_occ_3d=occ3d).eval()
        cn.load_state_dict({k: C.bf16_round(v) for k, v in seeded_state_dict(cn, seed).items()})
        cns.append(cn)
    return unet, cns


def run(mode, threads, nsteps=50):
    torch.set_num_threads(threads)
    unet, cns = build()
    inp = C.step_inputs(2)
    boxes = [inp["boxes_bg"], inp["boxes_fg"]]
    conds = [inp["cond_bg"], inp["cond_fg"]]
    ts, ratio = R.ddim_timesteps(50)
    acp = R.ddim_alphas()
    x = C.step_latents()
    dt = {"ref": None, "floor_f16": torch.float16, "floor_bf16": torch.bfloat16}[mode]
    out = {}
    import contextlib
    ctxs = contextlib.ExitStack()
    if dt is not None:
        for m in [unet] + cns:
            ctxs.enter_context(storage_emulation(m, dt))
    with torch.no_grad(), ctxs:
        for i in range(nsteps):
            t = int(ts[i])
            x = R.denoise_step(unet, cns, x, t, inp["text"], inp["camera_param"], boxes, conds, 2.0,
                               R.ddim_coefs(acp, t, ratio))
            if dt is not None:
                x = x.to(dt).float()                           # latents are stored in the model dtype
            if i + 1 in C.TRAJ_CHECKPOINTS:
                out["%s_%d" % (mode, i + 1)] = x[0].numpy().astype(np.float32)
                np.savez_compressed(os.path.join(OUT, "trajectory_part_%s.npz" % mode), **out)
                print(mode, "step", i + 1, "|x|", float(x.norm()), flush=True)


def merge():
    out = {}
    for mode in ("ref", "floor_f16", "floor_bf16"):
        p = os.path.join(OUT, "trajectory_part_%s.npz" % mode)
        if os.path.exists(p):
            with np.load(p) as z:
                out.update({k: z[k] for k in z.files})
    np.savez_compressed(os.path.join(HERE, "trajectory_ddim50.npz"), **out)
    print("wrote", sorted(out))


if __name__ == "__main__":
    if sys.argv[1] == "merge":
        merge()
    else:
        run(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 4, int(sys.argv[3]) if len(sys.argv) > 3 else 50)
