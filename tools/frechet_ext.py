"""EXTENSION report for BASELINE configs[4]: how far do fp8 attention weights (+ a folded LoRA) move the SAMPLES?

configs[4] asks for "FID vs reference on fixed-seed synthetic conds".  FID needs Inception weights (the reference
downloads them, misc/inception.py:13; offline here) and real images, so this reports what can be computed:
the Frechet distance between two sets of final LATENTS in a FIXED random-projection feature space
(seeded 64-d Gaussian projection of each view's 4 x 28 x 50 latent) — NOT FID, labelled as such:

    FD(A, B) = |mu_A - mu_B|^2 + Tr(S_A + S_B - 2 (S_A S_B)^(1/2))

Sets: N scenes (different noise / condition seeds) x 6 views, `steps` DDIM steps of the full dual-branch
sampler, (a) 16-bit weights, (b) fp8 projections; plus FD between two disjoint halves of (a) as the
sampling-noise yardstick.  python tools/frechet_ext.py [scenes] [steps] [lora_rank]"""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench                                                                      # noqa: E402
from dualdiff_amd.pipeline.pipeline_bev_controlnet import BEVDenoiser              # noqa: E402


def frechet(a, b):
    from scipy import linalg
    mu1, mu2 = a.mean(0), b.mean(0)
    s1, s2 = np.cov(a, rowvar=False), np.cov(b, rowvar=False)
    covmean, _ = linalg.sqrtm(s1.dot(s2), disp=False)
    covmean = covmean.real
    return float(((mu1 - mu2) ** 2).sum() + np.trace(s1) + np.trace(s2) - 2 * np.trace(covmean))


def sample(fp8, n_scenes, steps, lora_rank, dtype=torch.float16):
    dev = torch.device("cuda", 0)
    unet, cns = bench.build_models(dtype, dev, fp8=fp8, lora_rank=lora_rank)
    den = BEVDenoiser(unet, cns, guidance_scale=2.0, num_inference_steps=steps, use_graph=True)
    outs = []
    with torch.no_grad():
        for s in range(n_scenes):
            den.set_inputs(*bench.synthetic_inputs(1, dtype, dev, seed=5000 + s))
            den.run(steps)
            outs.append(den.latents.float().cpu()[0])                              # (6, 4, 28, 50)
    return torch.cat(outs).reshape(n_scenes * 6, -1)


def main():
    n_scenes = int(sys.argv[1]) if len(sys.argv) > 1 else 24
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    rank = int(sys.argv[3]) if len(sys.argv) > 3 else 4
    g = torch.Generator().manual_seed(2024)
    proj = torch.randn((4 * 28 * 50, 64), generator=g) / (4 * 28 * 50) ** 0.5
    a = sample(False, n_scenes, steps, rank)
    fa = (a @ proj).numpy()
    half = fa.shape[0] // 2
    out = {"metric": "Frechet distance in a fixed 64-d random projection of the final latents — NOT FID",
           "scenes": n_scenes, "views": 6, "ddim_steps": steps, "lora_rank_folded": rank,
           "fd_16bit_half_vs_half": frechet(fa[:half], fa[half:])}
    # "mfma": W8A8 on the fp8 matrix instruction (round 3, bench.py --fp8-weights); "weights": e4m3 weights only (round 2)
    for mode in ("mfma", "weights"):
        b = sample(mode, n_scenes, steps, rank)
        fb = (b @ proj).numpy()
        out["fd_fp8_%s_vs_16bit" % mode] = frechet(fa, fb)
        out["rel_l2_latents_fp8_%s_vs_16bit" % mode] = float((a - b).norm() / a.norm())
    print(json.dumps(out))


if __name__ == "__main__":
    main()
