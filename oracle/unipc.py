"""CPU restatement of the UniPC multistep scheduler (bh2, order 2, x0-prediction) — TEST INFRASTRUCTURE ONLY.

SURVEY.md §8f N4: the sampler the reference really uses
(/root/reference/MD_txt_con_fusion/magicdrive/misc/test_utils.py:15,161-162:
`pipe.scheduler = UniPCMultistepScheduler.from_config(pipe.scheduler.config)`), driven from the loop at
pipeline/pipeline_bev_controlnet.py:381-499 (`scheduler.set_timesteps`, `scheduler.step(noise_pred, t, latents)`).

PARITY UNPINNED: the scheduler lives in diffusers (pinned 0.17.1 by the reference's requirements), which is
not in /root/reference and not installed in this image, so there is no reference output to mint golden
vectors from.  This file restates the published algorithm (Zhao et al., "UniPC: A Unified
Predictor-Corrector Framework for Fast Sampling of Diffusion Models", 2023, B(h) = e^h - 1 variant, and the
procedure of diffusers' `UniPCMultistepScheduler.step / multistep_uni_p_bh_update /
multistep_uni_c_bh_update` with the SD-v1.5 config: scaled-linear betas 0.00085..0.012, 1000 train steps,
solver_order 2, epsilon prediction, predict_x0, lower_order_final, no thresholding, no corrector skips).
It is written procedurally (lists of past outputs, explicit predictor / corrector calls) on purpose: the
product folds every step into one linear combination with precomputed coefficients
(dualdiff_amd/pipeline/schedulers.py), and the tests check that folding against this step-by-step form.
"""
import math

import numpy as np
import torch


class UniPCRestated:
    def __init__(self, num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012, solver_order=2,
                 lower_order_final=True):
        betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps, dtype=torch.float64) ** 2
        acp = torch.cumprod(1.0 - betas, dim=0)
        self.alpha_t = acp.sqrt()
        self.sigma_t = (1 - acp).sqrt()
        self.lambda_t = self.alpha_t.log() - self.sigma_t.log()
        self.num_train_timesteps = num_train_timesteps
        self.solver_order = solver_order
        self.lower_order_final = lower_order_final

    def set_timesteps(self, num_inference_steps):
        ts = np.linspace(0, self.num_train_timesteps - 1, num_inference_steps + 1).round()[::-1][:-1].copy().astype(np.int64)
        _, first = np.unique(ts, return_index=True)          # duplicates removed, order kept
        self.timesteps = ts[np.sort(first)]
        self.model_outputs = [None] * self.solver_order
        self.timestep_list = [None] * self.solver_order
        self.lower_order_nums = 0
        self.last_sample = None
        self.this_order = None
        return self.timesteps

    # x0 = (x - sigma_t eps) / alpha_t
    def convert_model_output(self, eps, t, sample):
        return (sample - self.sigma_t[t] * eps) / self.alpha_t[t]

    def _rb(self, rks, h, order):
        hh = -h
        h_phi_1 = math.expm1(hh)
        h_phi_k = h_phi_1 / hh - 1
        B_h = math.expm1(hh)
        fact = 1
        R, b = [], []
        for i in range(1, order + 1):
            R.append([rk ** (i - 1) for rk in rks])
            b.append(h_phi_k * fact / B_h)
            fact *= i + 1
            h_phi_k = h_phi_k / hh - 1 / fact
        return np.array(R, dtype=np.float64), np.array(b, dtype=np.float64), h_phi_1, B_h

    def predictor(self, prev_t, sample, order):
        s0, t = self.timestep_list[-1], prev_t
        m0 = self.model_outputs[-1]
        lam_t, lam_s0 = float(self.lambda_t[t]), float(self.lambda_t[s0])
        a_t, sg_t, sg_s0 = float(self.alpha_t[t]), float(self.sigma_t[t]), float(self.sigma_t[s0])
        h = lam_t - lam_s0
        rks, D1s = [], []
        for i in range(1, order):
            si, mi = self.timestep_list[-(i + 1)], self.model_outputs[-(i + 1)]
            rk = (float(self.lambda_t[si]) - lam_s0) / h
            rks.append(rk)
            D1s.append((mi - m0) / rk)
        rks.append(1.0)
        R, b, h_phi_1, B_h = self._rb(rks, h, order)
        x_t_ = sg_t / sg_s0 * sample - a_t * h_phi_1 * m0
        if D1s:
            rhos_p = np.array([0.5]) if order == 2 else np.linalg.solve(R[:-1, :-1], b[:-1])
            pred = sum(float(r) * d for r, d in zip(rhos_p, D1s))
        else:
            pred = 0
        return x_t_ - a_t * B_h * pred

    def corrector(self, this_x0, this_t, last_sample, order):
        s0, t = self.timestep_list[-1], this_t
        m0 = self.model_outputs[-1]
        lam_t, lam_s0 = float(self.lambda_t[t]), float(self.lambda_t[s0])
        a_t, sg_t, sg_s0 = float(self.alpha_t[t]), float(self.sigma_t[t]), float(self.sigma_t[s0])
        h = lam_t - lam_s0
        rks, D1s = [], []
        for i in range(1, order):
            si, mi = self.timestep_list[-(i + 1)], self.model_outputs[-(i + 1)]
            rk = (float(self.lambda_t[si]) - lam_s0) / h
            rks.append(rk)
            D1s.append((mi - m0) / rk)
        rks.append(1.0)
        R, b, h_phi_1, B_h = self._rb(rks, h, order)
        rhos_c = np.array([0.5]) if order == 1 else np.linalg.solve(R, b)
        x_t_ = sg_t / sg_s0 * last_sample - a_t * h_phi_1 * m0
        corr = sum(float(r) * d for r, d in zip(rhos_c[:-1], D1s)) if D1s else 0
        return x_t_ - a_t * B_h * (corr + float(rhos_c[-1]) * (this_x0 - m0))

    def step(self, eps, t, sample):
        """One `scheduler.step(noise_pred, t, latents)`: returns the previous (less noisy) sample."""
        idx = int(np.nonzero(self.timesteps == t)[0][0])
        use_corrector = idx > 0 and self.last_sample is not None
        x0 = self.convert_model_output(eps, t, sample)
        if use_corrector:
            sample = self.corrector(x0, t, self.last_sample, self.this_order)
        prev_t = 0 if idx == len(self.timesteps) - 1 else int(self.timesteps[idx + 1])
        for i in range(self.solver_order - 1):
            self.model_outputs[i] = self.model_outputs[i + 1]
            self.timestep_list[i] = self.timestep_list[i + 1]
        self.model_outputs[-1] = x0
        self.timestep_list[-1] = t
        order = min(self.solver_order, len(self.timesteps) - idx) if self.lower_order_final else self.solver_order
        self.this_order = min(order, self.lower_order_nums + 1)
        self.last_sample = sample
        prev = self.predictor(prev_t, sample, self.this_order)
        if self.lower_order_nums < self.solver_order:
            self.lower_order_nums += 1
        return prev
