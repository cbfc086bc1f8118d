"""UniPC (N4): the folded per-step coefficient table against the step-by-step restatement, and
size-independent properties of the sampler itself.  CPU only."""
import numpy as np
import pytest
import torch

from dualdiff_amd.pipeline.schedulers import UNIPC_NCOEF, unipc_schedule, unipc_timesteps
from oracle.unipc import UniPCRestated


def folded_run(tab, x, eps_seq):
    last = torch.zeros_like(x)
    m1 = torch.zeros_like(x)
    m2 = torch.zeros_like(x)
    for row, eps in zip(tab.tolist(), eps_seq):
        a_x, a_e, use_c, c_l, c_1, c_2, c_0, p_x, p_0, p_1 = row
        x0 = a_x * x + a_e * eps
        xc = c_l * last + c_1 * m1 + c_2 * m2 + c_0 * x0 if use_c else x
        x, last, m2, m1 = p_x * xc + p_0 * x0 + p_1 * m1, xc, m1, x0
    return x


@pytest.mark.parametrize("steps", [1, 2, 3, 5, 20, 50])
def test_folded_table_equals_procedural_scheduler(steps):
    sch = UniPCRestated()
    ts = sch.set_timesteps(steps)
    ts2, tab = unipc_schedule(steps)
    assert tab.shape == (len(ts), UNIPC_NCOEF) and tab.dtype == torch.float32
    assert np.array_equal(ts2.numpy(), ts)
    g = torch.Generator().manual_seed(steps)
    x = torch.randn(257, generator=g, dtype=torch.float64)
    eps_seq = [torch.randn(257, generator=g, dtype=torch.float64) for _ in ts]
    ref = x.clone()
    for t, eps in zip(ts.tolist(), eps_seq):
        ref = sch.step(eps, t, ref)
    got = folded_run(tab.double(), x, eps_seq)
    assert (got - ref).abs().max().item() <= 5e-7 * ref.abs().max().item()


def test_timesteps_are_the_reference_samplers_spacing():
    ts = unipc_timesteps(20)
    assert ts[0] == 999 and len(ts) == 20 and np.all(np.diff(ts) < 0)
    assert list(ts[:3]) == [999, 949, 899]          # linspace(0, 999, 21).round() reversed, last dropped


@pytest.mark.parametrize("steps", [4, 20])
def test_ideal_denoiser_is_integrated_exactly(steps):
    """With a model whose x0-prediction never changes (eps consistent with one clean sample) every
    UniPC update is the exact solution of the probability-flow ODE: the sample stays on
    x_t = alpha_t x0 + sigma_t n for the initial noise n, down to t = 0."""
    sch = UniPCRestated()
    ts = sch.set_timesteps(steps)
    g = torch.Generator().manual_seed(3)
    x0 = torch.randn(64, generator=g, dtype=torch.float64)
    n = torch.randn(64, generator=g, dtype=torch.float64)
    x = sch.alpha_t[999] * x0 + sch.sigma_t[999] * n
    for i, t in enumerate(ts.tolist()):
        eps = (x - sch.alpha_t[t] * x0) / sch.sigma_t[t]
        x = sch.step(eps, t, x)
        nxt = 0 if i == len(ts) - 1 else int(ts[i + 1])
        want = sch.alpha_t[nxt] * x0 + sch.sigma_t[nxt] * n
        assert (x - want).abs().max().item() < 1e-9


def test_first_and_last_steps_are_first_order():
    _, tab = unipc_schedule(20)
    assert tab[0, 2] == 0 and tab[0, 9] == 0            # no corrector, order-1 predictor on the first step
    assert tab[-1, 9] == 0 and tab[-1, 2] == 1          # lower_order_final: last predictor is order 1
    assert (tab[1:, 2] == 1).all()
    assert (tab[1:-1, 9] != 0).all()
