"""Shared timing helper of the measurement scripts: microseconds per call of fn() inside a HIP graph of n back-to-back
launches (the way the step runs its kernels), averaged over `reps` replays."""
import time

import torch


def graph_time(fn, n=20, reps=10):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph(); s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        fn(); torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=s):
            for _ in range(n): fn()
    torch.cuda.current_stream().wait_stream(s)
    g.replay(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): g.replay()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps / n * 1e6
