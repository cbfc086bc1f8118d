#!/usr/bin/env python3
"""Two-launch vs in-launch split-K reduction (dd_gemm_desc.splitk_inkernel) on the step's split-K shapes: for every
(tile, split) of the tracked table that splits K, the cold-weight time of both forms (the tuner's own timing loop).
Run on the GPU box:  python tools/splitk_ab.py [fp16|bf16]"""
import ast
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dualdiff_amd import ops as O  # noqa: E402


def main():
    dt = torch.bfloat16 if len(sys.argv) > 1 and sys.argv[1] == "bf16" else torch.float16
    code = O.DD_F16 if dt == torch.float16 else O.DD_BF16
    with open(O.TUNE_TABLE_PATH) as f:
        entries = [(ast.literal_eval(k), v) for k, v in json.load(f)["entries"]]
    dev = torch.device("cuda")
    flush = torch.empty(320 << 20, dtype=torch.uint8, device=dev)

    def timed(fn, iters=15):
        ts = []
        for _ in range(iters):
            flush.zero_()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            fn()
            e1.record()
            e1.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3)
        ts.sort()
        return ts[len(ts) // 2]

    tot = [0.0, 0.0]
    for key, v in entries:
        tile, split = int(v[0]), int(v[1])
        if split <= 1:
            continue
        if key[0] == "g":
            _, rows, n, k, epi, kdt, has_a2, has_ln = key[:8]
            if kdt != code or has_ln or epi != 0:
                continue
            k1 = k // 2 if has_a2 else k
            a = torch.randn((rows, k1), device=dev).to(dt)
            a2 = torch.randn((rows, k - k1), device=dev).to(dt) if has_a2 else None
            w = (torch.randn((n, k), device=dev) * k ** -0.5).to(dt)
            res = torch.randn((rows, n), device=dev).to(dt) if "res" in key else None
            fn = lambda ink: O.gemm(a, w, a2=a2, res=res, tile=tile, split_k=split, splitk_inkernel=ink)   # noqa: E731
            label = "gemm %dx%dx%d" % (rows, n, k)
        else:
            _, m, hin, win, cin, cout, stride, hv, wv, kdt = key
            if kdt != code or stride != 1 or (hv, wv) != (hin, win):
                continue
            x = torch.randn((m * hin * win, cin), device=dev).to(dt)
            w = (torch.randn((cout, 9 * cin), device=dev) * (9 * cin) ** -0.5).to(dt)
            fn = lambda ink: O.conv3x3(x, w, None, m, hin, win, tile=tile, split_k=split, splitk_inkernel=ink)   # noqa: E731
            label = "conv %dx%dx%d (%dx%d)" % (m * hin * win, cout, 9 * cin, hin, win)
        y0, y1 = fn(0), fn(1)
        same = torch.equal(y0, y1)
        t0 = timed(lambda: fn(0))
        t1 = timed(lambda: fn(1))
        tot[0] += t0
        tot[1] += t1
        print("%-34s tile %2d split %2d  two-launch %7.1f us  in-launch %7.1f us  %+5.1f %%  bitwise %s"
              % (label, tile, split, t0, t1, (t1 / t0 - 1) * 100, same), flush=True)
    print("sum over table entries: two-launch %.1f us, in-launch %.1f us" % tuple(tot))


if __name__ == "__main__":
    main()
