"""Per-kernel micro-benchmarks on the real layer shapes of config 2 (M = 12 instances).

Sweeps the dd_gemm tile configs / split-K factors and times attention + norm kernels with HIP
events on torch's current stream.  Writes JSON lines to gpurun_out/bench_ops.jsonl; the
heuristics in csrc/gemm.hip are tuned from this output.
Usage: python tools/bench_ops.py [--dtype bf16|fp16] [--quick]
"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dualdiff_amd import ops as O  # noqa: E402

M = 12
TILES = [1, 2, 3, 4, 5, 11, 12, 13, 14, 15, 16, 17, 18, 19, 20]
LEVELS = [(28, 50, 320), (14, 25, 640), (7, 13, 1280), (4, 7, 1280)]


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e-3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--quick", action="store_true")
    ap.add_argument("--out", default="gpurun_out/bench_ops.jsonl")
    args = ap.parse_args()
    dt = torch.bfloat16 if args.dtype == "bf16" else torch.float16
    dev = torch.device("cuda:0")
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    out = open(args.out, "a")

    def emit(rec):
        rec["dtype"] = args.dtype
        line = json.dumps(rec)
        print(line, flush=True)
        out.write(line + "\n")
        out.flush()

    def r(*shape, s=1.0):
        return (torch.randn(*shape, device=dev) * s).to(dt)

    # ---------------- dense GEMMs
    dense = []
    for (h, w, c) in LEVELS[:3]:
        rows = M * h * w
        dense += [(rows, 3 * c, c, "qkv"), (rows, c, c, "proj"), (rows, c, 4 * c, "ff2")]
    dense += [(M * 98, 2 * 320, 768, "cross_kv_L0"), (M * 98, 2 * 1280, 768, "cross_kv_L2"),
              (M * 28, 3 * 1280, 1280, "qkv_mid"), (M * 28, 1280, 5120, "ff2_mid"),
              (M, 1280, 320, "time_emb")]
    if args.quick:
        dense = dense[:4]
    for rows, n, k, name in dense:
        a, wt, b = r(rows, k), r(n, k, s=0.05), r(n)
        res = r(rows, n)
        best = None
        for tile in TILES:
            for split in ([1] if rows > 2000 else [1, 2, 4, 8]):
                try:
                    t = timeit(lambda: O.gemm(a, wt, b, res=res, tile=tile, split_k=split))
                except RuntimeError as ex:
                    continue
                tf = 2.0 * rows * n * k / t / 1e12
                emit({"op": "gemm", "name": name, "rows": rows, "n": n, "k": k, "tile": tile,
                      "split": split, "us": t * 1e6, "tflops": tf})
                if best is None or t < best[0]:
                    best = (t, tile, split)
        t = timeit(lambda: O.gemm(a, wt, b, res=res))
        emit({"op": "gemm_auto", "name": name, "rows": rows, "n": n, "k": k, "us": t * 1e6,
              "tflops": 2.0 * rows * n * k / t / 1e12, "best_us": best[0] * 1e6,
              "best_tile": best[1], "best_split": best[2], "plan": O.gemm_kernel_name(rows, n, k, dt)})
    # GEGLU
    for (h, w, c) in LEVELS[:3]:
        rows = M * h * w
        a, wt, b = r(rows, c), r(8 * c, c, s=0.05), r(8 * c)
        for tile in [1, 3, 5, 11, 12, 14, 16, 20]:
            t = timeit(lambda: O.gemm(a, wt, b, epilogue=O.DD_EPI_GEGLU, tile=tile))
            emit({"op": "geglu", "rows": rows, "n": 4 * c, "k": c, "tile": tile, "us": t * 1e6,
                  "tflops": 2.0 * rows * 8 * c * c / t / 1e12})

    # ---------------- convs
    convs = [(28, 50, 320, 320, 1, None), (28, 50, 640, 320, 1, None), (28, 50, 960, 320, 1, None),
             (28, 50, 320, 320, 2, None),
             (14, 25, 640, 640, 1, None), (14, 25, 1280, 640, 1, None), (14, 25, 1920, 640, 1, None),
             (14, 25, 320, 640, 1, None), (14, 25, 640, 640, 1, (28, 50)),
             (7, 13, 1280, 1280, 1, None), (7, 13, 2560, 1280, 1, None), (7, 13, 640, 1280, 1, None),
             (4, 7, 1280, 1280, 1, None), (4, 7, 2560, 1280, 1, None)]
    if args.quick:
        convs = convs[:2]
    for (h, w, cin, cout, stride, up) in convs:
        x, wt, b = r(M * h * w, cin), r(cout, 9 * cin, s=0.02), r(cout)
        hv, wv = (h, w) if up is None else up
        ho, wo = (hv - 1) // stride + 1, (wv - 1) // stride + 1
        rows = M * ho * wo
        flops = 2.0 * rows * cout * 9 * cin
        best = None
        for tile in TILES:
            for split in ([1] if rows > 4000 else [1, 2, 4, 8, 16]):
                try:
                    t = timeit(lambda: O.conv3x3(x, wt, b, M, h, w, stride=stride, up_size=up,
                                                 tile=tile, split_k=split))
                except RuntimeError:
                    continue
                emit({"op": "conv", "hw": [h, w], "cin": cin, "cout": cout, "stride": stride,
                      "up": up, "tile": tile, "split": split, "us": t * 1e6, "tflops": flops / t / 1e12})
                if best is None or t < best[0]:
                    best = (t, tile, split)
        t = timeit(lambda: O.conv3x3(x, wt, b, M, h, w, stride=stride, up_size=up))
        emit({"op": "conv_auto", "hw": [h, w], "cin": cin, "cout": cout, "stride": stride, "up": up,
              "us": t * 1e6, "tflops": flops / t / 1e12, "best_us": best[0] * 1e6,
              "best_tile": best[1], "best_split": best[2]})

    # ---------------- attention
    for (lq, lk, d, name) in [(1400, 1400, 40, "self_L0"), (350, 350, 80, "self_L1"),
                              (91, 91, 160, "self_L2"), (28, 28, 160, "self_mid"),
                              (1400, 98, 40, "cross_L0"), (350, 98, 80, "cross_L1"),
                              (91, 98, 160, "cross_L2")]:
        c = 8 * d
        q, k, v = r(M * lq, c), r(M * lk, c), r(M * lk, c)
        for variant in [0]:
            t = timeit(lambda: O.attention(q, k, v, M, lq, lk, 8, d, variant=variant))
            emit({"op": "attention", "name": name, "lq": lq, "lk": lk, "d": d, "variant": variant,
                  "us": t * 1e6, "tflops": 4.0 * M * 8 * lq * lk * d / t / 1e12})

    # ---------------- norms (HBM-bound): bytes = read + write once
    for (h, w, c) in LEVELS:
        rows = M * h * w
        x, g, b = r(rows, c), r(c), r(c)
        t = timeit(lambda: O.groupnorm(x, g, b, M, h * w, 32, 1e-5, True))
        emit({"op": "groupnorm_silu", "rows": rows, "c": c, "us": t * 1e6,
              "gbps": 2.0 * rows * c * 2 / t / 1e9})
        t = timeit(lambda: O.layernorm(x, g, b))
        emit({"op": "layernorm", "rows": rows, "c": c, "us": t * 1e6,
              "gbps": 2.0 * rows * c * 2 / t / 1e9})
    a, b2 = r(M * 1400 * 320), r(M * 1400 * 320)
    t = timeit(lambda: O.add(a, b2))
    emit({"op": "add", "n": a.numel(), "us": t * 1e6, "gbps": 3.0 * a.numel() * 2 / t / 1e9})


if __name__ == "__main__":
    main()
