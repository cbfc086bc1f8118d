"""GPU box: ONE scene on ONE GPU with the two CFG halves as two CONCURRENT launch chains (6 view-instances each, three
streams per chain, one captured graph) against the production form (12 view-instances in one chain).
The step is T(M) = 3.9 ms + 0.62 ms x M with the 3.9 ms made of per-launch latencies; two half chains beside
each other could hide each other's latencies — or pay the 3.3 GB weight stream twice.  This measures which.
    python tools/cfg_concurrent.py [steps]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench  # noqa: E402
from dualdiff_amd.pipeline.pipeline_bev_controlnet import BEVDenoiser  # noqa: E402


def timed(fn, steps, warm=5):
    for i in range(warm):
        fn(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        fn(warm + i)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    # streams per half chain: 3 (the production fork / join: 6 in all, MORE than the 4 hardware queues, which the command
    # processor then time-slices — profiles/r03_hw_queues.txt) or 1 (serial branches: 2 in all)
    par = (sys.argv[2] if len(sys.argv) > 2 else "3") == "3"
    dev, dtype = torch.device("cuda:0"), torch.float16
    torch.cuda.set_device(dev)
    unet, cns = bench.build_models(dtype, dev, frames=1, fp8=False, lora_rank=0)
    with torch.no_grad():
        inputs = bench.synthetic_inputs(1, dtype, dev, seed=1234)
        den = BEVDenoiser(unet, cns, guidance_scale=2.0, num_inference_steps=50)
        den.set_inputs(*inputs)
        init = den.lat2.clone()
        den.capture()
        ms_one = timed(lambda i: den.step(i % 50), steps)
        den.lat2.copy_(init)
        for i in range(2):
            den.step(i)
        ref = den.latents.float().clone()

        box = {}
        halves = [BEVDenoiser(unet, cns, guidance_scale=2.0, num_inference_steps=50, use_graph=False, parallel_branches=par,
                              cfg_half=h, cfg_exchange=lambda e: box["eps2"]) for h in (0, 1)]
        for d in halves:
            d.set_inputs(*inputs)
        a, b = halves
        eps2 = box["eps2"] = torch.empty((2,) + tuple(a.lat2[0].shape), dtype=dtype, device=dev)
        side = torch.cuda.Stream()

        def half_body(d, k):
            eps2[k].copy_(d._step_body().reshape(eps2[k].shape))

        def combine():
            a._scheduler_step(eps2)
            b.lat2.copy_(a.lat2)

        def set_step(i):
            a._set_step(i)
            b._set_step(i)

        def restore():
            a.lat2.copy_(init)
            b.lat2.copy_(init)

        # two graphs (ONE graph holding both chains crashes hipGraphInstantiate on this ROCm), replayed on two streams
        set_step(0)
        lanes = [torch.cuda.Stream(), torch.cuda.Stream()]
        graphs = []
        for k, d in enumerate(halves):
            lanes[k].wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(lanes[k]):
                half_body(d, k)                                    # eager warm-up: tunes the 6-instance shapes
                half_body(d, k)
                torch.cuda.synchronize()
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, stream=lanes[k]):
                    half_body(d, k)
            graphs.append(g)
        torch.cuda.synchronize()
        restore()

        def body():                                                # eager form of the same schedule
            main_s = torch.cuda.current_stream()
            for k, d in enumerate(halves):
                lanes[k].wait_stream(main_s)
                with torch.cuda.stream(lanes[k]):
                    half_body(d, k)
            for k in range(2):
                main_s.wait_stream(lanes[k])
            combine()

        def step2(i):
            set_step(i % 50)
            main_s = torch.cuda.current_stream()
            for k in range(2):
                lanes[k].wait_stream(main_s)
                with torch.cuda.stream(lanes[k]):
                    graphs[k].replay()
            for k in range(2):
                main_s.wait_stream(lanes[k])
            combine()

        if len(sys.argv) > 3 and sys.argv[3] == "onegraph":
            # both chains inside ONE captured graph: a single fork / join (nested forks — a forked chain forking its
            # ControlNet branches again — crash hipGraphInstantiate on this ROCm, so this form needs 1 stream per chain)
            s1 = torch.cuda.Stream()
            s1.wait_stream(torch.cuda.current_stream())
            g1 = torch.cuda.CUDAGraph()
            with torch.cuda.stream(s1):
                body()
                torch.cuda.synchronize()
                restore()
                with torch.cuda.graph(g1, stream=s1):
                    body()
            torch.cuda.current_stream().wait_stream(s1)
            restore()

            def step2(i):                                          # noqa: F811
                set_step(i % 50)
                g1.replay()

        for i in range(2):
            step2(i)
        got = a.latents.float()
        err = float((got - ref).norm() / ref.norm())
        restore()
        ms_two = timed(step2, steps)
        restore()
        ms_eager = timed(lambda i: (set_step(i % 50), body()), 10, warm=2)
    print("one chain, 12 view-instances : %.3f ms/step  %.2f steps/s" % (ms_one, 1e3 / ms_one))
    print("streams per half chain: %d" % (3 if par else 1))
    print("two concurrent half chains   : %.3f ms/step  %.2f steps/s   (latents after 2 steps vs one chain: rel-L2 %.2e)"
          % (ms_two, 1e3 / ms_two, err))
    print("two half chains, eager       : %.3f ms/step" % ms_eager)


if __name__ == "__main__":
    main()
