"""UNet decoder on 12 view-instances in one chain vs the two CFG halves (6 + 6, fully independent: attn4 only couples the views
of one half) as two concurrent chains on two streams — each form as its own HIP graph, alone on the GPU."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from dualdiff_amd import ops as O
dtype = torch.float16; dev = torch.device("cuda:0")
unet, cns = bench.build_models(dtype, dev)
lat, prompt, cam, boxes, conds = bench.synthetic_inputs(1, dtype, dev, 1)
H, W, M = bench.H, bench.W, 12
lat2 = torch.cat([lat.reshape(6, 4, H, W)] * 2)
t = torch.full((M,), 500.0, device=dev)
def graph_time(fn, n=30):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph(); s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        fn(); torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=s): out = fn()
    torch.cuda.current_stream().wait_stream(s)
    for _ in range(3): g.replay()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): g.replay()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3, out
def half_state(st, lo, hi, m):
    def rows(x, hw): return x[lo * hw:hi * hw]
    out = dict(st)
    out["m"] = hi - lo
    out["x"] = rows(st["x"], st["h"] * st["w"])
    out["skips"] = [(rows(s, sh * sw), sh, sw) for s, sh, sw in st["skips"]]
    out["temb"] = {k: v[lo:hi] for k, v in st["temb"].items()}
    lc = st["lc"]
    out["ctx2d"] = st["ctx2d"][lo * lc:hi * lc]
    return out
with torch.no_grad():
    x8 = O.nchw_to_nhwc(lat2, 8)
    p = [cns[i].prepare_condition(cam, boxes[i], prompt, conds[i], False) for i in range(2)]
    r = [cns[i].forward_nhwc(x8, M, H, W, t, p[i], 1.0) for i in range(2)]
    unet.kv_bank = False                     # K/V projected per layer from the (sliced) context
    st = unet.encode_nhwc(x8, M, H, W, t, p[0]["ctx2d"], p[0]["lc"])
    down = [tuple((r[0][j][0], r[1][j][0])) for j in range(len(r[0]) - 1)]; mid = (r[0][-1][0], r[1][-1][0])
    ms, full = graph_time(lambda: unet.decode_nhwc(st, down, mid)); print("decode, 12 instances, one chain (no K/V bank): %.3f ms" % ms, flush=True)
    side = torch.cuda.Stream()
    def hw_of(j): return r[0][j][1] * r[0][j][2]
    def halves():
        main = torch.cuda.current_stream()
        outs = [None, None]
        side.wait_stream(main)
        for hf, stream in ((1, side), (0, main)):
            lo, hi = hf * 6, hf * 6 + 6
            with torch.cuda.stream(stream):
                sh = half_state(st, lo, hi, 6)
                dh = [tuple(b_[lo * hw_of(j):hi * hw_of(j)] for b_ in down[j]) for j in range(len(down))]
                mh = tuple(b_[lo * hw_of(len(r[0]) - 1):hi * hw_of(len(r[0]) - 1)] for b_ in mid)
                outs[hf] = unet.decode_nhwc(sh, dh, mh)
        main.wait_stream(side)
        return torch.cat(outs)
    ms2, split = graph_time(halves); print("decode, 6 + 6 instances on two streams:        %.3f ms" % ms2, flush=True)
    print("max |diff| %.3e" % (full.float() - split.float()).abs().max().item())
