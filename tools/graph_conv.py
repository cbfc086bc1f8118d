import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from dualdiff_amd import ops as O
M=12; dt=torch.bfloat16; dev=torch.device("cuda:0")
def r(*shape, s=1.0): return (torch.randn(*shape, device="cuda")*s).to(dt)
def graph_time(fn, n=40, reps=10):
    import time
    fn(); torch.cuda.synchronize()
    g=torch.cuda.CUDAGraph(); s=torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        fn(); torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=s):
            for _ in range(n): fn()
    torch.cuda.current_stream().wait_stream(s)
    g.replay(); torch.cuda.synchronize(); t0=time.perf_counter()
    for _ in range(reps): g.replay()
    torch.cuda.synchronize(); return (time.perf_counter()-t0)/reps/n*1e6
for (cin,cout,h,w) in ((320,320,28,50),(960,320,28,50),(640,640,14,25),(1280,640,14,25)):
    x, wt, b = r(M*h*w, cin), r(cout, 9*cin, s=0.02), r(cout)
    out = torch.empty(M*h*w, cout, device="cuda", dtype=dt)
    res=[]
    for tile in (11, 17, 28, 29, 13, 18, 15):
        for sp in ((1,) if h==28 else (1,2,3)):
            res.append((graph_time(lambda: O.conv3x3(x, wt, b, M, h, w, tile=tile, split_k=sp, out=out)), tile, sp))
    res.sort()
    print((cin,cout,h,w), " | ".join("t%d s%d %.1f" % (t,s,u) for u,t,s in res[:7]))
