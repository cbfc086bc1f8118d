"""GPU diagnostic: one dense GEMM per kernel family against torch, each in its own process (a faulting kernel kills only
its case).  DD_HIP_LIB picks the library, DD_XS_COLS the tile-order slicing."""
import os, subprocess, sys
CASES = [(44, 16800, 960, 320, 0), (44, 67200, 960, 320, 0), (72, 4200, 640, 640, 0), (72, 67200, 960, 320, 0), (75, 67200, 960, 320, 0),
         (75, 16800, 5120, 640, 1), (75, 4200, 1920, 640, 0), (73, 67200, 320, 320, 0), (78, 67200, 320, 320, 0), (0, 4200, 5120, 640, 1)]
if len(sys.argv) > 1:
    import torch
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from dualdiff_amd import ops as O
    tile, rows, n, k, geglu = CASES[int(sys.argv[1])]
    dev = torch.device("cuda")
    torch.manual_seed(0)
    x = torch.randn(rows, k, device=dev).half()
    w = (torch.randn(n * (2 if geglu else 1), k, device=dev) * k ** -0.5).half()
    b = torch.randn(n * (2 if geglu else 1), device=dev).half()
    O.workspace(256 << 20, dev)
    if geglu:
        y = O.gemm(x, w, b, tile=tile, epilogue=O.DD_EPI_GEGLU)
        r = x.float() @ w.float().t() + b.float()
        r = r[:, :n] * torch.nn.functional.gelu(r[:, n:])
    else:
        y = O.gemm(x, w, b, tile=tile, split_k=1)
        r = x.float() @ w.float().t() + b.float()
    torch.cuda.synchronize()
    err = float((y.float() - r).abs().max())
    print("max err %.4f %s" % (err, "OK" if err < 0.05 else "WRONG"))
    sys.exit(0)
for i, c in enumerate(CASES):
    r = subprocess.run([sys.executable, os.path.abspath(__file__), str(i)], capture_output=True, text=True, timeout=300)
    msg = (r.stdout.strip().splitlines() or [""])[-1]
    if r.returncode != 0:
        msg = "rc=%d %s" % (r.returncode, [l for l in (r.stdout + r.stderr).splitlines() if "fault" in l.lower() or "Error" in l][-1:])
    print("lib=%s xs=%s case %s: %s" % (os.path.basename(os.environ.get("DD_HIP_LIB", "product")), os.environ.get("DD_XS_COLS", "-"), c, msg), flush=True)
