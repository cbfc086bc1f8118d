import os, time, torch, torch.distributed as dist
dist.init_process_group("gloo")
r = dist.get_rank()
torch.cuda.set_device(0)
x = torch.randn(6, 4, 28, 50, device="cuda", dtype=torch.bfloat16)
outs = [torch.empty_like(x) for _ in range(2)]
for i in range(3):
    dist.all_gather(outs, x)
torch.cuda.synchronize(); t = time.perf_counter()
for i in range(10):
    dist.all_gather(outs, x)
torch.cuda.synchronize()
if r == 0: print("gloo all_gather of a CUDA tensor: %.2f ms" % ((time.perf_counter() - t) / 10 * 1e3))
y = x.cpu(); outs_c = [torch.empty_like(y) for _ in range(2)]
t = time.perf_counter()
for i in range(10):
    dist.all_gather(outs_c, y)
if r == 0: print("gloo all_gather of a CPU tensor: %.2f ms" % ((time.perf_counter() - t) / 10 * 1e3))
dist.destroy_process_group()
