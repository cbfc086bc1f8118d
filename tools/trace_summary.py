"""Timeline summary of a rocprofv3 --kernel-trace CSV: per-kernel-class busy time, union busy time,
idle gaps and average concurrency over the last N graph replays.
Usage: python tools/trace_summary.py <kernel_trace.csv> [steps]"""
import csv, sys, collections, re
csv.field_size_limit(1 << 30)
rows = []
with open(sys.argv[1], newline="") as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
# a step ends with the cfg_ddim kernel
ends = [i for i, r in enumerate(rows) if "cfg_ddim" in r[2]]
lo, hi = ends[-steps - 1] + 1, ends[-1] + 1
win = rows[lo:hi]
t0, t1 = win[0][0], max(r[1] for r in win)
def cls(n):
    for k in ("conv3s", "gemm3_kernel", "gemm2_kernel", "dd_gemm_kernel", "conv3x3_thin", "splitk", "attn", "gn_stats", "gn_apply", "gn_fused", "gn_coop", "layernorm", "dd_add", "dd_scale",
              "dd_silu", "conv3x3_small", "nchw", "nhwc", "timestep", "cfg_ddim"):
        if k in n:
            if k == "gemm3_kernel":                       # pipelined dense family (round 5): <T, WM, WN, TM, TN, NSTAGE, GEGLU>
                m = re.search(r"Lb([01])EE", n)
                return "gemm3 " + ("geglu" if m and m.group(1) == "1" else "dense")
            if k == "gemm2_kernel":
                m = re.search(r"Lb([01])ELb([01])E", n)
                return "gemm2 " + ("conv" if m and m.group(1) == "1" else ("geglu" if m and m.group(2) == "1" else "dense")) if m else "gemm2"
            return k
    return "torch/other"
busy = collections.Counter(); cnt = collections.Counter()
for s, e, n in win:
    busy[cls(n)] += e - s; cnt[cls(n)] += 1
ev = []
for s, e, n in win:
    ev.append((s, 1)); ev.append((e, -1))
ev.sort()
union = 0; depth = 0; last = None
for t, d in ev:
    if depth > 0: union += t - last
    depth += d; last = t
wall = t1 - t0
print("steps %d  wall %.3f ms/step  union-busy %.3f ms/step  idle %.1f%%  sum-of-kernels %.3f ms/step  avg concurrency %.2f  launches/step %d" % (
    steps, wall / steps / 1e6, union / steps / 1e6, 100 * (1 - union / wall), sum(busy.values()) / steps / 1e6,
    sum(busy.values()) / union, len(win) / steps))
for k, v in busy.most_common():
    print("  %-16s %5d launches/step  %8.3f ms/step  avg %7.1f us" % (k, cnt[k] / steps, v / steps / 1e6, v / cnt[k] / 1e3))
# the launches that are not this library's kernels, by (short) name
other = collections.defaultdict(lambda: [0, 0])
for s, e, n in win:
    if cls(n) == "torch/other":
        m = re.search(r"(\w+Functor\w*|\w+copy_kernel\w*|CatArray\w+|gather_kernel|index\w+|Cijk\w{0,20}|\w+_kernel)", n)
        key = (m.group(1) if m else n[:60]) + (" <Half>" if "Half" in n else "") + (" <float>" if "float" in n else "")
        other[key][0] += 1; other[key][1] += e - s
for k, (c, t) in sorted(other.items(), key=lambda kv: -kv[1][1]):
    print("    other: %-60s %5.1f /step  %7.1f us/step  avg %6.1f us" % (k[:60], c / steps, t / steps / 1e3, t / c / 1e3))
