"""BASELINE configs[2]: the SFA cross-attention at batch = 4 scenes (48 view-instances) on its own — every launch of
`txt_con_XFormersAttn` (and the `_plus` variant) timed hot in a graph chain and classified against the MFMA and HBM
roofs (2.5 PFLOP/s, 8 TB/s).  python tools/sfa_roofline.py [fp16|bf16]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dualdiff_amd import ops as O
from dualdiff_amd.networks.txt_con_fusion import txt_con_XFormersAttn, txt_con_XFormersAttn_plus
from tools._timing import graph_time
dt = torch.bfloat16 if (len(sys.argv) > 1 and sys.argv[1] == "bf16") else torch.float16
dev = torch.device("cuda")
M, LQ, LK, C, H = 48, 1400, 77, 320, 8
D = C // H
PEAK_F, PEAK_B = 2.5e15, 8.0e12


def r(*s, scale=1.0):
    return (torch.randn(*s, device=dev) * scale).to(dt)


def row(name, fn, flops, nbytes):
    t = graph_time(fn) * 1e-6
    fm, fb = flops / t / PEAK_F, nbytes / t / PEAK_B
    bound = "mfma" if flops / PEAK_F > nbytes / PEAK_B else "hbm"
    print("  %-44s %7.1f us  %7.1f TFLOP/s (%.3f of MFMA)  %6.0f GB/s (%.3f of HBM)  roof: %s, frac %.3f" %
          (name, t * 1e6, flops / t / 1e12, fm, nbytes / t / 1e9, fb, bound, fm if bound == "mfma" else fb))
    return t


x, e = r(M * LQ, C), r(M * LK, 768)
for cls in (txt_con_XFormersAttn, txt_con_XFormersAttn_plus):
    m = cls().to(dev, dt)
    for p in m.parameters():
        p.data.normal_(0, 0.03)
    print("%s, %d instances x %d tokens, %d text tokens, %s" % (cls.__name__, M, LQ, LK, str(dt).split(".")[-1]))
    with torch.no_grad():
        tot = row("whole module (all launches)", lambda: m.run(x, M, LQ, e, LK), 0.0, 0.0) if False else None
        tw = graph_time(lambda: m.run(x, M, LQ, e, LK))
        rows = M * LQ
        q = r(rows, C)
        kv = r(M * LK, 2 * C)
        t = 0.0
        t += row("to_q  %dx%dx%d" % (rows, C, C), lambda: O.gemm(x, m.to_q.weight if hasattr(m, "to_q") else m.to_q_occ.weight),
                 2.0 * rows * C * C, 2.0 * (2 * rows * C + C * C))
        if cls is txt_con_XFormersAttn:
            wkv = torch.cat([m.to_k.weight, m.to_v.weight], 0).contiguous()
            t += row("to_k|to_v (text) %dx%dx768" % (M * LK, 2 * C), lambda: O.gemm(e, wkv),
                     2.0 * M * LK * 2 * C * 768, 2.0 * (M * LK * 768 + M * LK * 2 * C + 2 * C * 768))
            t += row("SDPA occ->text lq=%d lk=%d d=%d" % (LQ, LK, D),
                     lambda: O.attention(q, kv[:, :C], kv[:, C:], M, LQ, LK, H, D, D ** -0.5),
                     4.0 * M * H * LQ * LK * D, 2.0 * (2 * rows * C + M * LK * 2 * C))
        else:
            ko, vo = r(rows, C), r(rows, C)
            t += 2 * row("to_k_occ / to_v_occ %dx%dx%d (each)" % (rows, C, C), lambda: O.gemm(x, m.to_k_occ.weight),
                         2.0 * rows * C * C, 2.0 * (2 * rows * C + C * C))
            t += 2 * row("to_k_txt / to_v_txt %dx%dx768 (each)" % (M * LK, C), lambda: O.gemm(e, m.to_k_txt.weight),
                         2.0 * M * LK * C * 768, 2.0 * (M * LK * 768 + M * LK * C + C * 768))
            t += row("SDPA occ->text lq=%d lk=%d d=%d" % (LQ, LK, D),
                     lambda: O.attention(q, kv[:, :C], kv[:, C:], M, LQ, LK, H, D, D ** -0.5),
                     4.0 * M * H * LQ * LK * D, 2.0 * (2 * rows * C + M * LK * 2 * C))
            t += row("SDPA occ self-style lq=lk=%d d=%d" % (LQ, D), lambda: O.attention(q, ko, vo, M, LQ, LQ, H, D, D ** -0.5),
                     4.0 * M * H * LQ * LQ * D, 2.0 * 4 * rows * C)
        t += row("to_out + bias + residual %dx%dx%d" % (rows, C, C), lambda: m.to_out[0].run(q, res=x),
                 2.0 * rows * C * C, 2.0 * (3 * rows * C + C * C))
        print("  sum of launches %.1f us; module as called %.1f us\n" % (t * 1e6, tw))

# ---- round 3: the fused kernel (dd_xattn320: to_q -> SDPA over the text keys -> to_out + bias + residual, one launch) ----
from dualdiff_amd.networks import layers as _layers
m = txt_con_XFormersAttn().to(dev, dt)
for p in m.parameters():
    p.data.normal_(0, 0.03)
with torch.no_grad():
    for inst in (48, 24, 18, 12):
        rows = inst * LQ
        xx, ee = r(rows, C), r(inst * LK, 768)
        kv = r(inst * LK, 2 * C)
        kvh = r(2 * H, inst * LK, D)
        print("fused txt_con_XFormersAttn (dd_xattn320), %d instances x %d tokens, %d text tokens, %s" % (inst, LQ, LK, str(dt).split(".")[-1]))
        fl = 4.0 * rows * C * C + 4.0 * inst * H * LQ * LK * D
        by = 2.0 * (3 * rows * C + 2 * C * C + inst * LK * 2 * C)          # x in, residual (the same tensor: L2), out
        row("to_q + SDPA + to_out + bias + residual (1 launch)",
            lambda: O.xattn320(xx, m.to_q.wx, m.to_out[0].wx, m.to_out[0].bias, kvh[:H], kvh[H:], inst, LQ, LK, D ** -0.5, res=xx),
            fl, by)
        row("  same, K / V as row-major column slices",
            lambda: O.xattn320(xx, m.to_q.wx, m.to_out[0].wx, m.to_out[0].bias, kv[:, :C], kv[:, C:], inst, LQ, LK, D ** -0.5, res=xx),
            fl, by)
        cap = O.XATTN_MAX_WGS
        for label, fused, wgs in (("fused kernel forced", True, 1 << 30), ("3 launches forced", False, cap),
                                  ("as DISPATCHED (fused iff <= %d workgroups)" % cap, True, cap)):
            _layers.XATTN_FUSED, O.XATTN_MAX_WGS = fused, wgs
            print("  module as called (to_k|to_v GEMM + %s): %.1f us" % (label, graph_time(lambda: m.run(xx, inst, LQ, ee, LK))))
        _layers.XATTN_FUSED, O.XATTN_MAX_WGS = True, cap
        print()
