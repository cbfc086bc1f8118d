"""Shared reporting of the end-to-end parity tests: metric, bound, and the tracked error table.

Metric: relative L2 error e(y) = ||y - ref||_2 / ||ref||_2 against the fp32 CPU oracle.
Bound:  e(HIP) <= max(1e-3, 1.1 * e_floor) where e_floor is the error of the oracle itself when every
inter-module tensor is rounded to the storage dtype (oracle/numerics.py) — north_star's 1e-3 wherever the
storage dtype allows it, and at most 10 % above the reference dtype's own rounding noise elsewhere.
Every comparison appends `name, dtype, e_hip, e_floor, bound` to the CSV named by DD_PARITY_CSV
(default gpurun_out/r02_parity.csv; the copy judged is profiles/r02_parity.csv).
"""
import os

import torch

FLOOR_SLACK = 1.1
_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def bound(e_floor):
    return max(1e-3, FLOOR_SLACK * e_floor)


def rel_l2(y, ref):
    y, ref = y.detach().float().cpu(), ref.float()
    return ((y - ref).norm() / (ref.norm() + 1e-20)).item()


def _csv_path():
    return os.environ.get("DD_PARITY_CSV", os.path.join(_ROOT, "gpurun_out", "r02_parity.csv"))


def log_row(name, dtype, e_hip, e_floor, bnd):
    path = _csv_path()
    try:
        os.makedirs(os.path.dirname(path), exist_ok=True)
        new = not os.path.exists(path)
        with open(path, "a") as f:
            if new:
                f.write("name,dtype,e_hip,e_floor,bound\n")
            f.write("%s,%s,%.4e,%.4e,%.4e\n" % (name.replace(",", ";"), str(dtype).split(".")[-1], e_hip, e_floor, bnd))
    except OSError:
        pass


def report(name, y, ref, dtype, record, emul=None):
    """Prints and logs e(HIP) and the reference-dtype noise floor; returns e(HIP) / bound."""
    e = rel_l2(y, ref)
    fl = rel_l2(emul, ref) if emul is not None else 0.0
    print("%-40s %-8s e_hip=%.3e  e_floor=%.3e  bound=%.3e" % (name, str(dtype).split(".")[-1], e, fl, bound(fl)))
    record.append((name, e, fl))
    log_row(name, dtype, e, fl, bound(fl))
    assert torch.isfinite(y).all(), name
    return e / bound(fl)
