"""Per-kernel MFMA-pipe utilisation from a rocprofv3 --pmc pass with
SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_WAVES.

Usage: python tools/pmc_mfma_summary.py <pass_dir> <out.csv>

gfx950: SQ_VALU_MFMA_BUSY_CYCLES sums the busy cycles of the MFMA pipes of all 1024 SIMDs (it equals
16 x the number of v_mfma_f32_16x16x32 instructions); GRBM_GUI_ACTIVE is summed over the 8 XCDs, so the
kernel's duration in shader clocks is GUI_ACTIVE / 8 and
    MFMA utilisation = MFMA_BUSY / (GUI_ACTIVE / 8 * 1024).
SQ_ACTIVE_INST_VALU is in quad-cycles: VALU issue utilisation = 4 * ACTIVE_INST_VALU / (GUI_ACTIVE / 8 * 1024).
"""
import csv
import glob
import os
import sys
from collections import defaultdict

csv.field_size_limit(1 << 30)


def main():
    d, out = sys.argv[1:3]
    steps = int(sys.argv[3]) if len(sys.argv) > 3 else 2
    acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
    recs = []
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(f, newline="") as fh:
            for r in csv.DictReader(fh):
                recs.append((int(r["Dispatch_Id"]), r["Kernel_Name"], r["Counter_Name"], float(r["Counter_Value"])))
    # only the dispatches of the last `steps` denoising steps (a step ends with the cfg_ddim kernel): the tuner's
    # and the warm-up's launches are not the population the bench line describes (tools/pmc_summary.py)
    recs.sort()
    ends = sorted({r[0] for r in recs if "cfg_ddim" in r[1]})
    if len(ends) >= steps + 1:
        lo, hi = ends[-steps - 1], ends[-1]
        recs = [r for r in recs if lo < r[0] <= hi]
    for _, name, cname, val in recs:
        a = acc[name][cname]
        a[0] += val
        a[1] += 1
    rows = []
    for name, c in acc.items():
        if "GRBM_GUI_ACTIVE" not in c or "SQ_VALU_MFMA_BUSY_CYCLES" not in c:
            continue
        n = c["GRBM_GUI_ACTIVE"][1]
        gui = c["GRBM_GUI_ACTIVE"][0] / n / 8.0
        mfma = c["SQ_VALU_MFMA_BUSY_CYCLES"][0] / n
        valu = 4.0 * c["SQ_ACTIVE_INST_VALU"][0] / n if "SQ_ACTIVE_INST_VALU" in c else float("nan")
        insts = c["SQ_INSTS_MFMA"][0] / n if "SQ_INSTS_MFMA" in c else float("nan")
        rows.append((gui * n, name, n, gui, mfma / (gui * 1024.0), valu / (gui * 1024.0), insts))
    rows.sort(reverse=True)
    with open(out, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["kernel", "launches", "avg_shader_cycles", "mfma_pipe_utilisation", "valu_issue_utilisation",
                    "mfma_instructions_per_launch"])
        for _, name, n, gui, mu, vu, insts in rows:
            w.writerow([name, n, "%.0f" % gui, "%.3f" % mu, "%.3f" % vu, "%.0f" % insts])
    tot = sum(r[0] for r in rows)
    print("kernels:", len(rows), " time-weighted MFMA utilisation of MFMA kernels: %.3f" %
          (sum(r[0] * r[4] for r in rows if r[4] > 0) / max(sum(r[0] for r in rows if r[4] > 0), 1.0)),
          " of all kernels: %.3f" % (sum(r[0] * r[4] for r in rows) / max(tot, 1.0)))


if __name__ == "__main__":
    main()
