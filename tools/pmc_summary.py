"""Summarise rocprofv3 --pmc counter_collection CSVs into per-kernel HBM traffic of the TIMED STEPS only.

Usage: python tools/pmc_summary.py <fetch_pass_dir> <write_pass_dir> <out.json> [<out.csv>] [--steps N]

FETCH_SIZE and WRITE_SIZE are collected in separate passes (they do not fit one pass on gfx950).
Units are KB.  Per /opt/skills/guides/MI355X_MICROARCH.md (HBM section): on gfx950 FETCH_SIZE
reports exactly half of the bytes of 16-B-per-lane streaming reads (global_load and
buffer_load...lds alike), so it is doubled; WRITE_SIZE is exact for 16-B-per-lane stores.

ATTRIBUTION.  Rounds 1-3 averaged every dispatch of a symbol in the process — including
the autotuner's launches, each of which follows a 320 MB cache-flush memset whose dirty lines are written
back under (and charged to) the launch behind it: that was the unexplained ~23 MB "floor" on the tiny GEMM
symbols, and the reason a 3-step run showed 510 dispatches of a symbol the step launches 131 times.  Now:
  * dispatches are ordered by Dispatch_Id and only those of the last N steps count (a step ends with the
    cfg_ddim kernel; N = --steps, default 2 = the command tools/refresh_profiles.sh runs);
  * rows are keyed by (symbol, grid size, workgroup size) — one row per launch SHAPE — and the per-symbol
    entries the bench line reads are the launch-weighted sums of those rows, so `traffic / algorithmic`
    compares the same population on both sides.
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

csv.field_size_limit(1 << 30)


def load(d, counter):
    """[(dispatch_id, kernel, grid, wg, value)] of one pass, in dispatch order."""
    rows = []
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(f, newline="") as fh:
            for r in csv.DictReader(fh):
                if r["Counter_Name"] != counter:
                    continue
                rows.append((int(r["Dispatch_Id"]), r["Kernel_Name"], int(r.get("Grid_Size") or 0),
                             int(r.get("Workgroup_Size") or 0), float(r["Counter_Value"])))
    rows.sort()
    return rows


def window(rows, steps):
    """Dispatches of the last `steps` denoising steps (a step ends with the cfg_ddim kernel)."""
    ends = [i for i, r in enumerate(rows) if "cfg_ddim" in r[1]]
    if len(ends) < steps + 1:
        raise SystemExit("pmc_summary: %d cfg_ddim dispatches, need %d (warm-up + %d steps)" % (len(ends), steps + 1, steps))
    return rows[ends[-steps - 1] + 1: ends[-1] + 1]


def collect(d, counter, steps):
    acc = defaultdict(lambda: [0.0, 0])
    win = window(load(d, counter), steps)
    for _, name, grid, wg, val in win:
        a = acc[(name, grid, wg)]
        a[0] += val
        a[1] += 1
    return acc, len(win)


def main():
    argv = list(sys.argv[1:])
    steps = 2
    if "--steps" in argv:
        i = argv.index("--steps")
        steps = int(argv[i + 1])
        del argv[i:i + 2]
    fdir, wdir, out_json = argv[:3]
    out_csv = argv[3] if len(argv) > 3 else None
    fetch, nf = collect(fdir, "FETCH_SIZE", steps)
    write, nw = collect(wdir, "WRITE_SIZE", steps)
    shapes = {}
    for key in sorted(set(fetch) | set(write)):
        fs, fn = fetch.get(key, (0.0, 0))
        ws, wn = write.get(key, (0.0, 0))
        if not fn or not wn:
            continue
        shapes[key] = {"launches_per_step": fn / steps, "fetch_bytes_per_launch": 2.0 * fs / fn * 1024.0,   # gfx950: x2, KB -> B
                       "write_bytes_per_launch": ws / wn * 1024.0}
        shapes[key]["hbm_bytes_per_launch"] = shapes[key]["fetch_bytes_per_launch"] + shapes[key]["write_bytes_per_launch"]
    kernels = {}
    for (name, grid, wg), r in shapes.items():
        k = kernels.setdefault(name, {"launches_fetch_pass": 0.0, "fetch_bytes_per_launch": 0.0, "write_bytes_per_launch": 0.0,
                                      "hbm_bytes_per_launch": 0.0, "shapes": []})
        n = r["launches_per_step"] * steps
        for f in ("fetch_bytes_per_launch", "write_bytes_per_launch", "hbm_bytes_per_launch"):
            k[f] = (k[f] * k["launches_fetch_pass"] + r[f] * n) / (k["launches_fetch_pass"] + n)
        k["launches_fetch_pass"] += n
        k["shapes"].append({"grid": grid, "workgroup": wg, "launches_per_step": r["launches_per_step"],
                            "hbm_bytes_per_launch": r["hbm_bytes_per_launch"]})
    for k in kernels.values():
        k["launches_per_step"] = k["launches_fetch_pass"] / steps
        k["launches_write_pass"] = k["launches_fetch_pass"]
    with open(out_json, "w") as f:
        json.dump({"unit": "bytes per launch (FETCH_SIZE x2 + WRITE_SIZE, KB->B)",
                   "population": "dispatches of the last %d steps only (%d / %d dispatches of the fetch / write pass), "
                                 "per (symbol, grid, workgroup) shape" % (steps, nf, nw),
                   "kernels": kernels}, f, indent=1)
    if out_csv:
        with open(out_csv, "w", newline="") as f:
            w = csv.writer(f)
            w.writerow(["kernel", "grid", "workgroup", "launches_per_step", "fetch_bytes_per_launch(x2 corrected)",
                        "write_bytes_per_launch", "hbm_bytes_per_launch"])
            for (name, grid, wg), r in sorted(shapes.items(), key=lambda kv: -kv[1]["hbm_bytes_per_launch"] * kv[1]["launches_per_step"]):
                w.writerow([name, grid, wg, "%.1f" % r["launches_per_step"], "%.0f" % r["fetch_bytes_per_launch"],
                            "%.0f" % r["write_bytes_per_launch"], "%.0f" % r["hbm_bytes_per_launch"]])
    print("kernel symbols: %d, launch shapes: %d, dispatches per step: %.0f" % (len(kernels), len(shapes), nf / steps))


if __name__ == "__main__":
    main()
