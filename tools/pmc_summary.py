#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc passes of bench.py per kernel (mean per launch).

    python tools/pmc_summary.py OUT.json [steps=N] NAME=DIR/PREFIX [NAME=DIR/PREFIX ...]

Each DIR/PREFIX names one pass (PREFIX_counter_collection.csv + PREFIX_kernel_trace.csv).  Writes a
JSON of {kernel: {launches, avg_ms, counters...}} and prints a table.  A kernel that is launched with
several grid sizes (the forward product serves three layers of different shapes) gets one more row per
grid size, keyed "kernel [grid=N]" (N = work-items), beside the row that averages all its launches.
With steps=N (the number of bench steps the profiled command ran, warm-up included), a kernel that is
launched m > 1 times per step -- the forward product: layer 1, mlp1's output layer, mlp2's layer 1, all
on the same grid -- also gets one row per position in the step, keyed "kernel [launch i/m]".
Derived columns:
  hbm_read_bytes   = FETCH_SIZE (KiB) * 1024 * 2   (gfx950 tallies 128-B read requests at 64 B for
                                                    16-B-per-lane reads: doubled, as measured on this chip)
  hbm_write_bytes  = WRITE_SIZE (KiB) * 1024
  hbm_read_bytes_by_request_size = 32 RDREQ_32B + 64 RDREQ_64B + 128 RDREQ_128B (when that pass is given:
                                   the fabric read requests by size, a cross-check of the doubling)
  mfma_util        = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs * GRBM_GUI_ACTIVE / 8 XCDs)
  clock_ghz        = (GRBM_GUI_ACTIVE / 8) / kernel duration
"""
import collections
import csv
import json
import re
import sys


def short(name):
    name = re.sub(r"nerf::\(anonymous namespace\)::", "", name).replace("void ", "")
    return re.sub(r"\((?:[^()]|\([^()]*\))*\)$", "", name)


def read_pass(prefix, steps=0):
    per = collections.defaultdict(lambda: collections.defaultdict(list))
    seen = {}
    grids = collections.defaultdict(set)
    rows = list(csv.DictReader(open(prefix + "_counter_collection.csv")))
    dispatches = collections.defaultdict(list)   # kernel -> dispatch ids in launch order
    for r in rows:
        k = short(r["Kernel_Name"])
        grids[k].add(r.get("Grid_Size", ""))
        if int(r["Dispatch_Id"]) not in dispatches[k]:
            dispatches[k].append(int(r["Dispatch_Id"]))
    position = {}
    for k, ids in dispatches.items():
        ids.sort()
        m = len(ids) // steps if steps and len(ids) % steps == 0 else 0
        if m > 1:
            for n, d in enumerate(ids):
                position[(k, d)] = f"{k} [launch {n % m + 1}/{m}]"
    for r in rows:
        k = short(r["Kernel_Name"])
        keys = [k] + ([f"{k} [grid={r['Grid_Size']}]"] if len(grids[k]) > 1 else [])
        if (k, int(r["Dispatch_Id"])) in position:
            keys.append(position[(k, int(r["Dispatch_Id"]))])
        for key in keys:
            per[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
            seen[(key, r["Dispatch_Id"])] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-6
    for (k, _), ms in seen.items():
        per[k]["_ms"].append(ms)
    return per


def main():
    out_path, passes = sys.argv[1], sys.argv[2:]
    steps = 0
    if passes and passes[0].startswith("steps="):
        steps = int(passes.pop(0).split("=", 1)[1])
    summary = collections.defaultdict(dict)
    for spec in passes:
        name, prefix = spec.split("=", 1)
        for k, counters in read_pass(prefix, steps).items():
            s = summary[k]
            ms = counters.pop("_ms")
            s.setdefault("launches", len(ms))
            s[f"avg_ms_{name}"] = sum(ms) / len(ms)
            for c, v in counters.items():
                s[c] = sum(v) / len(v)
    for k, s in summary.items():
        if "FETCH_SIZE" in s:
            s["hbm_read_bytes"] = s["FETCH_SIZE"] * 1024 * 2
        if "WRITE_SIZE" in s:
            s["hbm_write_bytes"] = s["WRITE_SIZE"] * 1024
        if "TCC_EA0_RDREQ_128B_sum" in s:
            s["hbm_read_bytes_by_request_size"] = (32 * s.get("TCC_EA0_RDREQ_32B_sum", 0) + 64 * s.get("TCC_EA0_RDREQ_64B_sum", 0)
                                                   + 128 * s["TCC_EA0_RDREQ_128B_sum"])
        if "GRBM_GUI_ACTIVE" in s and s["GRBM_GUI_ACTIVE"] > 0:
            cyc = s["GRBM_GUI_ACTIVE"] / 8
            ms = next(v for n, v in s.items() if n.startswith("avg_ms_") and "mfma" in n) if any(
                n.startswith("avg_ms_") and "mfma" in n for n in s) else None
            if "SQ_VALU_MFMA_BUSY_CYCLES" in s:
                s["mfma_util"] = s["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * cyc)
            if ms:
                s["clock_ghz"] = cyc / (ms * 1e-3) / 1e9
    json.dump(summary, open(out_path, "w"), indent=1, sort_keys=True)
    for k, s in sorted(summary.items(), key=lambda kv: -max(v for n, v in kv[1].items() if n.startswith("avg_ms"))):
        cols = " ".join(f"{n}={v:.4g}" for n, v in sorted(s.items()))
        print(f"{k[:60]:60s} {cols}")


if __name__ == "__main__":
    main()
