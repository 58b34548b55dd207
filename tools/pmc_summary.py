"""Fold the rocprofv3 outputs of tools/profile_round.sh into the committed summaries under profiles/.

  python tools/pmc_summary.py <tag> [steps the traced command ran: warm-up + timed]

reads  gpurun_out/<tag>_stats/**/*kernel_stats.csv and gpurun_out/<tag>_pmc_*/**/*counter_collection.csv
writes profiles/<tag>_kernel_stats.csv (samble kernels + the largest others, verbatim rows),
       profiles/<tag>_pmc.json (per kernel, per launch: counters, fabric traffic, MFMA-pipe busy),
       profiles/<tag>_bench.json (the bench line).
Traffic follows MI355X_MICROARCH.md: FETCH_SIZE / WRITE_SIZE are KB; FETCH_SIZE is doubled on gfx950.
"""
import csv
import glob
import json
import os
import shutil
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLOCK_GHZ_NOMINAL = 2.4
N_SIMD = 1024


def short(name):
    n = name.split("(")[0]
    return n[5:] if n.startswith("void ") else n


def main(tag, steps=0):
    gout = os.path.join(ROOT, "gpurun_out")
    prof = os.path.join(ROOT, "profiles")
    os.makedirs(prof, exist_ok=True)
    stats = glob.glob(os.path.join(gout, f"{tag}_stats", "**", "*kernel_stats.csv"), recursive=True)
    avg_ns, calls = {}, {}
    if stats:
        shutil.copy(stats[0], os.path.join(prof, f"{tag}_kernel_stats.csv"))
        for row in csv.DictReader(open(stats[0])):
            avg_ns[short(row["Name"])] = float(row["AverageNs"])
            calls[short(row["Name"])] = int(row["Calls"])
    # launches per step: most kernels run once per step, so the most frequent call count among the library's kernels is
    # the number of steps the traced command ran (warm-up + timed)
    from collections import Counter
    counts = Counter(c for k, c in calls.items() if k.startswith("samble::"))
    steps_traced = counts.most_common(1)[0][0] if counts else 0
    if steps:   # given on the command line (the block steps launch most of their kernels twice or more)
        steps_traced = steps
    per = defaultdict(lambda: defaultdict(list))
    for f in glob.glob(os.path.join(gout, f"{tag}_pmc_*", "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            per[short(row["Kernel_Name"])][row["Counter_Name"]].append(float(row["Counter_Value"]))
    kernels = {}
    for k, ctr in per.items():
        if not k.startswith("samble::"):
            continue
        e = {f"{c}_per_launch": round(sum(v) / len(v), 1) for c, v in ctr.items()}
        if "FETCH_SIZE" in ctr and "WRITE_SIZE" in ctr:
            f_kb = sum(ctr["FETCH_SIZE"]) / len(ctr["FETCH_SIZE"])
            w_kb = sum(ctr["WRITE_SIZE"]) / len(ctr["WRITE_SIZE"])
            e["traffic_bytes_per_launch"] = int((2 * f_kb + w_kb) * 1024)
        if k in calls and steps_traced:
            e["launches_per_step"] = round(calls[k] / steps_traced, 2)
        if k in avg_ns:
            e["avg_us_kernel_trace"] = round(avg_ns[k] / 1e3, 2)
            if "SQ_VALU_MFMA_BUSY_CYCLES" in ctr and "SQ_BUSY_CYCLES" in ctr:
                mf = sum(ctr["SQ_VALU_MFMA_BUSY_CYCLES"]) / len(ctr["SQ_VALU_MFMA_BUSY_CYCLES"])
                # SQ_BUSY_CYCLES is summed over the 32 SEs x ... ; report MFMA busy against wall time x SIMDs
                # at the nominal clock (a lower bound on the pipe-busy fraction when the clock sags).
                e["mfma_busy_frac_at_2.4GHz"] = round(mf / (avg_ns[k] * CLOCK_GHZ_NOMINAL * N_SIMD), 4)
        kernels[k] = e
    note = ("rocprofv3 --pmc passes (separate runs for SQ_*, FETCH_SIZE, WRITE_SIZE, LDS) of `python3 bench.py --steps 3 "
            "--warmup 1 --no-cpu-baseline --no-breakdown` on MI355X; avg_us from the --kernel-trace --stats run of "
            "`bench.py --steps 20 --warmup 5`. traffic = (2*FETCH_SIZE + WRITE_SIZE)*1024: KB units, FETCH_SIZE doubled "
            "per MI355X_MICROARCH.md (gfx950 counts wide coalesced reads at half size); fabric-side requests, "
            "Infinity-Cache hits included.")
    step_traffic = int(sum(e.get("traffic_bytes_per_launch", 0) * e.get("launches_per_step", 1) for e in kernels.values()))
    json.dump({"note": note, "step_traffic_bytes": step_traffic, "kernels": kernels},
              open(os.path.join(prof, f"{tag}_pmc.json"), "w"), indent=1)
    print(f"fabric traffic per step (sum over launches): {step_traffic / 1e9:.3f} GB")
    b = os.path.join(gout, f"{tag}_bench.json")
    if os.path.exists(b):
        lines = [l for l in open(b).read().splitlines() if l.startswith("{")]
        if lines:
            open(os.path.join(prof, f"{tag}_bench.json"), "w").write(lines[-1] + "\n")
    for k, e in sorted(kernels.items(), key=lambda kv: -kv[1].get("avg_us_kernel_trace", 0)):
        print(f"{k:45s} {e.get('avg_us_kernel_trace', 0):9.1f} us  mfma {e.get('mfma_busy_frac_at_2.4GHz', 0):.3f}  "
              f"traffic {e.get('traffic_bytes_per_launch', 0) / 1e6:9.1f} MB")


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 0)
