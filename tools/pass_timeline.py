#!/usr/bin/env python3
"""Timeline of ONE pass of bench.py out of a rocprofv3 --kernel-trace CSV: every dispatch of the last pass with its start offset,
duration and the idle gap in front of it (what the fixed tail of a pass is made of).

    rocprofv3 --kernel-trace --output-format csv -d DIR -- python3 bench.py --config c2 --steps 5 --warmup 1 --no-cpu-baseline
    python3 tools/pass_timeline.py DIR [--first KERNEL_SUBSTRING] [--skip N]

A pass starts at the dispatch whose name contains --first (default: init_red_kernel, the first launch of ftkx_slices_prepare /
series_begin_kernel of the series pass)."""
import csv
import glob
import json
import os
import sys


def short(name):
    name = name.replace("ftkx::", "").replace("(anonymous namespace)::", "")
    if "rocprim" in name:
        for key in ("block_sort", "block_merge", "radix_sort_onesweep", "histogram", "scan", "merge"):
            if key in name:
                return "rocprim:" + key
        return "rocprim"
    if name.startswith("void "):
        name = name[5:]
    return name.split("(")[0][:60]


def main():
    d = sys.argv[1]
    first = "init_red_kernel"
    if "--first" in sys.argv:
        first = sys.argv[sys.argv.index("--first") + 1]
    files = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
    if not files:
        sys.exit("no kernel_trace.csv under " + d)
    rows = []
    for f in files:
        for r in csv.DictReader(open(f)):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", ""), r.get("Stream_Id", "")))
    rows.sort()
    firsts = [s.strip() for s in first.split(",")]
    starts = [i for i, r in enumerate(rows) if any(s in r[2] for s in firsts)]
    if len(starts) < 2:
        sys.exit("fewer than two passes found")
    skip = int(sys.argv[sys.argv.index("--skip") + 1]) if "--skip" in sys.argv else 0   # passes to leave out at the end (bench.py's latency / per-kernel passes)
    if len(starts) < 2 + skip:
        sys.exit("fewer passes than --skip leaves")
    a, b = starts[-2 - skip], starts[-1 - skip]          # the last COMPLETE pass (before the skipped ones)
    t0 = rows[a][0]
    prev_end = t0
    out = []
    busy = 0
    for s, e, n, q, st in rows[a:b]:
        out.append({"kernel": short(n), "start_us": (s - t0) / 1e3, "dur_us": (e - s) / 1e3, "gap_before_us": (s - prev_end) / 1e3, "queue": q, "stream": st})
        busy += e - s
        prev_end = max(prev_end, e)
    span = (prev_end - t0) / 1e3
    print(f"# pass of {len(out)} dispatches: first start -> last end {span:.1f} us, sum of kernel durations {busy / 1e3:.1f} us, "
          f"pass period (start to next pass's start) {(rows[b][0] - t0) / 1e3:.1f} us")
    for o in out:
        print(f"{o['start_us']:9.1f} us  +{o['dur_us']:8.1f} us  (gap {o['gap_before_us']:7.1f})  q{o['queue']} s{o['stream']}  {o['kernel']}")
    if "--json" in sys.argv:
        json.dump({"span_us": span, "kernels_us": busy / 1e3, "period_us": (rows[b][0] - t0) / 1e3, "dispatches": out}, open(sys.argv[sys.argv.index("--json") + 1], "w"), indent=1)


if __name__ == "__main__":
    main()
