#!/usr/bin/env python3
"""Turns the raw rocprofv3 output of tools/collect_profiles.sh (under gpurun_out/) into the small, committed summaries under
profiles/ and updates profiles/traffic.json (read by bench.py for roofline.traffic).

HBM bytes per launch = FETCH_SIZE * 1024 * k + WRITE_SIZE * 1024, with k measured by the calibration run in the same call
(MI355X_MICROARCH.md, HBM: on gfx950 FETCH_SIZE reports half the bytes of a wide 16-B-per-lane stream; WRITE_SIZE is exact)."""
import collections
import csv
import glob
import json
import os
import re
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
cfg, tag = sys.argv[1], sys.argv[2]
src = os.path.join(ROOT, "gpurun_out", f"profiles_{tag}_{cfg}")
dst = os.path.join(ROOT, "profiles")
os.makedirs(dst, exist_ok=True)


def one(pattern):
    # gpurun MERGES into gpurun_out/: an earlier call's files may still be there -- take the newest match
    g = sorted(glob.glob(os.path.join(src, pattern)), key=os.path.getmtime)
    assert g, pattern
    return g[-1]


shutil.copy(one("trace/*/*kernel_stats.csv"), os.path.join(dst, f"{tag}_{cfg}_kernel_stats.csv"))
if os.path.exists(os.path.join(src, "timeline.txt")):
    shutil.copy(os.path.join(src, "timeline.txt"), os.path.join(dst, f"{tag}_{cfg}_pass_timeline.txt"))
if os.path.exists(os.path.join(src, "timeline_single.txt")):
    shutil.copy(os.path.join(src, "timeline_single.txt"), os.path.join(dst, f"{tag}_{cfg}_pass_timeline_single.txt"))
if os.path.exists(os.path.join(src, "timeline_overlap.txt")):      # the passes without events between the kernels (what `sustained` and the side configurations run): the split pass's tail next to the next mask kernel
    shutil.copy(os.path.join(src, "timeline_overlap.txt"), os.path.join(dst, f"{tag}_{cfg}_pass_timeline_plain.txt"))
import subprocess
try:
    commit = subprocess.run(["git", "rev-parse", "--short", "HEAD"], cwd=ROOT, capture_output=True, text=True).stdout.strip()
    dirty = bool(subprocess.run(["git", "status", "--porcelain", "--", "ftk_amd/csrc"], cwd=ROOT, capture_output=True, text=True).stdout.strip())
except Exception:   # noqa: BLE001
    commit, dirty = "unknown", False
shutil.copy(one("bench_plain.json"), os.path.join(dst, f"{tag}_{cfg}_bench.json"))


def pmc(path, counter):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            acc[r["Kernel_Name"].split("(")[0].replace("void ", "")].append(float(r["Counter_Value"]))
    return {k: (sum(v) / len(v), len(v)) for k, v in acc.items()}


calib = pmc(one("calib/*/*counter_collection.csv"), "FETCH_SIZE")
calib_kb = [v for k, v in calib.items() if "calib_read" in k][0][0]
streamed = 8 * (1 << 30)
k = streamed / (calib_kb * 1024.0)
fetch = pmc(one("pmc_fetch/*/*counter_collection.csv"), "FETCH_SIZE")
write = pmc(one("pmc_write/*/*counter_collection.csv"), "WRITE_SIZE")
summary = {"config": cfg, "fetch_size_calibration": {"streamed_bytes": streamed, "FETCH_SIZE_KB": calib_kb, "bytes_per_reported_byte": k}, "kernels": {}}
for name in sorted(set(fetch) | set(write)):
    if "ftkx" not in name:
        continue
    f, nf = fetch.get(name, (0.0, 0)); w, nw = write.get(name, (0.0, 0))
    summary["kernels"][name] = {"FETCH_SIZE_KB_per_dispatch": f, "WRITE_SIZE_KB_per_dispatch": w, "dispatches": [nf, nw],
                                "hbm_read_bytes_per_dispatch": f * 1024 * k, "hbm_write_bytes_per_dispatch": w * 1024,
                                "hbm_bytes_per_dispatch": f * 1024 * k + w * 1024}
json.dump(summary, open(os.path.join(dst, f"{tag}_{cfg}_pmc_summary.json"), "w"), indent=1)
bench = json.loads(open(one("bench_plain.json")).read().strip().splitlines()[-1])
dom = bench["roofline"].get("kernel_family", bench["roofline"]["kernel"]).replace("ftkx::", "").split("<")[0]
# bench.py labels kernel families; the marching mask kernel's pre-pass instantiation (<.., true>) is not the timed one
key = {"mask_kernel": "mask_", "cull_kernel": "cull_", "exact_kernel": "exact_", "tile_kernel": "tile_"}.get(dom, dom)
def is_prepass(n):   # REDUCE instantiations: mask_march4_kernel<ND, true, PD, RY>
    return n.rstrip().endswith(", true>") or re.search(r"mask_march4_kernel<\d, true", n) is not None


cand = [v for n, v in summary["kernels"].items() if key in n and not is_prepass(n)]
tj = os.path.join(dst, "traffic.json")
traffic = json.load(open(tj)) if os.path.exists(tj) else {}
if cand:
    best = max(cand, key=lambda v: v["hbm_bytes_per_dispatch"])
    traffic[cfg] = {"kernel": bench["roofline"]["kernel"], "hbm_bytes_per_launch": best["hbm_bytes_per_dispatch"],
                    "source": f"profiles/{tag}_{cfg}_pmc_summary.json", "kernel_sources_at": commit + (" + uncommitted changes" if dirty else "")}
json.dump(traffic, open(tj, "w"), indent=1)
print(json.dumps(summary, indent=1))
print("bench:", bench["value"], bench["roofline"]["frac"], bench["roofline"]["kernel_ms_per_pass"])
