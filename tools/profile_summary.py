#!/usr/bin/env python3
"""
Condenses gpurun_out/prof/* (tools/profile_round.sh) into the files kept under profiles/:
  profiles/<tag>_kernel_stats.csv      rocprofv3 --kernel-trace --stats summary of the bench command
  profiles/<tag>_pmc_per_kernel.csv    per-kernel averages of every collected counter
  profiles/pmc_traffic.json            HBM bytes per launch per bench kernel name (read by bench.py for roofline.traffic)
HBM bytes = 2 * FETCH_SIZE * 1024 + WRITE_SIZE * 1024: on gfx950 FETCH_SIZE reports half the bytes of a wide coalesced
read (MI355X_MICROARCH.md, HBM section); WRITE_SIZE is exact for 16-byte stores and float atomics.
"""
import collections
import csv
import glob
import json
import os
import re
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gpurun_out", "prof")
tag = sys.argv[1] if len(sys.argv) > 1 else "r03"


def short(name):
    m = re.search(r"(rowtile_wgrad_kernel<[^>]*>|rowtile_kernel<[^>]*>|weight_grad_kernel<[^>]*>|pack_weights_kernel<[^>]*>|prune_to_csr_kernel)", name)
    return (m.group(1) if m else name[:50]).replace("unsigned short", "bf16")


def bench_names(rows):
    """Map dispatches to bench.py kernel names by template arguments: rowtile_kernel<CT, IT, OT, BWD, VEC, NTW, KSMAX, DZIN, NWV>."""
    seen = collections.Counter()
    out = []
    for r in rows:
        n = short(r["Kernel_Name"])
        k = None
        if n.startswith("rowtile_wgrad"):
            k = "bwd_data0+wgrad1"                             # layer 0's backward-data launch carries layer 1's weight gradient
        elif n.startswith("rowtile"):
            args = [a.strip() for a in n[n.index("<") + 1:n.rindex(">")].split(",")]
            bwd, dzin = args[3] == "true", len(args) > 7 and args[7] == "true"
            if not bwd:
                k = "fwd%d" % (seen["fwd"] % 2)
                seen["fwd"] += 1
            else:
                k = "bwd_data0" if dzin else "bwd_data1"          # the top layer derives dZ itself, the layer below receives it
        elif n.startswith("weight_grad"):
            k = "bwd_weight0"                                  # the last launch of the sweep: the bottom layer's weight gradient
        elif n.startswith("pack"):
            k = "pack"
        elif n.startswith("prune"):
            k = "prune"
        out.append(k)
    return out


def per_kernel(run):
    files = glob.glob(os.path.join(SRC, run, "*", "*counter_collection.csv"))
    if not files:
        return {}
    rows = list(csv.DictReader(open(max(files, key=os.path.getmtime))))       # gpurun_out/ keeps earlier runs too
    by_disp = collections.OrderedDict()
    for r in rows:
        by_disp.setdefault(r["Dispatch_Id"], []).append(r)
    firsts = [v[0] for v in by_disp.values()]
    names = bench_names(firsts)
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for (d, rs), k in zip(by_disp.items(), names):
        if k is None:
            continue
        for r in rs:
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: {c: sum(v) / len(v) for c, v in cs.items()} for k, cs in agg.items()}


os.makedirs(os.path.join(ROOT, "profiles"), exist_ok=True)
stats = glob.glob(os.path.join(SRC, "trace", "*", "*kernel_stats.csv"))
if stats:
    shutil.copy(max(stats, key=os.path.getmtime), os.path.join(ROOT, "profiles", "%s_kernel_stats.csv" % tag))
stats = glob.glob(os.path.join(SRC, "trace_graph", "*", "*kernel_stats.csv"))
if stats:
    shutil.copy(max(stats, key=os.path.getmtime), os.path.join(ROOT, "profiles", "%s_kernel_stats_graph.csv" % tag))
for run, name in (("trace_c5", "c5_kernel_stats"), ("trace_c5_packed", "c5_packed_kernel_stats")):
    stats = glob.glob(os.path.join(SRC, run, "*", "*kernel_stats.csv"))
    if stats:
        shutil.copy(max(stats, key=os.path.getmtime), os.path.join(ROOT, "profiles", "%s_%s.csv" % (tag, name)))
merged = collections.defaultdict(dict)
for run in ("fetch", "write", "sq", "mfma"):
    for k, cs in per_kernel(run).items():
        merged[k].update(cs)
cols = sorted({c for cs in merged.values() for c in cs})
with open(os.path.join(ROOT, "profiles", "%s_pmc_per_kernel.csv" % tag), "w") as f:
    w = csv.writer(f)
    w.writerow(["kernel"] + cols)
    for k in sorted(merged):
        w.writerow([k] + ["%.1f" % merged[k][c] if c in merged[k] else "" for c in cols])
traffic = {k: int(2 * cs.get("FETCH_SIZE", 0) * 1024 + cs.get("WRITE_SIZE", 0) * 1024)
           for k, cs in merged.items() if "FETCH_SIZE" in cs and "WRITE_SIZE" in cs}
# bench.py quotes these bytes only for the workload they were recorded on (tools/profile_round.sh profiles the default bench command)
sha = os.popen("git -C %s rev-parse --short HEAD 2>/dev/null" % ROOT).read().strip() or "?"
sys.path.insert(0, ROOT)
import bench  # noqa: E402  (source_hash: the recording is only quoted for the kernel sources it was made with)
meta = dict(batch=50, seq=100, din=360, hidden=200, prune_k=1, dtype="bf16", lengths="full",
            kernels=["pack", "fwd0", "fwd1", "bwd_data1", "bwd_data0+wgrad1", "bwd_weight0"], git_sha="%s (%s)" % (sha, tag),
            source_hash=bench.source_hash())
json.dump(dict(meta=meta, traffic=traffic), open(os.path.join(ROOT, "profiles", "pmc_traffic.json"), "w"), indent=1, sort_keys=True)
print(json.dumps(traffic))
for k in sorted(merged):
    cs = merged[k]
    if "SQ_WAVES" in cs:
        w = cs["SQ_WAVES"]
        print("%-12s waves %5d  cyc/wave %6.0f  wait %3.0f%%  active %3.0f%%  valu/wave %5.0f salu/wave %5.0f" % (
            k, w, 4 * cs["SQ_WAVE_CYCLES"] / w, 100 * cs["SQ_WAIT_ANY"] / cs["SQ_WAVE_CYCLES"],
            100 * cs["SQ_ACTIVE_INST_ANY"] / cs["SQ_WAVE_CYCLES"], cs["SQ_INSTS_VALU"] / w, cs["SQ_INSTS_SALU"] / w))
