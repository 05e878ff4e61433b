"""Condenses the rocprofv3 outputs of tools/profile_round.sh into the small files kept under profiles/.

  gpurun_out/<tag>_kernel_stats.csv      per-kernel calls / total / average / percentage (the --stats summary)
  gpurun_out/<tag>_pmc_traffic.csv       per-kernel mean FETCH_SIZE / WRITE_SIZE (KB) and HBM bytes per launch
  gpurun_out/klt_traffic.json            the figure bench.py reports as roofline.traffic

Corrections follow /opt/skills/guides/MI355X_MICROARCH.md: FETCH_SIZE / WRITE_SIZE are in KB, and gfx950 reports
half of the fetched bytes (so fetch is doubled); the two counters come from separate passes.
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

import hashlib

tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
out = "gpurun_out"
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import source_digest  # noqa: E402  (comments and white space do not count)

KLT_DIGEST = source_digest.digest(["vo_klt.hip"])                       # bench.py refuses these constants for any other kernel source
KLT_FRAME_DIGEST = source_digest.digest(["vo_klt.hip", "vo_frame.hip"])  # ... and for another frame store (the 4x derivative format the tracker consumes)


def find(pattern):
    hits = sorted(glob.glob(os.path.join(out, pattern), recursive=True))
    return hits[-1] if hits else None


def short(name):
    name = name.split("(")[0]
    for pre in ("void ",):
        if name.startswith(pre):
            name = name[len(pre):]
    return name


stats = find(f"{tag}_stats/**/*kernel_stats.csv")
if stats:
    with open(stats) as f, open(os.path.join(out, f"{tag}_kernel_stats.csv"), "w") as g:
        g.write(f.read())

means = {}
for ctr, sub in (("FETCH_SIZE", "fetch"), ("WRITE_SIZE", "write")):
    path = find(f"{tag}_pmc_{sub}/**/*counter_collection.csv")
    if not path:
        continue
    acc = defaultdict(lambda: [0.0, 0])
    with open(path) as f:
        for row in csv.DictReader(f):
            if row["Counter_Name"] != ctr:
                continue
            a = acc[short(row["Kernel_Name"])]
            a[0] += float(row["Counter_Value"])
            a[1] += 1
    means[ctr] = {k: (v[0] / v[1], v[1]) for k, v in acc.items()}

if means:
    kernels = sorted(set().union(*[set(m) for m in means.values()]))
    with open(os.path.join(out, f"{tag}_pmc_traffic.csv"), "w", newline="") as g:
        wcsv = csv.writer(g)                 # (kernel names carry template commas: quoted)
        wcsv.writerow(["kernel", "launches", "fetch_size_kb_raw", "write_size_kb", "hbm_bytes_per_launch"])
        for k in kernels:
            fe, n = means.get("FETCH_SIZE", {}).get(k, (0.0, 0))
            wr, n2 = means.get("WRITE_SIZE", {}).get(k, (0.0, 0))
            wcsv.writerow([k, max(n, n2), "%.1f" % fe, "%.1f" % wr, int((2 * fe + wr) * 1024)])
    klt = [k for k in kernels if "k_klt_track" in k]
    if klt:
        k = klt[0]
        fe = means["FETCH_SIZE"][k][0] if "FETCH_SIZE" in means else 0.0
        wr = means["WRITE_SIZE"][k][0] if "WRITE_SIZE" in means else 0.0
        json.dump({
            "kernel": k, "klt_source_sha256_16": KLT_DIGEST, "klt_frame_source_sha256_16": KLT_FRAME_DIGEST, "measured": tag,
            "config": "one k_klt_track launch of the profiled command (bench.py default: ONE batched context of 256 sequences x 2000 points; see `klt_valu.json` sq_waves_per_launch = sequences x points of the launch the counters were averaged over)",
            "fetch_size_kb_per_launch": 2 * fe,
            "write_size_kb_per_launch": wr,
            "hbm_bytes_per_launch": int((2 * fe + wr) * 1024),
            "note": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over `bench.py --steps 4 "
                    "--warmup 2 --no-cpu-baseline` (tools/profile_round.sh), mean over the k_klt_track launches; "
                    "FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 reports half of the fetched bytes); counts "
                    "L2 fills incl. Infinity-Cache hits (each XCD L2 pulls its own copy of the pyramids it touches)",
        }, open(os.path.join(out, "klt_traffic.json"), "w"), indent=1)
path = find(f"{tag}_pmc_mfma/**/*counter_collection.csv")
if path:
    acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0.0, 0]))
    with open(path) as f:
        for row in csv.DictReader(f):
            a = acc[short(row["Kernel_Name"])][row["Counter_Name"]]
            v = float(row["Counter_Value"])
            a[0] += v; a[1] = max(a[1], v); a[2] += 1
    with open(os.path.join(out, f"{tag}_pmc_mfma.csv"), "w", newline="") as g:
        wcsv = csv.writer(g)
        wcsv.writerow(["kernel", "launches", "mfma_util_pct_mean", "mfma_util_pct_max", "mfma_flops_f64_mean", "mfma_flops_f64_max"])
        for k in sorted(acc):
            u, fl = acc[k].get("MfmaUtil", [0, 0, 0]), acc[k].get("MfmaFlopsF64", [0, 0, 0])
            n = max(u[2], fl[2], 1)
            wcsv.writerow([k, n, "%.3f" % (u[0] / n), "%.3f" % u[1], "%.4g" % (fl[0] / n), "%.4g" % fl[1]])
path = find(f"{tag}_pmc_valu/**/*counter_collection.csv")
if path:
    acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
    with open(path) as f:
        for row in csv.DictReader(f):
            a = acc[short(row["Kernel_Name"])][row["Counter_Name"]]
            a[0] += float(row["Counter_Value"]); a[1] += 1
    names = ["SQ_INSTS_VALU", "SQ_WAVES", "SQ_ACTIVE_INST_VALU", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "GRBM_GUI_ACTIVE"]
    with open(os.path.join(out, f"{tag}_pmc_valu.csv"), "w", newline="") as g:
        wcsv = csv.writer(g)
        wcsv.writerow(["kernel", "launches"] + [n.lower() + "_mean" for n in names] + ["valu_insts_per_wave"])
        for k in sorted(acc):
            n = max(v[1] for v in acc[k].values())
            m = {c: (acc[k][c][0] / acc[k][c][1] if acc[k][c][1] else 0.0) for c in names}
            wcsv.writerow([k, n] + ["%.6g" % m[c] for c in names] + ["%.1f" % (m["SQ_INSTS_VALU"] / max(m["SQ_WAVES"], 1))])
    klt = [k for k in acc if "k_klt_track" in k]
    if klt:
        k = klt[0]
        m = {c: (acc[k][c][0] / acc[k][c][1] if acc[k][c][1] else 0.0) for c in names}
        json.dump({"kernel": k, "klt_source_sha256_16": KLT_DIGEST, "klt_frame_source_sha256_16": KLT_FRAME_DIGEST, "measured": tag,
                   "sq_insts_valu_per_launch": m["SQ_INSTS_VALU"], "sq_waves_per_launch": m["SQ_WAVES"],
                   "sq_active_inst_valu": m["SQ_ACTIVE_INST_VALU"], "sq_wave_cycles": m["SQ_WAVE_CYCLES"],
                   "sq_busy_cycles": m["SQ_BUSY_CYCLES"], "grbm_gui_active": m["GRBM_GUI_ACTIVE"],
                   "note": "rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE over "
                           "`bench.py --steps 4 --warmup 2` (tools/profile_round.sh), mean over the k_klt_track launches (sq_waves_per_launch = "
                           "sequences x points of one launch); SQ_INSTS_VALU counts wave-instructions"},
                  open(os.path.join(out, "klt_valu.json"), "w"), indent=1)

# ---- one file with everything bench.py's roofline.kernels reports, per kernel of the default command: time share and average launch
# from the --stats summary, HBM bytes / vector instructions / MFMA figures per launch from the --pmc passes.  Every number in it can be
# recomputed from <tag>_kernel_stats.csv and <tag>_pmc_{traffic,valu,mfma}.csv, which are committed beside it.
def _rows(name):
    path = os.path.join(out, name)
    return list(csv.DictReader(open(path))) if os.path.exists(path) else []


kc = {"measured": tag, "csrc_sha256_16": source_digest.digest(),
      "command": "python3 bench.py --steps 20 --warmup 5 --regions 1 --no-extras --no-cpu-baseline (kernel stats); --steps 4 --warmup 2 (each --pmc pass)",
      "files": [f"profiles/{tag}_kernel_stats_default.csv", f"profiles/{tag}_pmc_traffic_default.csv", f"profiles/{tag}_pmc_valu_default.csv",
                f"profiles/{tag}_pmc_mfma_default.csv"],
      "kernels": []}
st = {short(r["Name"]): r for r in _rows(f"{tag}_kernel_stats.csv")}
# the same command on ONE stream (tools/kstats256.sh <tag>): a launch's duration without the other streams' kernels on the chip.  In the
# default three-stream layout a narrow kernel (k_ba_solve, k_ba_update_w) is "running" from the moment it is dispatched behind a
# 3 ms tracker launch on another stream, so its --stats duration there is mostly waiting for compute units
alone_path = find(f"{tag}_ks256/**/*kernel_stats.csv")
alone = {short(r["Name"]): r for r in csv.DictReader(open(alone_path))} if alone_path else {}
if alone_path:
    with open(alone_path) as f, open(os.path.join(out, f"{tag}_kernel_stats_one_stream.csv"), "w") as g:
        g.write(f.read())
    kc["files"].append(f"profiles/{tag}_kernel_stats_one_stream.csv")
tr = {r["kernel"]: r for r in _rows(f"{tag}_pmc_traffic.csv")}
va = {r["kernel"]: r for r in _rows(f"{tag}_pmc_valu.csv")}
mf = {r["kernel"]: r for r in _rows(f"{tag}_pmc_mfma.csv")}
for k, r in sorted(st.items(), key=lambda kv: -float(kv[1]["Percentage"])):
    if float(r["Percentage"]) < 0.5:
        continue
    e = {"kernel": k, "pct_of_kernel_time": float(r["Percentage"]), "calls": int(r["Calls"]), "avg_launch_us": float(r["AverageNs"]) / 1e3}
    if k in alone:
        e["one_stream_avg_launch_us"] = float(alone[k]["AverageNs"]) / 1e3; e["one_stream_pct_of_kernel_time"] = float(alone[k]["Percentage"])
    if k in tr:
        e["hbm_bytes_per_launch"] = int(tr[k]["hbm_bytes_per_launch"])
    if k in va:
        e["valu_insts_per_launch"] = float(va[k]["sq_insts_valu_mean"]); e["waves_per_launch"] = float(va[k]["sq_waves_mean"])
    if k in mf:
        e["mfma_util_pct_mean"] = float(mf[k]["mfma_util_pct_mean"]); e["mfma_util_pct_max"] = float(mf[k]["mfma_util_pct_max"])
        e["mfma_flops_f64_per_launch"] = float(mf[k]["mfma_flops_f64_mean"])
    kc["kernels"].append(e)
json.dump(kc, open(os.path.join(out, "kernel_counters.json"), "w"), indent=1)
print("summaries written for", tag)
