#!/usr/bin/env python3
"""Summarise gpurun_out/prof_r04/ (profiles/collect_r04.sh) into the committed round-4 evidence:
  profiles/r04_<workload>_kernel_stats.csv   per-kernel calls / total / average duration (kernel trace)
  profiles/r04_pmc_traffic.json               per-kernel bytes per launch: FETCH_SIZE x 2 (gfx950 counts a
                                              128-B request as 64 B, MI355X_MICROARCH.md "HBM") and WRITE_SIZE;
                                              both counters are reported in KiB by rocprofv3
"""
import collections
import csv
import glob
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PROF = os.path.join(ROOT, "gpurun_out", "prof_r04")


def short(name):
    n = name.split("(")[0].replace("void ", "").replace("pop::", "")
    return n.split("<")[0] if n.startswith("k_") else n


def kernel_stats(wl, sub="_stats", tag=""):
    fs = sorted(glob.glob(os.path.join(PROF, wl + sub, "*", "*kernel_trace.csv")), key=os.path.getmtime, reverse=True)   # newest collection first
    acc = collections.OrderedDict()
    for r in csv.DictReader(open(fs[0])):
        a = acc.setdefault(r["Kernel_Name"], [0, 0, []])
        a[0] += 1
        a[1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        a[2].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    tot = sum(a[1] for a in acc.values())
    out = os.path.join(ROOT, "profiles", "r04_%s%s_kernel_stats.csv" % (wl, tag))
    with open(out, "w") as f:
        f.write("kernel,calls,total_ms,avg_us,percent,median_us\n")      # (the mean of a short kernel carries the first launches' outliers)
        for k, a in sorted(acc.items(), key=lambda kv: -kv[1][1]):
            f.write('"%s",%d,%.3f,%.2f,%.2f,%.2f\n' % (k, a[0], a[1] / 1e6, a[1] / a[0] / 1e3, 100.0 * a[1] / tot, sorted(a[2])[len(a[2]) // 2] / 1e3))
    return out


def pmc(wl, ctr):
    fs = sorted(glob.glob(os.path.join(PROF, "%s_%s" % (wl, ctr), "*", "*counter_collection.csv")), key=os.path.getmtime, reverse=True)
    acc = collections.defaultdict(lambda: [0, 0.0, 0])
    if not fs:
        return acc
    for r in csv.DictReader(open(fs[0])):
        if r["Counter_Name"] != ctr:
            continue
        a = acc[short(r["Kernel_Name"])]
        a[0] += 1
        a[1] += float(r["Counter_Value"]) * 1024.0
        a[2] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    return acc


def sq_summary():
    """per-kernel SQ counters of the tx0.1v3 run (separate --pmc passes): waves, VALU instructions and busy share, wait
    share, VGPRs / LDS / workgroup size of the dispatch -> profiles/r04_sq_summary.json"""
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    nd = collections.defaultdict(lambda: collections.defaultdict(set))
    meta = {}
    dsum = collections.defaultdict(dict)           # kernel -> dispatch id -> duration [ns] (under the counter pass)
    newest = []                                    # one file per SQ pass: the newest collection
    for d in sorted(glob.glob(os.path.join(PROF, "tx0.1v3_SQ*"))):
        fs = sorted(glob.glob(os.path.join(d, "*", "*counter_collection.csv")), key=os.path.getmtime, reverse=True)
        newest += fs[:1]
    for f in newest:
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            if not k.startswith("k_"):
                continue
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
            if r.get("Start_Timestamp") and r.get("End_Timestamp"):
                dsum[k][r["Dispatch_Id"]] = float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
            nd[k][r["Counter_Name"]].add(r["Dispatch_Id"])
            meta[k] = {"vgprs": int(r.get("VGPR_Count") or 0), "lds_bytes": int(r.get("LDS_Block_Size") or 0),
                       "workgroup": int(r.get("Workgroup_Size") or 0), "scratch_bytes": int(r.get("Scratch_Size") or 0)}
    out = {}
    durations = {k: sum(v.values()) / len(v) for k, v in dsum.items() if v}
    for k, cs in acc.items():
        per = {c: v / max(len(nd[k][c]), 1) for c, v in cs.items()}
        e = dict(meta[k])
        e["per_dispatch"] = {c: round(v) for c, v in sorted(per.items())}
        wc = per.get("SQ_WAVE_CYCLES")
        if wc:
            # Units (MI355X_MICROARCH.md, PMC table): SQ_WAVE_CYCLES, SQ_WAIT_* and SQ_ACTIVE_INST_* all count QUAD-cycles of a wave, and
            # WAIT_ANY (parked: s_waitcnt / barrier) + WAIT_INST_ANY (ready but not issued: pipe busy or operand not ready) + ACTIVE_INST_*
            # ~ WAVE_CYCLES.  (The r3 summary multiplied ACTIVE_INST_VALU by four and so reported the momentum kernel as 0.83 "VALU-busy";
            # in the same units it is 0.21 of a wave's cycles.)  simd_valu_utilisation: the VALU quad-cycles of all waves over the SIMD
            # quad-cycles of the launch (1024 SIMDs x its duration at 2.4 GHz) -- what "instruction-issue-bound" would have to be near 1 for.
            e["wait_share_of_wave_cycles"] = round(per.get("SQ_WAIT_ANY", 0.0) / wc, 3) if "SQ_WAIT_ANY" in per else None
            e["issue_stall_share_of_wave_cycles"] = round(per.get("SQ_WAIT_INST_ANY", 0.0) / wc, 3) if "SQ_WAIT_INST_ANY" in per else None
            e["valu_active_share_of_wave_cycles"] = round(per.get("SQ_ACTIVE_INST_VALU", 0.0) / wc, 3) if "SQ_ACTIVE_INST_VALU" in per else None
            dur = durations.get(k)
            if dur and "SQ_ACTIVE_INST_VALU" in per:
                e["avg_us_under_pmc"] = round(dur / 1e3, 2)
                e["simd_valu_utilisation"] = round(per["SQ_ACTIVE_INST_VALU"] / (1024.0 * dur * 1e-9 * 2.4e9 / 4.0), 3)
        out[k] = e
    if out:
        p = os.path.join(ROOT, "profiles", "r04_sq_summary.json")
        with open(p, "w") as f:
            json.dump(out, f, indent=1, sort_keys=True)
        print(p)


def main():
    sq_summary()
    traffic = {}
    for wl in ("gx1v7", "tx0.1v3"):
        if not os.path.isdir(os.path.join(PROF, wl + "_stats")):
            continue
        print(kernel_stats(wl))
        if os.path.isdir(os.path.join(PROF, wl + "_deep_stats")):
            print(kernel_stats(wl, "_deep_stats", "_deep_state"))      # the deep-boundary-layer state, traced on its own
        fe, wr = pmc(wl, "FETCH_SIZE"), pmc(wl, "WRITE_SIZE")
        t = traffic.setdefault(wl, {})
        for k in fe:
            if not k.startswith("k_"):
                continue
            t[k] = {"launches_sampled": fe[k][0], "fetch_bytes": round(2.0 * fe[k][1] / fe[k][0]),
                    "write_bytes": round(wr[k][1] / wr[k][0]) if k in wr and wr[k][0] else 0,
                    "avg_us_under_pmc": round(fe[k][2] / fe[k][0] / 1e3, 2)}
    out = os.path.join(ROOT, "profiles", "r04_pmc_traffic.json")
    with open(out, "w") as f:
        json.dump(traffic, f, indent=1, sort_keys=True)
    print(out)


if __name__ == "__main__":
    main()
