"""Summarise rocprofv3 outputs of bench.py into profiles/ (markdown + json).

    python tools/profile_summary.py <tag> <dir with runs>

<dir> holds six rocprofv3 output directories, three per iteration engine:
    res_trace res_fetch res_write   (default engine at n=4096: resident, one k_resident launch per call)
    str_trace str_fetch str_write   (GRAPHLEARN_AMD_ENGINE=stream GRAPHLEARN_AMD_NO_GRAPH=1: k_edge + k_node per iteration)
*_trace: --kernel-trace --stats;  *_fetch: --pmc FETCH_SIZE;  *_write: --pmc WRITE_SIZE (separate passes).

PMC handling follows MI355X_MICROARCH.md "HBM": FETCH_SIZE / WRITE_SIZE are in KiB-units (x 1024 B);
on gfx950 FETCH_SIZE reports exactly 1/2 of the bytes of a wide coalesced (16 B/lane) streaming read,
so it is doubled; WRITE_SIZE is exact for 16 B/lane streaming stores.
"""
import csv, glob, json, os, statistics as st, sys

tag, base = sys.argv[1:3]
STEPS = int(sys.argv[3]) if len(sys.argv) > 3 else 2000  # iterations of the timed k_resident launch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out_dir = os.path.join(ROOT, "profiles")
os.makedirs(out_dir, exist_ok=True)
n = 4096
algo = 12 * n * (n - 1)


def one(pattern):
    files = sorted(glob.glob(pattern, recursive=True), key=lambda f: int(os.path.basename(f).split("_")[0]))
    return files[-1] if files else None  # the latest run id when gpurun_out/ still holds older passes


def short(name):
    return name.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]


def trace(d):
    rows = list(csv.DictReader(open(one(os.path.join(base, d, "**", "*kernel_trace.csv")))))
    by = {}
    for r in rows:
        by.setdefault(short(r["Kernel_Name"]), []).append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    return rows, by


def pmc(d, counter):
    f = one(os.path.join(base, d, "**", "*counter_collection.csv"))
    vals = {}
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == counter:
            vals.setdefault(short(r["Kernel_Name"]), []).append(float(r["Counter_Value"]))
    return vals


def table(by, lines):
    total = sum(sum(v) for v in by.values())
    lines += ["| kernel | calls | total ms | avg us | min us | max us | % |", "|---|---:|---:|---:|---:|---:|---:|"]
    summary = {}
    for k, v in sorted(by.items(), key=lambda kv: -sum(kv[1])):
        lines.append(f"| {k} | {len(v)} | {sum(v)/1e6:.3f} | {st.mean(v)/1e3:.2f} | {min(v)/1e3:.2f} | {max(v)/1e3:.2f} | {100*sum(v)/total:.1f} |")
        summary[k] = {"calls": len(v), "avg_us": st.mean(v) / 1e3, "total_ms": sum(v) / 1e6}
    return summary


lines = [f"# rocprofv3 summary ({tag}) -- LogModel n=4096 d=1024, bench.py --no-cpu-baseline --no-converge", ""]
out = {"tag": tag, "algorithmic_bytes_per_iteration": algo, "traffic": {}}

# ---------------------------------------------------------------- resident engine (default at this size)
rows, by = trace("res_trace")
lines += ["## resident engine (default): `rocprofv3 --kernel-trace --stats -- python3 bench.py --no-cpu-baseline --no-converge`", ""]
out["kernels_resident"] = table(by, lines)
res = sorted((r for r in rows if "k_resident" in r["Kernel_Name"]), key=lambda r: int(r["Start_Timestamp"]))
durs = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in res]
timed = max(durs)
r0 = res[durs.index(timed)]
lines += ["", f"k_resident, the timed launch ({STEPS} iterations in ONE launch): {timed/1e3:.1f} us = {timed/1e3/STEPS:.2f} us/iteration "
          f"-> {algo/(timed/STEPS):.0f} GB/s of algorithmic bytes (24 E = {algo/1e6:.1f} MB per iteration) = {algo/(timed/STEPS)/8000:.3f} of 8 TB/s",
          f"k_resident: VGPR {r0['VGPR_Count']}, SGPR {r0['SGPR_Count']}, grid {r0['Grid_Size_X']} x wg {r0['Workgroup_Size_X']}, LDS {r0['LDS_Block_Size']}, scratch {r0['Scratch_Size']}"]
out["k_resident_timed_launch_us"] = timed / 1e3
out["k_resident_us_per_iteration"] = timed / 1e3 / STEPS
fetch, write = pmc("res_fetch", "FETCH_SIZE"), pmc("res_write", "WRITE_SIZE")
f = max(v for k, vs in fetch.items() if k.startswith("k_resident") for v in vs)
w = max(v for k, vs in write.items() if k.startswith("k_resident") for v in vs)
rb, wb = f * 2 * 1024, w * 1024
out["traffic"]["k_resident"] = {"fetch_kib_raw": f, "write_kib": w, "read_bytes": rb, "write_bytes": wb,
                                "total_bytes": rb + wb, "iterations": STEPS, "total_bytes_per_iteration": (rb + wb) / STEPS}
lines += ["", "### fabric traffic of the timed k_resident launch (PMC, separate passes)", "",
          f"FETCH_SIZE {f:.0f} KiB raw (x2 x1024 = {rb/1e9:.2f} GB), WRITE_SIZE {w:.0f} KiB ({wb/1e9:.3f} GB): "
          f"{(rb+wb)/STEPS/1e6:.1f} MB per iteration against {algo/1e6:.1f} MB algorithmic -- the iterate never leaves the "
          f"register files / LDS; what is left is the stream of z (8 E = {algo/3/1e6:.1f} MB) plus the small slab exchange."]

# ---------------------------------------------------------------- streaming engine
rows, by = trace("str_trace")
lines += ["", "## streaming engine: `GRAPHLEARN_AMD_ENGINE=stream GRAPHLEARN_AMD_NO_GRAPH=1 rocprofv3 --kernel-trace --stats -- python3 bench.py ...` "
          "(eager launches: the profiler serialises graph replays)", ""]
out["kernels_streaming"] = table(by, lines)
edge = [r for r in rows if "k_edge<false" in r["Kernel_Name"]]
node = [r for r in rows if "k_node<false" in r["Kernel_Name"]]
seq = sorted(edge + node, key=lambda r: int(r["Start_Timestamp"]))
gaps = [int(seq[i + 1]["Start_Timestamp"]) - int(seq[i]["End_Timestamp"]) for i in range(len(seq) - 1)]
e_avg = st.mean(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in edge)
lines += ["", f"k_edge<false>: avg {e_avg/1e3:.2f} us -> {algo/e_avg:.0f} GB/s of algorithmic bytes = {algo/e_avg/8000:.3f} of 8 TB/s; "
          f"median gap between kernels under the profiler {st.median(gaps)/1e3:.2f} us",
          f"k_edge<false>: VGPR {edge[0]['VGPR_Count']}, SGPR {edge[0]['SGPR_Count']}, grid {edge[0]['Grid_Size_X']} x wg {edge[0]['Workgroup_Size_X']}, LDS {edge[0]['LDS_Block_Size']}, scratch {edge[0]['Scratch_Size']}"]
fetch, write = pmc("str_fetch", "FETCH_SIZE"), pmc("str_write", "WRITE_SIZE")


def pick(d, prefix):
    for name, v in d.items():
        if name.startswith(prefix):
            return v[5:] or v
    return [0.0]


lines += ["", "### fabric traffic per launch (PMC, separate passes)", "",
          "| kernel | FETCH_SIZE (KiB, raw) | WRITE_SIZE (KiB) | read bytes (FETCH x 2 x 1024) | write bytes | total MB |", "|---|---:|---:|---:|---:|---:|"]
for k in ("k_edge<false", "k_node<false"):
    f = st.mean(pick(fetch, k))
    w = st.mean(pick(write, k))
    rb, wb = f * 2 * 1024, w * 1024
    out["traffic"][k] = {"fetch_kib_raw": f, "write_kib": w, "read_bytes": rb, "write_bytes": wb, "total_bytes": rb + wb,
                         "iterations": 1, "total_bytes_per_iteration": rb + wb}
    lines.append(f"| {k} | {f:.1f} | {w:.1f} | {rb/1e6:.1f} MB | {wb/1e6:.1f} MB | {(rb+wb)/1e6:.1f} |")
lines += ["", f"algorithmic bytes per iteration (SURVEY 8d): 24 E = {algo/1e6:.1f} MB; the (w0, w1, z) working set is "
          f"3 x 68 MB = 204 MB < 256 MiB Infinity Cache, and rocprofv3's FETCH/WRITE_SIZE count the L2's fabric-side "
          f"requests (Infinity-Cache hits included), so they measure L2<->fabric traffic, not DRAM-only traffic."]
open(os.path.join(out_dir, f"{tag}_rocprof_summary.md"), "w").write("\n".join(lines) + "\n")
json.dump(out, open(os.path.join(out_dir, f"{tag}_rocprof_summary.json"), "w"), indent=1)
print("\n".join(lines))
