"""Summarise the rocprofv3 passes of tools/profile_r02.sh into profiles/r02_* (markdown + json + the raw stats CSVs).

    python tools/profile_summary_r02.py [gpurun_out/prof_r02]

PMC handling follows MI355X_MICROARCH.md "HBM": FETCH_SIZE / WRITE_SIZE are in KiB units; on gfx950 FETCH_SIZE
reports exactly 1/2 of the bytes of a wide coalesced (16 B/lane) streaming read, so it is doubled; WRITE_SIZE is
exact for 16 B/lane streaming stores.  (The records of the hand-offs are 16-byte accesses too.)
"""
import csv, glob, json, os, shutil, statistics as st, sys

base = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/prof_r02"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out_dir = os.path.join(ROOT, "profiles")
n = 4096
algo = 12 * n * (n - 1)


def one(d, suffix):
    # gpurun merges new outputs into gpurun_out/ beside older passes (run ids are pids of a fresh box): take the newest file
    f = sorted(glob.glob(os.path.join(base, d, "**", "*" + suffix), recursive=True), key=os.path.getmtime)
    return f[-1]


def short(name):
    return name.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]


def bench_line(d):
    for l in open(os.path.join(base, d + ".json")):
        if l.startswith("{"):
            return json.loads(l)
    return None


def headline(name):  # the headline's k_tile launches (not the rounds instantiation of bench.py's n = 8192 leg)
    n = short(name)
    return n.startswith("k_tile") and not n.endswith((", true>", ", 1>", ", 2>"))


def launches(d, prefix):
    rows = [r for r in csv.DictReader(open(one(d, "kernel_trace.csv"))) if headline(r["Kernel_Name"])]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    return [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows], rows


def table(d, lines):
    by = {}
    for r in csv.DictReader(open(one(d, "kernel_trace.csv"))):
        by.setdefault(short(r["Kernel_Name"]), []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    total = sum(sum(v) for v in by.values())
    lines += ["| kernel | calls | total ms | avg us | min us | max us | % |", "|---|---:|---:|---:|---:|---:|---:|"]
    for k, v in sorted(by.items(), key=lambda kv: -sum(kv[1])):
        lines.append(f"| {k} | {len(v)} | {sum(v)/1e3:.3f} | {st.mean(v):.2f} | {min(v):.2f} | {max(v):.2f} | {100*sum(v)/total:.1f} |")


out = {"tag": "r02", "algorithmic_bytes_per_iteration": algo, "traffic": {}, "iterations_per_launch": 2000}
lines = ["# rocprofv3 summary (r02) -- LogModel n=4096 d=1024, tile engine (k_tile)", "",
         "Commands: tools/profile_r02.sh (kernel trace + stats of `bench.py --steps 20 --warmup 5` -- the driver's command -- and of the "
         "default `bench.py` (2000 steps); `--pmc FETCH_SIZE` and `--pmc WRITE_SIZE` in passes of their own).", ""]

for d, steps in (("trace20", 20), ("trace2000", 2000)):
    b = bench_line(d)
    durs, rows = launches(d, "k_tile")
    # bench.py's k_tile launches in order: [warm-up when >= 16 steps,] first call (`steps`), timed call (`steps`), long call (max(steps, 2000)),
    # time-to-converge (with norms)
    lines += [f"## `bench.py --steps {steps}`" + (" --warmup 5" if steps == 20 else ""), ""]
    table(d, lines)
    names = [short(r["Kernel_Name"]) for r in rows]
    lines += ["", "k_tile launches in order (us): " + ", ".join(f"{x:.1f}" for x in durs),
              f"instantiations: {sorted(set(names))}",
              f"VGPR {rows[1]['VGPR_Count']}, SGPR {rows[1]['SGPR_Count']}, grid {rows[1]['Grid_Size_X']} x wg {rows[1]['Workgroup_Size_X']}, "
              f"LDS {rows[1]['LDS_Block_Size']}, scratch {rows[1]['Scratch_Size']}"]
    ti = 1 if steps == 20 else 2   # the default run's 100 warm-up steps are a k_tile launch of their own
    timed = durs[ti]
    out[f"steps{steps}"] = {"k_tile_launch_us": durs, "timed_launch_us": timed, "us_per_iteration": timed / steps,
                            "bench_avg_launch_us": b["roofline"]["avg_launch_us"], "bench_value": b["value"],
                            "bench_frac": b["roofline"]["frac"]}
    lines += ["", f"timed launch ({steps} iterations in ONE launch): {timed:.1f} us under the profiler = {timed/steps:.2f} us/iteration; "
              f"bench.py's HIP-event bracket of the same run: {b['roofline']['avg_launch_us']:.1f} us "
              f"(value {b['value']:.0f} it/s, frac {b['roofline']['frac']:.3f} of 8 TB/s in algorithmic bytes, "
              f"streamed_frac {b['roofline']['streamed_frac']:.3f})", ""]

# PMC: the 2000-iteration launches (the timed one and the long one are both 2000 steps in the default run)
vals = {}
for d, c in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
    v = []
    for r in csv.DictReader(open(one(d, "counter_collection.csv"))):
        if r["Counter_Name"] == c and headline(r["Kernel_Name"]):
            v.append(float(r["Counter_Value"]))
    vals[c] = v
f = st.median(sorted(vals["FETCH_SIZE"])[-2:])   # the two 2000-iteration launches are the largest
w = st.median(sorted(vals["WRITE_SIZE"])[-2:])
rb, wb = f * 2 * 1024, w * 1024
out["traffic"]["k_tile"] = {"fetch_kib_raw": f, "write_kib": w, "read_bytes": rb, "write_bytes": wb, "total_bytes": rb + wb,
                            "iterations": 2000, "total_bytes_per_iteration": (rb + wb) / 2000,
                            "all_launches_fetch_kib": vals["FETCH_SIZE"], "all_launches_write_kib": vals["WRITE_SIZE"]}
lines += ["## fabric traffic of a 2000-iteration k_tile launch (PMC, separate passes)", "",
          f"FETCH_SIZE {f:.0f} KiB raw (x 2 x 1024 = {rb/1e9:.2f} GB), WRITE_SIZE {w:.0f} KiB ({wb/1e9:.3f} GB): "
          f"{(rb+wb)/2000/1e6:.1f} MB per iteration against {algo/1e6:.1f} MB algorithmic (24 E).  w never moves (registers); "
          f"what streams is half of z (4 E = {algo/6/1e6:.1f} MB: the other half lives in LDS) plus the tagged records "
          f"(degree partials and node values, 16 bytes per double, re-read while polling).",
          "", "per launch, in order: FETCH_SIZE KiB " + ", ".join(f"{x:.0f}" for x in vals["FETCH_SIZE"]),
          "per launch, in order: WRITE_SIZE KiB " + ", ".join(f"{x:.0f}" for x in vals["WRITE_SIZE"])]

open(os.path.join(out_dir, "r02_rocprof_summary.md"), "w").write("\n".join(lines) + "\n")
json.dump(out, open(os.path.join(out_dir, "r02_rocprof_summary.json"), "w"), indent=1)
for d in ("trace20", "trace2000"):
    shutil.copy(one(d, "kernel_stats.csv"), os.path.join(out_dir, f"r02_kernel_stats_{d}.csv"))
    shutil.copy(os.path.join(base, d + ".json"), os.path.join(out_dir, f"r02_bench_{d}.json"))
print("\n".join(lines))
