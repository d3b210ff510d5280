"""Summarise the rocprofv3 passes of tools/profile_r04.sh into profiles/r04_* (markdown + json + the raw stats CSVs).

    python tools/profile_summary_r04.py [gpurun_out/prof_r04]

PMC handling follows MI355X_MICROARCH.md "HBM": FETCH_SIZE / WRITE_SIZE are in KiB units; on gfx950 FETCH_SIZE
reports exactly 1/2 of the bytes of a wide coalesced (16 B/lane) streaming read, so it is doubled; WRITE_SIZE is
exact for 16 B/lane streaming stores.  (The records of the hand-offs are 16-byte accesses too.)
"""
import csv, glob, json, os, shutil, statistics as st, sys

base = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/prof_r04"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out_dir = os.path.join(ROOT, "profiles")
n = 4096
algo = 12 * n * (n - 1)
TIMED = "k_tile<32, true, false, 0, false>"     # the instantiation bench.py's timed call launches (U = 32, no norms)


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


def launches(d):
    rows = [r for r in csv.DictReader(open(one(d, "kernel_trace.csv"))) if short(r["Kernel_Name"]) == TIMED]
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


def counter(d, c):   # per launch of the timed instantiation, in launch order
    rows = [r for r in csv.DictReader(open(one(d, "counter_collection.csv"))) if r["Counter_Name"] == c and short(r["Kernel_Name"]) == TIMED]
    rows.sort(key=lambda r: int(r.get("Dispatch_Id", 0)))
    return [float(r["Counter_Value"]) for r in rows]


out = {"tag": "r04", "algorithmic_bytes_per_iteration": algo, "timed_instantiation": TIMED, "traffic_bytes_per_launch": {"k_tile": {}}}
lines = ["# rocprofv3 summary (r04) -- LogModel n=4096 d=1024, tile engine (k_tile)", "",
         "Commands: tools/profile_r04.sh (kernel trace + stats of `bench.py --steps 20 --warmup 5` -- the driver's command -- and of the "
         "default `bench.py` (2000 steps); `--pmc FETCH_SIZE` and `--pmc WRITE_SIZE` of both commands in passes of their own).", ""]

for d, steps in (("trace20", 20), ("trace2000", 2000)):
    b = bench_line(d)
    durs, rows = launches(d)
    # bench.py's launches of the timed instantiation in order: warm-up (>= 4 steps), first call (`steps`), timed call (`steps`),
    # long call (max(steps, 2000)); time-to-converge runs the instantiation with the norms
    lines += [f"## `bench.py --steps {steps}`" + (" --warmup 5" if steps == 20 else ""), ""]
    table(d, lines)
    lines += ["", f"{TIMED} launches in order (us): " + ", ".join(f"{x:.1f}" for x in durs),
              f"VGPR {rows[1]['VGPR_Count']}, SGPR {rows[1]['SGPR_Count']}, grid {rows[1]['Grid_Size_X']} x wg {rows[1]['Workgroup_Size_X']}, "
              f"LDS {rows[1]['LDS_Block_Size']}, scratch {rows[1]['Scratch_Size']}"]
    ti = 2   # the warm-up steps (5 or 100: at least the engine's shortest call of 4) are a launch of their own
    timed = durs[ti]
    r = b["roofline"]
    out[f"steps{steps}"] = {"launch_us": durs, "timed_launch_us": timed, "us_per_iteration": timed / steps,
                            "bench_avg_launch_us": r["avg_launch_us"], "bench_value": b["value"], "bench_frac": r["frac"], "bench_bound": r["bound"]}
    lines += ["", f"timed launch ({steps} iterations in ONE launch): {timed:.1f} us under the profiler = {timed/steps:.2f} us/iteration; "
              f"bench.py's HIP-event bracket of the same run: {r['avg_launch_us']:.1f} us "
              f"(value {b['value']:.0f} it/s; roofline bound {r['bound']}: "
              + (f"{r['achieved']:.2f} of {r['peak']} {r['unit']} = {r['frac']:.3f} in real flops, {r['valu_issue_frac']:.3f} of the fp64 issue slots; "
                 if r.get('achieved') is not None else "census stale: no figure; ")
              + f"SURVEY 8d's algorithmic-bytes figure {r['hbm_algorithmic']['frac']:.2f} x 8 TB/s)", ""]

lines += ["## fabric traffic of the timed launches (PMC, separate passes; FETCH_SIZE x 2 x 1024 + WRITE_SIZE x 1024 bytes)", ""]
for steps, ti in ((20, 2), (2000, 2)):
    f, w = counter(f"fetch{steps}", "FETCH_SIZE"), counter(f"write{steps}", "WRITE_SIZE")
    rb, wb = f[ti] * 2 * 1024, w[ti] * 1024
    out["traffic_bytes_per_launch"]["k_tile"][str(steps)] = rb + wb
    out[f"traffic{steps}"] = {"fetch_kib_raw": f[ti], "write_kib": w[ti], "read_bytes": rb, "write_bytes": wb, "total_bytes": rb + wb,
                              "per_iteration": (rb + wb) / steps, "all_launches_fetch_kib": f, "all_launches_write_kib": w}
    lines += [f"`--steps {steps}`: timed launch FETCH_SIZE {f[ti]:.0f} KiB raw (= {rb/1e6:.1f} MB), WRITE_SIZE {w[ti]:.0f} KiB ({wb/1e6:.1f} MB): "
              f"{(rb+wb)/1e6:.1f} MB per launch = {(rb+wb)/steps/1e6:.2f} MB per iteration against {algo/1e6:.1f} MB algorithmic (24 E).",
              "  per launch, in order: FETCH_SIZE KiB " + ", ".join(f"{x:.0f}" for x in f) + "; WRITE_SIZE KiB " + ", ".join(f"{x:.0f}" for x in w), ""]
t20, t2000 = out["traffic_bytes_per_launch"]["k_tile"]["20"], out["traffic_bytes_per_launch"]["k_tile"]["2000"]
per_it = (t2000 - t20) / 1980
lines += [f"A call's fixed traffic (w into the registers, half of 2 gamma z into LDS, w back): {(t20 - 20 * per_it)/1e6:.1f} MB; per iteration "
          f"{per_it/1e6:.2f} MB (w never moves: what streams is half of z, 4 E = {algo/6/1e6:.1f} MB, plus the tagged records).", ""]

stamps = os.path.join(base, "call_stamps_20.txt")
if os.path.exists(stamps):
    shutil.copy(stamps, os.path.join(out_dir, "r04_call_stamps_20.txt"))
    lines += ["## phase stamps of 20-iteration calls (`tools/call_stamps.py 4096 20`, GRAPHLEARN_AMD_RES_STAMPS=1)", "", "```"]
    lines += [l.rstrip() for l in open(stamps) if l.startswith(("--", "[tile"))]
    lines += ["```", ""]

tl = os.path.join(base, "stamps_timeline.txt")
if os.path.exists(tl):
    keep = [l.rstrip() for l in open(tl) if l.startswith(("==", "   "))]
    open(os.path.join(out_dir, "r04_stamps_timeline.txt"), "w").write("\n".join(keep) + "\n")
    lines += ["## chip-wide timeline of an iteration (`tools/stamps_raw.py 4096 40`: every wave's stamps)", "", "```"] + keep + ["```", ""]

open(os.path.join(out_dir, "r04_rocprof_summary.md"), "w").write("\n".join(lines) + "\n")
json.dump(out, open(os.path.join(out_dir, "r04_rocprof_summary.json"), "w"), indent=1)
for d in ("trace20", "trace2000"):
    shutil.copy(one(d, "kernel_stats.csv"), os.path.join(out_dir, f"r04_kernel_stats_{d}.csv"))
    shutil.copy(os.path.join(base, d + ".json"), os.path.join(out_dir, f"r04_bench_{d}.json"))
print("\n".join(lines))
