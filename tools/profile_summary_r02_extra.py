"""Summarise the rocprofv3 passes of tools/profile_r02_extra.sh into profiles/r02_extra_summary.md and copy the raw
kernel-stats CSVs beside it.

    python tools/profile_summary_r02_extra.py [gpurun_out/prof_r02x]

PMC handling as in tools/profile_summary_r02.py (MI355X_MICROARCH.md "HBM": FETCH_SIZE x 2 x 1024 B on gfx950 for wide
coalesced reads, WRITE_SIZE x 1024 B)."""
import csv, glob, os, re, shutil, statistics as st, sys

base = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/prof_r02x"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out_dir = os.path.join(ROOT, "profiles")


def one(d, suffix):
    f = sorted(glob.glob(os.path.join(base, d, "**", "*" + suffix), recursive=True), key=os.path.getmtime)
    return f[-1]


def short(name):
    return re.sub(r"\(anonymous namespace\)::|void ", "", name).split("(")[0]


def table(d, lines, top=8):
    by = {}
    for r in csv.DictReader(open(one(d, "kernel_trace.csv"))):
        by.setdefault(short(r["Kernel_Name"]), []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    total = sum(sum(v) for v in by.values())
    lines += ["| kernel | calls | total ms | avg us | min us | max us | % |", "|---|---:|---:|---:|---:|---:|---:|"]
    for k, v in sorted(by.items(), key=lambda kv: -sum(kv[1]))[:top]:
        lines.append(f"| {k} | {len(v)} | {sum(v)/1e3:.3f} | {st.mean(v):.2f} | {min(v):.2f} | {max(v):.2f} | {100*sum(v)/total:.1f} |")
    return by


def log_tail(name, pat):
    for l in open(os.path.join(base, name)):
        if re.search(pat, l):
            return l.strip()
    return ""


lines = ["# rocprofv3 summary (r02, kernels beside the headline)", "",
         "Commands: tools/profile_r02_extra.sh (each pass: `rocprofv3 --output-format csv --kernel-trace --stats -- python3 tools/<bench>`;",
         "FETCH_SIZE and WRITE_SIZE at n = 8192 in passes of their own).  Raw stats: `profiles/r02_*_kernel_stats.csv`.", ""]
for n in (1024, 2048, 8192):
    d = f"tile{n}"
    shutil.copy(one(d, "kernel_stats.csv"), os.path.join(out_dir, f"r02_tile_n{n}_kernel_stats.csv"))
    lines += [f"## LogModel loop, n = {n}: `tools/iter_bench.py {n} 500` (one 128-iteration warm-up call, three 500-iteration calls)", ""]
    by = table(d, lines, 4)
    kt = [k for k in by if k.startswith("k_tile")][0]
    calls = sorted(by[kt])
    per_it = st.median(calls[-3:]) / 500
    e = n * (n - 1) // 2
    lines += ["", f"`{kt}`: 500-iteration launches {', '.join(f'{c/1e3:.2f}' for c in calls[-3:])} ms = **{per_it:.2f} us per iteration**"
              f" (24 E bytes / that = {24*e/per_it/1e6:.2f} TB/s = {24*e/per_it/1e6/8:.2f} of 8 TB/s).  Bench line of the same process: "
              + log_tail(d + ".log", r"^n=\d+ steps"), ""]
    if n == 8192:
        tr = {}
        for pmc, mult in (("fetch8192", 2 * 1024), ("write8192", 1024)):
            rows = [r for r in csv.DictReader(open(one(pmc, "counter_collection.csv"))) if short(r["Kernel_Name"]).startswith("k_tile")]
            rows.sort(key=lambda r: int(r["Start_Timestamp"]))
            per = {}
            for r in rows:
                per.setdefault(r["Dispatch_Id"], 0.0)
                per[r["Dispatch_Id"]] += float(r["Counter_Value"])
            vals = list(per.values())[-3:]  # the three 500-iteration launches
            tr[pmc] = st.mean(vals) * mult / 500
        lines += [f"PMC (per iteration, 500-iteration launches): FETCH_SIZE x 2 x 1024 = {tr['fetch8192']/1e6:.1f} MB, WRITE_SIZE x 1024 = "
                  f"{tr['write8192']/1e6:.1f} MB, together {(tr['fetch8192']+tr['write8192'])/1e6:.1f} MB against 24 E = {24*e/1e6:.1f} MB "
                  f"algorithmic (the resident quarter of w never moves) -> {(tr['fetch8192']+tr['write8192'])/per_it/1e6:.2f} TB/s between the L2s and the fabric "
                  "(the counters sit in front of the Infinity Cache: the streamed w that it serves is included).", ""]
shutil.copy(one("lgmrf", "kernel_stats.csv"), os.path.join(out_dir, "r02_lgmrf_kernel_stats.csv"))
lines += ["## LGMRF n = 2048, d = 512: the first 256 node updates of a cycle (`GRAPHLEARN_AMD_LGMRF_MAX_NODES=256 tools/lgmrf_bench.py 2048 512 1`)", ""]
by = table("lgmrf", lines, 10)
adm = [k for k in by if k.startswith("k_lg_admm")]
if adm:
    v = by[adm[0]]
    per_cycle = sum(sum(x) for k, x in by.items() if k.startswith(("k_gj_", "k_lg_prepare", "k_lg_shift")))
    per_node = sum(sum(x) for k, x in by.items() if k.startswith("k_") and not k.startswith(("k_gj_", "k_lg_prepare", "k_lg_shift")))
    lines += ["", f"`{adm[0]}`: {len(v)} launches (one per node update, all ADMM steps inside), average {st.mean(v):.1f} us, median {st.median(v):.1f} us; "
              f"the kernels of a node update together {per_node/len(v):.1f} us; once per cycle (Theta + rho I inverted from scratch, "
              f"`k_gj_*`): {per_cycle/1e3:.1f} ms = {per_cycle/2048:.1f} us per node update of a full cycle.", ""]
for bits in (64, 32):
    d = f"gd{bits}"
    shutil.copy(one(d, "kernel_stats.csv"), os.path.join(out_dir, f"r02_graphdict_fp{bits}_kernel_stats.csv"))
    lines += [f"## GraphDictLog K = 8, n = 1024, T = 4096, {'fp64' if bits == 64 else 'fp32 contractions'} (`GD_BITS={bits} tools/graphdict_bench.py`)", ""]
    table(d, lines, 8)
    lines += ["", "Bench lines of the same process: " + " | ".join(l.strip() for l in open(os.path.join(base, d + ".log")) if "per step" in l or "ms" in l and "rocprof" not in l)[:600], ""]
open(os.path.join(out_dir, "r02_extra_summary.md"), "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
