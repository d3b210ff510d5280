"""Summarise rocprofv3 outputs of bench.py into profiles/ (markdown + json).

    python tools/profile_summary.py <round tag> <kernel-trace dir> <pmc FETCH_SIZE dir> <pmc WRITE_SIZE dir>

PMC handling follows MI355X_MICROARCH.md "HBM": FETCH_SIZE / WRITE_SIZE are in KiB-units (x 1024 B);
on gfx950 FETCH_SIZE reports exactly 1/2 of the bytes of a wide coalesced (16 B/lane) streaming read,
so it is doubled for the edge pass; WRITE_SIZE is exact for 16 B/lane streaming stores.
"""
import csv, glob, json, os, statistics as st, sys

tag, trace_dir, fetch_dir, write_dir = sys.argv[1:5]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out_dir = os.path.join(ROOT, "profiles")
os.makedirs(out_dir, exist_ok=True)


def one(pattern):
    files = glob.glob(pattern, recursive=True)
    return files[0] if files else None


def short(name):
    return name.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]


rows = list(csv.DictReader(open(one(os.path.join(trace_dir, "**", "*kernel_trace.csv")))))
by = {}
for r in rows:
    by.setdefault(short(r["Kernel_Name"]), []).append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
total = sum(sum(v) for v in by.values())
lines = [f"# rocprofv3 --kernel-trace --stats summary ({tag})", "",
         "command: `GRAPHLEARN_AMD_NO_GRAPH=1 rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --no-cpu-baseline --no-converge` (eager launches: the profiler serialises graph replays)", "",
         "| kernel | calls | total ms | avg us | min us | max us | % |", "|---|---:|---:|---:|---:|---:|---:|"]
summary = {}
for k, v in sorted(by.items(), key=lambda kv: -sum(kv[1])):
    lines.append(f"| {k} | {len(v)} | {sum(v)/1e6:.3f} | {st.mean(v)/1e3:.2f} | {min(v)/1e3:.2f} | {max(v)/1e3:.2f} | {100*sum(v)/total:.1f} |")
    summary[k] = {"calls": len(v), "avg_us": st.mean(v) / 1e3, "total_ms": sum(v) / 1e6}

edge = [r for r in rows if "k_edge<false" in r["Kernel_Name"]]
node = [r for r in rows if "k_node<false" in r["Kernel_Name"]]
seq = sorted(edge + node, key=lambda r: int(r["Start_Timestamp"]))
gaps = [int(seq[i + 1]["Start_Timestamp"]) - int(seq[i]["End_Timestamp"]) for i in range(len(seq) - 1)]
span = (int(seq[-1]["End_Timestamp"]) - int(seq[0]["Start_Timestamp"])) / max(len(edge), 1)
lines += ["", f"iteration = k_edge + k_node: span per iteration {span/1e3:.2f} us under the profiler (graph-replayed launches are "
          f"serialised by the tool; un-profiled: see BENCH line), median gap between kernels {st.median(gaps)/1e3:.2f} us",
          f"k_edge<false>: VGPR {edge[0]['VGPR_Count']}, SGPR {edge[0]['SGPR_Count']}, grid {edge[0]['Grid_Size_X']} x wg {edge[0]['Workgroup_Size_X']}, LDS {edge[0]['LDS_Block_Size']}, scratch {edge[0]['Scratch_Size']}"]


def pmc(d, counter):
    f = one(os.path.join(d, "**", "*counter_collection.csv"))
    vals = {}
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == counter:
            vals.setdefault(short(r["Kernel_Name"]), []).append(float(r["Counter_Value"]))
    return vals


fetch, write = pmc(fetch_dir, "FETCH_SIZE"), pmc(write_dir, "WRITE_SIZE")
n = 4096
algo = 12 * n * (n - 1)
lines += ["", "## HBM traffic per launch (PMC, separate passes)", "",
          "| kernel | FETCH_SIZE (KiB, raw) | WRITE_SIZE (KiB) | read bytes (FETCH x 2 x 1024) | write bytes | total MB |", "|---|---:|---:|---:|---:|---:|"]
traffic = {}
def pick(d, prefix):
    for name, v in d.items():
        if name.startswith(prefix):
            return v[5:] or v
    return [0.0]


for k in ("k_edge<false", "k_node<false"):
    f = st.mean(pick(fetch, k))
    w = st.mean(pick(write, k))
    rb, wb = f * 2 * 1024, w * 1024
    traffic[k] = {"fetch_kib_raw": f, "write_kib": w, "read_bytes": rb, "write_bytes": wb, "total_bytes": rb + wb}
    lines.append(f"| {k} | {f:.1f} | {w:.1f} | {rb/1e6:.1f} MB | {wb/1e6:.1f} MB | {(rb+wb)/1e6:.1f} |")
lines += ["", f"algorithmic bytes per iteration (SURVEY 8d): 24 E = {algo/1e6:.1f} MB; the (w0, w1, z) working set is "
          f"3 x 68 MB = 204 MB < 256 MiB Infinity Cache, and rocprofv3's FETCH/WRITE_SIZE count the L2's fabric-side "
          f"requests (Infinity-Cache hits included), so they measure L2<->fabric traffic, not DRAM-only traffic."]
open(os.path.join(out_dir, f"{tag}_rocprof_summary.md"), "w").write("\n".join(lines) + "\n")
json.dump({"tag": tag, "kernels": summary, "span_per_iteration_us": span / 1e3, "traffic": traffic,
           "algorithmic_bytes_per_iteration": algo}, open(os.path.join(out_dir, f"{tag}_rocprof_summary.json"), "w"), indent=1)
print("\n".join(lines))
