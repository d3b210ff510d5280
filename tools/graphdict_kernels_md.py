"""Dev tool: profiles/r04_graphdict_kernels.md from a rocprofv3 kernel-stats CSV of tools/graphdict_bench.py (GD_BITS=32).
usage: python tools/graphdict_kernels_md.py gpurun_out/<run>/prof/runc_kernel_stats.csv "<step line of graphdict_bench.py>" """
import csv, os, sys

rows = {}
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
    rows[n] = (int(r["Calls"]), float(r["AverageNs"]) / 1e3)
# the same kernels at the end of round 3 (VERDICT r03 "weak" 10 and this round's first profile, gpurun_out/r04m: us per launch)
before = {"k_gd_update2<8>": 71.8, "k_gd_dual_update<8>": 46.4, "k_gd_m<8>": 35.2, "k_gd_finish3<8>": 32.8, "k_gd_reduce2": 28.5,
          "k_gd_to_f32": 33.0, "k_gd_coef_commit": 5.6, "k_gd_coef_stat": 4.0, "k_gd_grad_c<float, 8, 2>": 467.0, "k_gd_grad_w<float, 8, 8>": 404.0}
if "k_gd_grad_c<float, 8, 2, 4>" in rows:  # (the x_j of three columns ahead: the same kernel as far as this table goes)
    before["k_gd_grad_c<float, 8, 2, 4>"] = before.pop("k_gd_grad_c<float, 8, 2>")
if "k_gd_grad_c<float, 8, 3>" in rows:  # (round 4's last change: strips of three rows; the same kernel as far as this table goes)
    before["k_gd_grad_c<float, 8, 3>"] = before.pop("k_gd_grad_c<float, 8, 2>")
out = ["# GraphDictLog K = 8, n = 1024, T = 4096, fp32 contractions: the kernels of a step", "",
       "`rocprofv3 --kernel-trace --stats -- python3 tools/graphdict_bench.py` (GD_BITS=32; two `fit` calls of GD_STEPS steps + the gradient",
       "timing aid), average us per launch.  Round 3: the same kernels before this round's work on the tail (`k_gd_to_f32`: three",
       "launches per step then, 33 us together; now only at set-up).", "",
       "| kernel | round 3, us | now: launches | us |", "|---|---:|---:|---:|"]
for n in sorted(set(before) | {k for k in rows if k.startswith("k_gd")}, key=lambda k: -max(before.get(k, 0), rows.get(k, (0, 0))[1])):
    c, u = rows.get(n, (0, 0.0))
    out.append(f"| {n} | " + (f"{before[n]:.1f}" if n in before else "(set-up)") + f" | {c} | {u:.2f} |")
tail_now = sum(rows.get(k, (0, 0))[1] for k in ("k_gd_update2<8>", "k_gd_dual_update<8>", "k_gd_m<8>", "k_gd_finish3<8>", "k_gd_reduce2", "k_gd_coef_commit", "k_gd_coef_stat"))
out += ["", f"Behind the two contractions: 257 us per step at the end of round 3 (with the conversions), {tail_now:.0f} us now."]
if len(sys.argv) > 2:
    out += ["", "`tools/graphdict_bench.py` of the same build: " + sys.argv[2]]
out.append("")
p = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "r04_graphdict_kernels.md")
open(p, "w").write("\n".join(out))
print("\n".join(out))
