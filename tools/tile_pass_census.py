"""Instruction census of the tile engine's pass, from the assembly hipcc wrote for the shipped build flags.

usage: python tools/tile_pass_census.py [--kernel ILi32ELb1ELb0ELi0E] [--out profiles/r04_tile_census.json]

Compiles graph_learning_amd/csrc/gl_tile.hip with the Makefile's flags plus -save-temps (tools/build_tile.sh), takes the
kernel bench.py times (k_tile<32, true, false, 0, false>: U = 32, no norms), and counts the instructions of the basic blocks that
make up the unrolled pass (the blocks holding the v_max_f64 of `p = max(0, .)`, 64 per pass) by class.  bench.py prices the
pass's fp64 vector instructions against the fp64 VALU issue rate (MI355X: 256 CUs x 4 SIMDs x 16 lanes per clock x 2.4 GHz
= 39.3 T lane-operations/s, i.e. the 78.6 TFLOP/s spec with an FMA counted twice) -- the roof that binds this kernel.
"""
import argparse, hashlib, json, os, re, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ap = argparse.ArgumentParser()
ap.add_argument("--kernel", default="ILi32ELb1ELb0ELi0E")
ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "r04_tile_census.json"))
ap.add_argument("--no-build", action="store_true")
a = ap.parse_args()
if not a.no_build:
    subprocess.run([os.path.join(ROOT, "tools", "build_tile.sh")], check=True, stdout=subprocess.DEVNULL)
asm = os.path.join(ROOT, "graph_learning_amd", "csrc", "build", "report", "gl_tile-hip-amdgcn-amd-amdhsa-gfx950.s")
s = open(asm).read()
m = re.search(r"^(\S*k_tile%s\S*):" % re.escape(a.kernel), s, flags=re.M)
body = s[m.start():s.index(".Lfunc_end", m.start())].split("\n")
blocks, cur = [], []
for l in body:
    if re.match(r"^\.LBB\d+_\d+:", l):
        blocks.append(cur); cur = []
    cur.append(l)
blocks.append(cur)
CLASSES = [("valu_f64_fma", r"^v_(fma|fmac)_f64"), ("valu_f64_other", r"^v_(add|mul|max|min)_f64"), ("valu_other", r"^v_"), ("lds", r"^ds_"),
           ("vmem", r"^(global_|buffer_|scratch_)"), ("wait", r"^s_(waitcnt|nop)"), ("salu", r"^s_")]
def census(lines):
    c = {k: 0 for k, _ in CLASSES}
    c["scratch"] = 0
    for l in lines:
        t = l.strip()
        if not l.startswith("\t") or t.startswith((".", ";")):
            continue
        op = t.split()[0]
        if op.startswith("scratch_"):
            c["scratch"] += 1
        for k, p in CLASSES:
            if re.match(p, op):
                c[k] += 1
                break
    return c
pass_blocks = [b for b in blocks if sum("v_max_f64" in l for l in b) >= 8]
# the kernel holds the pass twice: iteration 0 is an instantiation of its own in front of the loop (it runs underneath the
# prologue's loads); the census is of the loop's copy, the last 64 v_max_f64 in program order
keep, cnt = [], 0
for b in reversed(pass_blocks):
    if cnt >= 64:
        break
    keep.append(b)
    cnt += sum("v_max_f64" in l for l in b)
pass_blocks = keep[::-1]
tot = {}
for b in pass_blocks:
    for k, v in census(b).items():
        tot[k] = tot.get(k, 0) + v
tot["valu_f64"] = tot["valu_f64_fma"] + tot["valu_f64_other"]      # fp64 vector instructions (issue slots of the fp64 pipe)
tot["flop_f64"] = 2 * tot["valu_f64_fma"] + tot["valu_f64_other"]  # real flops per lane: a fused multiply-add is 2, add / mul / max 1
nmax = sum(sum("v_max_f64" in l for l in b) for b in pass_blocks)
assert nmax == 64, f"expected the 64 v_max_f64 of one pass (32 row pairs x 2 edges per lane), found {nmax}"
# the sources the timed kernel is compiled from (the kernel file and the headers its pass draws on)
SOURCES = ["gl_tile.hip", "gl_lanes.h", "gl_tile_plan.h", "gl_gridbar.h", "gl_logdeg.h", "gl_common.h"]
src = b"".join(open(os.path.join(ROOT, "graph_learning_amd", "csrc", f), "rb").read() for f in SOURCES)
out = {"kernel": m.group(1), "kernel_demangled": "k_tile<32, true, false, 0, false> (U = 32, write-back mask only, no norms, single graph)",
       "source_sha16": hashlib.sha256(src).hexdigest()[:16], "sources": SOURCES, "row_pairs_per_pass": 32, "edges_per_lane_and_row_pair": 2,
       "per_pass": tot, "per_row_pair": {k: v / 32.0 for k, v in tot.items()},
       "fp64_valu_peak_lane_ops_per_s": 256 * 4 * 16 * 2.4e9,
       "note": "basic blocks of the unrolled pass (those holding its 64 v_max_f64); per lane and row pair = 2 edges"}
os.makedirs(os.path.dirname(a.out), exist_ok=True)
json.dump(out, open(a.out, "w"), indent=1)
print(json.dumps(out["per_row_pair"]), "->", a.out)
