"""Dev tool: where a kernel touches scratch (per basic block), from the -save-temps assembly.
usage: python tools/scratch_sites.py <mangled-name-substring> [asm file]"""
import re, sys
asm = sys.argv[2] if len(sys.argv) > 2 else "graph_learning_amd/csrc/build/report/gl_tile-hip-amdgcn-amd-amdhsa-gfx950.s"
s = open(asm).read()
m = re.search(r"^(\S*%s\S*):" % re.escape(sys.argv[1]), s, flags=re.M)
i = m.start(); j = s.index(".Lfunc_end", i)
b = s[i:j].split("\n")
lab, cnt, order = "entry", {}, []
for k, l in enumerate(b):
    if re.match(r"^\.LBB\d+_\d+:", l):
        lab = l.strip()
    if "scratch_" in l:
        if lab not in cnt:
            cnt[lab] = [0, 0, k]; order.append(lab)
        cnt[lab][0 if "store" in l else 1] += 1
nmax = sum(1 for l in b if "v_max_f64" in l)
first_max = next((k for k, l in enumerate(b) if "v_max_f64" in l), -1)
last_max = max((k for k, l in enumerate(b) if "v_max_f64" in l), default=-1)
print(f"{m.group(1)}: {len(b)} lines, pass spans lines {first_max}..{last_max}")
for lab in order:
    st, ld, k = cnt[lab]
    print(f"  line {k:5d} {lab[:60]:60s} stores {st:3d} loads {ld:3d}")
