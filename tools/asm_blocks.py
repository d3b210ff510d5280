"""Dev tool: per-basic-block census (scratch, global, fp64) of one kernel in a gfx950 assembly file.
usage: python tools/asm_blocks.py build/gl_tile.s <mangled-name-substring>"""
import re, sys
s = open(sys.argv[1]).read()
parts = re.split(r'\n(_Z\w+):[^\n]*\n', s)
for i in range(1, len(parts), 2):
    if sys.argv[2] not in parts[i]:
        continue
    body = parts[i + 1].split('.Lfunc_end')[0]
    blk, cur, blocks = 'entry', [], []
    for l in body.split('\n'):
        m = re.match(r'^(\.LBB\d+_\d+):', l)
        if m:
            blocks.append((blk, cur)); blk, cur = m.group(1), []
        elif l.startswith('\t') and not l.strip().startswith(('.', ';')):
            cur.append(l)
    blocks.append((blk, cur))
    print(parts[i])
    for b, c in blocks:
        cnt = lambda p: sum(bool(re.search(p, l)) for l in c)
        if cnt('scratch') or cnt('global_store') or cnt(r'v_\w+_f64') > 4 or cnt('s_barrier'):
            print(f"  {b:12s} n {len(c):4d} f64 {cnt(r'v_[a-z]+_f64'):4d} gload {cnt('global_load'):3d} gstore {cnt('global_store'):3d} "
                  f"scr_ld {cnt('scratch_load'):3d} scr_st {cnt('scratch_store'):3d} ds {cnt(r'ds_'):3d} barrier {cnt('s_barrier')}")
