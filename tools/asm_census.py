"""Dev tool: per-kernel instruction census of a gfx950 assembly file (hipcc -S --cuda-device-only).
usage: python tools/asm_census.py build/gl_tile.s [name-filter]"""
import re, sys
s = open(sys.argv[1]).read()
flt = sys.argv[2] if len(sys.argv) > 2 else ""
parts = re.split(r'\n(_Z\w+):[^\n]*\n', s)
pats = [('scratch_load', r'scratch_load'), ('scratch_store', r'scratch_store'), ('acc_read', 'v_accvgpr_read'),
        ('acc_write', 'v_accvgpr_write'), ('ds_read', r'\bds_read'), ('ds_write', r'\bds_write'), ('bpermute', 'bpermute'),
        ('global_load', 'global_load'), ('global_store', 'global_store'), ('buffer_load', 'buffer_load'), ('buffer_store', 'buffer_store'),
        ('f64', r'v_(add|mul|fma|max|min|sqrt|rcp|div\w*)_f64'), ('dpp', r'(row_|quad_perm|row_ror|half_mirror)'),
        ('permlane', 'v_permlane'), ('readlane', 'v_readlane'), ('writelane', 'v_writelane'), ('cndmask', 'v_cndmask'),
        ('v_mov', r'\bv_mov_b'), ('waitcnt', 's_waitcnt'), ('barrier', 's_barrier')]
for i in range(1, len(parts), 2):
    name, body = parts[i], parts[i + 1].split('.Lfunc_end')[0]
    if flt not in name:
        continue
    lines = [l for l in body.split('\n') if l.startswith('\t') and not l.strip().startswith(('.', ';'))]
    print(name, 'instructions', len(lines))
    print('   ' + '  '.join(f"{k} {sum(1 for l in lines if re.search(p, l))}" for k, p in pats))
