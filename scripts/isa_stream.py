"""Print the instruction-category stream of one kernel from a -save-temps .s file (M = MFMA, v = VALU, T = transcendental,
r / w = LDS read / write, G = global / buffer load, | = s_barrier, . = s_waitcnt, B = branch, L = label, s = scalar)."""
import sys


def cat(l):
    op = l.split()[0]
    if op.startswith('v_mfma'): return 'M'
    if op.startswith('ds_read'): return 'r'
    if op.startswith('ds_write'): return 'w'
    if op.startswith(('buffer_load', 'global_load')): return 'G'
    if op.startswith(('buffer_store', 'global_store')): return 'S'
    if op.startswith('scratch_'): return 'X'
    if op.startswith('s_barrier'): return '|'
    if op.startswith('s_waitcnt'): return '.'
    if op.startswith(('v_exp', 'v_rcp', 'v_log', 'v_rsq', 'v_sqrt')): return 'T'
    if op.startswith('v_'): return 'v'
    if op.startswith(('s_cbranch', 's_branch')): return 'B'
    if l.endswith(':'): return 'L'
    return 's'


s = open(sys.argv[1]).read()
k = s[s.index(sys.argv[2]):]
k = k[k.index(':'):k.index('.end_amdhsa_kernel')] if '.end_amdhsa_kernel' in k else k
lines = [l.strip() for l in k.split('\n') if l.strip() and not l.strip().startswith((';', '.', '//'))]
print(''.join(cat(l) for l in lines))
