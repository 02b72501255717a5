"""Compressed view of a kernel's K loop from an ISA listing: one letter per instruction (M mfma, r ds_read, W ds_write, D LDS-DMA,
G global load, | s_waitcnt, B barrier, s scalar, v vector).  usage: isa_seq.py <file.s> <regex of the kernel's mangled name>"""
import re, sys
lines = open(sys.argv[1]).read().split('\n')
st = [i for i, l in enumerate(lines) if re.match(r'^_Z\d+' + sys.argv[2] + r'.*:\s', l)][0]
end = st
while 's_endpgm' not in lines[end]: end += 1
fn = [l for l in lines[st:end] if l.strip() and not l.strip().startswith(';')]
mf = [i for i, l in enumerate(fn) if 'v_mfma' in l]
lo = mf[0]
while not fn[lo].startswith('.LBB'): lo -= 1
hi = mf[-1]
while 's_cbranch' not in fn[hi] and 's_branch' not in fn[hi]: hi += 1
seq = []
for l in fn[lo:hi + 1]:
    t = l.strip().split()[0]
    if t.startswith('v_mfma'): seq.append('M')
    elif t.startswith('ds_read'): seq.append('r')
    elif t.startswith('ds_write'): seq.append('W')
    elif t.startswith('global_load_lds'): seq.append('D')
    elif t.startswith('global_load'): seq.append('G')
    elif t.startswith('s_waitcnt'): seq.append('|')
    elif t.startswith('s_barrier'): seq.append('B')
    elif t.startswith('s_cbranch') or t.startswith('s_branch'): seq.append('J')
    elif t.startswith('s_'): seq.append('s')
    elif t.startswith('v_'): seq.append('v')
    elif t.startswith('.LBB'): seq.append('\n' + t + ' ')
    else: seq.append('?')
print(''.join(seq))
