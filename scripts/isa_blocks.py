"""Per basic block of every kernel in a gfx950 ISA dump (hipcc -S --cuda-device-only): instruction, MFMA, scratch
(spill) and LDS-DMA counts -- shows whether spill traffic sits inside the MFMA loops.

    python scripts/isa_blocks.py /tmp/k32.s [name-substring]
"""
import re
import sys

lines = open(sys.argv[1]).read().split('\n')
pat = sys.argv[2] if len(sys.argv) > 2 else ''
fn = blk = None
info, order = {}, []
for l in lines:
    m = re.match(r'^(_Z\w+):', l) or re.match(r'^(\.LBB\d+_\d+):', l)
    if m:
        if m.group(1).startswith('_Z'):
            fn, blk = m.group(1), 'entry'
        else:
            blk = m.group(1)
        if fn:
            info[(fn, blk)] = dict(n=0, mfma=0, ld=0, st=0, dma=0)
            order.append((fn, blk))
        continue
    if fn is None or not l.startswith('\t') or l.startswith('\t.') or l.startswith('\t;'):
        continue
    i = info[(fn, blk)]
    i['n'] += 1
    i['mfma'] += 'v_mfma' in l
    i['ld'] += 'scratch_load' in l
    i['st'] += 'scratch_store' in l
    i['dma'] += ' lds' in l and 'global_load' in l
for k in order:
    i = info[k]
    if pat in k[0] and (i['mfma'] or i['ld'] or i['st']):
        print(k[0][-44:], k[1], i)
