"""Scan gfx950 assembly (hipcc -S --cuda-device-only) for 96/128-bit buffer stores with an SGPR soffset whose data registers are
written by one of the next two VALU instructions: the compiler inserts no wait state there, the hardware needs one (half_io.h).
usage: python tools/scan_store_hazard.py file.s"""
import re, sys
lines = open(sys.argv[1]).read().split('\n')
kern = None
hits = 0
stores = 0
for i, l in enumerate(lines):
    m = re.match(r'\s*\.globl\s+(\S+)', l)
    if m: kern = m.group(1)
    m = re.match(r'\s*buffer_store_dwordx([34])\s+v\[(\d+):(\d+)\],\s*(\S+),\s*s\[\d+:\d+\],\s*(\S+)', l)
    if not m: continue
    soff = m.group(5)
    if not soff.startswith('s'): continue
    stores += 1
    lo, hi = int(m.group(2)), int(m.group(3))
    # next two real instructions
    nxt = []
    j = i + 1
    while j < len(lines) and len(nxt) < 2:
        t = lines[j].strip()
        j += 1
        if not t or t.startswith('.') or t.startswith(';') or t.endswith(':'): continue
        nxt.append(t)
    for k, t in enumerate(nxt):
        if not t.startswith('v_'): continue
        d = re.match(r'v_\S+\s+v\[(\d+):(\d+)\]|v_\S+\s+v(\d+)', t)
        if not d: continue
        if d.group(3) is not None: dlo = dhi = int(d.group(3))
        else: dlo, dhi = int(d.group(1)), int(d.group(2))
        if dlo <= hi and dhi >= lo:
            hits += 1
            print('%s: line %d: %s  ->  [+%d] %s' % (kern[:70], i + 1, l.strip(), k + 1, t))
print('stores with SGPR soffset:', stores, 'followed within 2 instructions by a VALU write of the data registers:', hits)
