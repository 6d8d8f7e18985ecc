#!/usr/bin/env python3
"""Static instruction census of one kernel in a hipcc -S -gline-tables-only listing.

usage: isa_phase_count.py LISTING.s MANGLED_KERNEL_NAME [--bb]
Prints, per basic block: instruction counts by class (VALU / MFMA / SALU / LDS / VMEM / other), the source lines the
block's instructions come from (file:line as in the .loc directives; inlined callees keep their own file) and the
block's successors.  tools/chain_valu_model.py multiplies these by trip counts."""
import re, sys, collections

def classify(op):
    if op.startswith('v_mfma') or op.startswith('v_smfma'): return 'MFMA'
    if op.startswith('v_'): return 'VALU'
    if op.startswith('ds_'): return 'LDS'
    if op.startswith(('global_', 'buffer_', 'flat_', 'scratch_')): return 'VMEM'
    if op.startswith('s_load') or op.startswith('s_buffer_load'): return 'SMEM'
    if op in ('s_waitcnt', 's_nop', 's_barrier', 's_sleep') : return 'WAIT'
    if op.startswith('s_'): return 'SALU'
    return 'OTHER'

def parse(path, kernel):
    lines = open(path).read().split('\n')
    start = next(i for i, l in enumerate(lines) if l.startswith(kernel + ':'))
    files = {}
    for l in lines:
        m = re.match(r'\s*\.file\s+(\d+)\s+"([^"]*)"\s+"([^"]*)"', l)
        if m: files[int(m.group(1))] = m.group(3)
    blocks = []  # (label, counts, locs, succ, ops)
    cur = dict(label='entry', counts=collections.Counter(), locs=collections.Counter(), succ=[], ops=[])
    loc = None
    for l in lines[start + 1:]:
        if l.startswith('.Lfunc_end'): break
        m = re.match(r'(\.LBB\d+_\d+):', l)
        if m:
            blocks.append(cur)
            cur = dict(label=m.group(1), counts=collections.Counter(), locs=collections.Counter(), succ=[], ops=[])
            continue
        m = re.match(r'\s*\.loc\s+(\d+)\s+(\d+)', l)
        if m:
            loc = (files.get(int(m.group(1)), m.group(1)), int(m.group(2)))
            continue
        s = l.strip()
        if not s or s.startswith((';', '.', '//')): continue
        op = s.split()[0]
        if not re.match(r'[a-z]', op): continue
        c = classify(op)
        cur['counts'][c] += 1
        cur['ops'].append(op)
        if loc: cur['locs'][(loc, c)] += 1
        if op.startswith('s_cbranch') or op == 's_branch':
            cur['succ'].append(s.split()[-1])
    blocks.append(cur)
    return blocks

if __name__ == '__main__':
    blocks = parse(sys.argv[1], sys.argv[2])
    tot = collections.Counter()
    for b in blocks:
        tot.update(b['counts'])
        c = b['counts']
        byline = collections.Counter()
        for (loc, cl), n in b['locs'].items():
            if cl in ('VALU', 'MFMA'): byline[loc] += n
        top = ', '.join(f"{f.split('/')[-1].replace('tmac_','')}:{ln}x{n}" for (f, ln), n in sorted(byline.items(), key=lambda kv: (kv[0][0], kv[0][1])))
        print(f"{b['label']:12s} VALU {c['VALU']:4d} MFMA {c['MFMA']:3d} SALU {c['SALU']:4d} WAIT {c['WAIT']:3d} LDS {c['LDS']:3d} VMEM {c['VMEM']:3d} -> {','.join(b['succ'])}")
        if '--bb' in sys.argv and top: print('             ', top)
    print('TOTAL emitted', dict(tot))
