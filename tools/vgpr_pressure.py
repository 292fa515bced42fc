#!/usr/bin/env python3
"""VGPR liveness over a straight stretch of gfx9 assembly (a loop body), cyclic: how many vector registers are live at every
instruction of the compiler's output -- where a kernel at the edge of the register file spends them.

  tools/vgpr_pressure.py file.s kernel_name_substring first_line last_line   (line numbers relative to the kernel's label)
"""
import re
import sys


def regs(tok):
    out = []
    for m in re.finditer(r'\bv\[(\d+):(\d+)\]|\bv(\d+)\b', tok):
        if m.group(1) is not None:
            out += list(range(int(m.group(1)), int(m.group(2)) + 1))
        else:
            out.append(int(m.group(3)))
    return out


def parse(line):
    line = line.split(';')[0].strip()
    if not line or line.endswith(':') or line.startswith('.'):
        return None
    op, _, rest = line.partition(' ')
    ops = [o.strip() for o in rest.split(',')] if rest else []
    all_use = op.startswith(('ds_write', 'buffer_store', 'scratch_store', 'global_store', 'v_cmp', 'v_readlane', 'v_readfirstlane', 's_', 'ds_bpermute')) and not op.startswith('v_cmpx')
    if op.startswith('v_cmp') or op.startswith('v_readlane') or op.startswith('v_readfirstlane') or op.startswith('s_'):
        return op, [], [r for o in ops for r in regs(o)]
    if all_use:
        return op, [], [r for o in ops for r in regs(o)]
    if not ops:
        return op, [], []
    d = regs(ops[0])
    u = [r for o in ops[1:] for r in regs(o)]
    if 'dpp' in op or 'dpp' in line or op.startswith(('v_fmac', 'v_writelane', 'v_mac')):
        u += d
    return op, d, u


def main():
    path, name, a, b = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
    lines = open(path).read().split('\n')
    start = [i for i, l in enumerate(lines) if name in l and l.rstrip().endswith(':') or (name in l and l.startswith('_Z') and ':' in l)][0]
    body = [(i, parse(lines[start + i])) for i in range(a, b)]
    body = [(i, p) for i, p in body if p]
    n = len(body)
    live = set()
    pressure = [0] * n
    for _ in range(2):  # cyclic: two backward passes
        for k in range(n - 1, -1, -1):
            _, (op, d, u) = body[k]
            live -= set(d)
            live |= set(u)
            pressure[k] = len(live)
    step = max(1, n // 60)
    for k in range(0, n, step):
        print(body[k][0], pressure[k], body[k][1][0])
    peak = max(range(n), key=lambda k: pressure[k])
    print('peak', pressure[peak], 'at line', body[peak][0])


main()
