#!/usr/bin/env python3
"""k_raster requests a triangle's record with two hand-written s_load_dwordx16 and waits for it one visit later (scan_wait()); the
compiler does not know that the 32 destination registers are in flight in between.  This check reads the kernel ISA
(hipcc --cuda-device-only -S), builds the control-flow graph of every k_raster variant and propagates the set of scalar registers
with a request outstanding (added by s_load_dwordx16, emptied by `s_waitcnt lgkmcnt(0)`; union at joins) to a fixed point; it fails if
any instruction reads or writes a register while that register is in the set.
usage: check_scan_regs.py kernels_raster.s"""
import re, sys

RNG = re.compile(r'\bs\[(\d+):(\d+)\]')
ONE = re.compile(r'\bs(\d+)\b')
LABEL = re.compile(r'^([.\w$]+):')


def regs_of(text):
    used = set()
    for a, b in RNG.findall(text):
        used.update(range(int(a), int(b) + 1))
    for a in ONE.findall(text):
        used.add(int(a))
    return used


def check_kernel(name, lines, first_line):
    # basic blocks
    blocks, cur, labels = [], {"label": None, "ins": []}, {}
    for off, raw in enumerate(lines):
        t = raw.split(';')[0].strip()
        m = LABEL.match(t)
        if m:
            if cur["ins"] or cur["label"]:
                blocks.append(cur)
            cur = {"label": m.group(1), "ins": []}
            continue
        if not t or t.startswith('.'):
            continue
        cur["ins"].append((first_line + off, t))
        op = t.split()[0]
        if op.startswith('s_cbranch') or op == 's_branch' or op == 's_endpgm' or op.startswith('s_setpc'):
            blocks.append(cur); cur = {"label": None, "ins": []}
    blocks.append(cur)
    for i, b in enumerate(blocks):
        if b["label"]:
            labels[b["label"]] = i
    succ = []
    for i, b in enumerate(blocks):
        s = []
        last = b["ins"][-1][1] if b["ins"] else ""
        op = last.split()[0] if last else ""
        if op == 's_branch':
            s.append(labels[last.split()[1]])
        elif op.startswith('s_cbranch'):
            s.append(labels[last.split()[1]])
            if i + 1 < len(blocks): s.append(i + 1)
        elif op == 's_endpgm' or op.startswith('s_setpc'):
            pass
        elif i + 1 < len(blocks):
            s.append(i + 1)
        succ.append(s)
    inset = [set() for _ in blocks]
    work = [0]
    seen_in = [None] * len(blocks)
    conflicts = {}
    requests = 0
    while work:
        i = work.pop()
        fl = set(inset[i])
        if seen_in[i] is not None and seen_in[i] == fl:
            continue
        seen_in[i] = set(fl)
        for ln, t in blocks[i]["ins"]:
            op = t.split()[0]
            if op == 's_waitcnt' and 'lgkmcnt(0)' in t:
                fl = set()
                continue
            if op == 's_load_dwordx16':
                ops = t[len(op):].split(',')
                dst = regs_of(ops[0])
                hit = regs_of(','.join(ops[1:])) & fl
                if hit: conflicts[ln] = (t, sorted(hit))
                fl |= dst
                requests += 1
                continue
            hit = regs_of(t) & fl
            if hit:
                conflicts[ln] = (t, sorted(hit))
        for j in succ[i]:
            if not fl <= inset[j]:
                inset[j] |= fl
                work.append(j)
            elif seen_in[j] is None:
                work.append(j)
    for ln in sorted(conflicts):
        print(f"{name[:64]}: line {ln}: `{conflicts[ln][0]}` touches s{conflicts[ln][1]} while a request for it is outstanding")
    return len(conflicts)


src = open(sys.argv[1]).read().split('\n')
starts = [(i, re.match(r'^(_ZN\S*k_raster\S*):', l).group(1)) for i, l in enumerate(src) if re.match(r'^(_ZN\S*k_raster\S*):', l)]
bad = 0
for i, name in starts:
    j = i + 1
    while j < len(src) and not src[j].strip().startswith('s_endpgm'):
        j += 1
    # the kernel's text runs to the last s_endpgm before the next function symbol
    k = j
    while k < len(src) and not re.match(r'^_Z\S*:', src[k]) and not src[k].startswith('\t.section'):
        k += 1
    bad += check_kernel(name, src[i + 1:k], i + 2)
print(f"{len(starts)} k_raster variants checked, {bad} conflicts")
sys.exit(1 if bad else 0)
