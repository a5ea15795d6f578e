#!/usr/bin/env python3
"""Developer tool: CFG-aware check of a gfx950 assembly listing (hipcc -S --cuda-device-only) for VGPR reads / overwrites that no
s_waitcnt covers on SOME path.

    python tools/slp_hazard/waitcheck.py dev.s [kernel-name-substring]

Per kernel: basic blocks from labels and s_branch / s_cbranch_*; forward dataflow to a fixed point.  State per program point:
for every VGPR that is the destination of a possibly outstanding load, the SMALLEST number of younger operations of its counter
over all paths (vmcnt: global / buffer / scratch loads, stores and atomics retire in issue order, so `s_waitcnt vmcnt(N)` completes
every operation with at least N younger ones; lgkmcnt: DS operations likewise, while a scalar load is outstanding only lgkmcnt(0)
is trusted).  Any instruction that reads, or overwrites, a VGPR still in that state is reported with the load it races.
Built to find the missing wait behind the non-deterministic SLP build of tail_bf16.hip."""
import re
import sys
from collections import defaultdict

path = sys.argv[1]
want = sys.argv[2] if len(sys.argv) > 2 else ""
REG = re.compile(r"\bv(\d+)\b|\bv\[(\d+):(\d+)\]")
AREG = re.compile(r"\ba(\d+)\b|\ba\[(\d+):(\d+)\]")


def regs(tok):
    out = set()
    for rx, base in ((REG, 0), (AREG, 1000)):
        for m in rx.finditer(tok):
            if m.group(1) is not None:
                out.add(base + int(m.group(1)))
            else:
                out.update(range(base + int(m.group(2)), base + int(m.group(3)) + 1))
    return out


def split_ops(rest):
    ops, depth, cur = [], 0, ""
    for ch in rest:
        depth += ch == "["
        depth -= ch == "]"
        if ch == "," and depth == 0:
            ops.append(cur.strip()); cur = ""
        else:
            cur += ch
    if cur.strip():
        ops.append(cur.strip())
    return ops


def parse(line):
    """-> (kind, dst regs, src regs, wait (vm, lgkm) or None)"""
    parts = line.split(None, 1)
    op = parts[0]
    rest = parts[1] if len(parts) > 1 else ""
    ops = split_ops(rest)
    if op == "s_waitcnt":
        vm = re.search(r"vmcnt\((\d+)\)", rest)
        lg = re.search(r"lgkmcnt\((\d+)\)", rest)
        if re.fullmatch(r"\s*(0|0x0)\s*", rest):
            return "wait", set(), set(), (0, 0)
        return "wait", set(), set(), (int(vm.group(1)) if vm else None, int(lg.group(1)) if lg else None)
    vload = op.startswith(("global_load", "buffer_load", "scratch_load", "flat_load"))
    vstore = op.startswith(("global_store", "buffer_store", "scratch_store", "flat_store", "global_atomic", "buffer_atomic", "flat_atomic"))
    dsread = op.startswith(("ds_read", "ds_bpermute", "ds_permute", "ds_swizzle", "ds_consume", "ds_append"))
    dswrite = op.startswith("ds_") and not dsread
    dst, src = set(), set()
    if vstore or dswrite:
        for o in ops:
            src |= regs(o)
        return ("vm" if vstore else "ds"), set(), src, None
    if op.startswith("s_load") or op.startswith("s_buffer_load"):
        return "smem", set(), set(), None
    if op.startswith(("s_", "v_cmp_", "v_cmpx_")) and not op.startswith("s_nop"):
        for o in ops:
            src |= regs(o)
        return "other", set(), src, None
    if op in ("v_readfirstlane_b32", "v_readlane_b32"):
        for o in ops[1:]:
            src |= regs(o)
        return "other", set(), src, None
    if ops:
        dst = regs(ops[0])
        for o in ops[1:]:
            src |= regs(o)
        if "fmac" in op or "_mac_" in op or op.startswith("v_dot2c") or op == "v_writelane_b32" or op.startswith("v_cndmask") and False:
            src |= dst
    if vload:
        if "lds" in rest.split():
            return "vm", set(), src, None
        return "vmload", dst, src, None
    if dsread:
        return "dsload", dst, src, None
    return "other", dst, src, None


def kernels(path):
    name, body = None, []
    for ln, raw in enumerate(open(path), 1):
        line = raw.split(";")[0].strip()
        if not line:
            continue
        if line.endswith(":") and not line.startswith(".L") and not line.startswith("."):
            if name and body:
                yield name, body
            name, body = line[:-1], []
            continue
        if name is None:
            continue
        if line.startswith(".") and not line.startswith(".L"):
            continue
        body.append((ln, line))
        if line.startswith("s_endpgm"):
            yield name, body
            name, body = None, []


def analyse(name, body):
    # basic blocks
    starts = {0}
    label_at = {}
    for i, (ln, line) in enumerate(body):
        if line.endswith(":"):
            label_at[line[:-1]] = i
            starts.add(i)
        elif line.startswith(("s_branch", "s_cbranch", "s_endpgm", "s_setpc")):
            starts.add(i + 1)
    order = sorted(s for s in starts if s < len(body))
    bid = {s: k for k, s in enumerate(order)}
    succ = defaultdict(list)
    for k, s in enumerate(order):
        e = order[k + 1] if k + 1 < len(order) else len(body)
        last = body[e - 1][1]
        if last.startswith("s_branch"):
            t = last.split()[1]
            if t in label_at:
                succ[k].append(bid[label_at[t]])
        elif last.startswith("s_cbranch"):
            t = last.split()[1]
            if t in label_at:
                succ[k].append(bid[label_at[t]])
            if e < len(body):
                succ[k].append(bid[e])
        elif last.startswith(("s_endpgm", "s_setpc")):
            pass
        elif e < len(body):
            succ[k].append(bid[e])
    # state: (vm: {reg: (k, line)}, ds: {reg: (k, line)}, smem: bool)
    instate = {0: ({}, {}, False)}
    work = [0]
    reports = {}

    def merge(a, b):
        vm, ds = dict(a[0]), dict(a[1])
        ch = False
        for dst_, src_ in ((vm, b[0]), (ds, b[1])):
            for r, (k, l) in src_.items():
                if r not in dst_ or dst_[r][0] > k:
                    dst_[r] = (k, l); ch = True
        sm = a[2] or b[2]
        return (vm, ds, sm), ch or sm != a[2]

    it = 0
    while work:
        k = work.pop()
        it += 1
        if it > 20000:
            print("  (fixed point not reached)"); break
        vm, ds, sm = ({r: v for r, v in instate[k][0].items()}, {r: v for r, v in instate[k][1].items()}, instate[k][2])
        s = order[k]
        e = order[k + 1] if k + 1 < len(order) else len(body)
        for i in range(s, e):
            ln, line = body[i]
            if line.endswith(":"):
                continue
            kind, dst, src, wait = parse(line)
            if kind == "wait":
                v, l = wait
                if v is not None:
                    vm = {r: x for r, x in vm.items() if x[0] < v}
                if l is not None:
                    if l == 0:
                        ds, sm = {}, False
                    elif not sm:
                        ds = {r: x for r, x in ds.items() if x[0] < l}
                continue
            for pend, what in ((vm, "vmcnt"), (ds, "lgkmcnt")):
                rd = src & pend.keys()
                wr = (dst & pend.keys()) if kind not in ("vmload", "dsload") else set()
                for r in sorted(rd | wr):
                    key = (ln, pend[r][1])
                    if key not in reports:
                        reports[key] = "line %d: %-70s %s v%d: outstanding %s load of line %d" % (
                            ln, line[:70], "READS" if r in rd else "OVERWRITES", r, what, pend[r][1])
            if kind in ("vm", "vmload"):
                vm = {r: (x[0] + 1, x[1]) for r, x in vm.items()}
                for r in dst:
                    vm[r] = (0, ln)
            elif kind in ("ds", "dsload"):
                ds = {r: (x[0] + 1, x[1]) for r, x in ds.items()}
                for r in dst:
                    ds[r] = (0, ln)
            elif kind == "smem":
                sm = True
            else:
                for r in dst:          # a VALU write ends the register's pending state (already reported if it raced)
                    vm.pop(r, None); ds.pop(r, None)
        out = (vm, ds, sm)
        for t in succ[k]:
            if t not in instate:
                instate[t] = out; work.append(t)
            else:
                m, ch = merge(instate[t], out)
                if ch:
                    instate[t] = m; work.append(t)
    return reports


total = 0
for name, body in kernels(path):
    if want not in name:
        continue
    rep = analyse(name, body)
    print("%s: %d instructions, %d reports" % (name[:90], len(body), len(rep)))
    for key in sorted(rep)[:30]:
        print("   " + rep[key])
    total += len(rep)
print("total reports:", total)
