"""Loops of one kernel in a `hipcc -S --cuda-device-only` listing, from its control-flow graph (basic blocks at labels and behind
branches; loops = strongly connected components, nested ones found by removing a loop's entry block): for each loop its static size
and instruction mix -- the hot loops of the land kernel are where an instruction counts several times per column-step.

usage: isa_loops.py listing.s [kernel-name-substring] [min VALU instructions]"""
import collections
import re
import sys

sys.setrecursionlimit(100000)
from isa_stats import kernel_body  # noqa: E402


def classify(op):
    if not op.startswith("v_"):
        return None
    if "f64" in op:
        return "f64"
    if op.startswith(("v_mov", "v_accvgpr")):
        return "mov"
    if op.startswith("v_cndmask"):
        return "cndmask"
    if op.startswith("v_cmp"):
        return "cmp"
    if op.startswith("v_div_"):
        return "div"
    if re.match(r"v_(rcp|sqrt|rsq|exp|log)_", op):
        return "trans"
    if op.startswith(("v_readlane", "v_writelane", "v_readfirstlane")):
        return "lane"
    if op.startswith("v_cvt"):
        return "cvt"
    if op.startswith("v_pk_"):
        return "pk"
    if re.match(r"v_(fma|mul|add|sub|subrev|mac|fmac|max|min)_f32", op):
        return "f32"
    return "int"


def blocks_of(body):
    blocks, cur, label_of = [], {"ins": [], "labels": []}, {}
    for ln in body:
        t = ln.strip()
        if not t or t.startswith((";", "//")):
            continue
        m = re.match(r"^(\.LBB\d+_\d+):", t)
        if m:
            if cur["ins"]:
                blocks.append(cur)
                cur = {"ins": [], "labels": []}
            cur["labels"].append(m.group(1))
            continue
        if t.startswith(".") or t.endswith(":"):
            continue
        cur["ins"].append(t)
        if t.startswith(("s_branch", "s_cbranch", "s_endpgm", "s_setpc")):
            blocks.append(cur)
            cur = {"ins": [], "labels": []}
    if cur["ins"]:
        blocks.append(cur)
    for i, b in enumerate(blocks):
        for lb in b["labels"]:
            label_of[lb] = i
    succ = []
    for i, b in enumerate(blocks):
        s, last = [], b["ins"][-1] if b["ins"] else ""
        m = re.match(r"s_c?branch\w*\s+(\.LBB\d+_\d+)", last)
        if m and m.group(1) in label_of:
            s.append(label_of[m.group(1)])
        if not last.startswith(("s_branch", "s_endpgm", "s_setpc")) and i + 1 < len(blocks):
            s.append(i + 1)
        succ.append(s)
    return blocks, succ


def sccs(nodes, succ):
    nodes = set(nodes)
    index, low, stack, on, out, n = {}, {}, [], set(), [], [0]

    def visit(v):
        index[v] = low[v] = n[0]; n[0] += 1
        stack.append(v); on.add(v)
        for w in succ[v]:
            if w not in nodes:
                continue
            if w not in index:
                visit(w); low[v] = min(low[v], low[w])
            elif w in on:
                low[v] = min(low[v], index[w])
        if low[v] == index[v]:
            comp = []
            while True:
                w = stack.pop(); on.discard(w); comp.append(w)
                if w == v:
                    break
            if len(comp) > 1 or v in succ[v]:
                out.append(comp)
    for v in sorted(nodes):
        if v not in index:
            visit(v)
    return out


def report(blocks, succ, comp, depth, minv, pred):
    ops = [t.split()[0] for b in sorted(comp) for t in blocks[b]["ins"]]
    cls = collections.Counter(c for c in map(classify, ops) if c)
    valu = sum(cls.values())
    if valu >= minv:
        salu = sum(1 for o in ops if o.startswith("s_") and not o.startswith(("s_waitcnt", "s_nop")))
        print("%sloop @block %d (%d blocks): VALU %5d  SALU %4d  s_nop %3d  waitcnt %3d  ds %3d  vmem %3d | %s" % (
            "    " * depth, min(comp), len(comp), valu, salu, ops.count("s_nop"), ops.count("s_waitcnt"), sum(1 for o in ops if o.startswith("ds_")),
            sum(1 for o in ops if o.startswith(("global_", "buffer_", "scratch_", "flat_"))), " ".join("%s %d" % kv for kv in cls.most_common())))
    cs = set(comp)
    entries = [v for v in comp if any(p not in cs for p in pred[v])] or [min(comp)]
    inner = cs - set(entries)
    for sub in sorted(sccs(inner, succ), key=min):
        report(blocks, succ, sub, depth + 1, minv, pred)


def main():
    path = sys.argv[1]
    key = sys.argv[2] if len(sys.argv) > 2 else "ELb1ELi1EE"
    minv = int(sys.argv[3]) if len(sys.argv) > 3 else 40
    name, body = kernel_body(open(path).read().splitlines(), key)
    blocks, succ = blocks_of(body)
    pred = collections.defaultdict(list)
    for i, s in enumerate(succ):
        for w in s:
            pred[w].append(i)
    print(name, "basic blocks", len(blocks))
    for comp in sorted(sccs(range(len(blocks)), succ), key=min):
        report(blocks, succ, comp, 0, minv, pred)


if __name__ == "__main__":
    main()
