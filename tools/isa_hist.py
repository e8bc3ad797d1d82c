#!/usr/bin/env python3
"""Instruction-class histogram of one kernel in a hipcc -S listing: tools/isa_hist.py file.s kernel_substring [loop]
With 'loop', restricts to the largest backward-branch loop body."""
import collections
import re
import sys


def classify(op):
    if op.startswith("v_mfma"): return "mfma"
    if op.startswith("ds_"): return "lds"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")): return "vmem"
    if op.startswith("s_waitcnt"): return "waitcnt"
    if op.startswith("s_barrier"): return "barrier"
    if op.startswith("s_"): return "salu"
    if op.startswith(("v_exp", "v_log", "v_rcp", "v_rsq", "v_sqrt", "v_sin", "v_cos")): return "trans"
    if op.startswith("v_"): return "valu"
    return "other"


def main():
    path, key = sys.argv[1], sys.argv[2]
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if re.match(r"^[A-Za-z_][\w$]*:", l) and key in l.split(":")[0])
    end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
    body = lines[start:end + 1]
    if len(sys.argv) > 3:
        labels = {l.split(":")[0]: i for i, l in enumerate(body) if re.match(r"^\.LBB\d+_\d+:", l)}
        best = (0, 0, 0)
        for i, l in enumerate(body):
            m = re.search(r"s_cbranch_\w+\s+(\.LBB\d+_\d+)", l) or re.search(r"s_branch\s+(\.LBB\d+_\d+)", l)
            if m and m.group(1) in labels and labels[m.group(1)] < i and i - labels[m.group(1)] > best[0]:
                best = (i - labels[m.group(1)], labels[m.group(1)], i)
        body = body[best[1]:best[2] + 1]
        print("loop body lines", best)
    h = collections.Counter()
    ops = collections.Counter()
    for l in body:
        l = l.strip()
        if not l or l.startswith((".", ";", "//")) or l.endswith(":"):
            continue
        op = l.split()[0]
        h[classify(op)] += 1
        ops[op] += 1
    print(dict(h), "total", sum(h.values()))
    print(ops.most_common(40))


if __name__ == "__main__":
    main()
