#!/usr/bin/env python3
"""Instruction mix of one kernel in a device .s file (development aid): static counts per basic block, so that the hot loops of a
kernel can be priced in wavefront-instructions before a GPU visit.  usage: isa_count.py <file.s> <kernel name substring> [-b]"""
import re, sys, collections
path, pat = sys.argv[1], sys.argv[2]
per_block = "-b" in sys.argv
cur, blocks, name = None, None, None
kern = {}
for line in open(path):
    m = re.match(r"^(_Z\w+):", line)
    if m:
        name = m.group(1); cur = "entry"; blocks = kern.setdefault(name, collections.OrderedDict()); blocks[cur] = []
        continue
    if name is None: continue
    if line.startswith(".Lfunc_end"): name = None; continue
    m = re.match(r"^(\.LBB\w+):", line)
    if m: cur = m.group(1); blocks[cur] = []; continue
    m = re.match(r"^\s+([a-z_0-9]+)\s", line)
    if m and not line.strip().startswith("."): blocks[cur].append(m.group(1))
def cls(op):
    if op.startswith("v_"): return "valu"
    if op.startswith("s_"): return "salu"
    if op.startswith("ds_"): return "lds"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")): return "vmem"
    return "other"
for k, blocks in kern.items():
    if pat not in k: continue
    tot = collections.Counter()
    for b, ops in blocks.items():
        c = collections.Counter(cls(o) for o in ops)
        tot.update(c)
        if per_block and len(ops) >= 8: print("  %-16s %s" % (b, dict(c)))
    print(k[:120], dict(tot))
