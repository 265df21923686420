#!/usr/bin/env python3
"""isa_hist.py <file.s> <kernel-name-substring> [--blocks]: mnemonic histogram of one kernel of a hipcc --save-temps listing,
whole body and per basic block (label to label), so that the executed path of a kernel can be counted by hand.
Classes: valu arithmetic (add/mul/fma/sub, packed ones apart), moves/selects, integer/address, LDS, vector memory, scalar, waits."""
import re, sys, collections
src, pat = sys.argv[1], sys.argv[2]
per_block = "--blocks" in sys.argv
lines = open(src).read().split("\n")
start = next(i for i, l in enumerate(lines) if re.match(r"^[A-Za-z_][\w$.]*:", l) and pat in l.split(":")[0])
end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith("s_endpgm"))
def cls(m):
    if m.startswith("v_pk_mov"): return "valu_pk_mov"
    if m.startswith("v_pk_"): return "valu_pk_f32"
    if re.match(r"v_(add|sub|mul|fma|fmac|mac|mad)_f32", m): return "valu_f32"
    if re.match(r"v_(min|max|min3|max3|med3)_f32", m): return "valu_minmax"
    if re.match(r"v_(mov|cndmask|accvgpr|readfirstlane|readlane|writelane|swap)", m): return "valu_move_select"
    if m.startswith("v_cmp") or m.startswith("v_cmpx"): return "valu_cmp"
    if m.startswith("v_"): return "valu_int_other"
    if m.startswith("ds_"): return "lds"
    if m.startswith("buffer_") or m.startswith("global_") or m.startswith("flat_") or m.startswith("scratch_"): return "vmem"
    if m.startswith("s_waitcnt") or m.startswith("s_barrier") or m.startswith("s_nop") or m.startswith("s_sleep"): return "wait_barrier"
    if m.startswith("s_"): return "salu"
    return "other"
tot, totc = collections.Counter(), collections.Counter()
blk, blkc, name = collections.Counter(), collections.Counter(), "entry"
def flush():
    if per_block and sum(blk.values()):
        print("%-14s %5d  " % (name, sum(blk.values())) + " ".join("%s=%d" % kv for kv in sorted(blkc.items())))
for l in lines[start + 1:end + 1]:
    s = l.strip()
    if not s or s.startswith(";") or s.startswith("."):
        if re.match(r"^\.LBB\d+_\d+:", s):
            flush(); blk.clear(); blkc.clear(); name = s.split(":")[0]
        continue
    m = s.split()[0]
    tot[m] += 1; totc[cls(m)] += 1; blk[m] += 1; blkc[cls(m)] += 1
flush()
print("== %s: %d instructions" % (lines[start], sum(tot.values())))
for c, n in sorted(totc.items(), key=lambda kv: -kv[1]): print("  %-18s %6d" % (c, n))
print("  top mnemonics: " + ", ".join("%s %d" % kv for kv in tot.most_common(28)))
