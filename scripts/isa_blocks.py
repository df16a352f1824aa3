#!/usr/bin/env python
"""Static instruction mix of one kernel of a hipcc -S listing, basic block by basic block.
    python scripts/isa_blocks.py /tmp/k.s _Z11k_pso_eval2ILi2ELb0ELb0EE
Columns: block label, line, instructions, VALU, of which FP64-rate (v_*_f64, conversions to/from f64), SALU, LDS, VMEM, branch target."""
import re, sys
path, sym = sys.argv[1], sys.argv[2]
lines = open(path).read().split("\n")
start = next(i for i, l in enumerate(lines) if l.startswith(sym) and l.rstrip().split(":")[0].startswith(sym) and ":" in l)
end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith(".Lfunc_end"))
blocks = []; cur = ["entry", start, []]
for i in range(start + 1, end):
    l = lines[i].strip()
    m = re.match(r"^(\.LBB\d+_\d+):", l)
    if m:
        blocks.append(cur); cur = [m.group(1), i, []]; continue
    if not l or l.startswith(";") or l.startswith("."): continue
    cur[2].append(l.split(";")[0].strip())
blocks.append(cur)
tot = [0] * 6
print("%-12s %7s %5s %5s %5s %5s %4s %5s  %s" % ("block", "line", "n", "valu", "f64", "salu", "lds", "vmem", "branches"))
for name, ln, ins in blocks:
    valu = [x for x in ins if x.startswith("v_")]
    f64 = [x for x in valu if re.match(r"v_\w*f64", x.split()[0]) or "cvt" in x.split()[0] and "f64" in x.split()[0]]
    salu = [x for x in ins if x.startswith("s_") and not x.startswith("s_waitcnt") and not x.startswith("s_cbranch") and not x.startswith("s_branch") and not x.startswith("s_nop")]
    lds = [x for x in ins if x.startswith("ds_")]
    vmem = [x for x in ins if x.startswith("global_") or x.startswith("buffer_") or x.startswith("flat_") or x.startswith("scratch_")]
    br = [x.split()[-1] for x in ins if x.startswith("s_cbranch") or x.startswith("s_branch")]
    print("%-12s %7d %5d %5d %5d %5d %4d %5d  %s" % (name, ln + 1, len(ins), len(valu), len(f64), len(salu), len(lds), len(vmem), " ".join(br)))
    for k, v in enumerate((len(ins), len(valu), len(f64), len(salu), len(lds), len(vmem))): tot[k] += v
print("total        %7s %5d %5d %5d %5d %4d %5d" % ("", *tot))
