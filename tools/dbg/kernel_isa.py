#!/usr/bin/env python3
"""Cut one kernel out of a tools/dbg/disasm.sh listing: kernel_isa.py /tmp/isa/k_rollout_ahead.s 'iter_ahead_kernel<30, 6, 17, 0, 8, 1, 1>' [step]
Prints the VALU / MFMA / LDS / VMEM instruction counts; with `step`, the instructions between the 12th and the 15th MFMA
(one unrolled rollout step of the tile kernels)."""
import re, sys
src, name = sys.argv[1], sys.argv[2]
lines = open(src).read().splitlines()
start = next(i for i, l in enumerate(lines) if re.match(r"^[0-9a-f]+ <", l) and name in l)
end = next((i for i in range(start + 1, len(lines)) if re.match(r"^[0-9a-f]+ <", lines[i])), len(lines))
body = [l.split("//")[0].rstrip() for l in lines[start + 1:end] if l.strip()]
ins = [l.split()[0] for l in body if l.split()]
cnt = lambda p: sum(1 for i in ins if re.match(p, i))
print(f"{name}: {len(ins)} instructions, VALU {cnt(r'v_(?!mfma)')}, MFMA {cnt(r'v_mfma')}, DS {cnt(r'ds_')}, VMEM {cnt(r'(global|buffer|scratch)_')}, scratch {cnt(r'scratch_')}, s_nop {cnt(r's_nop')}")
if len(sys.argv) > 3:
    n1, n2 = (int(x) for x in sys.argv[4:6]) if len(sys.argv) > 5 else (12, 15)
    m = [i for i, l in enumerate(body) if l.split() and l.split()[0].startswith("v_mfma")]
    seg = body[m[n1]:m[n2] + 1]
    print("\n".join(seg))
    si = [l.split()[0] for l in seg]
    print(f"-- segment: VALU {sum(1 for i in si if re.match(r'v_(?!mfma)', i))}, MFMA {sum(1 for i in si if i.startswith('v_mfma'))}")
