"""List the compiler-inserted `s_waitcnt vmcnt(..)` inside the loops of the LDS-DMA kernels (hand-counted waits live in
ASMSTART/ASMEND blocks and are skipped).  A compiler wait inside a DMA ring serialises the ring: the pieces just
issued are waited for on the spot.  Usage: hipcc --offload-arch=gfx950 -O3 -std=c++17 --cuda-device-only -S f.hip -o f.s;
python tools/isa_waits.py f.s [kernel-substring]"""
import re, sys
lines = open(sys.argv[1]).read().split("\n")
want = sys.argv[2] if len(sys.argv) > 2 else ""
kern, start = None, 0
kernels = []
for i, l in enumerate(lines):
    m = re.match(r"^(_Z\w+):", l)
    if m:
        if kern: kernels.append((kern, start, i))
        kern, start = m.group(1), i
if kern: kernels.append((kern, start, len(lines)))
for kern, a, b in kernels:
    if want not in kern: continue
    body = lines[a:b]
    if not any("global_load_lds" in x for x in body): continue
    labels = {}
    for i, l in enumerate(body):
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m: labels[m.group(1)] = i
    loops = []
    for i, l in enumerate(body):
        m = re.search(r"s_cbranch_\w+\s+(\.LBB\d+_\d+)|s_branch\s+(\.LBB\d+_\d+)", l)
        if m:
            t = m.group(1) or m.group(2)
            if t in labels and labels[t] < i: loops.append((labels[t], i))
    in_asm = False
    print(f"== {kern[:70]}  loops: {[(x, y, sum('v_mfma' in z for z in body[x:y])) for x, y in loops]}")
    for i, l in enumerate(body):
        if "#ASMSTART" in l: in_asm = True
        if "#ASMEND" in l: in_asm = False
        if "s_waitcnt" in l and "vmcnt" in l and not in_asm:
            inside = [(x, y) for x, y in loops if x <= i <= y and any("global_load_lds" in z for z in body[x:y])]
            if inside:
                prev = [z.strip() for z in body[max(0, i - 3):i]]
                nxt = [z.strip() for z in body[i + 1:i + 3]]
                print(f"   line {i}: {l.strip()}   in loop {inside[-1]}   before: {prev[-1][:50]} | after: {nxt[0][:60]}")
