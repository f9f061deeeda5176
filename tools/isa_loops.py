"""ISA audit of streaming loops: for every loop (a backward branch) of the named kernels in a gfx950 assembly file, count
the buffer/global loads, LDS ops, scratch (spill) accesses and `s_waitcnt vmcnt(0)` inside the loop body.  A scratch reload
inside a streaming loop is followed by vmcnt(0), which drains the loads in flight.

  hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -S --cuda-device-only -I xclim_amd/csrc x.hip -o x.s
  python tools/isa_loops.py x.s k_hs_fused
"""
import re
import sys


def main():
    path, pat = sys.argv[1], sys.argv[2]
    lines = open(path).read().split("\n")
    # kernel bodies: from "<name>:" to ".Lfunc_end"
    i = 0
    while i < len(lines):
        m = re.match(r"^(_Z\w+):", lines[i])
        if m and pat in m.group(1):
            name = m.group(1)
            j = i
            while not lines[j].startswith(".Lfunc_end"):
                j += 1
            audit(name, lines[i:j])
            i = j
        i += 1


def audit(name, body):
    print("==", name, len(body), "lines")
    label_at = {}
    for k, l in enumerate(body):
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m:
            label_at[m.group(1)] = k
    loops = []
    for k, l in enumerate(body):
        m = re.search(r"\bs_cbranch_\w+\s+(\.LBB\d+_\d+)|\bs_branch\s+(\.LBB\d+_\d+)", l)
        if m:
            tgt = m.group(1) or m.group(2)
            if tgt in label_at and label_at[tgt] <= k:
                loops.append((label_at[tgt], k, tgt))
    for a, b, tgt in sorted(loops):
        seg = body[a:b + 1]
        cnt = lambda rx: sum(1 for l in seg if re.search(rx, l))
        nload = cnt(r"\b(buffer_load|global_load)")
        if nload == 0 and cnt(r"\bscratch_") == 0:
            continue
        c = dict(ld=nload, st=cnt(r"(buffer|global)_store"), ds=cnt(r"\bds_"), sl=cnt(r"scratch_load"), ss=cnt(r"scratch_store"),
                 w0=cnt(r"vmcnt\(0\)"), valu=cnt(r"^\s+v_"))
        print("  loop %-12s lines %6d-%6d (%5d instr)  vmem loads %3d  stores %3d  ds %4d  scratch ld/st %3d/%3d  vmcnt(0) %2d  valu %5d"
              % (tgt, a, b, b - a, c["ld"], c["st"], c["ds"], c["sl"], c["ss"], c["w0"], c["valu"]))


main()
