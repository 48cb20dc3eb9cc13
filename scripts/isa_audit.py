#!/usr/bin/env python3
"""Static audit of the gfx950 code of the rti_kernel instantiations: registers, spills, WHERE the scratch instructions sit, and
the instruction mix of the backward / forward sweep (the block with the most matrix instructions).

    python scripts/isa_audit.py [extra hipcc flags]  > profiles/rNN_isa_audit.txt

Compiles csrc/ndp_hip.hip with -save-temps into a temporary directory (nothing is written into the tree; no GPU needed).
"""
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "ndp_nmpc_qd_amd", "csrc")
KERNELS = [   # (mangled template arguments, what it is)
    ("ILi3ELi4ELb1ELi20ELi0ELi1ELi0EE", "N = 20, 1 RTI iteration, fused downwash, automatic QP mode (the headline launch)"),
    ("ILi3ELi4ELb0ELi20ELi0ELi1ELi0EE", "N = 20, 1 RTI iteration, no downwash"),
    ("ILi3ELi4ELb1ELi20ELi0ELi1ELi1EE", "N = 20 work-list producer (fused downwash)"),
    ("ILi3ELi4ELb0ELi20ELi0ELi1ELi2EE", "N = 20 work-list consumer"),
    ("ILi5ELi2ELb0ELi40ELi0ELi2ELi0EE", "N = 40, 2 RTI iterations, in place (config 5 without the work list)"),
    ("ILi5ELi2ELb0ELi40ELi0ELi2ELi1EE", "N = 40, 2 RTI iterations, work-list producer (config 5 default at batch 4096)"),
    ("ILi5ELi2ELb0ELi40ELi0ELi2ELi2EE", "N = 40, 2 RTI iterations, work-list consumer"),
]


def main():
    extra = sys.argv[1:]
    with tempfile.TemporaryDirectory() as td:
        cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-mllvm", "-amdgpu-mfma-vgpr-form", "-fPIC", "-shared",
               "-save-temps", "-I" + CSRC, "-o", os.path.join(td, "x.so"), os.path.join(CSRC, "ndp_hip.hip")] + extra
        subprocess.run(cmd, cwd=td, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        asm = [f for f in os.listdir(td) if f.endswith("gfx950.s")][0]
        text = open(os.path.join(td, asm)).read()
    print("hipcc " + " ".join(cmd[1:8] + extra))
    meta = {}
    for m in re.finditer(r"\.name:\s+(\S+)\n(.*?)\.wavefront_size", text, re.S):
        def g(k, body=m.group(2)):
            mm = re.search(k + r":\s+(\d+)", body)
            return int(mm.group(1)) if mm else None
        meta[m.group(1)] = dict(vgpr=g(r"\.vgpr_count"), spill=g(r"\.vgpr_spill_count"), scratch=g(r"\.private_segment_fixed_size"))
    lines = text.split("\n")
    for targs, what in KERNELS:
        name = "_ZN3ndp10rti_kernel" + targs + "EvNS_8KernArgsE"
        try:
            a = lines.index(next(ln for ln in lines if ln.startswith(name + ":")))
        except StopIteration:
            print(f"\n== {what}: not in this build")
            continue
        b = next(i for i in range(a, len(lines)) if lines[i].startswith(".Lfunc_end"))
        md = meta.get(name, {})
        print(f"\n== {what}\n   {name}\n   VGPRs (arch + acc) {md.get('vgpr')}, spilled {md.get('spill')}, scratch {md.get('scratch')} B per lane, "
              f"{b - a} lines")
        blocks, cur = collections.OrderedDict(), "entry"
        blocks[cur] = []
        for ln in lines[a + 1:b]:
            s = ln.strip()
            lab = re.match(r"(\.LBB\w+):", s)
            if lab:
                cur = lab.group(1)
                blocks[cur] = []
            elif s and not s.startswith((";", ".")):
                blocks[cur].append(s.split()[0])
        tot_l = sum(sum(1 for o in ops if o.startswith("scratch_load")) for ops in blocks.values())
        tot_s = sum(sum(1 for o in ops if o.startswith("scratch_store")) for ops in blocks.values())
        print(f"   scratch instructions in the code: {tot_l} loads, {tot_s} stores; by basic block (blocks holding matrix instructions = the sweeps):")
        with_m = [(k, ops) for k, ops in blocks.items() if any(o.startswith("v_mfma") for o in ops)]
        in_sweeps_l = sum(sum(1 for o in ops if o.startswith("scratch_load")) for _, ops in with_m)
        in_sweeps_s = sum(sum(1 for o in ops if o.startswith("scratch_store")) for _, ops in with_m)
        print(f"     inside blocks with matrix instructions: {in_sweeps_l} loads, {in_sweeps_s} stores "
              f"({len(with_m)} such blocks, {sum(len(o) for _, o in with_m)} instructions)")
        outside = [(k, sum(1 for o in ops if o.startswith("scratch_load")), sum(1 for o in ops if o.startswith("scratch_store")), len(ops))
                   for k, ops in blocks.items() if not any(o.startswith("v_mfma") for o in ops)]
        outside = [x for x in outside if x[1] + x[2] >= 8]
        for k, l, s, n in outside:
            print(f"     {k}: {l} loads, {s} stores in {n} instructions (no matrix instruction: set-up / linearise / interior-point bookkeeping)")
        # instruction mix of the block with the most matrix instructions (the unrolled first sweep of a compile-time horizon)
        k, ops = max(blocks.items(), key=lambda kv: sum(1 for o in kv[1] if o.startswith("v_mfma")))
        c = collections.Counter(ops)
        nm16 = sum(v for o, v in c.items() if o.startswith("v_mfma_f64_16x16x4"))
        nm4 = sum(v for o, v in c.items() if o.startswith("v_mfma_f64_4x4x4"))
        nmo = sum(v for o, v in c.items() if o.startswith("v_mfma")) - nm16 - nm4
        valu = sum(v for o, v in c.items() if o.startswith("v_") and not o.startswith("v_mfma"))
        f64 = sum(v for o, v in c.items() if o.startswith("v_") and "f64" in o and not o.startswith("v_mfma"))
        print(f"   largest sweep block {k}: {len(ops)} instructions = {nm16} v_mfma_f64_16x16x4 + {nm4} v_mfma_f64_4x4x4 + {nmo} other matrix, "
              f"{valu} VALU ({f64} f64), {sum(v for o, v in c.items() if o.startswith('ds_'))} LDS, {c.get('s_nop', 0)} s_nop, "
              f"{c.get('s_waitcnt', 0)} s_waitcnt, {sum(v for o, v in c.items() if o.startswith('v_accvgpr'))} accvgpr moves")
        top = ", ".join(f"{o} {v}" for o, v in c.most_common(12))
        print(f"     most frequent: {top}")


if __name__ == "__main__":
    main()
