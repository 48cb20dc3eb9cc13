#!/usr/bin/env python3
"""Static audit of the gfx950 code of the rti_kernel instantiations: registers, spills, WHERE the scratch instructions sit, and
the instruction mix of the backward / forward sweep (the block with the most matrix instructions).

    python scripts/isa_audit.py [extra hipcc flags]  > profiles/rNN_isa_audit.txt

Compiles csrc/ndp_hip.hip to device assembly (--cuda-device-only -S) in a temporary directory (nothing is written into the tree; no
GPU needed).  (-save-temps, used before, round-trips the module through bitcode, which this compiler's reader rejects for the
current source: "Invalid cast".)
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
        cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-mllvm", "-amdgpu-mfma-vgpr-form", "-fPIC",
               "--cuda-device-only", "-S", "-I" + CSRC, "-o", os.path.join(td, "x.s"), os.path.join(CSRC, "ndp_hip.hip")] + extra
        subprocess.run(cmd, cwd=td, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        text = open(os.path.join(td, "x.s")).read()
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
        full = {cur: []}
        for ln in lines[a + 1:b]:
            s = ln.strip()
            lab = re.match(r"(\.LBB\w+):", s)
            if lab:
                cur = lab.group(1)
                blocks[cur] = []
                full[cur] = []
            elif s and not s.startswith((";", ".")):
                blocks[cur].append(s.split()[0])
                full[cur].append(s)
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
        # (the cofactor-path sweep: one v_rcp_f64 per stage; the interior-point loop's LDL' sweeps hold four and are left aside)
        cand = {kk: oo for kk, oo in blocks.items() if sum(1 for o in oo if o.startswith("v_rcp")) <= max(1, sum(1 for o in oo if o.startswith("v_mfma_f64_16x16x4")) // 7 + 2)}
        k, ops = max(cand.items(), key=lambda kv: sum(1 for o in kv[1] if o.startswith("v_mfma")))
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
        classify_nops(full[k])


def _regs(tok):
    """register numbers named by an operand like v[16:23], v7, -v[2:3]"""
    m = re.match(r"-?\|?v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r"-?\|?v(\d+)", tok)
    return {int(m.group(1))} if m else set()


def classify_nops(lines):
    """Every s_nop of a block by the hazard it pads: what the instruction in front of it produced and what the instruction behind it
    consumes (VERDICT r3 #6b).  Wait states = sum of (N + 1) over s_nop N."""
    import collections
    ins = [ln.split(";")[0].strip() for ln in lines]
    ins = [x for x in ins if x]
    cls = collections.Counter()
    cyc = collections.Counter()
    i = 0
    while i < len(ins):
        if not ins[i].startswith("s_nop"):
            i += 1
            continue
        j, n = i, 0
        while j < len(ins) and ins[j].startswith("s_nop"):
            n += int(ins[j].split()[1]) + 1
            j += 1
        prev = next((ins[p] for p in range(i - 1, -1, -1) if not ins[p].startswith(("s_waitcnt", "s_nop", ";"))), "")
        nxt = ins[j] if j < len(ins) else ""
        # the producer the pad is for: the nearest matrix / transcendental / VALU instruction in front whose result the next one reads
        prod = ""
        use = set()
        for tok in nxt.replace(",", " ").split()[1:]:
            use |= _regs(tok)
        for p in range(i - 1, max(-1, i - 12), -1):
            ops = ins[p].replace(",", " ").split()
            if len(ops) > 1 and (_regs(ops[1]) & use):
                prod = ops[0]
                break
        prod = prod or prev.split()[0] if prev else "?"
        cons = nxt.split()[0] if nxt else "?"
        if prod.startswith("v_mfma") and cons.startswith("v_mfma"):
            key = "matrix result -> matrix instruction's A / B operand (dependent chain)"
        elif prod.startswith("v_mfma") and cons.startswith("ds_"):
            key = "matrix result -> LDS store"
        elif prod.startswith("v_mfma"):
            key = "matrix result -> vector instruction"
        elif "dpp" in nxt:
            key = "vector write -> DPP read"
        elif prod.startswith(("v_rcp", "v_rsq", "v_sqrt")):
            key = "transcendental result -> use"
        elif cons.startswith("v_mfma"):
            key = "vector write -> matrix instruction operand"
        elif cons.startswith(("v_readlane", "v_readfirstlane", "s_")) or prod.startswith("v_cmp"):
            key = "vector -> scalar / lane read"
        else:
            key = f"other ({prod} -> {cons})"
        cls[key] += 1
        cyc[key] += n
        i = j
    tot = sum(cyc.values())
    print(f"   s_nop pads of that block by hazard ({sum(cls.values())} pads, {tot} wait states):")
    for k2, v in sorted(cyc.items(), key=lambda kv: -kv[1]):
        print(f"     {v:5d} wait states in {cls[k2]:4d} pads: {k2}")


if __name__ == "__main__":
    main()
    # the shipped library's properties the ISA tests assert (tests/test_isa_properties.py)
    sys.path.insert(0, ROOT)
    from ndp_nmpc_qd_amd import build, isa_inspect as I
    co = I.CodeObject(build.build())
    kk = co.kernels()
    sym = [n for n in kk if "mlp_stream_kernel" in n][0]
    print("\n== shipped library (ndp_nmpc_qd_amd/libndp_nmpc_hip.so)")
    print("   mlp_stream_kernel epoch store:", I.epoch_store_is_ordered(co.disassemble(sym))[1])
    print("   rti_kernel instantiations: %d, with scratch: %d" % (sum("rti_kernel" in n for n in kk), sum(1 for n, v in kk.items() if "rti_kernel" in n and v["scratch"])))
    for n, v in kk.items():
        if "rti_kernel" in n:
            print(f"     {n[19:53]:36s} vgpr {v['vgpr']:3d} (acc {v['agpr']:3d}) spilled {v['spill']:3d} scratch {v['scratch']:4d} B")
