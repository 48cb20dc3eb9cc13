"""Looks INTO the built libndp_nmpc_hip.so: the gfx950 code object it carries, the kernels' resource metadata and their ISA.

Used by tests/test_isa_properties.py (no GPU needed) and scripts/isa_audit.py: properties of the shipped binary that no run-time
test can see -- scratch (spill) bytes per lane of every rti_kernel instantiation, and that the epoch word of the downwash-ahead
kernel is stored only after the force rows have COMPLETED (s_waitcnt vmcnt(0)).
"""
import os
import re
import struct
import subprocess
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def device_code_object(so_path, arch="gfx950"):
    """The bytes of the `arch` code object inside the clang offload bundle of a HIP shared library."""
    data = open(so_path, "rb").read()
    i = data.find(MAGIC)
    if i < 0:
        raise RuntimeError(f"{so_path}: no clang offload bundle")
    n = struct.unpack_from("<Q", data, i + len(MAGIC))[0]
    off = i + len(MAGIC) + 8
    for _ in range(n):
        o, sz, tl = struct.unpack_from("<QQQ", data, off)
        off += 24
        triple = data[off:off + tl].decode()
        off += tl
        if arch in triple and sz:
            return data[i + o:i + o + sz]
    raise RuntimeError(f"{so_path}: no {arch} code object in the bundle")


class CodeObject:
    def __init__(self, so_path, arch="gfx950"):
        self._td = tempfile.TemporaryDirectory()
        self.path = os.path.join(self._td.name, "dev.co")
        with open(self.path, "wb") as fh:
            fh.write(device_code_object(so_path, arch))
        self._meta = None

    def kernels(self):
        """{mangled name: {vgpr, agpr, sgpr, spill, scratch, lds}} from the code object's metadata note."""
        if self._meta is None:
            txt = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", self.path], check=True, capture_output=True, text=True).stdout
            meta = {}
            for blk in re.split(r"\n\s+- \.agpr_count:", txt)[1:]:
                blk = ".agpr_count:" + blk

                def g(k, b=blk):
                    m = re.search(r"\." + k + r":\s+(\S+)", b)
                    return m.group(1) if m else None
                name = g("name")
                if name is None:
                    continue
                meta[name] = dict(vgpr=int(g("vgpr_count")), agpr=int(g("agpr_count")), sgpr=int(g("sgpr_count")),
                                  spill=int(g("vgpr_spill_count")), sgpr_spill=int(g("sgpr_spill_count")),
                                  scratch=int(g("private_segment_fixed_size")), lds=int(g("group_segment_fixed_size")))
            self._meta = meta
        return self._meta

    def disassemble(self, symbol):
        """Instruction lines (mnemonic + operands, comments stripped) of one kernel."""
        txt = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", "--no-show-raw-insn", f"--disassemble-symbols={symbol}", self.path],
                             check=True, capture_output=True, text=True).stdout
        out = []
        for ln in txt.split("\n"):
            s = ln.split("//")[0].strip()
            if not s or s.endswith(":") or s.startswith(("Disassembly", self.path, "/")) or "file format" in s:
                continue
            out.append(s)
        return out


def rti_kernel_name(nslot, waves, fused, nc=0, prec=0, nrc=None, qmode=0, tick=False):
    nrc = (1 if nc else 0) if nrc is None else nrc
    return (f"_ZN3ndp10rti_kernelILi{nslot}ELi{waves}ELb{1 if fused else 0}ELi{nc}ELi{prec}ELi{nrc}ELi{qmode}ELb{1 if tick else 0}EEEvNS_8KernArgsE")


def epoch_store_is_ordered(insns):
    """mlp_stream_kernel (prefetch form): every epoch store (the kernel's only 8-byte global store) must be preceded -- after the
    last force-row store in front of it -- by `s_waitcnt vmcnt(0)` (possibly combined with other counters).  Returns (ok, detail)."""
    ep = [i for i, s in enumerate(insns) if s.startswith("global_store_dwordx2")]
    if not ep:
        return False, "no epoch store (global_store_dwordx2) found"
    for e in ep:
        rows = [i for i in range(e) if insns[i].startswith("global_store_dword ") or insns[i].startswith("global_store_dword\t")]
        if not rows:
            return False, "no row store in front of the epoch store"
        between = insns[rows[-1] + 1:e]
        if not any(s.startswith("s_waitcnt") and re.search(r"vmcnt\(0\)", s) for s in between):
            return False, "no s_waitcnt vmcnt(0) between the last row store and the epoch store: " + " | ".join(between[-6:])
    return True, f"{len(ep)} epoch store(s), each behind s_waitcnt vmcnt(0)"
