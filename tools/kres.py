#!/usr/bin/env python3
"""Per-kernel resources (VGPRs, AGPRs, SGPRs, scratch, LDS, occupancy) of a built libqsmc_hip.so, without a GPU: the device
code object is the ELF with e_machine 224 embedded in the host library; its AMDGPU metadata note lists every kernel.

    python tools/kres.py [lib.so] [name-substring ...]        # table
    python tools/kres.py --diff old.so new.so [substr ...]    # kernels whose numbers differ
"""
import re
import subprocess
import sys
import tempfile

READELF = "/opt/rocm/lib/llvm/bin/llvm-readelf"
DEFAULT = "/root/repo/python-qinfer_amd/qinfer_amd/lib/libqsmc_hip.so"


def code_object(path):
    blob = open(path, "rb").read()
    out = []
    at = 0
    while True:
        at = blob.find(b"\x7fELF", at)
        if at < 0:
            break
        if blob[at + 18:at + 20] == b"\xe0\x00":          # e_machine = 224 (EM_AMDGPU)
            # e_shoff + e_shnum * e_shentsize bounds the object
            shoff = int.from_bytes(blob[at + 40:at + 48], "little")
            shentsize = int.from_bytes(blob[at + 58:at + 60], "little")
            shnum = int.from_bytes(blob[at + 60:at + 62], "little")
            out.append(blob[at:at + shoff + shentsize * shnum])
        at += 4
    return out


def kernels(path):
    res = {}
    for obj in code_object(path):
        with tempfile.NamedTemporaryFile(suffix=".co") as f:
            f.write(obj)
            f.flush()
            txt = subprocess.run([READELF, "--notes", f.name], capture_output=True, text=True).stdout
        for blk in re.split(r"\n\s+- \.", "\n" + txt):
            name = re.search(r"\.name:\s+(\S+)", blk)
            if not name or ".vgpr_count" not in blk:
                continue
            g = lambda k, d=0: int((re.search(r"\.%s:\s+(\d+)" % k, blk) or [None, d])[1])      # noqa: E731
            dem = subprocess.run(["c++filt", name.group(1)], capture_output=True, text=True).stdout.strip()
            dem = re.sub(r"\(.*", "", dem).replace("void ", "")
            v, a = g("vgpr_count"), g("agpr_count")
            tot = v + a if a else v
            occ = max(1, min(8, 512 // max(8, (tot + 7) // 8 * 8)))
            res[dem] = dict(vgpr=v, agpr=a, sgpr=g("sgpr_count"), scratch=g("private_segment_fixed_size"),
                            lds=g("group_segment_fixed_size"), vspill=g("vgpr_spill_count"), sspill=g("sgpr_spill_count"), occ=occ)
    return res


def fmt(n, r):
    return "%-70s v%-3d a%-3d s%-3d scratch %-4d lds %-6d vspill %-3d sspill %-3d occ %d" % (
        n[:70], r["vgpr"], r["agpr"], r["sgpr"], r["scratch"], r["lds"], r["vspill"], r["sspill"], r["occ"])


def main():
    args = sys.argv[1:]
    if args and args[0] == "--diff":
        a, b = kernels(args[1]), kernels(args[2])
        subs = args[3:]
        for n in sorted(set(a) | set(b)):
            if subs and not any(s in n for s in subs):
                continue
            if a.get(n) != b.get(n):
                print("-", fmt(n, a[n]) if n in a else "(absent) " + n)
                print("+", fmt(n, b[n]) if n in b else "(absent) " + n)
        return
    path = args[0] if args and args[0].endswith(".so") else DEFAULT
    subs = [x for x in args if not x.endswith(".so")]
    for n, r in sorted(kernels(path).items()):
        if subs and not any(s in n for s in subs):
            continue
        print(fmt(n, r))


if __name__ == "__main__":
    main()
