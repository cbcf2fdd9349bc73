#!/usr/bin/env python3
"""Static check of the device code: no DS (LDS) instruction may take a NEGATIVE base address register.

hipcc's loop strength reduction can emit `ds_read_b128 v[..], vB offset:N` with vB = -K + i*stride (a 32-bit negative value
whose sum with the immediate wraps to the intended address).  On gfx950 that form returned wrong data for a quarter of the
wave whenever the workgroup shared its CU with workgroups of another kernel (found with tools/race_probe.py: the 7x7 stems'
weight reads; results were correct when the kernel ran alone).  The stems now read their weights through the scalar cache;
this scan flags the pattern should the compiler produce it elsewhere.
usage: scan_ds_negative_base.py   (compiles every csrc/*.hip to assembly; ~1 min)"""
import glob, os, re, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "temporally-consistent-stereo-matching_amd", "csrc")
FLAGS = ["-O3", "--offload-arch=gfx950", "-std=c++17", "-munsafe-fp-atomics", "-mllvm", "-amdgpu-kernarg-preload-count=16",
         "-mllvm", "-amdgpu-mfma-vgpr-form", "-S", "--cuda-device-only", "-I" + os.path.join(ROOT, "include")]
SDEF = re.compile(r"s_(?:movk_i32|mov_b32) (s\d+), (0x[0-9a-f]+)")


def negative(instr: str, lit: str) -> bool:
    v = int(lit, 16)
    return v >= 0xffff0000 or ("s_movk_i32" in instr and len(lit) == 6 and v >= 0x8000)


def scan(path: str):
    lines = open(path).read().split("\n")
    kname, hits = None, []
    for i, l in enumerate(lines):
        m = re.match(r"^(_Z\w+|k_\w+):", l)
        if m:
            kname = m.group(1)
        m = re.search(r"\b(ds_\w+)\s+(.*)", l)
        if not m or kname is None:
            continue
        ops = [o.strip() for o in m.group(2).split(",")]
        addr = (ops[0] if m.group(1).startswith(("ds_write", "ds_store")) else (ops[1] if len(ops) > 1 else "")).split()[0] if ops else ""
        if not re.match(r"v\d+$", addr):
            continue
        for j in range(i - 1, max(i - 400, 0), -1):
            lj = lines[j]
            if re.match(r"^(_Z\w+|k_\w+):", lj):
                break
            md = re.search(r"\bv_mov_b32_e32\s+" + addr + r"\b,\s*(\S+)", lj)
            if md:
                src = md.group(1)
                if re.match(r"0x[0-9a-f]+$", src) and negative("v_mov", src):
                    hits.append((kname, l.strip(), lj.strip()))
                elif re.match(r"s\d+$", src):
                    for k in range(j - 1, max(j - 400, 0), -1):
                        mk = SDEF.search(lines[k])
                        if mk and mk.group(1) == src:
                            if negative(lines[k], mk.group(2)):
                                hits.append((kname, l.strip(), lines[k].strip()))
                            break
                break
            if re.search(r"\bv_\w+\s+" + addr + r"\b", lj):
                break                      # some other definition (computed address): not the constant-base form
    return hits


def main() -> int:
    bad = 0
    with tempfile.TemporaryDirectory() as tmp:
        for src in sorted(glob.glob(os.path.join(CSRC, "*.hip"))):
            out = os.path.join(tmp, os.path.basename(src)[:-4] + ".s")
            subprocess.run(["/opt/rocm/bin/hipcc", *FLAGS, src, "-o", out], check=True, stderr=subprocess.DEVNULL)
            for kname, ds, definition in scan(out):
                bad += 1
                print(f"{os.path.basename(src)}: {kname}: `{ds}` takes a negative base (`{definition}`)")
    print("negative-base DS instructions:", bad)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
