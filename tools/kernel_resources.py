#!/usr/bin/env python3
"""Register / LDS / scratch allocation of every kernel in the built library, read from the code objects' metadata (no GPU, no recompilation):
the `.hip_fatbin` section of libtcs_mi355.so is a sequence of clang offload bundles (one per translation unit); each is unbundled for gfx950
and its AMDGPU metadata note parsed.  What the numbers decide: waves per SIMD = 512 // ceil8(vgpr_count) (capped at 8), and the K loop of
k_conv_s16 is only as fast as that occupancy lets it be ("Register occupancy").
usage: python tools/kernel_resources.py [substring]      -> one line per kernel; tests/test_host.py asserts the bounds."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"
FIELDS = ("vgpr_count", "agpr_count", "sgpr_count", "vgpr_spill_count", "sgpr_spill_count", "group_segment_fixed_size",
          "private_segment_fixed_size", "max_flat_workgroup_size")


def default_lib() -> str:
    return os.path.join(ROOT, "temporally-consistent-stereo-matching_amd", "lib", "libtcs_mi355.so")


def kernel_resources(lib: str = None, arch: str = "gfx950") -> dict:
    """{mangled kernel name: {field: int}} over all translation units of the library."""
    lib = lib or default_lib()
    out = {}
    with tempfile.TemporaryDirectory() as tmp:
        fat = os.path.join(tmp, "fat.bin")
        subprocess.run([f"{LLVM}/llvm-objcopy", "--dump-section", f".hip_fatbin={fat}", lib, os.path.join(tmp, "copy.so")], check=True)
        blob = open(fat, "rb").read()
        starts = [m.start() for m in re.finditer(re.escape(MAGIC), blob)]
        for n, (a, b) in enumerate(zip(starts, starts[1:] + [len(blob)])):
            part, co = os.path.join(tmp, f"b{n}.bin"), os.path.join(tmp, f"b{n}.co")
            with open(part, "wb") as f:
                f.write(blob[a:b])
            r = subprocess.run([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", f"--input={part}",
                                f"--targets=hipv4-amdgcn-amd-amdhsa--{arch}", f"--output={co}"], capture_output=True, text=True)
            if r.returncode != 0 or not os.path.exists(co) or os.path.getsize(co) == 0:
                continue
            notes = subprocess.run([f"{LLVM}/llvm-readelf", "--notes", co], capture_output=True, text=True, check=True).stdout
            for block in re.split(r"\n\s+- \.", notes)[1:]:
                block = "." + block
                name = re.search(r"\.name:\s+(\S+)", block)
                if not name or ".vgpr_count" not in block:
                    continue
                out[name.group(1)] = {k: int(m.group(1)) for k in FIELDS for m in [re.search(rf"\.{k}:\s+(\d+)", block)] if m}
    return out


def demangle(names):
    names = list(names)
    try:
        r = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True)
        nice = r.stdout.splitlines()
        return nice if r.returncode == 0 and len(nice) == len(names) else names
    except OSError:
        return names


def waves_per_simd(vgprs: int) -> int:
    return max(1, min(8, 512 // (-(-max(vgprs, 1) // 8) * 8)))


if __name__ == "__main__":
    only = sys.argv[1] if len(sys.argv) > 1 else ""
    res = kernel_resources()
    names = sorted(res)
    for mangled, nice in zip(names, demangle(names)):
        if only and only not in nice:
            continue
        r = res[mangled]
        print(f"{nice[:88]:88s} vgpr {r['vgpr_count']:3d} ({waves_per_simd(r['vgpr_count'])} waves/SIMD)  spill {r.get('vgpr_spill_count', 0)}"
              f"  lds {r.get('group_segment_fixed_size', 0):6d}  scratch {r.get('private_segment_fixed_size', 0)}")
