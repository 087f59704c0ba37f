"""Per-kernel resource usage of the built libbsched.so: VGPRs, SGPRs, LDS, scratch (`.private_segment_fixed_size` must be 0 for
the hot kernels: a kernel that uses scratch pays for the scratch set-up on every launch).  Usage: python tools/kernel_resources.py [substring ...]"""
import os
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
LIB = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "batch-scheduler_amd", "libbsched.so")


def resources(lib=LIB):
    """every kernel of every code object in the library (one offload bundle per translation unit)"""
    out = {}
    with tempfile.TemporaryDirectory() as d:
        fat = os.path.join(d, "fat")
        subprocess.run([f"{LLVM}/llvm-objcopy", "--dump-section", f".hip_fatbin={fat}", lib], check=True)
        blob = open(fat, "rb").read()
        magic = b"__CLANG_OFFLOAD_BUNDLE__"
        starts = [m.start() for m in re.finditer(re.escape(magic), blob)]
        for n, a in enumerate(starts):
            part, co = os.path.join(d, f"fat{n}"), os.path.join(d, f"co{n}")
            open(part, "wb").write(blob[a:(starts[n + 1] if n + 1 < len(starts) else len(blob))])
            subprocess.run([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", f"--input={part}", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={co}"], check=True)
            notes = subprocess.run([f"{LLVM}/llvm-readelf", "--notes", co], capture_output=True, text=True, check=True).stdout
            for blk in notes.split("- .agpr_count:")[1:]:
                name = re.search(r"\.name:\s+(\S+)", blk)
                if not name:
                    continue
                get = lambda k: int(re.search(rf"\.{k}:\s+(\d+)", blk).group(1))
                rec = dict(vgpr=get("vgpr_count"), sgpr=get("sgpr_count"), lds=get("group_segment_fixed_size"), scratch=get("private_segment_fixed_size"),
                           agpr=int(blk.split()[0]), unit=n)
                assert name.group(1) not in out, f"{name.group(1)} is emitted by two translation units"
                out[name.group(1)] = rec
    return out


if __name__ == "__main__":
    pats = sys.argv[1:]
    for k, v in sorted(resources().items()):
        if not pats or any(p in k for p in pats):
            print(f"{k[:90]:90s} vgpr {v['vgpr']:3d} sgpr {v['sgpr']:3d} lds {v['lds']:6d} scratch {v['scratch']}")
