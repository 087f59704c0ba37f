"""Disassembly of one kernel of the built libbsched.so (every code object of the fat binary is searched).
usage: python tools/kernel_disasm.py <mangled-name-substring> [out.s]   (prints the path of the .s file and a few counts)"""
import os
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
LIB = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "batch-scheduler_amd", "libbsched.so")


def code_objects(lib, d):
    fat = os.path.join(d, "fat")
    subprocess.run([f"{LLVM}/llvm-objcopy", "--dump-section", f".hip_fatbin={fat}", lib], check=True)
    blob = open(fat, "rb").read()
    starts = [m.start() for m in re.finditer(re.escape(b"__CLANG_OFFLOAD_BUNDLE__"), blob)]
    out = []
    for n, a in enumerate(starts):
        part, co = os.path.join(d, f"fat{n}"), os.path.join(d, f"co{n}")
        open(part, "wb").write(blob[a:(starts[n + 1] if n + 1 < len(starts) else len(blob))])
        subprocess.run([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", f"--input={part}", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={co}"], check=True)
        out.append(co)
    return out


def main():
    pat = sys.argv[1]
    dst = sys.argv[2] if len(sys.argv) > 2 else os.path.join(tempfile.gettempdir(), "kernel.s")
    with tempfile.TemporaryDirectory() as d:
        for co in code_objects(LIB, d):
            syms = subprocess.run([f"{LLVM}/llvm-readelf", "-sW", co], capture_output=True, text=True).stdout
            names = [ln.split()[-1] for ln in syms.splitlines() if pat in ln and " FUNC " in ln]
            for nm in names:
                txt = subprocess.run([f"{LLVM}/llvm-objdump", "-d", co, f"--disassemble-symbols={nm}"], capture_output=True, text=True).stdout
                body = "\n".join(ln.split("//")[0].rstrip() for ln in txt.splitlines())
                open(dst, "w").write(body)
                ins = [ln.split()[0] for ln in body.splitlines() if ln.startswith("\t")]
                cnt = lambda p: sum(1 for i in ins if re.match(p, i))
                print(nm)
                print(f"  {dst}: {len(ins)} instructions; scratch/buffer spill {cnt(r'scratch_|buffer_(load|store)_dword$')}, v_writelane {cnt('v_writelane')}, v_readlane {cnt('v_readlane')}, "
                      f"s_load {cnt('s_load')}, global/flat load {cnt('(global|flat)_load')}, ds {cnt('ds_')}, s_waitcnt {cnt('s_waitcnt')}, s_barrier {cnt('s_barrier')}")
                return
    print("no kernel matches", pat)


if __name__ == "__main__":
    main()
