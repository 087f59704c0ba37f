import subprocess, sys, re, tempfile, os, hashlib
LLVM="/opt/rocm/lib/llvm/bin"
def kernels(lib):
    d=tempfile.mkdtemp()
    fat,co=os.path.join(d,"fat"),os.path.join(d,"co")
    subprocess.run([f"{LLVM}/llvm-objcopy","--dump-section",f".hip_fatbin={fat}",lib],check=True)
    subprocess.run([f"{LLVM}/clang-offload-bundler","--unbundle","--type=o",f"--input={fat}","--targets=hipv4-amdgcn-amd-amdhsa--gfx950",f"--output={co}"],check=True)
    dis=subprocess.run([f"{LLVM}/llvm-objdump","-d","--no-show-raw-insn","--no-leading-addr",co],capture_output=True,text=True,check=True).stdout
    out={}; cur=None
    for line in dis.splitlines():
        m=re.match(r"^[0-9a-f]* ?<(\S+)>:$",line)
        if m: cur=m.group(1); out[cur]=[]; continue
        if cur and line.strip():
            # strip addresses in branch targets comments
            l=re.sub(r"//.*$","",line).strip()
            l=re.sub(r"<\S+\+0x[0-9a-f]+>","",l)
            out[cur].append(l)
    return {k:hashlib.md5("\n".join(v).encode()).hexdigest()+":"+str(len(v)) for k,v in out.items()}
a=kernels(sys.argv[1]); b=kernels(sys.argv[2])
same=[k for k in a if k in b and a[k]==b[k]]
diff=[k for k in a if k in b and a[k]!=b[k]]
print("same",len(same),"diff",len(diff),"only old",[k for k in a if k not in b],"only new",[k for k in b if k not in a])
for k in diff: print("DIFF",k,a[k],b[k])
