import re,sys,subprocess
src=sys.argv[1]; pat=sys.argv[2] if len(sys.argv)>2 else ""
r=subprocess.run(["/opt/rocm/bin/hipcc","--offload-arch=gfx950","-O3","-fPIC","-std=c++17","-fno-gpu-rdc","-ffp-contract=off","-c",src,"-o","/tmp/_res.o","-Rpass-analysis=kernel-resource-usage"],capture_output=True,text=True)
cur=None; rows={}
for l in r.stderr.split("\n"):
    m=re.search(r"Function Name: (\S+)",l)
    if m: cur=m.group(1); rows[cur]={}
    for key in ("VGPRs","AGPRs","ScratchSize \[bytes/lane\]","Occupancy \[waves/SIMD\]","VGPRs Spill","SGPRs"):
        m=re.search(r"remark:\s+"+key+r": (\d+)",l)
        if m and cur: rows[cur][key.split(" [")[0].replace("\\","")]=int(m.group(1))
import subprocess as sp
for k,v in rows.items():
    if pat in k:
        name=sp.run(["c++filt",k],capture_output=True,text=True).stdout.strip()
        print(name[:110], v)
if r.returncode: print(r.stderr[-3000:])
