"""Scans hipcc --save-temps assembly (gfx950 .s files) for the hottest loop of every kernel (the loop with the most
MFMAs) and prints what sits inside it: MFMA / vector-load / scratch counts and the s_waitcnt vmcnt values.  A healthy
register- or DMA-prefetch ring shows COUNTED waits (vmcnt(5), vmcnt(12) ...); a column of zeros means the loads are
drained before every use -- seen in round 3 on the bf16 GEMM (run-time operand layouts + conditional prefetch loads) and
on the pointwise kernel's fragment reads (run-time ReLU branch).
    cd /tmp && hipcc -O3 -std=c++17 --offload-arch=gfx950 -c <repo>/xmcgan_image_generation_amd/csrc/gemm_f32.hip --save-temps -o /tmp/x.o
    python tools/isa_loop_scan.py /tmp/gemm_f32-hip-amdgcn-amd-amdhsa-gfx950.s"""
import re,sys
for f in sys.argv[1:]:
    s=open(f).read()
    for m in re.finditer(r"^(_Z\S+):[^\n]*\n(.*?)\.end_amdhsa_kernel", s, re.S|re.M):
        name=m.group(1); body=m.group(2).splitlines()
        # find loops: label ... branch back to label
        labels={}
        for i,l in enumerate(body):
            t=l.strip()
            mm=re.match(r"(\.LBB\d+_\d+):", t)
            if mm: labels[mm.group(1)]=i
        best=None
        for i,l in enumerate(body):
            t=l.strip().split()
            if t and t[0].startswith("s_cbranch") and len(t)>1 and t[1] in labels and labels[t[1]]<i:
                seg=body[labels[t[1]]:i]
                nm=sum(1 for x in seg if "v_mfma" in x)
                if nm and (best is None or nm>best[0]): best=(nm,seg)
        if not best: continue
        nm,seg=best
        vm=[re.search(r"vmcnt\((\d+)\)",x).group(1) for x in seg if "s_waitcnt" in x and "vmcnt" in x]
        ld=sum(1 for x in seg if re.search(r"\b(buffer_load|global_load)",x))
        sc=sum(1 for x in seg if "scratch_" in x)
        short=re.sub(r"_ZN12_GLOBAL__N_1\d+","",name)[:60]
        print(f"{short:62s} mfma={nm:4d} loads={ld:3d} scratch={sc:3d} vmcnt waits: {' '.join(vm[:24])}")
