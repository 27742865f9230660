import sys, os, time
sys.path.insert(0,'.'); sys.path.insert(0,'tests'); sys.path.insert(0,'tools')
import numpy as np
from speedseq_amd import capi
import oracle_py, common
orc = oracle_py.Oracle('oracle/liboracle.so')
lib = capi.Lib(sys.argv[3] if len(sys.argv)>3 else 'speedseq_amd/libssgpu.so')
opt = lib.opt_init()
oidx=orc.idx_load(common.EXAMPLE_FA); gidx=lib.index_load(common.EXAMPLE_FA)
N=int(sys.argv[1]); rl=int(sys.argv[2]) if len(sys.argv)>2 else 150
pairs, seqs, seq, off = common.sim_reads(N, 7, rl)
names=[]; 
for nm,_,_ in pairs: names += [nm,nm]
quals=['I'*len(s) for s in seqs]
t0=time.time(); res=capi.mem_process_pairs(lib,gidx,opt,seq,off,id0=0); print("gpu/emu pe",time.time()-t0, res.stats[:5])
text,so=capi.sam_format(lib,gidx,opt,res,names,seq,off,quals,"grp1")
t0=time.time(); otext,oso,opes=orc.process_pairs(oidx,seq,off,names,quals,0,"grp1",8); print("oracle pe",time.time()-t0)
print("pes", res.pes[1], opes[1])
a=text.split('\n'); b=otext.split('\n')
print(len(a),len(b))
bad=0
for i,(x,y) in enumerate(zip(a,b)):
    if x!=y:
        bad+=1
        if bad<6: print("DIFF\n ",x[:70],x.split('\t',11)[-1][-120:],"\n ",y[:70],y.split('\t',11)[-1][-120:])
print("sam line mismatches",bad,"of",len(a), "identical" if text==otext else "DIFFERENT")
