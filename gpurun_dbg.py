import sys, os
sys.path.insert(0,'.')
import numpy as np
import torch; torch.cuda.init()
from esparse_loader import load
esp=load()
n=256; N=n**3
A=esp.ExtendableSparseMatrix(N,N,capacity_hint=12*n*n*(n-1)+6*n*n)
for it in range(3):
    A.reset(); A.generate_fdrand(n,n,n,rand_mode=1)
    if it==2: os.environ["ESP_LOCAL_STAMPS"]="gpurun_out/stamps.bin"
    A.flush()
st=np.fromfile("gpurun_out/stamps.bin",dtype=np.uint64).reshape(-1,8).astype(np.int64)
d=np.diff(st,axis=1)*10.0/1000.0   # 100 MHz ticks -> us
names=["seg->loads arrived","count+scan","scatter->LDS","column sort+fold","compaction","look-back","LDS compact+stores"]
print("segments",len(st),"block lifetime us: median %.2f mean %.2f"%(np.median(st[:,7]-st[:,0])*0.01, np.mean(st[:,7]-st[:,0])*0.01))
for i,nm in enumerate(names): print("%-22s median %6.2f  mean %6.2f  p90 %6.2f"%(nm,np.median(d[:,i]),np.mean(d[:,i]),np.percentile(d[:,i],90)))
span=(st[:,7].max()-st[:,0].min())*0.01
print("kernel span us %.1f ; sum of lifetimes / span = %.1f concurrent blocks"%(span, (st[:,7]-st[:,0]).sum()*0.01/span))
