#!/bin/bash
for st in 1 2 3 4 0; do
ESP_LOCAL_STOP=$st timeout 300 python - <<PY
import os,sys
sys.path.insert(0,'.')
from esparse_loader import load
esp=load()
n=256; N=n**3
A=esp.ExtendableSparseMatrix(N,N,capacity_hint=12*n*n*(n-1)+6*n*n)
A.timing_enable(True)
for it in range(4):
    A.reset(); A.generate_fdrand(n,n,n,rand_mode=1); 
    try:
        A._d.flush(0)
    except Exception as e:
        pass
    if it==0: A.timing()
t=A.timing()
print("stop", os.environ.get("ESP_LOCAL_STOP"), "local ms/launch %.3f"%(t["local"][0]/max(t["local"][1],1)))
PY
done
