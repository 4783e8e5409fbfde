import sys, numpy as np, torch
sys.path.insert(0, '.')
from oracle import binding as O
from tests.helpers import *
import cactus_amd.hip as H
H.init(0)
M,K,N=16,1024,2048
rng=np.random.default_rng(M+K+N)
B,Bs=rand_weights_int8(rng,N,K,128)
A16=rand_act_f16(rng,M,K)
A,As=O.quantize_activations(A16)
oc,og=O.matmul_int8(A,As,B,Bs,128,True)
d=lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
c=H.matmul_int8(d(A),d(As),d(B),d(Bs),128).cpu().view(torch.int16).numpy().view(np.uint16)
bad=np.argwhere(c!=oc)
print("nbad",len(bad),"of",c.size)
print("rows hist",np.bincount(bad[:,0],minlength=16))
print("cols mod16 hist",np.bincount(bad[:,1]%16,minlength=16))
for m,n in bad[:5]:
    g=og[m,n].astype(np.float64); s=Bs[n].astype(np.float64)
    exact=(g*s).sum()*float(As[m])
    print(m,n,hex(c[m,n]),hex(oc[m,n]),f16(c[m:m+1,n:n+1])[0,0],f16(oc[m:m+1,n:n+1])[0,0],exact)
# per-M check
for MM in (1,4,5,8,12,15,16):
    c2=H.matmul_int8(d(A[:MM]),d(As[:MM]),d(B),d(Bs),128).cpu().view(torch.int16).numpy().view(np.uint16)
    print(MM,"bad",(c2!=oc[:MM]).sum())
