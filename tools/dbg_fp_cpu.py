"""Debug aid (CPU only): the GPU five-point solver's source compiled for the host against the oracle,
on the hypotheses of a saved pair (gpurun_out/dbg_pair_pts.npz)."""
import sys, ctypes as C
sys.path.insert(0,'.')
import numpy as np
from oracle import oracle as O
from visual_odometry_amd import synth
z=np.load('gpurun_out/dbg_pair_pts.npz'); p1,p2=z['p1'],z['p2']
K=synth.TUM_K
og,oE,om,oit,osamp=O.find_essential(p1,p2,K,fixed_iters=int(sys.argv[1]) if len(sys.argv)>1 else 0)
fx,fy,cx,cy=K
x1=np.stack([(p1[:,0].astype(np.float64)-cx)/fx,(p1[:,1].astype(np.float64)-cy)/fy],1)
x2=np.stack([(p2[:,0].astype(np.float64)-cx)/fx,(p2[:,1].astype(np.float64)-cy)/fy],1)
H=C.CDLL('build_ab/libfp_host.so')
def host(a,b):
    Es=np.zeros((10,9)); n=H.host_five_point(a.ctypes.data_as(C.c_void_p),b.ctypes.data_as(C.c_void_p),Es.ctypes.data_as(C.c_void_p)); return Es[:n].reshape(-1,3,3)
def resid(E,a,b):
    r=[np.array([*b[k],1])@E@np.array([*a[k],1]) for k in range(5)]
    EEt=E@E.T; T=2*EEt@E-np.trace(EEt)*E
    return max(np.max(np.abs(r)),abs(np.linalg.det(E)),np.max(np.abs(T)))
nd=0
for h in range(len(osamp)):
    a=np.ascontiguousarray(x1[osamp[h]]); b=np.ascontiguousarray(x2[osamp[h]])
    g=host(a,b); o=O.five_point(a,b)
    only_g=[E for E in g if min([min(np.abs(E-F).max(),np.abs(E+F).max()) for F in o]+[9])>1e-8]
    only_o=[F for F in o if min([min(np.abs(E-F).max(),np.abs(E+F).max()) for E in g]+[9])>1e-8]
    if only_g or only_o:
        nd+=1
        print('hyp',h,'host',len(g),'oracle',len(o),'only host',['%.1e'%resid(E,a,b) for E in only_g],'only oracle',['%.1e'%resid(F,a,b) for F in only_o])
print('differing hypotheses',nd,'of',len(osamp))
