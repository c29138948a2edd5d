"""Debug aid (CPU only): completeness of the five-point solvers over many minimal samples of the
synthetic sequence.  Ground truth per sample = union of everything any variant finds with a
constraint residual below 1e-12."""
import sys, ctypes as C
sys.path.insert(0,'.')
import numpy as np
from oracle import oracle as O
from visual_odometry_amd import synth
H=C.CDLL('build_ab/libfp_host.so')
K=synth.TUM_K; fx,fy,cx,cy=K
def host(a,b,chart):
    Es=np.zeros((10,9)); n=H.host_five_point_chart(a.ctypes.data_as(C.c_void_p),b.ctypes.data_as(C.c_void_p),Es.ctypes.data_as(C.c_void_p),chart); return [e.reshape(3,3) for e in Es[:n]]
def resid(E,a,b):
    r=[np.array([*b[k],1])@E@np.array([*a[k],1]) for k in range(5)]
    EEt=E@E.T; T=2*EEt@E-np.trace(EEt)*E
    return max(np.max(np.abs(r)),abs(np.linalg.det(E)),np.max(np.abs(T)))
def same(E,F): return min(np.abs(E-F).max(),np.abs(E+F).max())<1e-7
def union(*sets):
    out=[]
    for s in sets:
        for E in s:
            if not any(same(E,F) for F in out): out.append(E)
    return out
frames=synth.sequence(0x5EED0000, 60)
pairs=[int(x) for x in sys.argv[1].split(',')] if len(sys.argv)>1 else [3,20,40,55]
nh=int(sys.argv[2]) if len(sys.argv)>2 else 200
tot={k:[0,0,0] for k in ('g0','g0+odd','g01','g012','o4','o1')}   # [missed, marginal extras, samples with any difference]
N=0; ntrue=0
for p in pairs:
    ka,da=O.orb_detect(frames[p],1000); kb,db=O.orb_detect(frames[p+1],1000)
    qi,ti,_=O.match(da,db)
    p1=np.stack([ka['x'][qi],ka['y'][qi]],1); p2=np.stack([kb['x'][ti],kb['y'][ti]],1)
    x1=np.stack([(p1[:,0].astype(np.float64)-cx)/fx,(p1[:,1].astype(np.float64)-cy)/fy],1)
    x2=np.stack([(p2[:,0].astype(np.float64)-cx)/fx,(p2[:,1].astype(np.float64)-cy)/fy],1)
    _,_,_,_,samp=O.find_essential(p1,p2,K,fixed_iters=nh)
    for h in range(nh):
        a=np.ascontiguousarray(x1[samp[h]]); b=np.ascontiguousarray(x2[samp[h]])
        g=[host(a,b,c) for c in range(3)]
        O.set_five_point_charts(4); o4=list(O.five_point(a,b))
        O.set_five_point_charts(1); o1=list(O.five_point(a,b))
        O.set_five_point_charts(4)
        allm=union(g[0],g[1],g[2],o4)
        truth=[E for E in allm if resid(E,a,b)<1e-12]
        variants={'g0':g[0],'g0+odd':union(g[0],g[1]) if len(g[0])%2 else g[0],'g01':union(g[0],g[1]),'g012':union(*g),'o4':o4,'o1':o1}
        N+=1; ntrue+=len(truth)
        for k,v in variants.items():
            miss=sum(1 for T in truth if not any(same(T,E) for E in v))
            extra=sum(1 for E in v if not any(same(T,E) for T in truth))
            tot[k][0]+=miss; tot[k][1]+=extra; tot[k][2]+= (miss>0 or extra>0)
print('samples',N,'true solutions',ntrue)
for k,v in tot.items(): print(k,'missed',v[0],'marginal extras',v[1],'samples differing',v[2])
