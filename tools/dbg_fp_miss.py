import sys, ctypes as C
sys.path.insert(0,'.')
import numpy as np
from oracle import oracle as O
from visual_odometry_amd import synth
H=C.CDLL('build_ab/libfp_host.so')
K=synth.TUM_K; fx,fy,cx,cy=K
vp=lambda a:a.ctypes.data_as(C.c_void_p)
def host(a,b,chart):
    Es=np.zeros((10,9)); n=H.host_five_point_chart(vp(a),vp(b),vp(Es),chart); return [e.reshape(3,3) for e in Es[:n]]
def same(E,F): return min(np.abs(E-F).max(),np.abs(E+F).max())<1e-7
frames=synth.sequence(0x5EED0000, 60)
for p in (3,20,40,55):
    ka,da=O.orb_detect(frames[p],1000); kb,db=O.orb_detect(frames[p+1],1000)
    qi,ti,_=O.match(da,db)
    p1=np.stack([ka['x'][qi],ka['y'][qi]],1); p2=np.stack([kb['x'][ti],kb['y'][ti]],1)
    x1=np.stack([(p1[:,0].astype(np.float64)-cx)/fx,(p1[:,1].astype(np.float64)-cy)/fy],1)
    x2=np.stack([(p2[:,0].astype(np.float64)-cx)/fx,(p2[:,1].astype(np.float64)-cy)/fy],1)
    _,_,_,_,samp=O.find_essential(p1,p2,K,fixed_iters=200)
    for h in range(200):
        a=np.ascontiguousarray(x1[samp[h]]); b=np.ascontiguousarray(x2[samp[h]])
        g=host(a,b,0); o=list(O.five_point(a,b))
        miss=[T for T in o if not any(same(T,E) for E in g)]
        if not miss: continue
        cf=np.zeros(11); bm=np.zeros((3,13)); EE=np.zeros((4,9))
        H.host_front(vp(a),vp(b),0,vp(cf),vp(bm),vp(EE))
        rr=np.roots(cf[::-1]); 
        roots=np.zeros(10); nr=H.host_real_roots(vp(cf),vp(roots))
        print('pair',p,'hyp',h,'host models',len(g),'oracle',len(o),'missed',len(miss))
        print('  cf',cf)
        print('  numpy roots',np.sort_complex(rr))
        print('  host real roots',roots[:nr])
        # coordinates of missed solutions in this chart: E = x EE0 + y EE1 + z EE2 + w EE3
        for T in miss:
            co=np.linalg.lstsq(EE.T,T.reshape(9),rcond=None)[0]
            print('  missed solution chart coords (x,y,z)/w', co[:3]/co[3], 'w',co[3])
