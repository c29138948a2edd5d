import sys, numpy as np
sys.path.insert(0,'/root/repo')
from oracle import oracle as O
from visual_odometry_amd import capi, synth
ctx = capi.Context(0)
a,b = synth.frame_pair(21)
ka,da = O.orb_detect(a,2000); kb,db = O.orb_detect(b,2000)
qi,ti,_ = O.match(da,db)
p1 = np.stack([ka['x'][qi],ka['y'][qi]],1); p2=np.stack([kb['x'][ti],kb['y'][ti]],1)
K = synth.TUM_K
og,oE,om,oit,osamp = O.find_essential(p1,p2,K, fixed_iters=16)
gg,gE,gm,git,gsamp = ctx.find_essential(p1,p2,K, fixed_iters=16)
print('samples equal', np.array_equal(osamp,gsamp), og, gg)
def norm(p): 
    return np.stack([(p[:,0].astype(np.float64))*(1/K[0]) + (-K[2]*(1/K[0])), p[:,1].astype(np.float64)*(1/K[1]) + (-K[3]*(1/K[1]))],1)
m1,m2 = norm(p1),norm(p2)
thr = (1.0/((K[0]+K[1])/2))**2
def count(E):
    x1 = np.c_[m1,np.ones(len(m1))]; x2=np.c_[m2,np.ones(len(m2))]
    Ex1 = x1@E.T; Etx2 = x2@E
    num = (x2*Ex1).sum(1)**2; den = Ex1[:,0]**2+Ex1[:,1]**2+Etx2[:,0]**2+Etx2[:,1]**2
    return int(((num/den).astype(np.float32) <= np.float32(thr)).sum())
for h in range(16):
    idx = osamp[h]
    Eo = O.five_point(m1[idx], m2[idx])
    Eg = ctx.five_point(m1[idx][None], m2[idx][None])[0]
    print(h, 'oracle', len(Eo), [count(E) for E in Eo], 'gpu', len(Eg), [count(E) for E in Eg])
    for E in Eg:
        d = min(min(np.abs(E-F).max(), np.abs(E+F).max()) for F in Eo) if len(Eo) else -1
        print('    gpu model nearest oracle diff %.2e' % d)
og,oE,om,oit,osamp = O.find_essential(p1,p2,K)
gg,gE,gm,git,gsamp = ctx.find_essential(p1,p2,K)
print('adaptive', og, oit, gg, git)
