import sys, time, numpy as np
sys.path.insert(0,'' + __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))) + ''); sys.path.insert(0,'/root/repo/tests')
import osmo_tetra_amd as T
n=int(sys.argv[1]) if len(sys.argv)>1 else 200000
pat=np.array([3,0,1,0,1,0,1,0],np.uint8); types=np.tile(pat,n//8+1)[:n]
slots=T.synth_slots(np.concatenate([[3],types]).astype(np.uint8),seed=11,scramb_init=0x41802A07)
stream=np.concatenate([np.zeros(100,np.uint8),slots.reshape(-1),np.zeros(700,np.uint8)])
r0=T.sync_walk(stream[:100+510*40+700].copy())
anchor=r0["slots"][0][0]
print("anchor",anchor)
ng=(len(stream)-anchor)//510
g=np.arange(ng)
# grid slot g is stream slot (anchor-100)//510 + g ; types index offset
k0=(anchor-100)//510
ty=np.concatenate([[3],types])[k0:k0+ng].astype(np.uint32)
if len(ty)<ng: ty=np.concatenate([ty,np.full(ng-len(ty),0xff,np.uint32)])
off=np.where(ty==3,214,244).astype(np.uint32)
cls=np.where(ty==0xff,0xff,ty|(off<<8)).astype(np.uint32)
ys=np.where(ty==3,214,0xffff).astype(np.uint16)
for kw in (dict(grid=True,burst_events=False),dict(burst_events=False),dict(grid=True)):
    for _ in range(3):
        t0=time.perf_counter(); r=T.sync_walk(stream,chunk=64,anchor=anchor,cls=cls,ysum=ys,**kw); el=time.perf_counter()-t0
    print(kw, "%.3f ms"%(el*1e3), r["nslots"], r["noffgrid"], len(r["event_arr"]))
rng=np.random.default_rng(1)
for nbad in (100, 1000):
    c2=cls.copy()
    idx=rng.choice(np.arange(100,ng-100),nbad,replace=False)
    c2[idx]=0|(100<<8)          # a spurious NORM_1 sequence at offset 100: burst dropped, lock kept
    for _ in range(3):
        t0=time.perf_counter(); r=T.sync_walk(stream,chunk=64,anchor=anchor,cls=c2,ysum=ys,grid=True,burst_events=False); el=time.perf_counter()-t0
    print(nbad,"misplaced: %.3f ms"%(el*1e3), r["nslots"], len(r["event_arr"]))
