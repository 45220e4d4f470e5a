import os, sys, time
sys.path.insert(0, "/root/repo/nf-isam_amd"); sys.path.insert(0, "/root/repo")
import numpy as np, torch
import nfisam_hip as nh, bench as BM
K,H,B,L=9,8,5.0,1
dev=torch.device("cuda:0")
for nc in (2,4,8,16,32):
    xs,kps=[],[]
    for c in range(nc):
        rng=np.random.RandomState(c)
        xs.append(torch.from_numpy(rng.randn(2000,15).astype(np.float32)).to(dev))
        kps.append(nh.pack(torch.from_numpy(BM.init_blob_np(15,K,H,L,c)).to(dev),15,K,H,L))
    tb=nh.TrainBatch(xs,kps,K,H,B,L,lr=0.01,max_iters=400,early_stop=False)
    tb.prepare(True); torch.cuda.synchronize(); t0=time.perf_counter(); tb.run(True); torch.cuda.synchronize()
    dt=time.perf_counter()-t0
    print("nc=%d: %.1f us/iter %.3e samples/s"%(nc,dt/400*1e6,nc*2000*400/dt))
