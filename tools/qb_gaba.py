import sys; sys.path.insert(0,'tests')
import numpy as np, gabalib as G, time
hip = G.Hip(**G.PACBIO)
rng = np.random.default_rng(3)
ref = rng.integers(0,4,2_000_000,dtype=np.uint8)
jobs=[]
N=int(sys.argv[1]); L=int(sys.argv[2]); TR=int(sys.argv[3]) if len(sys.argv)>3 else 1
for i in range(N):
    s=int(rng.integers(0,len(ref)-L)); a=ref[s:s+L]
    b=G.mutate(rng,a,0.012,0.072,0.036)
    jobs.append((a,0,0,b,0,0,0,TR))
for rep in range(2):
    t=time.time(); r=hip.extend_batch(jobs); dt=time.time()-t
    s=hip.stats()
    print('N',N,'L',L,'wall',round(dt,3),'kernel_ms',round(s.kernel_ms,3),'vectors',s.vectors,'Gvec/s',round(s.vectors/s.kernel_ms/1e6,3),'trace_steps',s.trace_steps, 'GB/s(40.5B/vec)', round(s.vectors*40.5/s.kernel_ms/1e6,1))
print(sum(x['traced']==1 for x in r), 'traced')
