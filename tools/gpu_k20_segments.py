"""DIAGNOSTIC: host-call / device segments of the driver's 20-step rollout (K = 20, one fused launch per
slice) in a tight repeat loop, for 1 / 2 / 4 stream slices."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, numpy as np
from bench import build_env
from gym_solo_amd import abi
n=4096
for k,spl,streams in ((20,20,2),(20,20,1),(20,20,4)):
  env=build_env(n,0,'float32',steps_per_launch=spl,rollout_streams=streams); eng=env.engine
  g=torch.Generator(device='cuda').manual_seed(1234)
  acts=(torch.rand(k,n,12,device='cuda',generator=g)*2-1)*6.283
  out=eng.rollout_buffers(k)
  eng.rollout(acts[:5], abi.STEP_ALL)
  torch.cuda.synchronize()
  res=[]
  for rep in range(12):
    sb=eng.stats.clone(); torch.cuda.synchronize()
    t0=time.perf_counter()
    eng.rollout(acts, abi.STEP_ALL, out=out)
    t1=time.perf_counter()
    torch.cuda.synchronize()
    t2=time.perf_counter()
    st=(eng.stats-sb).clone()
    t3=time.perf_counter()
    torch.cuda.synchronize()
    t4=time.perf_counter()
    res.append(((t1-t0)*1e3,(t2-t0)*1e3,(t3-t2)*1e3,(t4-t2)*1e3))
  r=np.array(res)
  print('K=%d S=%d streams=%d: median ms: rollout host call %.3f ; rollout to sync %.3f ; stats host %.3f ; stats to sync %.3f ; => %.3g env-steps/s'%(k,spl,streams,*np.median(r,axis=0), n*k/(np.median(r[:,1])+np.median(r[:,3]))*1e3))
  env._close()
