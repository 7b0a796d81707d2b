#!/usr/bin/env python3
"""Times the cable (radial profile) adjoint for 4M rays x ~512 steps with rays in random order and in source-pixel
order (neighbouring rays at nearly the same radius -> same-bin LDS adds, the case the pair / quad pre-reduction
of k_backtrace_cable is for).  Two JSON lines."""
import sys, json, torch, numpy as np
sys.path.insert(0,'.')
from adjointnonlinearraytracing_amd import drrt
drrt.options.check_failed=False
dev=torch.device('cuda:0')
rres, radius = 257, 1.0
ds = radius/(rres-1)/2; length = 512*ds
prof = torch.sqrt(2.0 - torch.linspace(0,1,rres)**2).to(dev)
n=2048*2048
g=torch.Generator(device=dev).manual_seed(0)
ang=torch.rand(n,device=dev,generator=g)*6.2831853; rad=0.9*radius*torch.sqrt(torch.rand(n,device=dev,generator=g))
pos=torch.stack([radius+rad*torch.cos(ang), torch.full((n,),0.37*ds,device=dev), radius+rad*torch.sin(ang)],-1)
vel=torch.randn(n,3,device=dev,generator=g)*0.05; vel[:,1]=1; vel/=vel.norm(dim=1,keepdim=True)
tg=torch.tensor([[radius,0.75*length,radius]],device=dev).expand(n,3).contiguous()
T=drrt.TracerC()
xt,vt,d2=T.trace_cable(prof,radius,length,pos,vel,tg,ds)
one=torch.ones_like(xt)
def timed(fn,reps=5):
    fn(); torch.cuda.synchronize()
    a,b=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize(); return a.elapsed_time(b)/reps
print(json.dumps(dict(adj_random_order_ms=timed(lambda: T.backtrace_cable(prof,radius,length,xt,vt,one,one,ds)))))
# pixel-ordered source (neighbouring rays at nearly the same radius)
side=2048
i=torch.arange(side,device=dev,dtype=torch.float32)
X,Z=torch.meshgrid(i,i,indexing='ij')
px=(X.flatten()+0.5)/side*2*radius; pz=(Z.flatten()+0.5)/side*2*radius
pos2=torch.stack([px, torch.full((n,),0.37*ds,device=dev), pz],-1)
vel2=torch.zeros(n,3,device=dev); vel2[:,1]=1
xt2,vt2,_=T.trace_cable(prof,radius,length,pos2,vel2,tg,ds)
print(json.dumps(dict(adj_pixel_order_ms=timed(lambda: T.backtrace_cable(prof,radius,length,xt2,vt2,one,one,ds)))))
