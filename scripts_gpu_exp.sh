#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -q -x > gpurun_out/pytest_gpu.log 2>&1; echo "pytest_rc=$?" >> gpurun_out/pytest_gpu.log
tail -12 gpurun_out/pytest_gpu.log
python - <<'PY'
import torch, time, sys
sys.path.insert(0,'.')
from adjointnonlinearraytracing_amd import sensor
dev=torch.device('cuda:0'); torch.manual_seed(0)
n,res,span=1<<20,512,1.0
x=torch.rand(n,3,device=dev)*0.8+0.1; x[:,1]=1.0
v=torch.randn(n,3,device=dev)*0.1; v[:,1]=1.0
p=torch.tensor([[.5,1.1,.5]],device=dev); nn=torch.tensor([[0.,1.,0.]],device=dev); tt=torch.tensor([[0.,0.,1.]],device=dev)
xs=x.clone().requires_grad_(True); vs=v.clone().requires_grad_(True)
gI=torch.randn(res,res,device=dev)
for it in range(3):
    torch.cuda.synchronize(); t0=time.perf_counter()
    img=sensor.generate_sensor((xs,vs),1.0,(p,nn),res,span,tt)
    torch.cuda.synchronize(); t1=time.perf_counter()
    (img*gI).sum().backward()
    torch.cuda.synchronize(); t2=time.perf_counter()
print('sensor splat 1M rays -> 512^2: fwd %.3f ms, bwd(+loss) %.3f ms'%((t1-t0)*1e3,(t2-t1)*1e3))
PY
