#!/usr/bin/env python3
"""CPU study behind the Winograd decision: the whole SR3 16->128 UNet evaluated with every 3x3 stride-1
convolution replaced by an fp32 emulation of F(2x2,3x3) (torch einsum transforms, fp32 throughout), against a float64
run and the plain fp32 run of the oracle.  Output of the round-2 run (8 threads, seeds as below):
  val   |ref|max 1.456  direct32 err 1.17e-06  wino32 err 1.35e-06  (tol 2.91e-05)
  train |ref|max 2.722  direct32 err 1.87e-06  wino32 err 2.28e-06  (tol 5.44e-05)
i.e. Winograd F(2x2,3x3) in fp32 is in the direct convolution's error class, 20x inside the stated tolerance.
The GPU kernel itself is compared with float64 per layer shape in tests/test_gpu_ops.py and over the whole network at
the benchmarked batch sizes in tests/test_gpu_bench_configs.py."""
import os
import sys
import time

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in (ROOT, os.path.join(ROOT, 'image-super-resolution-via-iterative-refinement_amd'), os.path.join(ROOT, 'tests')):
    sys.path.insert(0, _p)
from oracle import sr3_oracle as O
import bench
import model.networks as networks
torch.set_num_threads(8)
BT=torch.tensor([[1,0,-1,0],[0,1,1,0],[0,-1,1,0],[0,1,0,-1]],dtype=torch.float32)
G=torch.tensor([[1,0,0],[.5,.5,.5],[.5,-.5,.5],[0,0,1]],dtype=torch.float32)
AT=torch.tensor([[1,1,1,0],[0,1,-1,-1]],dtype=torch.float32)
def wino_conv(x,w,b):
    # x (B,C,H,W) fp32, w (O,C,3,3); pad 1; H,W even
    Bn,C,H,W=x.shape
    xp=F.pad(x,(1,1,1,1))
    # tiles: 4x4 patches stride 2
    p=xp.unfold(2,4,2).unfold(3,4,2)  # B,C,H/2,W/2,4,4
    V=torch.einsum('ir,bcyxrs,js->bcyxij',BT,p,BT)   # fp32
    U=torch.einsum('ir,ocrs,js->ocij',G,w,G)
    M=torch.einsum('bcyxij,ocij->boyxij',V,U)
    Y=torch.einsum('pi,boyxij,qj->boyxpq',AT,M,AT)   # B,O,H/2,W/2,2,2
    Y=Y.permute(0,1,2,4,3,5).reshape(Bn,w.shape[0],H,W)
    return Y+b.view(1,-1,1,1)
orig=F.conv2d
mode={'w':False}
def conv2d(x,w,b=None,stride=1,padding=0,**kw):
    if mode['w'] and w.shape[2]==3 and stride==1 and padding==1 and x.shape[2]%2==0 and x.dtype==torch.float32 and w.shape[0]>4 and w.shape[1]>8:
        return wino_conv(x,w,b if b is not None else torch.zeros(w.shape[0]))
    return orig(x,w,b,stride=stride,padding=padding,**kw)
O.F.conv2d=conv2d
for phase in ('val','train'):
    opt=bench.config_opt('sr3_16_128',phase=phase); opt['gpu_ids']=None
    torch.manual_seed(11)
    netG=networks.define_G(opt)
    sd={k:v.clone() for k,v in netG.state_dict().items()}
    desc=O.desc_from_opt(opt)
    g=torch.Generator().manual_seed(5)
    x=torch.randn(1,6,128,128,generator=g); t=torch.tensor([[0.7312]])
    with torch.no_grad():
        mode['w']=False
        ref64=O.unet_forward({k:(v.double() if v.is_floating_point() else v) for k,v in sd.items()},desc,x.double(),t.double())
        d32=O.unet_forward(sd,desc,x,t)
        mode['w']=True
        t0=time.time(); w32=O.unet_forward(sd,desc,x,t); print('wino time',time.time()-t0)
    print(phase,'|ref|max %.3f  direct32 err %.2e  wino32 err %.2e  (tol %.2e)'%(ref64.abs().max(), (d32-ref64).abs().max(), (w32-ref64).abs().max(), 2e-5*max(1,ref64.abs().max())))
