import sys, torch
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import mixgan_tts_amd as mg
from mixgan_tts_amd import ops, autograd as A
import torch.nn.functional as F
a=torch.randn(2,64,80).cuda(); b=torch.randn(2,64,80).cuda()
ref=torch.cat([a,b],-1).transpose(1,2).contiguous()
out=ops.cat_transpose(a,b)
print('cat_transpose err', (out-ref).abs().max().item())
w=torch.randn(160,160).cuda()/12
y=A.conv1d(out, w[:,:,None], None)
yr=F.conv1d(ref, w[:,:,None])
print('k1 conv err', (y-yr).abs().max().item(), yr.abs().max().item())
w3=torch.randn(64,160,3).cuda()/20; b3=torch.randn(64).cuda()
y3=A.conv1d(yr, w3, b3, 1, 1, 'lrelu')
y3r=F.leaky_relu(F.conv1d(yr,w3,b3,padding=1),0.2)
print('k3 conv err', (y3-y3r).abs().max().item())
