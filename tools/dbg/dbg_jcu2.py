import sys, torch, json, numpy as np
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import mixgan_tts_amd as mg
from mixgan_tts_amd import ops, autograd as A
from helpers import golden, T, hot_path_configs, load_seeded, seeded
from oracle import refmath as R
import torch.nn.functional as F
manifest=json.load(open('/root/repo/tests/golden/manifest.json'))
g=golden('jcu_ms0_L64')
_, pre, mc, tr = hot_path_configs(stats_dir='.')
D = mg.JCUDiscriminator(pre, mc, tr)
load_seeded(D, manifest, 'jcu_ms0', 41)
W,_=seeded(manifest,'jcu_ms0',41)
D=D.cuda()
x_ts, fake = T(g['x_ts']).cuda(), T(g['fake']).cuda()
t=T(g['t']).cuda()
with torch.no_grad():
    x = A.cat_transpose(fake, x_ts)
    xr = torch.cat([T(g['fake']), T(g['x_ts'])], -1).transpose(1,2)
    print('cat', (x.cpu()-xr).abs().max().item())
    x1 = A.conv1d(x, D.input_projection.linear.weight[:, :, None], None)
    x1r = F.linear(torch.cat([T(g['fake']), T(g['x_ts'])], -1), W['input_projection.linear.weight']).transpose(1,2)
    print('inproj', (x1.cpu()-x1r).abs().max().item(), x1r.abs().max().item())
    l0=D.conv_block[0]
    print(l0.stride, l0.padding, l0.conv.weight.shape)
    y = D._lrelu_conv(l0, x1)
    yr = F.leaky_relu(F.conv1d(x1r, W['conv_block.0.conv.weight'], W['conv_block.0.conv.bias'], padding=1), 0.2)
    print('conv0', (y.cpu()-yr).abs().max().item(), yr.abs().max().item(), (yr-T(g['fc0'])).abs().max().item())
    fc, fu = D(x_ts, fake, None, t)
    print('fc0 full', (fc[0].cpu()-T(g['fc0'])).abs().max().item())
