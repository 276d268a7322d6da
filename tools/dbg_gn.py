import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import torch, torch.nn as nn, torch.nn.functional as F
from helpers import hu
from tcvom_amd import ops
from tcvom_amd.weights import ConvSpec, WeightBank, bank_token
from tcvom_amd.synthetic import formula_tensor
DEV='cuda'
bf = lambda t: t.to(torch.bfloat16).float()
nhwc = lambda t: t.permute(0, 2, 3, 1).contiguous().to(torch.bfloat16).to(DEV)
nchw = lambda t: t.detach().permute(0, 3, 1, 2).float().cpu()
cin, cout, k, NF, H, W = 128, 64, 1, 3, 12, 20
w = nn.Parameter((formula_tensor('w', (cout, cin, k, k))).to(DEV))
gn = nn.GroupNorm(32, cout).to(DEV)
bank = WeightBank()
spec = ConvSpec('t', w, None, None, None, False, 1, 0, 'frame', ws=True)
bank.register(spec)
cfg = ops.ConvCfg(bank, spec, bn=gn, act=1)
x = bf(hu('x', (NF, cin, H, W)))
r = bf(hu('r', (NF, cout, H, W)))
xg, rg = nhwc(x).requires_grad_(True), nhwc(r).requires_grad_(True)
token = bank_token(bank, NF, True)
bank.frames_per_op = NF
z = ops.conv_bn_act(cfg, xg, token, True, res1=rg)
bank.frames_per_op = 1
wr = w.detach().cpu().clone().requires_grad_(True)
ws = wr - wr.mean(dim=(1, 2, 3), keepdim=True)
ws = ws / (torch.sqrt(torch.var(ws.reshape(cout, -1), dim=1) + 1e-12).reshape(-1, 1, 1, 1) + 1e-5)
xr, rr = x.clone().requires_grad_(True), r.clone().requires_grad_(True)
y = F.group_norm(F.conv2d(xr, ws), 32, gn.weight.detach().cpu(), gn.bias.detach().cpu(), 1e-5)
pre = y + rr
zr = F.relu(pre)
g = bf(hu('g', tuple(zr.shape)))
z.backward(nhwc(g)); zr.backward(g)
dres = nchw(rg.grad)
print('z err', float((nchw(z) - zr).abs().max()))
d = (dres - rr.grad).abs()
print('dres err max', float(d.max()), 'mean', float(d.mean()), 'frac wrong', float((d > 1e-3).float().mean()))
for f in range(NF):
    print(' frame', f, 'frac wrong', float((d[f] > 1e-3).float().mean()))
bad = (d > 1e-3)
print('pre at bad: |pre| mean', float(pre.detach()[bad].abs().mean()), 'overall |pre| mean', float(pre.detach().abs().mean()))
print('dres sample', dres[0, :3, 0, :4], rr.grad[0, :3, 0, :4], g[0,:3,0,:4])
