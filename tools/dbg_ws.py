import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), 'tests'))
import torch, torch.nn as nn, torch.nn.functional as F
from helpers import hu
from tcvom_amd.synthetic import formula_tensor
from tcvom_amd import _lib as L, ops
from tcvom_amd.conv_plan import ConvGeometry
from tcvom_amd.weights import ConvSpec, WeightBank
DEV='cuda'
bf=lambda t: t.to(torch.bfloat16).float()
torch.set_printoptions(linewidth=200, precision=2, sci_mode=False)
cin,N,H,W = 64,1,8,32
for mode in ('one',):
    tag='d%d_%d_%d'%(cin,H,W)
    w = nn.Parameter((formula_tensor('conv.%s.weight' % tag, (cin, cin, 3, 3)) * 0.2).to(DEV))
    bv = torch.full((cin,), 0.05) if mode == 'const' else (torch.zeros(cin).index_fill_(0, torch.tensor([5]), 0.1) if mode == 'one' else torch.arange(cin).float() * 0.001)
    b = nn.Parameter(bv.to(DEV))
    bank = WeightBank(); spec = ConvSpec(tag, w, None, None, b, False, 1, 1, 'frame'); bank.register(spec); bank.prepare(1, True)
    geo = ConvGeometry(spec, N, H, W)
    x = hu('x.'+tag, (N,cin,H,W)) - 0.5
    xg = x.permute(0,2,3,1).contiguous().to(torch.bfloat16).to(DEV)
    y = torch.empty(N,H,W,cin, device=DEV, dtype=torch.bfloat16)
    ng = L.call('tcvom_conv_stats_groups', geo.fwd[0], 1)
    stats = torch.full((ng*2*cin,), float('nan'), device=DEV)
    ops._launch_conv(geo.fwd, xg, bank.fwd_ptr(spec,0), y, b, stats, 0, L.stream_ptr())
    torch.cuda.synchronize()
    c0 = F.conv2d(bf(x), bf(w.detach().cpu()), None, 1, 1)
    yr = c0 + bv.view(1,-1,1,1)
    got = y.float().cpu().permute(0,3,1,2)
    st = stats.view(ng,2,cin).double().cpu()
    print(mode, 'out err %.4f' % float((got-yr).abs().max()), 'groups', ng)
    print(' (sum - ref)/px per channel:', ((st.sum(0)[0]-yr.double().sum((0,2,3)))/(N*H*W)).float())
    print(' group0 - group-ref0:', ((st[0,0]-yr[:,:,:4].double().sum((0,2,3)))/(N*4*W)).float())
    dev = (st.sum(0)[0]-yr.double().sum((0,2,3)))
    rows = yr.double().sum((0,3))      # [C, H]
    cols = yr.double().sum((0,2))      # [C, W]
    import numpy as np
    A = torch.cat([rows, cols], 1)     # [C, H+W]
    for k in range(H):
        r = rows[:, k]
        print('  row', k, 'corr %.3f scale %.3f' % (float(torch.corrcoef(torch.stack([dev, r]))[0,1]), float((dev*r).sum()/(r*r).sum())))
    dev2 = (st.sum(0)[1]-(yr.double()**2).sum((0,2,3)))
    print('  sq dev/px', (dev2/(N*H*W)).float()[:16])
    # per-lane hypothesis: lanes (columns x) subsets
    for k in range(0, W, 4):
        r = cols[:, k:k+4].sum(1)
        print('  cols', k, 'corr %.3f scale %.3f' % (float(torch.corrcoef(torch.stack([dev, r]))[0,1]), float((dev*r).sum()/(r*r).sum())))
