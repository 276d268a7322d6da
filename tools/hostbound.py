import sys, time, torch
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
import bench
from tcvom_amd.facade import train_step_loss
from tcvom_amd.optim import FusedAdam
dev = torch.device('cuda', 0)
model, a, fg, bg = bench.build(dev, 1088, 1920, 0)
params = [p for p in model.parameters() if p.requires_grad]
opt = FusedAdam(params, lr=1e-4, weight_decay=1e-4)
def step():
    out = model(a, fg, bg); loss = train_step_loss(out); model.zero_grad(set_to_none=True); loss.backward(); opt.step()
for _ in range(2): step()
torch.cuda.synchronize()
t0 = time.time()
for _ in range(5): step()
t1 = time.time()
torch.cuda.synchronize()
t2 = time.time()
print('host enqueue %.1f ms/step, total %.1f ms/step; adam table rebuilds %d in 7 steps' % ((t1-t0)*200, (t2-t0)*200, opt.table_rebuilds))
# forward-only / backward-only host split
torch.cuda.synchronize(); t0=time.time(); out = model(a,fg,bg); loss = train_step_loss(out); t1=time.time(); torch.cuda.synchronize(); t2=time.time()
print('fwd host %.1f ms, fwd total %.1f ms' % ((t1-t0)*1e3, (t2-t0)*1e3))
model.zero_grad(set_to_none=True); torch.cuda.synchronize(); t0=time.time(); loss.backward(); t1=time.time(); torch.cuda.synchronize(); t2=time.time()
print('bwd host %.1f ms, bwd total %.1f ms' % ((t1-t0)*1e3, (t2-t0)*1e3))
