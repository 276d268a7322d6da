"""Host time of each segment of the training step in a free-running loop (no synchronisation): a segment whose host time is far
above its enqueue cost is where the host waits for the GPU."""
import sys, time, torch
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
import bench
from tcvom_amd.facade import train_step_loss
from tcvom_amd.optim import FusedAdam
dev = torch.device('cuda', 0)
model, a, fg, bg = bench.build(dev, 1088, 1920, 0)
params = [p for p in model.parameters() if p.requires_grad]
opt = FusedAdam(params, lr=1e-4, weight_decay=1e-4)
seg = {}
def lap(name, t0):
    t = time.time(); seg.setdefault(name, []).append((t - t0) * 1e3); return t
def step():
    t = time.time()
    out = model(a, fg, bg); t = lap('forward', t)
    loss = train_step_loss(out); t = lap('loss', t)
    model.zero_grad(set_to_none=True); t = lap('zero_grad', t)
    loss.backward(); t = lap('backward', t)
    opt.step(); t = lap('adam', t)
for _ in range(3): step()
torch.cuda.synchronize(); seg.clear()
t0 = time.time()
for _ in range(6): step()
t1 = time.time(); torch.cuda.synchronize(); t2 = time.time()
print('host %.1f ms/step, total %.1f ms/step' % ((t1 - t0) / 6 * 1e3, (t2 - t0) / 6 * 1e3))
for k, v in seg.items():
    print('%-10s %s' % (k, ' '.join('%6.2f' % x for x in v)))
