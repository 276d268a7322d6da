import sys; sys.path.insert(0,'/root/repo')
import torch, torch.nn.functional as TF
import oracle, oracle.gca_net as G, oracle.tam as T
from oracle.state_spec import vmn_gca_state_spec
from tcvom_amd.synthetic import formula_tensor, synthetic_window
torch.set_num_threads(8)
def bf(t): return t.to(torch.bfloat16).float()
class FP:
    def __init__(s, W, Y, Z, B): s.W, s.Y, s.Z, s.B = W, Y, Z, B
    def __getattr__(s, n): return getattr(TF, n)
    def conv2d(s, x, w, b=None, *a, **k):
        if s.W: w = bf(w)
        y = TF.conv2d(x, w, b, *a, **k)
        return bf(y) if s.Y else y
    def conv_transpose2d(s, x, w, b=None, *a, **k):
        if s.W: w = bf(w)
        y = TF.conv_transpose2d(x, w, b, *a, **k)
        return bf(y) if s.Y else y
    def relu(s, x): 
        y = TF.relu(x); return bf(y) if s.Z else y
    def leaky_relu(s, x, a): 
        y = TF.leaky_relu(x, a); return bf(y) if s.Z else y
    def batch_norm(s, *a, **k):
        y = TF.batch_norm(*a, **k); return bf(y) if s.B else y
def run(W,Y,Z,B,H=256,Wd=320):
    G.F = FP(W,Y,Z,B); T.F = FP(W,Y,Z,B)
    state = {k: formula_tensor(k, s, torch.int64 if k.endswith('num_batches_tracked') else torch.float32) for k, s in vmn_gca_state_spec().items()}
    a,fg,bg = synthetic_window(1,3,H,Wd,seed=0)
    with torch.no_grad():
        out,_ = oracle.window_forward(state,a,fg,bg,window=7,dilate_kernel=12,training=True)
    return out
ref = run(0,0,0,0)
um = ref[6].isclose(torch.tensor(128/255.))
for cfg in [(1,0,0,0),(0,1,0,0),(0,0,1,0),(0,0,0,1),(1,1,1,1),(1,0,1,1),(0,0,1,1)]:
    o = run(*cfg)
    d = (o[7]-ref[7])
    print('W,Y,Z,B=',cfg,'mse_unk %.3e mse_all %.3e'%(float((d[um]**2).mean()), float((d**2).mean())))
print('--- per stage (all sources on inside the stage)')
import oracle.window as Wn
orig = dict(enc=G.encoder_frame, front=G.decoder_front, tail=G.decoder_tail, tam=T.tam_forward)
def staged(active):
    on, off = FP(1,1,1,1), FP(0,0,0,0)
    def wrap(name, fn):
        def f(*a, **k):
            G.F = T.F = on if name in active else off
            r = fn(*a, **k)
            G.F = T.F = off
            return r
        return f
    Wn.encoder_frame = wrap('enc', orig['enc']); Wn.decoder_front = wrap('front', orig['front'])
    Wn.decoder_tail = wrap('tail', orig['tail']); Wn.tam_forward = wrap('tam', orig['tam'])
    state = {k: formula_tensor(k, s, torch.int64 if k.endswith('num_batches_tracked') else torch.float32) for k, s in vmn_gca_state_spec().items()}
    a,fg,bg = synthetic_window(1,3,256,320,seed=0)
    with torch.no_grad():
        out,_ = oracle.window_forward(state,a,fg,bg,window=7,dilate_kernel=12,training=True)
    return out
for act in [('enc',),('front',),('tam',),('tail',),('enc','front','tam','tail')]:
    o = staged(act); d = o[7]-ref[7]
    print(act, 'mse_unk %.3e'%float((d[um]**2).mean()))
print('--- inside the encoder')
Wn.encoder_frame, Wn.decoder_front, Wn.decoder_tail, Wn.tam_forward = orig['enc'], orig['front'], orig['tail'], orig['tam']
sub = dict(layer=G._enc_layer, short=G._shortcut, guid=G._guidance_head, gca=G.guided_context_attention)
def sub_staged(active):
    on, off = FP(1,1,1,1), FP(0,0,0,0)
    base = on if 'stem' in active else off
    def wrap(name, fn):
        def f(state, p, *a, **k):
            key = name
            if name == 'layer': key = p.split('.')[-1]
            if name == 'gca': key = 'gca_' + p.split('.')[0]
            if name == 'short': key = 'short' + p.split('.')[-1]
            prev = G.F
            G.F = on if (key in active and p.startswith('encoder')) else (off if p.startswith('encoder') else prev)
            r = fn(state, p, *a, **k)
            G.F = prev
            return r
        return f
    G._enc_layer = wrap('layer', sub['layer']); G._shortcut = wrap('short', sub['short'])
    G._guidance_head = wrap('guid', sub['guid']); G.guided_context_attention = wrap('gca', sub['gca'])
    def enc(state, x, training, prefix='encoder'):
        G.F = base
        r = orig['enc'](state, x, training, prefix)
        G.F = off
        return r
    Wn.encoder_frame = enc
    T.F = off
    state = {k: formula_tensor(k, s, torch.int64 if k.endswith('num_batches_tracked') else torch.float32) for k, s in vmn_gca_state_spec().items()}
    a,fg,bg = synthetic_window(1,3,256,320,seed=0)
    with torch.no_grad():
        out,_ = oracle.window_forward(state,a,fg,bg,window=7,dilate_kernel=12,training=True)
    return out
for act in [('stem',),('guid',),('layer1',),('layer2',),('gca_encoder',),('layer3',),('layer_bottleneck',),('short0',),('short1',),('short2',),('short3',),('short4',)]:
    o = sub_staged(act); d = o[7]-ref[7]
    print(act, 'mse_unk %.3e'%float((d[um]**2).mean()))
print('--- stem+layer1+layer2 by source')
def sub_src(W,Y,Z,B, keys=('stem','layer1','layer2')):
    global FP_ON
    on, off = FP(W,Y,Z,B), FP(0,0,0,0)
    def wrap(name, fn):
        def f(state, p, *a, **k):
            key = name
            if name == 'layer': key = p.split('.')[-1]
            if name == 'gca': key = 'gca_' + p.split('.')[0]
            if name == 'short': key = 'short' + p.split('.')[-1]
            prev = G.F
            G.F = on if (key in keys and p.startswith('encoder')) else (off if p.startswith('encoder') else prev)
            r = fn(state, p, *a, **k)
            G.F = prev
            return r
        return f
    G._enc_layer = wrap('layer', sub['layer']); G._shortcut = wrap('short', sub['short'])
    G._guidance_head = wrap('guid', sub['guid']); G.guided_context_attention = wrap('gca', sub['gca'])
    def enc(state, x, training, prefix='encoder'):
        G.F = on if 'stem' in keys else off
        r = orig['enc'](state, x, training, prefix)
        G.F = off
        return r
    Wn.encoder_frame = enc
    T.F = off
    state = {k: formula_tensor(k, s, torch.int64 if k.endswith('num_batches_tracked') else torch.float32) for k, s in vmn_gca_state_spec().items()}
    a,fg,bg = synthetic_window(1,3,256,320,seed=0)
    with torch.no_grad():
        out,_ = oracle.window_forward(state,a,fg,bg,window=7,dilate_kernel=12,training=True)
    return out
for cfg in [(1,0,0,0),(0,1,0,0),(0,0,1,1)]:
    o = sub_src(*cfg); d = o[7]-ref[7]
    print('stem+l1+l2', cfg, 'mse_unk %.3e'%float((d[um]**2).mean()))
for cfg in [(1,0,0,0),(0,1,0,0),(0,0,1,1)]:
    o = sub_src(*cfg, keys=('stem',)); d = o[7]-ref[7]
    print('stem only', cfg, 'mse_unk %.3e'%float((d[um]**2).mean()))
