"""Oracle: key -> shape map of ``FullModel_VMD('vmn_gca').NET.state_dict()``
(584 tensors).  The checkpoint layout is part of the
drop-in contract (train_ddp.py:263,338); tests pin it against the list captured
from the real reference in tests/golden/state_keys.npz.

TEST INFRASTRUCTURE — never imported by the product path.
"""
from collections import OrderedDict


def _sn(d, p, shape):
    # SpectralNorm._make_params (models/GCA/ops.py:57-72): u [height], v [width], bar
    h = shape[0]
    w = 1
    for s in shape[1:]:
        w *= s
    d[p + '.module.weight_u'] = (h,)
    d[p + '.module.weight_v'] = (w,)
    d[p + '.module.weight_bar'] = tuple(shape)


def _bn(d, p, c):
    d[p + '.weight'] = (c,)
    d[p + '.bias'] = (c,)
    d[p + '.running_mean'] = (c,)
    d[p + '.running_var'] = (c,)
    d[p + '.num_batches_tracked'] = ()


def _gca(d, p):
    d[p + '.guidance_conv.weight'] = (64, 128, 1, 1)
    d[p + '.guidance_conv.bias'] = (64,)
    d[p + '.W.0.weight'] = (128, 128, 1, 1)
    _bn(d, p + '.W.1', 128)


def vmn_gca_state_spec():
    d = OrderedDict()
    e = 'encoder'
    # stem (resnet_enc.py:70-78)
    _sn(d, e + '.conv1', (32, 6, 3, 3))
    _sn(d, e + '.conv2', (32, 32, 3, 3))
    _sn(d, e + '.conv3', (64, 32, 3, 3))
    _bn(d, e + '.bn1', 32)
    _bn(d, e + '.bn2', 32)
    _bn(d, e + '.bn3', 64)
    inpl = 64
    for lname, planes, blocks, stride in (('layer1', 64, 3, 1), ('layer2', 128, 4, 2),
                                          ('layer3', 256, 4, 2), ('layer_bottleneck', 512, 2, 2)):
        for b in range(blocks):
            p = '%s.%s.%d' % (e, lname, b)
            cin = inpl if b == 0 else planes
            _sn(d, p + '.conv1', (planes, cin, 3, 3))
            _bn(d, p + '.bn1', planes)
            _sn(d, p + '.conv2', (planes, planes, 3, 3))
            _bn(d, p + '.bn2', planes)
            if b == 0 and stride != 1:
                _sn(d, p + '.downsample.1', (planes, cin, 1, 1))
                _bn(d, p + '.downsample.2', planes)
        inpl = planes
    for i, (cin, cout) in enumerate(((6, 32), (32, 32), (64, 64), (128, 128), (256, 256))):
        p = '%s.shortcut.%d' % (e, i)
        _sn(d, p + '.0', (cout, cin, 3, 3))
        _bn(d, p + '.2', cout)
        _sn(d, p + '.3', (cout, cout, 3, 3))
        _bn(d, p + '.5', cout)
    for ci, bi, cin, cout in ((1, 3, 3, 16), (5, 7, 16, 32), (9, 11, 32, 128)):
        _sn(d, '%s.guidance_head.%d' % (e, ci), (cout, cin, 3, 3))
        _bn(d, '%s.guidance_head.%d' % (e, bi), cout)
    _gca(d, e + '.gca')

    dd = 'decoder'
    _sn(d, dd + '.conv1', (32, 32, 4, 4))               # ConvTranspose2d weight [in, out, 4, 4]
    _bn(d, dd + '.bn1', 32)
    d[dd + '.conv2.weight'] = (1, 32, 3, 3)
    d[dd + '.conv2.bias'] = (1,)
    inpl = 512
    for lname, planes, blocks in (('layer1', 256, 2), ('layer2', 128, 3), ('layer3', 64, 3), ('layer4', 32, 2)):
        for b in range(blocks):
            p = '%s.%s.%d' % (dd, lname, b)
            cin = inpl if b == 0 else planes
            if b == 0:
                _sn(d, p + '.conv1', (cin, cin, 4, 4))   # ConvTranspose2d(cin, cin)
            else:
                _sn(d, p + '.conv1', (cin, cin, 3, 3))
            _bn(d, p + '.bn1', cin)
            _sn(d, p + '.conv2', (planes, cin, 3, 3))
            _bn(d, p + '.bn2', planes)
            if b == 0:
                _sn(d, p + '.upsample.1', (planes, cin, 1, 1))
                _bn(d, p + '.upsample.2', planes)
        inpl = planes
    _gca(d, dd + '.gca')
    for n in ('key_conv', 'query_conv', 'value_conv'):
        d['%s.fam.%s.weight' % (dd, n)] = (128, 128, 3, 3)
        d['%s.fam.%s.bias' % (dd, n)] = (128,)
    return d
