"""Which fp32 arithmetic does ATen's CPU bilinear resize use for the tensors dataset/VMD.py:62-66 feeds it
([H, W, C] numpy -> permute -> unsqueeze: a channels-last strided NCHW view)?  Emulates candidate orders in numpy and counts
the elements that differ from F.interpolate; the order with 0 differences is what csrc/frontend.hip implements."""
import itertools

import numpy as np
import torch
import torch.nn.functional as F

f32 = np.float32


def mul(x, y):
    return (x.astype(f32) * y.astype(f32)).astype(f32)


def fma(x, y, z):
    return (x.astype(np.float64) * y.astype(np.float64) + z.astype(np.float64)).astype(f32)


def main():
    torch.manual_seed(0)
    print('threads', torch.get_num_threads())
    for C in (3, 1):
        nh, nw, Ho, Wo = 200, 200, 161, 161
        hwc = torch.randint(0, 256, (nh + 7, nw + 9, C)).float().numpy()
        crop = hwc[3:3 + nh, 5:5 + nw]
        src = torch.from_numpy(crop).permute(2, 0, 1).unsqueeze(0)
        ref = F.interpolate(src, [Ho, Wo], mode='bilinear', align_corners=True)[0].numpy()
        rh, rw = f32(nh - 1) / f32(Ho - 1), f32(nw - 1) / f32(Wo - 1)
        h1r, w1r = (rh * np.arange(Ho, dtype=f32)).astype(f32), (rw * np.arange(Wo, dtype=f32)).astype(f32)
        h1, w1 = h1r.astype(np.int64), w1r.astype(np.int64)
        h1p, w1p = (h1 < nh - 1).astype(np.int64), (w1 < nw - 1).astype(np.int64)
        h1l, w1l = (h1r - h1.astype(f32)).astype(f32), (w1r - w1.astype(f32)).astype(f32)
        h0l, w0l = (f32(1) - h1l).astype(f32), (f32(1) - w1l).astype(f32)
        H0, H1, W0, W1 = h0l[:, None], h1l[:, None], w0l[None, :], w1l[None, :]
        for c in range(C):
            a = crop[..., c]
            A, B, Cc, D = a[h1][:, w1], a[h1][:, w1 + w1p], a[h1 + h1p][:, w1], a[h1 + h1p][:, w1 + w1p]
            res = {}
            for mi, mo in itertools.product(range(3), range(3)):           # two-stage forms
                def comb(m, x0, p, x1, q):
                    return (mul(x0, p) + mul(x1, q)).astype(f32) if m == 0 else (fma(x0, p, mul(x1, q)) if m == 1 else fma(x1, q, mul(x0, p)))
                res['two-stage %d%d' % (mi, mo)] = comb(mo, H0, comb(mi, W0, A, W1, B), H1, comb(mi, W0, Cc, W1, D))
            w00, w01, w10, w11 = mul(H0, W0), mul(H0, W1), mul(H1, W0), mul(H1, W1)
            res['4-weight sum, no fma'] = (((mul(w00, A) + mul(w01, B)).astype(f32) + mul(w10, Cc)).astype(f32) + mul(w11, D)).astype(f32)
            res['4-weight fma chain'] = fma(w11, D, fma(w10, Cc, fma(w01, B, mul(w00, A))))
            for m1, m2, m3 in itertools.product(range(3), range(2), range(2)):
                s1 = (mul(w00, A) + mul(w01, B)).astype(f32) if m1 == 0 else (fma(w00, A, mul(w01, B)) if m1 == 1 else fma(w01, B, mul(w00, A)))
                s2 = (s1 + mul(w10, Cc)).astype(f32) if m2 == 0 else fma(w10, Cc, s1)
                s3 = (s2 + mul(w11, D)).astype(f32) if m3 == 0 else fma(w11, D, s2)
                res['4-weight left-assoc %d%d%d' % (m1, m2, m3)] = s3
            res['4-weight fma chain rev'] = fma(w00, A, fma(w01, B, fma(w10, Cc, mul(w11, D))))
            best = sorted((int((v != ref[c]).sum()), k) for k, v in res.items())[:3]
            print('C=%d plane %d:' % (C, c), best)


if __name__ == '__main__':
    main()
    torch.set_num_threads(1)                 # what a DataLoader worker process runs with
    main()
