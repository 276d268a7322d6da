"""TEST INFRASTRUCTURE — numpy restatement of the reference's evaluation metrics (calc_metric.py:22-46 and
utils/utils.py:70-123: SAD, MSE, SSDA, dtSSD, flow-warped MESSDdt).  Only tests/ may import it.  Pinned by
tests/golden/metrics.npz (values from the reference's own functions, tests/golden/gen_golden.py:gen_metrics)."""
import numpy as np


def _sample_zero(im, x, y):
    """F.grid_sample(bilinear, align_corners=True, padding_mode='zeros') at pixel coordinates (x, y)."""
    H, W = im.shape
    x0, y0 = np.floor(x).astype(np.int64), np.floor(y).astype(np.int64)
    lx, ly = x - x0, y - y0
    out = np.zeros_like(x, dtype=np.float64)
    for dy, wy in ((0, 1 - ly), (1, ly)):
        for dx, wx in ((0, 1 - lx), (1, lx)):
            yy, xx = y0 + dy, x0 + dx
            ok = (yy >= 0) & (yy < H) & (xx >= 0) & (xx < W)
            out += np.where(ok, wy * wx * im[np.clip(yy, 0, H - 1), np.clip(xx, 0, W - 1)], 0.0)
    return out


def frame_metrics(a, g, tri, ha=None, hg=None, flow=None):
    """a, g, ha, hg float [H,W]; tri uint8 [H,W]; flow float [H,W,2] with NaN = invalid."""
    m = (tri > 0) & (tri < 255)
    d = (a.astype(np.float64) - g)[m]
    out = {'pixels': int(m.sum()), 'SAD': float(np.mean(np.abs(d))), 'MSE': float(np.mean(d ** 2)), 'SSDA': float(np.sqrt(np.sum(d ** 2)))}
    if ha is not None:
        out['dtSSD'] = float(np.sqrt(np.sum((((a.astype(np.float64) - ha) - (g.astype(np.float64) - hg))[m]) ** 2)))
    if flow is not None:
        H, W = a.shape
        ys, xs = np.meshgrid(np.arange(H), np.arange(W), indexing='ij')
        bad = np.isnan(flow).any(-1)
        fx, fy = np.where(bad, 0.0, flow[..., 0]), np.where(bad, 0.0, flow[..., 1])
        pa = _sample_zero(ha.astype(np.float64), xs + fx, ys + fy)
        pg = _sample_zero(hg.astype(np.float64), xs + fx, ys + fy)
        v = m & ~bad
        e1, e2 = (a.astype(np.float64) - g)[v], (pa - pg)[v]
        out['MESSDdt'] = (float(np.abs(e1 - e2).sum()), float(np.abs(e1 ** 2 - e2 ** 2).sum()), int(v.sum()))
    return out
