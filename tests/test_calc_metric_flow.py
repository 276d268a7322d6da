"""calc_metric.py flow decoding: the VideoMatting108 flow PNGs are written by OpenCV, i.e. the FILE holds
R = validity, G = y displacement, B = x displacement (cv2 hands them over as B, G, R and the reference slices that,
calc_metric.py:65-71).  A fixture written in that layout must decode to (x, y) flow with NaN at invalid pixels, through the
dependency-free 16-bit PNG reader (every row-filter type)."""
import struct
import zlib

import numpy as np
import pytest


def write_png16(path, img, filters):
    """img uint16 [H, W, C] in FILE channel order; filters[y] = PNG filter type of row y."""
    H, W, C = img.shape
    be = np.zeros((H, W, C, 2), np.uint8)
    be[..., 0], be[..., 1] = img >> 8, img & 255
    rows = be.reshape(H, -1).astype(np.int64)
    bpp = 2 * C
    out, prev = bytearray(), np.zeros(W * bpp, np.int64)
    for y in range(H):
        cur, ft = rows[y], filters[y]
        left = np.concatenate([np.zeros(bpp, np.int64), cur[:-bpp]])
        ul = np.concatenate([np.zeros(bpp, np.int64), prev[:-bpp]])
        if ft == 0:
            f = cur
        elif ft == 1:
            f = cur - left
        elif ft == 2:
            f = cur - prev
        elif ft == 3:
            f = cur - ((left + prev) >> 1)
        else:
            pa, pb, pc = np.abs(prev - ul), np.abs(left - ul), np.abs(left + prev - 2 * ul)
            pred = np.where((pa <= pb) & (pa <= pc), left, np.where(pb <= pc, prev, ul))
            f = cur - pred
        out.append(ft)
        out += bytes((f & 255).astype(np.uint8))
        prev = cur
    def chunk(t, b):
        return struct.pack('>I', len(b)) + t + b + struct.pack('>I', zlib.crc32(t + b) & 0xffffffff)
    with open(path, 'wb') as fh:
        fh.write(b'\x89PNG\r\n\x1a\n' + chunk(b'IHDR', struct.pack('>IIBBBBB', W, H, 16, {3: 2, 4: 6}[C], 0, 0, 0)) +
                 chunk(b'IDAT', zlib.compress(bytes(out))) + chunk(b'IEND', b''))


@pytest.mark.parametrize('channels', [3, 4])
def test_flow_png_is_decoded_in_opencv_channel_order(tmp_path, channels):
    import calc_metric
    rng = np.random.RandomState(0)
    H, W = 10, 7
    fx = rng.randint(-3000, 3000, (H, W)).astype(np.int16)
    fy = rng.randint(-3000, 3000, (H, W)).astype(np.int16)
    valid = (rng.rand(H, W) > 0.3)
    cv_order = [fx.view(np.uint16), fy.view(np.uint16)]
    if channels == 4:
        cv_order.append(np.full((H, W), 777, np.uint16))              # an unused third flow channel of a BGRA file
    cv_order.append(np.where(valid, 65535, 0).astype(np.uint16))      # cv2: last channel = mask
    cv_img = np.stack(cv_order, -1)
    file_img = cv_img[..., ::-1] if channels == 3 else cv_img[..., [2, 1, 0, 3]]
    path = str(tmp_path / 'flow_0001_0002.png')
    write_png16(path, np.ascontiguousarray(file_img), [y % 5 for y in range(H)])
    assert np.array_equal(calc_metric._read_png16(path), file_img)
    flow = calc_metric._flow(path)
    assert flow.shape == (H, W, 2) and flow.dtype == np.float32
    assert np.array_equal(np.isnan(flow[..., 0]), ~valid) and np.array_equal(np.isnan(flow[..., 1]), ~valid)
    assert np.allclose(flow[..., 0][valid], fx[valid] / 100.0) and np.allclose(flow[..., 1][valid], fy[valid] / 100.0)
    assert calc_metric._flow(str(tmp_path / 'missing.png')) is None
    bad = str(tmp_path / 'bad.png')
    open(bad, 'wb').write(b'not a png')
    with pytest.raises(Exception):
        calc_metric._flow(bad)
