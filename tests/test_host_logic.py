"""Host-side logic of the frame-batched window that needs no GPU: frame stacking without a copy, the row-range gradient
deposit of `ops.frame_slice`, and the fused loss sum of train_ddp.py:56-61."""
import pytest
import torch


def test_stack_frames_is_a_view_of_consecutive_slices():
    from tcvom_amd.vmn import _stack_frames
    buf = torch.arange(2 * 3 * 4 * 5, dtype=torch.float32).reshape(1, 6, 4, 5)          # facade layout [B = 1, S, ...]
    frames = [buf[:, s] for s in range(3)]
    X = _stack_frames(frames)
    assert X.shape == (3, 4, 5) and X.data_ptr() == buf.data_ptr() and torch.equal(X, buf[0, :3])
    # B = 2: the frames of one clip are not consecutive in a [B, S, ...] buffer -> a real concatenation, frame-major
    buf2 = torch.arange(2 * 3 * 4, dtype=torch.float32).reshape(2, 3, 4)
    frames2 = [buf2[:, s].contiguous() for s in range(3)]
    X2 = _stack_frames(frames2)
    assert X2.shape == (6, 4) and torch.equal(X2, torch.cat(frames2, 0))
    assert _stack_frames([buf2[:, 0]]) is not None


def test_frame_slice_deposits_a_row_range_gradient_or_pads_with_zeros():
    from tcvom_amd import ops
    # a producer that takes row-range gradients: frame_slice hands the small gradient over and returns nothing to autograd
    t = torch.randn(6, 3, requires_grad=True)
    y = t * 2.0
    stash = ops._GradStash()
    y._tcvom_grad_stash = stash
    y._tcvom_tail_rows = (2, 4)
    z = ops.frame_slice(y, 2, 4)
    assert z.shape == (2, 3) and torch.equal(z, y[2:4])
    z.sum().backward()
    assert t.grad is None or float(t.grad.abs().sum()) == 0.0          # nothing flowed through autograd ...
    assert len(stash) == 1 and stash[0][0] == 'rows' and stash[0][2:] == (2, 4) and torch.equal(stash[0][1], torch.ones(2, 3))
    # any other tensor: the plain slice, whose gradient is the zero-padded one
    t2 = torch.randn(6, 3, requires_grad=True)
    ops.frame_slice(t2 * 1.0, 2, 4).sum().backward()
    want = torch.zeros(6, 3)
    want[2:4] = 1.0
    assert torch.equal(t2.grad, want)
    # a different row range than a tail-only producer skips: not deposited
    y3 = (t2 * 1.0)
    y3._tcvom_grad_stash, y3._tcvom_tail_rows = ops._GradStash(), (2, 4)
    assert ops.frame_slice(y3, 0, 2).grad_fn.__class__.__name__ != '_FrameSliceBackward'
    # a conv + BatchNorm producer that runs for all frames (a stash, no tail rows): deposited, any frame range
    y4 = (t2 * 1.0)
    y4._tcvom_grad_stash = ops._GradStash()
    ops.frame_slice(y4, 0, 2).sum().backward()
    assert len(y4._tcvom_grad_stash) == 1 and y4._tcvom_grad_stash[0][2:] == (0, 2)


@pytest.mark.parametrize('S', [3, 5])
def test_neighbour_slices_gradient_equals_three_plain_slices(S):
    from tcvom_amd import ops
    B = 2
    t = torch.randn(S * B, 4, requires_grad=True)
    w = [torch.randn((S - 2) * B, 4) for _ in range(3)]
    c, p, n = ops.neighbour_slices(t * 1.0, B, S)
    assert torch.equal(c, t[B:(S - 1) * B]) and torch.equal(p, t[:(S - 2) * B]) and torch.equal(n, t[2 * B:])
    ((c * w[0]).sum() + (p * w[1]).sum() + (n * w[2]).sum()).backward()
    got = t.grad.clone()
    t.grad = None
    ((t[B:(S - 1) * B] * w[0]).sum() + (t[:(S - 2) * B] * w[1]).sum() + (t[2 * B:] * w[2]).sum()).backward()
    assert torch.allclose(got, t.grad, atol=1e-6)
    # one of the three unused: still the zero-padded sum of the others
    t.grad = None
    c, p, n = ops.neighbour_slices(t * 1.0, B, S)
    (c * w[0]).sum().backward()
    want = torch.zeros_like(t)
    want[B:(S - 1) * B] = w[0]
    assert torch.allclose(t.grad, want)


def test_train_step_loss_matches_the_reference_formula():
    from tcvom_amd.facade import train_step_loss
    vals = [torch.tensor(v, requires_grad=True) for v in (0.3, 0.0, 0.0, 0.7, 1.9)]
    loss = train_step_loss(vals + [None] * 7)
    want = vals[0].mean() + vals[1].mean() + vals[2].mean() + 0.5 * vals[3].mean() + 0.25 * vals[4].mean()
    assert abs(float(loss) - float(want)) < 1e-7
    loss.backward()
    assert [float(v.grad) for v in vals] == [1.0, 1.0, 1.0, 0.5, 0.25]


def test_loss_scaler_halves_once_per_overflow_and_ignores_the_stale_read_back():
    """ops.LossScaler reads the device counter one step late: when step t's overflow becomes known, step t+1's backward has
    already run at the OLD scale.  Its read-back must not halve the scale a second time."""
    from tcvom_amd import ops

    class Ev(object):
        def synchronize(self):
            pass

    sc = ops.LossScaler(65536.0)
    host = torch.zeros((2, 2), dtype=torch.int32)
    sc.counters[0] = (None, host, [None, None])

    def finish_step(bwd):              # what after_step() leaves behind for the next before_step()
        slot = sc.t & 1
        host[slot, 0] = bwd
        sc.counters[0][2][slot] = Ev()
        sc.t += 1

    assert sc.before_step('cuda:0') is False                 # nothing recorded yet
    finish_step(0)
    assert sc.before_step('cuda:0') is False and sc.scale == 65536.0
    finish_step(5)                                           # step 1 overflowed
    finish_step_skipped = sc.before_step('cuda:0')
    assert finish_step_skipped is True and sc.scale == 32768.0 and sc.skipped_steps == 1
    finish_step(7)                                           # step 2 ran its backward at the old scale: overflowed again
    assert sc.before_step('cuda:0') is True                  # the device dropped it (counter != 0) -> step counters taken back ...
    assert sc.scale == 32768.0 and sc.skipped_steps == 2     # ... but the scale is halved ONCE
    finish_step(0)
    assert sc.before_step('cuda:0') is False and sc.scale == 32768.0
    finish_step(3)                                           # a fresh overflow at the new scale halves again
    assert sc.before_step('cuda:0') is True and sc.scale == 16384.0


@pytest.mark.parametrize('arch', ['vmn_gca', 'vmn_dim', 'vmn_fba', 'vmn_index'])
def test_agg_reduction_other_than_one_is_refused_like_the_reference_fails(arch):
    """models/VMN/__init__.py:11 accepts agg_reduction, but the reference's own forward raises for every value != 1 (the TAM reshapes
    C / reduction-channel keys as C channels, VMN_model.py:33-37 -- verified on the imported reference for all four archs): the
    product refuses at construction instead of at the first window.  agg_window != 7 IS supported: golden window_s3_64x96_w5."""
    from tcvom_amd.vmn import get_VMN_models
    with pytest.raises(ValueError, match='agg_reduction'):
        get_VMN_models(arch, 7, agg_reduction=2)
    from models.model import FullModel_VMD
    with pytest.raises(ValueError, match='agg_reduction'):
        FullModel_VMD(arch, agg_window=5, agg_reduction=2)
