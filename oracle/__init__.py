"""CPU oracle for the TCVOM per-frame-window hot path (GCA base + TAM).

TEST INFRASTRUCTURE ONLY.  This package is a from-scratch, functional, pure
PyTorch-fp32 restatement of the reference algorithm (yunkezhang/TCVOM,
`models/model.py`, `models/VMN/*`, `models/GCA/*`, `utils/loss_func.py`).
It exists to *check* the HIP product path (`tcvom_amd/`) and to be timed as
the `cpu_baseline` leg of `bench.py`.  Only `tests/`, `__graft_entry__.smoke()`
and `bench.py`'s cpu_baseline leg may import it; the product package never
does (tests/test_layout.py enforces that).

Parity status: PINNED.  `tests/golden/gen_golden.py` imports the real
reference from /root/reference (this container only), runs it on
formula-initialised weights/inputs and stores inputs' parameters + outputs in
`tests/golden/*.npz`; `tests/test_oracle_golden.py` checks every function of
this package against those vectors.  The reference ships no tests or golden
vectors of its own.

`oracle.dim_net` (DIM base, config 1) and `oracle.fba_net` (FBA base + TAM, config 5) are imported explicitly
by their tests; both are pinned the same way (tests/golden/dim_*.npz, fba_*.npz).

Everything operates on a flat ``state`` dict that uses exactly the key layout
of the reference's ``FullModel_VMD(...).NET.state_dict()`` (584 tensors for
``vmn_gca``) so identical weights can be fed to reference, oracle and product.
"""
from .gca_net import (spectral_weight, guided_context_attention, encoder_frame,
                      decoder_front, decoder_tail)
from .tam import temporal_attention, tam_forward
from .window import (preprocess, make_trimap, l1_mask, attention_loss, dtssd_loss,
                     vmn_forward, window_forward, train_step_loss)

__all__ = [n for n in dir() if not n.startswith('_')]
