"""Drop-in import surface: `from models.model import FullModel_VMD` etc. resolve to the MI355X
implementation in `tcvom_amd` exactly where the reference's entry scripts expect them
(train_ddp.py:22, pred_vmn.py:23, pred_test.py)."""
