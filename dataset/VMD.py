"""Import-path shim: `from dataset.VMD import VideoMattingDataset` as in the reference's train_ddp.py:21 / pred_vmn.py:22."""
from tcvom_amd.data import VideoMattingDataset  # noqa: F401
