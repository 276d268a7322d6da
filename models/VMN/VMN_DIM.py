"""Reference import path `models.VMN.VMN_DIM` (models/VMN/VMN_DIM.py:6-136) -> the HIP implementation."""
from tcvom_amd.dim_net import DIMDecoder, DIMEncoder  # noqa: F401
