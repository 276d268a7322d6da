"""Reference import path `models.VMN.VMN_FBA` (models/VMN/VMN_FBA.py:6-59) -> the HIP implementation."""
from tcvom_amd.fba_net import vmn_fba_decoder  # noqa: F401
