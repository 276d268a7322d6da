"""Reference import path `models.VMN.VMN_GCA` (models/VMN/VMN_GCA.py:8-48) -> the HIP implementation."""
from tcvom_amd.gca_net import ResGuidedCxtAtten_FAM_Dec  # noqa: F401
