"""Reference import path `models.FBA.models` (models/FBA/models.py:7-353) -> the HIP implementation."""
from tcvom_amd.fba_net import FBA, MattingModule, ResnetDilated  # noqa: F401
