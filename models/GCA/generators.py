"""Reference import path `models.GCA.generators` (models/GCA/generators.py:8-46) -> the HIP implementation."""
from tcvom_amd.gca_net import GCA, Generator  # noqa: F401
