"""Same public names as the reference's models/GCA/ops.py:12-259."""
from tcvom_amd.gca_net import GuidedCxtAtten, SpectralNorm  # noqa: F401
