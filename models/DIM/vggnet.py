"""Reference import path `models.DIM.vggnet` (models/DIM/vggnet.py:10-133) -> the HIP implementation."""
from tcvom_amd.dim_net import DeepMatting, DIM_VGG  # noqa: F401
