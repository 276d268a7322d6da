"""Same public names as the reference's models/VMN/__init__.py:1-29."""
from tcvom_amd.vmn import get_VMN_models, VMN  # noqa: F401
