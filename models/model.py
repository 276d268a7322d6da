"""Same public names as the reference's models/model.py:15-453."""
from tcvom_amd.facade import FullModel, FullModel_VMD, EvalModel  # noqa: F401
