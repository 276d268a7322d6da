"""Same public names as the reference's models/VMN/VMN_model.py:9-113."""
from tcvom_amd.vmn import FeatureAggregationModule, VMN  # noqa: F401
