"""Input-pipeline pieces on the hot path's upstream side."""
from .multicrop import GPUMultiCrop, ViewSpec, sample_view_params  # noqa: F401
