"""Input-pipeline pieces on the hot path's upstream side."""
from .multicrop import GPUMultiCrop, ViewSpec, params_as_list, sample_batch_params, sample_view_params  # noqa: F401
