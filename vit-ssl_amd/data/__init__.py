"""Input-pipeline pieces on the hot path's upstream side."""
# see utils/__init__.py: data.data_builder / data.datasets of a reference checkout later on
# sys.path stay importable (the reference's train.py:8 does `from data.data_builder import ...`)
from pkgutil import extend_path

__path__ = extend_path(__path__, __name__)

from .multicrop import GPUMultiCrop, ViewSpec, params_as_list, sample_batch_params, sample_view_params  # noqa: F401
