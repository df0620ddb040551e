"""Host-side binding of the MI355X HIP library for the vit_core hot path."""
from ._lib import VitsslError, lib, header_symbols, LIB_PATH  # noqa: F401
