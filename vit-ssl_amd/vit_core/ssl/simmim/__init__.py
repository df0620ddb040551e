from .masking import simple_masking
from .model import SimMIMViT
