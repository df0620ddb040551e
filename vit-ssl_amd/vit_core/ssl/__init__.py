"""SSL model families (reference: vit_core/ssl/__init__.py exports DINOViT)."""


def __getattr__(name):
    if name == "DINOViT":
        from .dino.model import DINOViT
        return DINOViT
    raise AttributeError(name)
