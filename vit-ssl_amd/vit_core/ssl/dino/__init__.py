from .model import DINOViT
from .head import DINOHead
from .dino_utils import DINOMomentumScheduler, DINOTeacherTempScheduler
