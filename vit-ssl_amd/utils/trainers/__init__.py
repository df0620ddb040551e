from .base_trainer import BaseTrainer
from .dino_trainer import DINOTrainer
from .simmim_trainer import SimMIMTrainer
from .supervised_trainer import SupervisedTrainer
