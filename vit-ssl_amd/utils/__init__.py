"""Drop-in for the reference's ``utils`` package, hot-path subset:
model factory, optimizer/criterion/scheduler factories and the trainers' step loops.
The reference's Logger / MetricHandler / TrainingHistory (rich TUI, ignite, torcheval,
matplotlib) are out of scope and intentionally not mirrored."""
from .model_builder import build_model, freeze_backbone, load_weights
from .train_utils import get_transforms, make_criterion, make_optimizer, make_schedulers, setup_device
