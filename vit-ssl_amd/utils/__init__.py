"""Drop-in for the reference's ``utils`` package, hot-path subset:
model factory, optimizer/criterion/scheduler factories and the trainers' step loops.
The reference's Logger / MetricHandler / TrainingHistory (rich TUI, ignite, torcheval,
matplotlib) are out of scope and intentionally not mirrored."""
# The reference has packages of the same names (`utils`, `data`).  With this directory in front
# of the reference checkout on sys.path, the modules this package does NOT replace
# (utils.schemas, utils.logger, utils.metrics, utils.history, data.data_builder, data.datasets:
# what train.py:7-11 imports next to the model factory) must still resolve to the reference:
# extend the package search path with every later same-named package on sys.path.  Modules that
# exist here (model_builder, train_utils, schedulers, trainers) win because this directory stays
# first in __path__.
from pkgutil import extend_path

__path__ = extend_path(__path__, __name__)

from .model_builder import build_model, freeze_backbone, load_weights
from .train_utils import get_transforms, make_criterion, make_optimizer, make_schedulers, setup_device
