"""Common trainer scaffolding (reference: utils/trainers/base_trainer.py:16-123):
criterion / optimizer / scheduler factories, epoch loop, checkpoint dict layout
{epoch, model_state_dict, optimizer_state_dict, best_val_loss, config}.  The reference's
rich Logger, MetricHandler and TrainingHistory are out of scope; stdlib logging reports
the loss.  New: data-parallel gradient reduction when torch.distributed is initialised."""
import logging
import math
import os
from abc import ABC, abstractmethod

import torch
import torch.distributed as dist

from vit_core._runtime import limit_host_threads

from .._config import cfg_get, to_plain
from ..train_utils import make_criterion, make_optimizer, make_schedulers

logger = logging.getLogger(__name__)


class BaseTrainer(ABC):
    def __init__(self, model, save_path: str, config, train_loader, val_loader, device):
        self.model = model
        self.config = config
        self.train_loader = train_loader
        self.val_loader = val_loader
        self.device = device
        self.save_path = save_path
        self.warmup_epochs = cfg_get(config, "training", "warmup_epochs")
        self.num_epochs = cfg_get(config, "training", "num_epochs")
        self.eval_interval = cfg_get(config, "eval", "interval", default=0)

        if torch.device(device).type == "cuda":
            limit_host_threads()
        self.criterion = self.create_criterion()
        self.optimizer = make_optimizer(config, model)
        self.schedulers = make_schedulers(config, self.optimizer, self.num_epochs, self.warmup_epochs * len(train_loader))
        self.best_val_loss = math.inf
        self.current_epoch = 0
        self.start_epoch = 0

        self.world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
        self.rank = dist.get_rank() if self.world > 1 else 0
        self.reducer = None
        if self.world > 1:
            self._setup_data_parallel()

    # ---- data parallel -----------------------------------------------------------
    def _dp_store(self):
        return self.model.trainable_store() if hasattr(self.model, "trainable_store") else self.model.flat_store()

    def _setup_data_parallel(self):
        from vitssl_hip.engine import GradReducer
        for store in self.model.all_stores() if hasattr(self.model, "all_stores") else [self.model.flat_store()]:
            dist.broadcast(store.flat, 0)                      # identical weights on every rank
            store.mark_dirty()
        self.reducer = GradReducer(self._dp_store().gflat)

    def _is_fused(self) -> bool:
        from vitssl_hip.optim import FusedAdamW
        return isinstance(self.optimizer, FusedAdamW) and hasattr(self.model, "train_step")

    def _generic_reduce(self):
        """Reference-style path (loss.backward through autograd): average p.grad across ranks."""
        if self.world == 1:
            return
        for p in self.model.parameters():
            if p.grad is not None:
                dist.all_reduce(p.grad)
                p.grad.div_(self.world)

    # ---- abstract ------------------------------------------------------------------
    @abstractmethod
    def train_epoch(self, epoch):
        pass

    @abstractmethod
    def validate(self):
        pass

    def create_criterion(self):
        return make_criterion(self.config)

    # ---- loop ----------------------------------------------------------------------
    def fit(self, num_epochs: int):
        end_epoch = self.start_epoch + num_epochs
        for epoch in range(self.start_epoch + 1, end_epoch + 1):
            self.current_epoch = epoch
            train_metrics = self.train_epoch(epoch)
            val_metrics = self.validate()
            self._update_schedulers(epoch)
            self._log_metrics(epoch, train_metrics, val_metrics)
            self._save_if_best(epoch, val_metrics["Loss"])
            self._save_last(epoch)

    def _update_schedulers(self, epoch):
        if epoch > self.warmup_epochs:
            self.schedulers["main"].step()

    def _warmup_step(self, epoch):
        if self.schedulers["warmup"] is not None and epoch <= self.warmup_epochs:
            self.schedulers["warmup"].step()

    def _log_metrics(self, epoch, train_metrics, val_metrics):
        if self.rank == 0:
            logger.info(f"epoch {epoch}: train {train_metrics} | val {val_metrics}")

    def _checkpoint(self, epoch, **extra):
        ckpt = {"epoch": epoch, "model_state_dict": self.model.state_dict(),
                "optimizer_state_dict": self.optimizer.state_dict(), "config": to_plain(self.config)}
        ckpt.update(extra)
        return ckpt

    def _save_if_best(self, epoch, val_loss):
        if self.rank == 0 and self.best_val_loss >= val_loss:
            self.best_val_loss = val_loss
            logger.info(f"New best validation loss: {self.best_val_loss:.4f}. Saving model...")
            os.makedirs(self.save_path, exist_ok=True)
            torch.save(self._checkpoint(epoch, best_val_loss=self.best_val_loss), os.path.join(self.save_path, "best_model.pth"))

    def _save_last(self, epoch):
        if self.rank == 0:
            os.makedirs(self.save_path, exist_ok=True)
            torch.save(self._checkpoint(epoch), os.path.join(self.save_path, "last_model.pth"))
