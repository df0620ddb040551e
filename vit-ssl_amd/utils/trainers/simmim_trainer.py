"""SimMIM trainer (reference: utils/trainers/simmim_trainer.py:53-135).

Step body = the reference's zero_grad -> forward -> criterion -> backward -> step
(-> warm-up).  With nn.L1Loss(mean) + AdamW (the reference's configs/simmim/training.yaml)
the whole step runs fused in the HIP engine (SimMIMViT.train_step); any other criterion /
optimizer takes the reference-style autograd path through the same kernels.  The loss is
accumulated on the device and read once per epoch (the reference syncs every step)."""
import logging

import torch
from torch import nn

from .base_trainer import BaseTrainer

logger = logging.getLogger(__name__)


class SimMIMTrainer(BaseTrainer):
    def _fused_ok(self):
        return self._is_fused() and isinstance(self.criterion, nn.L1Loss) and self.criterion.reduction == "mean"

    def train_epoch(self, epoch: int):
        self.model.train()
        total = 0
        running = None
        fused = self._fused_ok()
        for idx, inputs in enumerate(self.train_loader):
            inputs = inputs.to(self.device, non_blocking=True)
            if fused:
                loss = self.model.train_step(inputs, self.optimizer, self.reducer)
            else:
                self.optimizer.zero_grad(set_to_none=True)
                preds, targets = self.model(inputs)
                loss = self.criterion(preds, targets)
                loss.backward()
                self._generic_reduce()
                self.optimizer.step()
                loss = loss.detach()
            self._warmup_step(epoch)
            running = loss if running is None else running + loss
            total += 1
        return {"Loss": float(running) / max(total, 1)}

    def validate(self):
        self.model.eval()
        total, running = 0, None
        with torch.no_grad():
            for idx, inputs in enumerate(self.val_loader):
                inputs = inputs.to(self.device, non_blocking=True)
                preds, targets = self.model(inputs)
                loss = self.criterion(preds, targets)
                running = loss if running is None else running + loss
                total += 1
        return {"Loss": float(running) / max(total, 1) if total else float("nan")}
