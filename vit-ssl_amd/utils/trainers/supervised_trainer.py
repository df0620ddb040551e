"""Supervised / fine-tune trainer (reference: utils/trainers/supervised_trainer.py:30-48):
logits = model(x); CrossEntropy; backward; step.  Runs the reference-style autograd path
through the HIP engine (the supervised config is the reference's small plumbing case)."""
import logging

import torch

from .base_trainer import BaseTrainer

logger = logging.getLogger(__name__)


class SupervisedTrainer(BaseTrainer):
    def train_epoch(self, epoch: int):
        self.model.train()
        total, correct, seen, running = 0, 0, 0, None
        for idx, (inputs, labels) in enumerate(self.train_loader):
            inputs, labels = inputs.to(self.device), labels.to(self.device)
            self.optimizer.zero_grad(set_to_none=True)
            logits = self.model(inputs)
            loss = self.criterion(logits, labels)
            loss.backward()
            self._generic_reduce()
            self.optimizer.step()
            self._warmup_step(epoch)
            running = loss.detach() if running is None else running + loss.detach()
            correct += int((logits.argmax(1) == labels).sum())
            seen += labels.numel()
            total += 1
        return {"Loss": float(running) / max(total, 1), "Accuracy": correct / max(seen, 1)}

    def validate(self):
        self.model.eval()
        total, correct, seen, running = 0, 0, 0, None
        with torch.no_grad():
            for idx, (inputs, labels) in enumerate(self.val_loader):
                inputs, labels = inputs.to(self.device), labels.to(self.device)
                logits = self.model(inputs)
                loss = self.criterion(logits, labels)
                running = loss if running is None else running + loss
                correct += int((logits.argmax(1) == labels).sum())
                seen += labels.numel()
                total += 1
        return {"Loss": float(running) / max(total, 1) if total else float("nan"), "Accuracy": correct / max(seen, 1)}
