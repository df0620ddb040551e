"""DINO trainer (reference: utils/trainers/dino_trainer.py:14-173): per-EPOCH cosine
schedules for the teacher temperature and the EMA momentum, multi-crop step, teacher
momentum update after the optimizer step.  Views arrive as a list of V tensors (G global
first).  AdamW => the fused engine step; otherwise the reference-style autograd path.

New: when the loader yields ONE uint8 tensor [B, H, W, 3] of decoded images
instead of the view list, the views are produced on the GPU by `data.GPUMultiCrop` from the
`transforms.globals` / `transforms.locals` recipes of the config (the lists the reference
feeds to torchvision on the CPU, data/datasets.py:80-123)."""
import logging

import torch

from vit_core.ssl.dino.dino_utils import DINOMomentumScheduler, DINOTeacherTempScheduler
from vit_core.ssl.dino.loss import DINOLoss

from .._config import cfg_get
from .base_trainer import BaseTrainer

logger = logging.getLogger(__name__)


class DINOTrainer(BaseTrainer):
    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        t = lambda k, d=None: cfg_get(self.config, "training", k, default=d)  # noqa: E731
        self.num_global_views = cfg_get(self.config, "data", "num_global_views", default=t("num_global_views", 2))
        self.temp_sched, self.mom_sched = self.build_schedules(self.config, self.num_epochs)

    @staticmethod
    def teacher_temp0(config):
        """`training.teacher_temp` (reference key); `teacher_temp_start` is accepted as the
        spelling this package's round-1 configs used."""
        temp0 = cfg_get(config, "training", "teacher_temp", default=None)
        if temp0 is None:
            temp0 = cfg_get(config, "training", "teacher_temp_start", default=0.04)
        return temp0

    @staticmethod
    def build_schedules(config, num_epochs):
        """(temperature, momentum) schedulers exactly as the reference trainer builds them
        (utils/trainers/dino_trainer.py:16-29 with configs/dino/training.yaml): `teacher_temp` is
        the start, `teacher_temp_final` defaults to it, BOTH run over num_epochs,
        `teacher_temp_scheduler` picks cosine | linear.  They are evaluated at `epoch`, which
        starts at 1 (reference :46, :80) -- see epoch_schedule."""
        t = lambda k, d=None: cfg_get(config, "training", k, default=d)  # noqa: E731
        temp0 = DINOTrainer.teacher_temp0(config)
        temp1 = t("teacher_temp_final")
        if temp1 is None:
            temp1 = temp0
        temp = DINOTeacherTempScheduler(temp0, temp1, num_epochs, t("teacher_temp_scheduler", "cosine"))
        mom = DINOMomentumScheduler(t("teacher_momentum_start", 0.996), t("teacher_momentum_final", 1.0), num_epochs)
        return temp, mom

    def epoch_schedule(self, epoch: int):
        """(teacher temperature, EMA momentum) of 1-based `epoch`."""
        return self.temp_sched.get_temp(epoch), self.mom_sched.get_momentum(epoch)

    def _views(self, inputs):
        if isinstance(inputs, torch.Tensor) and inputs.dtype == torch.uint8 and inputs.dim() == 4:
            if getattr(self, "_multicrop", None) is None:
                from data import GPUMultiCrop, ViewSpec
                tf = cfg_get(self.config, "transforms")
                t = lambda k, d=None: cfg_get(self.config, "training", k, default=d)  # noqa: E731
                self._multicrop = GPUMultiCrop(ViewSpec.from_config(tf["globals"]), ViewSpec.from_config(tf["locals"]),
                                               t("num_all_views", 10), self.num_global_views)
            return self._multicrop(inputs.to(self.device, non_blocking=True))
        return [v.to(self.device, non_blocking=True) for v in inputs]

    def create_criterion(self):
        t = lambda k, d=None: cfg_get(self.config, "training", k, default=d)  # noqa: E731
        return DINOLoss(teacher_temp=self.teacher_temp0(self.config), student_temp=t("student_temp", 0.1))

    def _loss(self, views):
        G = self.num_global_views
        teacher, student = self.model(views, G)
        B = views[0].shape[0]
        K = teacher.shape[-1]
        return self.criterion(teacher.view(G, B, K), student.view(len(views), B, K), self.model.center)

    def train_epoch(self, epoch: int):
        self.model.train()
        self.criterion.teacher_temp, momentum = self.epoch_schedule(epoch)
        fused = self._is_fused()
        total, running = 0, None
        for idx, inputs in enumerate(self.train_loader):
            views = self._views(inputs)
            if fused:
                loss = self.model.train_step(views, self.num_global_views, self.criterion, self.optimizer, self.reducer, momentum)
            else:
                self.optimizer.zero_grad(set_to_none=True)
                loss = self._loss(views)
                loss.backward()
                self._generic_reduce()
                self.optimizer.step()
                self.model.momentum_update_teacher(momentum)
                loss = loss.detach()
            self._warmup_step(epoch)
            running = loss if running is None else running + loss
            total += 1
        return {"Loss": float(running) / max(total, 1), "TeacherTemp": self.criterion.teacher_temp, "Momentum": momentum}

    def validate(self):
        self.model.eval()
        total, running = 0, None
        with torch.no_grad():
            for idx, inputs in enumerate(self.val_loader):
                views = self._views(inputs)
                loss = self._loss(views)
                running = loss if running is None else running + loss
                total += 1
        return {"Loss": float(running) / max(total, 1) if total else float("nan")}
