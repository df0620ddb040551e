"""DINOLoss (reference: vit_core/ssl/dino/loss.py:7-29): teacher softmax with centring and
sharpening, student log-softmax, summed over ALL (teacher view, student view) pairs and
averaged over (G, B, K) exactly as the reference writes it.  One fused HIP pass computes the
loss and the student-logit gradient; the [G,V,B,K] product is never materialised."""
import torch
from torch import nn
from torch.autograd import Function

from ... import _runtime as R
from ..._runtime import BF16, F32, ops


class _DinoLossFn(Function):
    @staticmethod
    def forward(ctx, teacher, student, center, t_temp, s_temp):
        G, B, K = teacher.shape
        V = student.shape[0]
        t2 = R.as_f32(teacher.detach()).reshape(G * B, K)
        s2 = R.as_f32(student).reshape(V * B, K)
        c = R.as_f32(center.detach()).reshape(-1)
        dev = s2.device
        t_ws = torch.empty(ops.dino_tws_floats(G, B, K), dtype=F32, device=dev)
        loss = torch.zeros(1, dtype=F32, device=dev)
        dstudent = torch.empty(V * B, K, dtype=BF16, device=dev) if student.requires_grad else None
        ops.dino_loss(t2, s2, c, t_ws, loss, dstudent, G, V, B, K, t_temp, s_temp, 1.0)
        ctx.save_for_backward(dstudent)
        ctx.shape = tuple(student.shape)
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        (dstudent,) = ctx.saved_tensors
        return None, (dstudent.float() * g).view(ctx.shape), None, None, None


class DINOLoss(nn.Module):
    def __init__(self, teacher_temp: float, student_temp: float):
        super().__init__()
        self.teacher_temp = teacher_temp
        self.student_temp = student_temp

    def forward(self, teacher_output: torch.Tensor, student_output: torch.Tensor, center: torch.Tensor):
        R.require_gpu(student_output, "DINOLoss")
        return _DinoLossFn.apply(teacher_output, student_output, center, float(self.teacher_temp), float(self.student_temp))
