"""Host-side composition of the HIP kernels into the vit_core forward/backward.

Design (MI355X-first, not a translation of the reference's eager module graph):
  * every parameter of a model lives in ONE flat fp32 buffer (FlatStore) with a
    matching flat gradient buffer; q/k/v weights are adjacent, so the fused QKV weight
    [3D, D] and its gradient are free views.  AdamW, EMA and the data-parallel
    all-reduce operate on the flat buffers (one kernel / few large collectives).
  * bf16 copies of every GEMM weight, plus the transposed copy the dgrad GEMM reads,
    are refreshed only when the flat buffer's version changes (once per optimizer step).
  * activations needed by backward are kept in step-persistent workspaces (288 GB of
    HBM: nothing is recomputed except dropout masks, which are counter-based).
  * backward is an explicit reverse schedule (no autograd graph inside the stack), so
    gradient ranges become ready in a known order and the reducer overlaps their
    all-reduce with the remaining backward on a side stream.
"""
from __future__ import annotations

from typing import Callable, Dict, List, Optional, Tuple

import torch
from torch import nn

from . import _lib as L
from . import ops

BF16 = torch.bfloat16
F32 = torch.float32
FP8 = ops.FP8
ALIGN = 64  # elements; keeps every parameter 256-byte aligned inside the flat buffer

# Operand type of the FORWARD Linear products inside the encoder blocks: "bf16" (default, the reference's
# autocast contract) or "fp8" (OCP e4m3, BASELINE.json configs[4]; backward products stay bf16).  Read when an
# EncoderStack is built: set VITSSL_LINEAR_OPERANDS=fp8 or call set_linear_operands("fp8") before the model's
# first forward.
import os as _os

_LINEAR_OPERANDS = _os.environ.get("VITSSL_LINEAR_OPERANDS", "bf16")


def set_linear_operands(kind: str):
    global _LINEAR_OPERANDS
    if kind not in ("bf16", "fp8"):
        raise L.VitsslError(f"linear operands must be 'bf16' or 'fp8', got {kind!r}")
    _LINEAR_OPERANDS = kind


def linear_operands() -> str:
    if _LINEAR_OPERANDS not in ("bf16", "fp8"):
        raise L.VitsslError(f"VITSSL_LINEAR_OPERANDS must be 'bf16' or 'fp8', got {_LINEAR_OPERANDS!r}")
    return _LINEAR_OPERANDS


def _round_up(n: int, a: int) -> int:
    return (n + a - 1) // a * a


class Workspace:
    """Named device buffers reused across steps (no allocation in steady state)."""

    def __init__(self):
        self._bufs: Dict[str, torch.Tensor] = {}

    def get(self, name: str, shape, dtype, device) -> torch.Tensor:
        t = self._bufs.get(name)
        shape = tuple(int(s) for s in shape)
        if t is None or t.shape != shape or t.dtype != dtype or t.device != device:
            t = torch.empty(shape, dtype=dtype, device=device)
            self._bufs[name] = t
        return t

    def clear(self):
        self._bufs.clear()


class FlatStore:
    """Flat fp32 parameter / gradient buffers behind an nn.Module tree.

    The module keeps ordinary nn.Parameters with the reference's state_dict keys; their
    storage is re-pointed into `self.flat` (and `.grad` into `self.gflat` on request).
    """

    def __init__(self, module: nn.Module, device: torch.device, only: Optional[Callable[[str], bool]] = None,
                 forward_only: bool = False):
        self.module = module
        self.device = device
        # forward_only: no backward ever runs through these weights (DINO's teacher): the transposed bf16 images, operands of
        # the input-gradient GEMMs only, are not kept (a third of the cast's bytes per optimizer step)
        self.forward_only = bool(forward_only)
        self.names: List[str] = []
        self.params: List[nn.Parameter] = []
        self.offsets: Dict[str, Tuple[int, int]] = {}
        off = 0
        for name, p in module.named_parameters():
            if only is not None and not only(name):
                continue
            n = p.numel()
            self.names.append(name)
            self.params.append(p)
            self.offsets[name] = (off, n)
            off = _round_up(off + n, ALIGN)
        self.numel = off
        self.flat = torch.zeros(off, dtype=F32, device=device)
        self.gflat = torch.zeros(off, dtype=F32, device=device)
        with torch.no_grad():
            for name, p in zip(self.names, self.params):
                o, n = self.offsets[name]
                view = self.flat[o:o + n].view(p.shape)
                view.copy_(p.data)
                p.data = view
        self._bf16: Dict[str, torch.Tensor] = {}
        self._bf16_key = None
        self.generation = 0   # bumped whenever the flat buffer is rewritten behind torch's back
        self._cast_jobs: List[Tuple[str, Callable[[], torch.Tensor], bool, bool]] = []
        self._fp8: Dict[str, torch.Tensor] = {}
        self._fp8_jobs: List[Tuple[str, Callable[[], torch.Tensor]]] = []
        self._fp8_plan = None
        self._fp8_index: Dict[str, int] = {}

    # ---- bookkeeping -------------------------------------------------------
    def is_attached(self) -> bool:
        """True while every Parameter still aliases the flat buffer (a .to()/.cuda() or
        load with assign=True re-allocates them)."""
        base = self.flat.data_ptr()
        for name, p in zip(self.names, self.params):
            o, _ = self.offsets[name]
            if p.data_ptr() != base + 4 * o:
                return False
        return True

    def view(self, name: str, shape=None) -> torch.Tensor:
        o, n = self.offsets[name]
        v = self.flat[o:o + n]
        return v.view(shape) if shape is not None else v

    def gview(self, name: str, shape=None) -> torch.Tensor:
        o, n = self.offsets[name]
        v = self.gflat[o:o + n]
        return v.view(shape) if shape is not None else v

    def span(self, first: str, last: str) -> Tuple[int, int]:
        lo = self.offsets[first][0]
        o, n = self.offsets[last]
        return lo, o + n

    def span_view(self, first: str, last: str, shape, grad=False) -> torch.Tensor:
        lo, hi = self.span(first, last)
        buf = self.gflat if grad else self.flat
        v = buf[lo:hi]
        if v.numel() != int(torch.Size(shape).numel()):
            raise L.VitsslError(f"parameters {first}..{last} are not contiguous in the flat store")
        return v.view(shape)

    def attach_grads(self):
        """Make every p.grad a view of the flat gradient buffer (fused-step mode)."""
        for name, p in zip(self.names, self.params):
            p.grad = self.gview(name, p.shape)

    def mark_dirty(self):
        """Called by everything that rewrites the flat buffer through a raw pointer (the HIP
        AdamW / EMA kernels, broadcasts): torch's version counters do not see those writes."""
        self.generation += 1

    def weights_key(self):
        """Changes whenever any parameter value may have changed.  `p.data = flat_view` gives
        every Parameter its OWN version counter, so in-place updates through the Parameters
        (any torch optimizer, load_state_dict, p.copy_ / p.mul_ ...) bump p._version and leave
        flat._version alone: the key has to include both (about 150 integers per model)."""
        pv = 0
        for p in self.params:
            pv += p._version
        return (self.generation, self.flat._version, pv)

    # ---- bf16 weight caches --------------------------------------------------
    def register_weight(self, key: str, src: Callable[[], torch.Tensor], transposed_too: bool = True, plain: bool = True):
        """Declare a 2-D GEMM weight [N,K]; caches `key` (bf16 [N,K], unless plain=False) and, if asked,
        `key + '.T'` (bf16 [K,N], the operand of the dgrad GEMM)."""
        self._cast_jobs.append((key, src, transposed_too and not self.forward_only, plain))

    def register_fp8_weight(self, key: str, src: Callable[[], torch.Tensor]):
        """Declare a 2-D GEMM weight [N,K] whose GEMM operands are e4m3 images with a per-tensor power-of-two
        scale: `w8(key)` returns (image [N,K], dequantisation factor as a 1-element device tensor), `w8t(key)`
        the transposed image [K,N] (operand of the input-gradient GEMM) with the same factor."""
        self._fp8_index[key] = len(self._fp8_jobs)
        self._fp8_jobs.append((key, src))

    def refresh_weights(self):
        key = self.weights_key()
        if key == self._bf16_key and (self._bf16 or self._fp8):
            return
        jobs = []
        for wname, src, tr, plain in self._cast_jobs:     # (never `key`: that is the cache key stored below)
            w = src()
            R, Cn = w.shape
            dst = self._bf16.get(wname) if plain else None
            if plain and (dst is None or dst.shape != (R, Cn)):
                dst = torch.empty(R, Cn, dtype=BF16, device=self.device)
                self._bf16[wname] = dst
            dst_t = None
            if tr:
                dst_t = self._bf16.get(wname + ".T")
                if dst_t is None or dst_t.shape != (Cn, R):
                    dst_t = torch.empty(Cn, R, dtype=BF16, device=self.device)
                    self._bf16[wname + ".T"] = dst_t
            jobs.append((w.contiguous(), dst, dst_t))
        if jobs:      # every weight of the store in one launch (was ~50 launches of ~12 us)
            if getattr(self, "_cast_plan", None) is None:
                self._cast_plan = ops.CastPlan()
            self._cast_plan.run(jobs)
        if self._fp8_jobs:
            jobs8 = []
            for k8, src in self._fp8_jobs:
                w = src()
                R, Cn = w.shape
                dst, dst_t = self._fp8.get(k8), self._fp8.get(k8 + ".T")
                if dst is None or dst.shape != (R, Cn):
                    dst = torch.empty(R, Cn, dtype=FP8, device=self.device)
                    dst_t = torch.empty(Cn, R, dtype=FP8, device=self.device)
                    self._fp8[k8], self._fp8[k8 + ".T"] = dst, dst_t
                jobs8.append((w.contiguous(), dst, dst_t))
            if self._fp8_plan is None:
                self._fp8_plan = ops.Fp8WeightPlan()
            self._fp8_plan.run(jobs8)
        self._bf16_key = key

    def w(self, key: str) -> torch.Tensor:
        return self._bf16[key]

    def w8(self, key: str) -> Tuple[torch.Tensor, torch.Tensor]:
        j = self._fp8_index[key]
        return self._fp8[key], self._fp8_plan.alpha[j:j + 1]

    def w8t(self, key: str) -> Tuple[torch.Tensor, torch.Tensor]:
        j = self._fp8_index[key]
        return self._fp8[key + ".T"], self._fp8_plan.alpha[j:j + 1]


RESERVED_CUS_DEFAULT = 8


def configure_collectives(reserve_cus: int = RESERVED_CUS_DEFAULT):
    """Size the collective library to the CUs the persistent GEMM grids leave alone.  Call BEFORE
    torch.distributed.init_process_group (RCCL reads its environment when the communicator is built).

    RCCL launches one workgroup per channel and a ring only progresses while all of them are resident.  RCCL 2.26 on
    gfx950 sets up 128 collective channels (tools/rccl_probe.py, NCCL_DEBUG=INFO, one MI355X); with the GEMM grids holding
    every other CU for a whole kernel, a collective of more channels than reserved CUs would only start at kernel
    boundaries.  So the channel count is capped at the reserve (unless the user has set NCCL_MAX_NCHANNELS): 8 channels
    move the 345 MB of ViT-B gradients (604 MB per GPU through the ring) well inside one backward pass.
    Returns the values in force."""
    if "NCCL_MAX_NCHANNELS" not in _os.environ:
        _os.environ["NCCL_MAX_NCHANNELS"] = str(max(1, int(reserve_cus)))
    if "NCCL_MIN_NCHANNELS" not in _os.environ:
        _os.environ["NCCL_MIN_NCHANNELS"] = str(min(int(_os.environ["NCCL_MAX_NCHANNELS"]), max(1, int(reserve_cus))))
    return {k: _os.environ.get(k) for k in ("NCCL_MAX_NCHANNELS", "NCCL_MIN_NCHANNELS")}


class GradReducer:
    """Data-parallel gradient averaging over RCCL (torch.distributed 'nccl' on ROCm),
    overlapped with backward: the engine calls `ready(lo, hi)` as soon as a contiguous
    range of the flat gradient buffer is final; ranges are coalesced into buckets of
    >= `bucket_elems` and all-reduced (sum) on a side stream.  The 1/world factor is
    folded into the optimizer kernel.  On CPU tensors (gloo, tests) it runs inline."""

    def __init__(self, gflat: torch.Tensor, group=None, bucket_mb: float = 32.0, expect: Optional[Tuple[int, int]] = None,
                 reserve_cus: int = RESERVED_CUS_DEFAULT):
        """`expect` = (lo, hi) range of the flat buffer that one backward must hand over exactly once
        (default: the whole buffer); finish() checks it, so a schedule change that forgets or repeats a
        range fails loudly on one GPU instead of silently de-synchronising replicas on eight."""
        import torch.distributed as dist
        self.dist = dist
        self.gflat = gflat
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
        if self.world > 1 and gflat.is_cuda:
            # The persistent GEMM workgroups take a CU's whole register file: the collective library's kernels can only run
            # beside them on CUs the GEMM grids leave alone.  Explicit library state, re-read on
            # every launch: it does not matter whether a forward ran before this reducer was built.  VITSSL_RESERVE_CUS
            # (if set) is the user's choice and wins; otherwise `reserve_cus`.
            if "VITSSL_RESERVE_CUS" not in _os.environ:
                L.call("vitssl_set_reserved_cus", int(reserve_cus))
        self.bucket_elems = int(bucket_mb * 1024 * 1024 / 4)
        self.pending: List[Tuple[int, int]] = []
        self.pending_elems = 0
        self.handles = []
        self.cuda = gflat.is_cuda
        # high priority: when a CU frees up, the collective's workgroups are dispatched ahead of the queued GEMM workgroups
        self.comm_stream = torch.cuda.Stream(device=gflat.device, priority=-1) if self.cuda else None
        self.launched: List[Tuple[int, int]] = []
        self.expect = expect if expect is not None else (0, gflat.numel())
        self.check_coverage = True
        # diagnostics (bench.py, first real multi-GPU run): with `timing` on, every bucket is bracketed by events on the
        # communication stream, and begin() drops one on the compute stream, so bucket_times() can tell whether buckets
        # queue behind the persistent GEMMs (start late / run long) or finish under the backward
        self.timing = False
        self._t0 = None
        self._bucket_events = []

    def _flush(self):
        if not self.pending:
            return
        # merge adjacent ranges (the backward schedule hands them over high-to-low)
        rng = sorted(self.pending)
        merged = [list(rng[0])]
        for lo, hi in rng[1:]:
            if lo <= merged[-1][1]:
                merged[-1][1] = max(merged[-1][1], hi)
            else:
                merged.append([lo, hi])
        self.pending, self.pending_elems = [], 0
        for lo, hi in merged:
            self.launched.append((lo, hi))
            if self.world == 1:
                continue
            chunk = self.gflat[lo:hi]
            if self.cuda:
                self.comm_stream.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(self.comm_stream):
                    if self.timing:
                        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                        e0.record(self.comm_stream)
                    self.dist.all_reduce(chunk, group=self.group)
                    if self.timing:
                        e1.record(self.comm_stream)
                        self._bucket_events.append((4 * (hi - lo), e0, e1))
            else:
                self.handles.append(self.dist.all_reduce(chunk, group=self.group, async_op=True))

    def begin(self):
        self.pending, self.pending_elems, self.handles, self.launched = [], 0, [], []
        if self.timing and self.cuda:
            self._bucket_events = []
            self._t0 = torch.cuda.Event(enable_timing=True)
            self._t0.record(torch.cuda.current_stream())

    def bucket_times(self):
        """[(bytes, ms from begin() to the bucket's start on the communication stream, ms the all-reduce took)] of the last
        step run with `timing` on (synchronises)."""
        if not self._bucket_events:
            return []
        torch.cuda.synchronize()
        return [(nbytes, round(self._t0.elapsed_time(e0), 3), round(e0.elapsed_time(e1), 3)) for nbytes, e0, e1 in self._bucket_events]

    def ready(self, lo: int, hi: int):
        self.pending.append((lo, hi))
        self.pending_elems += hi - lo
        if self.pending_elems >= self.bucket_elems:
            self._flush()

    def finish(self):
        """Launch what is left and make the compute stream wait for all buckets."""
        self._flush()
        for h in self.handles:
            h.wait()
        self.handles = []
        if self.cuda and self.world > 1:
            torch.cuda.current_stream().wait_stream(self.comm_stream)
        if self.check_coverage:
            self._assert_covered()

    def _assert_covered(self):
        """Every element of the expected range was handed over exactly once (alignment padding between
        parameters may be skipped: it is never read)."""
        pos = self.expect[0]
        for lo, hi in sorted(self.launched):
            if lo < pos:
                raise L.VitsslError(f"GradReducer: gradient range [{lo}, {hi}) was reduced twice (previous range ended at {pos})")
            if lo - pos >= ALIGN:
                raise L.VitsslError(f"GradReducer: gradient range [{pos}, {lo}) was never handed to the reducer")
            pos = hi
        if self.expect[1] - pos >= ALIGN:
            raise L.VitsslError(f"GradReducer: gradient range [{pos}, {self.expect[1]}) was never handed to the reducer")

    def stats(self):
        """(number of buckets, bytes) of the last step."""
        return len(self.launched), 4 * sum(hi - lo for lo, hi in self.launched)

    @property
    def grad_scale(self) -> float:
        return 1.0 / self.world


class EncoderStack:
    """L pre-LN transformer blocks (vit_core/encoder_block.py:40-53) on a flat store."""

    def __init__(self, store: FlatStore, block_prefixes: List[str], D: int, H: int, F: int, p_drop: float,
                 site_base: int = 0):
        if D % H != 0:
            raise AssertionError(f"d_model({D}) must be cleanly divisible by num_heads({H})!")
        self.store = store
        self.bp = list(block_prefixes)     # e.g. ["encoder_blocks.0.", ...] or [""] for a lone block
        self.L, self.D, self.H, self.F = len(self.bp), D, H, F
        self.dh = D // H
        self.p = float(p_drop)
        self.site_base = site_base
        self.ws = Workspace()
        self._saved = {}
        self.fp8 = linear_operands() == "fp8"
        if self.fp8 and (D % 128 != 0 or F % 128 != 0):
            raise L.VitsslError(f"fp8 linear operands need embed_dim ({D}) and mlp_dim ({F}) to be multiples of 128")
        for b in self.bp:
            a = b + "self_attention."
            srcs = {
                "wqkv": lambda a=a, D=D: store.span_view(a + "w_query.weight", a + "w_value.weight", (3 * D, D)),
                "wo": lambda a=a: store.view(a + "final_linear.weight", (D, D)),
                "w1": lambda b=b: store.view(b + "feed_forward.linear_in.weight", (F, D)),
                "w2": lambda b=b: store.view(b + "feed_forward.linear_out.weight", (D, F)),
            }
            for leaf, src in srcs.items():
                if self.fp8:      # e4m3 images [N,K] (forward) and [K,N] (input gradients); no bf16 copies
                    store.register_fp8_weight(b + leaf, src)
                else:
                    store.register_weight(b + leaf, src)
        # fp8: per block, scales of the four gradient tensors that feed an input-gradient GEMM
        # (0: d(FFN out) -> FC2, 1: d(FFN hidden) -> FC1, 2: d(attention out) -> out-projection, 3: dQKV -> QKV).
        # Delayed scaling: a step quantises with the power-of-two scale derived from the PREVIOUS step's max |value|
        # (one bit of headroom); the very first backward derives each scale from the tensor itself (one extra pass).
        # The state is kept PER SLOT: DINO runs the student stack's backward twice per step (local crops, then global
        # crops: different row counts and gradient magnitudes), and a scale taken from the other pass's maxima would
        # clamp or lose low bits whenever the two differ by more than the one bit of headroom.
        self._gs_by_slot: Dict[str, dict] = {}    # slot -> {"scale", "inv", "amax", "used": [L, 4] device tensors, "valid": bool}
        self._gs = None                           # state of the slot whose backward is running / ran last

    # names ----------------------------------------------------------------
    def _n(self, i, leaf):
        return self.bp[i] + leaf

    def block_span(self, i) -> Tuple[int, int]:
        names = [n for n in self.store.names if n.startswith(self.bp[i])]
        return self.store.span(names[0], names[-1])

    def _drop(self, i, which, seed, training):
        if not training or self.p <= 0.0:
            return ops.NO_DROP
        return ops.make_dropout(self.p, seed, self.site_base + 3 * i + which)

    # forward ----------------------------------------------------------------
    def forward(self, x: torch.Tensor, B: int, T: int, training: bool, seed: int, save: bool, slot: str = "a",
                return_attn: bool = False):
        """x: fp32 [B*T, D] (left untouched).  Returns (x_out fp32 [B*T, D], probs or None)."""
        st, D, H, F, dh = self.store, self.D, self.H, self.F, self.dh
        M = B * T
        dev = x.device
        g = self.ws.get
        probs = None
        rec = {"B": B, "T": T, "seed": seed, "training": training, "blocks": []}
        cur = x
        for i in range(self.L):
            tag = f"{slot}.{i}." if save else f"{slot}.tmp."
            h1 = None if self.fp8 else g(tag + "h1", (M, D), BF16, dev)
            mean1 = g(tag + "mean1", (M,), F32, dev)
            rstd1 = g(tag + "rstd1", (M,), F32, dev)
            qkv = g(tag + "qkv", (M, 3 * D), BF16, dev)
            att = g(tag + "att", (M, D), BF16, dev)
            lse = g(tag + "lse", (B, H, T), F32, dev)
            xmid = g(tag + "xmid", (M, D), F32, dev)
            h2 = None if self.fp8 else g(tag + "h2", (M, D), BF16, dev)
            mean2 = g(tag + "mean2", (M,), F32, dev)
            rstd2 = g(tag + "rstd2", (M,), F32, dev)
            u = g(tag + "u", (M, F), BF16, dev)
            a = None if self.fp8 else g(tag + "a", (M, F), BF16, dev)
            # block output: a fresh buffer per block when saving (it is the next block's
            # LN input), otherwise ping-pong
            xout = g(f"{slot}.{i}.xout" if save else f"{slot}.tmp.xout{i & 1}", (M, D), F32, dev)

            if return_attn and i == self.L - 1:
                probs = torch.empty(B, H, T, T, dtype=F32, device=dev)
            ln1 = (cur, st.view(self._n(i, "layer_norm1.weight")), st.view(self._n(i, "layer_norm1.bias")))
            ln2 = (xmid, st.view(self._n(i, "layer_norm2.weight")), st.view(self._n(i, "layer_norm2.bias")))
            b1, b2 = st.view(self._n(i, "feed_forward.linear_in.bias")), st.view(self._n(i, "feed_forward.linear_out.bias"))
            if self.fp8:
                # products on e4m3 operands: every producer also writes the e4m3 image of its output, kept per block for
                # the weight-gradient GEMMs of the backward (one byte per element next to the bf16 images)
                h1_8 = g(tag + "h1_8", (M, D), FP8, dev)
                att8 = g(tag + "att8", (M, D), FP8, dev)
                h2_8 = g(tag + "h2_8", (M, D), FP8, dev)
                a8 = g(tag + "a8", (M, F), FP8, dev)
                (wq, aq), (wo, ao), (w1, a1), (w2, a2) = (st.w8(self._n(i, k)) for k in ("wqkv", "wo", "w1", "w2"))
                ops.layernorm_fwd_fp8(*ln1, None, h1_8, mean1, rstd1)     # (no bf16 images of h1 / h2 / a: nothing reads them)
                ops.gemm_fp8_nt(h1_8, wq, qkv, L.EPI_BF16, alpha=aq)
                ops.attn_fwd(qkv, att, lse, B, T, H, dh, probs=probs if i == self.L - 1 else None, out_fp8=att8)
                ops.gemm_fp8_nt(att8, wo, xmid, L.EPI_RESID, alpha=ao, aux=cur, drop=self._drop(i, 0, seed, training))
                ops.layernorm_fwd_fp8(*ln2, None, h2_8, mean2, rstd2)
                ops.gemm_fp8_nt(h2_8, w1, u, L.EPI_GELU, alpha=a1, bias=b1, out_fp8=a8, drop=self._drop(i, 1, seed, training))
                ops.gemm_fp8_nt(a8, w2, xout, L.EPI_RESID, alpha=a2, bias=b2, aux=xmid, drop=self._drop(i, 2, seed, training))
            else:
                ops.layernorm_fwd(*ln1, h1, mean1, rstd1)
                ops.gemm_nt(h1, st.w(self._n(i, "wqkv")), qkv, L.EPI_BF16)
                ops.attn_fwd(qkv, att, lse, B, T, H, dh, probs=probs if i == self.L - 1 else None)
                ops.gemm_nt(att, st.w(self._n(i, "wo")), xmid, L.EPI_RESID, aux=cur, drop=self._drop(i, 0, seed, training))
                ops.layernorm_fwd(*ln2, h2, mean2, rstd2)
                ops.gemm_nt(h2, st.w(self._n(i, "w1")), u, L.EPI_GELU, bias=b1, out1=a, drop=self._drop(i, 1, seed, training))
                ops.gemm_nt(a, st.w(self._n(i, "w2")), xout, L.EPI_RESID, bias=b2, aux=xmid, drop=self._drop(i, 2, seed, training))
            if save:
                rec["blocks"].append(dict(xin=cur, h1=h1, mean1=mean1, rstd1=rstd1, qkv=qkv, att=att, lse=lse, xmid=xmid,
                                          h2=h2, mean2=mean2, rstd2=rstd2, u=u, a=a))
                if self.fp8:
                    rec["blocks"][-1].update(h1_8=h1_8, att8=att8, h2_8=h2_8, a8=a8)
            cur = xout
        if save:
            self._saved[slot] = rec
        return cur, probs

    # backward ---------------------------------------------------------------
    def backward(self, g: torch.Tensor, slot: str = "a", reducer: Optional[GradReducer] = None,
                 scale_key: Optional[str] = None) -> torch.Tensor:
        """g: fp32 [M, D] gradient wrt the stack output (overwritten in place; returned
        holding the gradient wrt the stack input).  Parameter gradients are ACCUMULATED
        into the store's flat gradient buffer.  `scale_key` (fp8 operands only) names the delayed-scaling state this pass
        uses; default = the slot, i.e. one state per kind of pass (callers that merely rotate slot names over the same kind
        of pass -- vit_core._functions.StackRunner -- pass one fixed key)."""
        if self.fp8:
            return self._backward_fp8(g, slot, reducer, scale_key or slot)
        st, D, H, F, dh = self.store, self.D, self.H, self.F, self.dh
        rec = self._saved[slot]
        B, T, seed, training = rec["B"], rec["T"], rec["seed"], rec["training"]
        M = B * T
        dev = g.device
        w = self.ws.get
        bw = f"bwd{M}."     # keyed by the row count: slots of different sizes (DINO local / global crops) keep their own buffers
        gm = w(bw + "gm", (M, D), BF16, dev)
        du = w(bw + "du", (M, F), BF16, dev)
        dh_ = w(bw + "dh", (M, D), BF16, dev)
        dqkv = w(bw + "dqkv", (M, 3 * D), BF16, dev)
        delta = w(bw + "delta", (B, H, T), F32, dev)
        gv = st.gview
        last = self.L - 1
        # The four weight gradients of a block run as ONE launch at the end of the block's backward (ops.gemm_tn_batch: one split
        # count and one reduce pass for all their tiles instead of four rounds of the CUs with 66 MB of partial tiles each).  Their
        # gradient operands must then all be alive at that point: the LayerNorm-2 backward writes its bf16 image into a second
        # buffer (gm2) instead of over gm.  VITSSL_TN_BATCH=0: one launch per gradient, as before.
        batch = _os.environ.get("VITSSL_TN_BATCH", "1") != "0"
        gm2 = w(bw + "gm2", (M, D), BF16, dev) if batch else gm
        # top of the chain: dropout-mask + cast of g, and the last block's linear_out bias grad
        ops.grad_mask_cast(g, gm, gv(self._n(last, "feed_forward.linear_out.bias")), self._drop(last, 2, seed, training))
        for i in range(last, -1, -1):
            s = rec["blocks"][i]
            a_ = self.bp[i] + "self_attention."
            # MLP
            wgrads = [(gm, s["a"], gv(self._n(i, "feed_forward.linear_out.weight"), (D, F))),
                      (du, s["h2"], gv(self._n(i, "feed_forward.linear_in.weight"), (F, D))),
                      (gm2, s["att"], gv(a_ + "final_linear.weight", (D, D))),
                      (dqkv, s["h1"], st.span_view(a_ + "w_query.weight", a_ + "w_value.weight", (3 * D, D), grad=True))]
            ops.gemm_nt(gm, st.w(self._n(i, "w2") + ".T"), du, L.EPI_DGELU, aux=s["u"],
                        colsum=gv(self._n(i, "feed_forward.linear_in.bias")))   # s["u"] holds g' = keep*scale*gelu'(u)
            if not batch:
                ops.gemm_tn(*wgrads[0])
            ops.gemm_nt(du, st.w(self._n(i, "w1") + ".T"), dh_, L.EPI_BF16)
            if not batch:
                ops.gemm_tn(*wgrads[1])
            ops.layernorm_bwd(dh_, s["xmid"], s["mean2"], s["rstd2"], st.view(self._n(i, "layer_norm2.weight")), g, g, gm2,
                              gv(self._n(i, "layer_norm2.weight")), gv(self._n(i, "layer_norm2.bias")), None,
                              self._drop(i, 0, seed, training))
            # attention
            ops.gemm_nt(gm2, st.w(self._n(i, "wo") + ".T"), dh_, L.EPI_BF16)
            if not batch:
                ops.gemm_tn(*wgrads[2])
            ops.attn_bwd(s["qkv"], s["att"], dh_, s["lse"], dqkv, delta, B, T, H, dh)
            ops.gemm_nt(dqkv, st.w(self._n(i, "wqkv") + ".T"), dh_, L.EPI_BF16)
            if batch:
                ops.gemm_tn_batch(wgrads)      # before the LayerNorm-1 backward below overwrites gm for the next block
            else:
                ops.gemm_tn(*wgrads[3])
            if i > 0:
                ops.layernorm_bwd(dh_, s["xin"], s["mean1"], s["rstd1"], st.view(self._n(i, "layer_norm1.weight")), g, g, gm,
                                  gv(self._n(i, "layer_norm1.weight")), gv(self._n(i, "layer_norm1.bias")),
                                  gv(self._n(i - 1, "feed_forward.linear_out.bias")), self._drop(i - 1, 2, seed, training))
            else:
                ops.layernorm_bwd(dh_, s["xin"], s["mean1"], s["rstd1"], st.view(self._n(i, "layer_norm1.weight")), g, g, None,
                                  gv(self._n(i, "layer_norm1.weight")), gv(self._n(i, "layer_norm1.bias")), None, ops.NO_DROP)
            if reducer is not None:
                # every gradient of block i is final, except linear_out.bias of block i-1,
                # which belongs to the next (lower) range
                lo, hi = self.block_span(i)
                reducer.ready(lo, hi)
        return g

    # fp8 input-gradient GEMMs ----------------------------------------------------
    @staticmethod
    def _scale_from_amax(amax: torch.Tensor, margin: int) -> torch.Tensor:
        """2^(floor(log2(448 / amax)) - margin) on the binary representation (the rule of csrc/fp8.hip)."""
        mant, e = torch.frexp(amax)                     # amax = mant * 2^e, mant in [0.5, 1)
        k = (9 - e - (mant > 0.875).to(e.dtype) - margin).clamp(-120, 120)
        # an exact power of two from its exponent field (torch.ldexp multiplies by pow(2, k), which is not exact on the GPU)
        return torch.bitwise_left_shift(k.to(torch.int32) + 127, 23).view(torch.float32)

    def _grad_scale_state(self, dev, slot: str = "a"):
        gs = self._gs_by_slot.get(slot)
        if gs is None or gs["scale"].device != dev:
            mk = lambda v: torch.full((self.L, 4), v, dtype=F32, device=dev)  # noqa: E731
            gs = {"scale": mk(1.0), "inv": mk(1.0), "amax": mk(0.0), "used": mk(1.0), "valid": False}
            self._gs_by_slot[slot] = gs
        self._gs = gs
        return gs

    def _rescale(self, sel, margin: int):
        """scale / inv of the selected entries of the running slot from their recorded max |value| (left alone where that is 0
        or not finite)."""
        gs = self._gs
        a = gs["amax"][sel]
        ok = (a > 0) & torch.isfinite(a)
        new = torch.where(ok, self._scale_from_amax(torch.where(ok, a, torch.ones_like(a)), margin), gs["scale"][sel])
        gs["scale"][sel] = new
        gs["inv"][sel] = 1.0 / new

    def fp8_grad_scales(self, slot: Optional[str] = None) -> torch.Tensor:
        """[L, 4] scales the last backward (of `slot`, default: whichever ran last) quantised its gradient operands with
        (tests hand them to the oracle)."""
        gs = self._gs if slot is None else self._gs_by_slot[slot]
        return gs["used"].clone()

    def recalibrate_fp8(self):
        """Derive the gradient scales from the tensors themselves in the next backward of every slot (as the first one does)."""
        for gs in self._gs_by_slot.values():
            gs["valid"] = False

    def _backward_fp8(self, g: torch.Tensor, slot: str, reducer: Optional[GradReducer], scale_key: str) -> torch.Tensor:
        """The schedule of backward() with all eight backward GEMMs of every block on e4m3 operands: the transposed
        weight images of the store, the activation images saved by the forward, and gradient images written by the
        producers of the gradients (LayerNorm backward, the dGELU epilogue, the attention backward's store phase)."""
        st, D, H, F, dh = self.store, self.D, self.H, self.F, self.dh
        rec = self._saved[slot]
        B, T, seed, training = rec["B"], rec["T"], rec["seed"], rec["training"]
        M = B * T
        dev = g.device
        w = self.ws.get
        bw = f"bwd{M}."     # keyed by the row count: slots of different sizes (DINO local / global crops) keep their own buffers
        gm8, du8, dq8 = w(bw + "gm8", (M, D), FP8, dev), w(bw + "du8", (M, F), FP8, dev), w(bw + "dq8", (M, 3 * D), FP8, dev)
        batch = _os.environ.get("VITSSL_TN_BATCH", "1") != "0"       # as in backward(): the block's four weight gradients in one launch
        gm8b = w(bw + "gm8b", (M, D), FP8, dev) if batch else gm8
        dh_ = w(bw + "dh", (M, D), BF16, dev)
        delta = w(bw + "delta", (B, H, T), F32, dev)
        gv = st.gview
        gs = self._grad_scale_state(dev, scale_key)
        sc = lambda i, t: gs["scale"][i, t:t + 1]   # noqa: E731
        inv = lambda i, t: gs["inv"][i, t:t + 1]    # noqa: E731
        am = lambda i, t: gs["amax"][i, t:t + 1]    # noqa: E731
        jit = not gs["valid"]
        # bf16 images of the gradient operands exist only in the self-calibrating first backward (settle() re-quantises
        # from them); afterwards every consumer reads the e4m3 image and the producers skip the bf16 store
        gm = w(bw + "gm", (M, D), BF16, dev) if jit else None
        du = w(bw + "du", (M, F), BF16, dev) if jit else None
        dqkv = w(bw + "dqkv", (M, 3 * D), BF16, dev) if jit else None

        def settle(i, t, x, x8):
            """first backward only: the producer has just recorded max|x|; take the scale from it and quantise again"""
            if jit:
                self._rescale((slice(i, i + 1), slice(t, t + 1)), margin=0)
                ops.quantize_fp8(x, x8, scale=sc(i, t))

        last = self.L - 1
        ops.grad_mask_cast_fp8(g, gm, gm8, sc(last, 0), am(last, 0), gv(self._n(last, "feed_forward.linear_out.bias")),
                               self._drop(last, 2, seed, training))
        settle(last, 0, gm, gm8)
        for i in range(last, -1, -1):
            s = rec["blocks"][i]
            a_ = self.bp[i] + "self_attention."
            (w2t, a2), (w1t, a1), (wot, ao), (wqt, aq) = (st.w8t(self._n(i, k)) for k in ("w2", "w1", "wo", "wqkv"))
            # MLP
            ops.gemm_fp8_nt(gm8, w2t, du, L.EPI_DGELU, alpha=a2, alpha2=inv(i, 0), aux=s["u"],
                            colsum=gv(self._n(i, "feed_forward.linear_in.bias")), out_fp8=du8, out_scale=sc(i, 1), out_amax=am(i, 1))
            settle(i, 1, du, du8)
            wgrads = [(gm8, s["a8"], gv(self._n(i, "feed_forward.linear_out.weight"), (D, F)), None, inv(i, 0)),
                      (du8, s["h2_8"], gv(self._n(i, "feed_forward.linear_in.weight"), (F, D)), None, inv(i, 1)),
                      (gm8b, s["att8"], gv(a_ + "final_linear.weight", (D, D)), None, inv(i, 2)),
                      (dq8, s["h1_8"], st.span_view(a_ + "w_query.weight", a_ + "w_value.weight", (3 * D, D), grad=True), None, inv(i, 3))]
            single = lambda j: ops.gemm_fp8_tn(wgrads[j][0], wgrads[j][1], wgrads[j][2], alpha2=wgrads[j][4])   # noqa: E731
            if not batch:
                single(0)
            ops.gemm_fp8_nt(du8, w1t, dh_, L.EPI_BF16, alpha=a1, alpha2=inv(i, 1))
            if not batch:
                single(1)
            ops.layernorm_bwd_fp8(dh_, s["xmid"], s["mean2"], s["rstd2"], st.view(self._n(i, "layer_norm2.weight")), g, g, gm, gm8b,
                                  sc(i, 2), am(i, 2), gv(self._n(i, "layer_norm2.weight")), gv(self._n(i, "layer_norm2.bias")), None,
                                  self._drop(i, 0, seed, training))
            settle(i, 2, gm, gm8b)
            # attention
            ops.gemm_fp8_nt(gm8b, wot, dh_, L.EPI_BF16, alpha=ao, alpha2=inv(i, 2))
            if not batch:
                single(2)
            ops.attn_bwd(s["qkv"], s["att"], dh_, s["lse"], dqkv, delta, B, T, H, dh, dqkv_fp8=dq8, scale=sc(i, 3), amax=am(i, 3))
            settle(i, 3, dqkv, dq8)
            ops.gemm_fp8_nt(dq8, wqt, dh_, L.EPI_BF16, alpha=aq, alpha2=inv(i, 3))
            if batch:
                ops.gemm_fp8_tn_batch(wgrads)      # before the LayerNorm-1 backward below overwrites gm8 for the next block
            else:
                single(3)
            if i > 0:
                ops.layernorm_bwd_fp8(dh_, s["xin"], s["mean1"], s["rstd1"], st.view(self._n(i, "layer_norm1.weight")), g, g, gm, gm8,
                                      sc(i - 1, 0), am(i - 1, 0), gv(self._n(i, "layer_norm1.weight")), gv(self._n(i, "layer_norm1.bias")),
                                      gv(self._n(i - 1, "feed_forward.linear_out.bias")), self._drop(i - 1, 2, seed, training))
                settle(i - 1, 0, gm, gm8)
            else:
                ops.layernorm_bwd(dh_, s["xin"], s["mean1"], s["rstd1"], st.view(self._n(i, "layer_norm1.weight")), g, g, None,
                                  gv(self._n(i, "layer_norm1.weight")), gv(self._n(i, "layer_norm1.bias")), None, ops.NO_DROP)
            if reducer is not None:
                lo, hi = self.block_span(i)
                reducer.ready(lo, hi)
        # scales of the NEXT backward: this step's max |value| with one bit of headroom
        gs["used"].copy_(gs["scale"])
        self._rescale((slice(None), slice(None)), margin=1)
        gs["amax"].zero_()
        gs["valid"] = True
        return g
