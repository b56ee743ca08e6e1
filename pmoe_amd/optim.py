"""Fused optimizer tail of the stage-2 step -- drop-ins for the three torch calls of the
reference trainer (``trainer/train_2.py:157-165,184``; hyper-parameters ``conf/stage_2_pmoe.yaml:11,137-144``):

    torch.nn.utils.clip_grad_norm_(model.parameters(), 1.0)   ->  pmoe_amd.optim.clip_grad_norm_
    torch.optim.Adam(params, lr, betas, eps, wd, amsgrad=True) ->  pmoe_amd.optim.FusedAdam   (same constructor)
    torch.optim.swa_utils.AveragedModel(model)                 ->  pmoe_amd.optim.FusedAveragedModel

Each call is one or two HIP launches over a chunk table (``csrc/optim.hip``) instead of several launches -- and, in
``check_grad_norm`` (utils/nn.py:10-19), one ``.item()`` host sync -- per parameter tensor (620 for E=4).  The
returned gradient norm is a device tensor; nothing here synchronises with the host.  ``FusedAdam.step(clip=norm)``
consumes the clip coefficient straight from device memory, so clip + step is: 2 launches for the norm, 1 for Adam.
"""
import ctypes as C

import numpy as np
import torch

from . import hip
from .hip import check, load, stream_ptr

CHUNK = 16384                      # PMOE_OPT_CHUNK (include/pmoe_hip.h)
F32 = torch.float32


class OptTensor(C.Structure):      # pmoe_opt_tensor
    _fields_ = [("param", C.c_void_p), ("grad", C.c_void_p), ("exp_avg", C.c_void_p), ("exp_avg_sq", C.c_void_p),
                ("max_exp_avg_sq", C.c_void_p), ("swa", C.c_void_p), ("numel", C.c_int64), ("bc1", C.c_float),
                ("bc2_sqrt", C.c_float)]


_ROW = np.dtype([("param", "<u8"), ("grad", "<u8"), ("exp_avg", "<u8"), ("exp_avg_sq", "<u8"), ("max_exp_avg_sq", "<u8"),
                 ("swa", "<u8"), ("numel", "<i8"), ("bc1", "<f4"), ("bc2_sqrt", "<f4")])
assert _ROW.itemsize == C.sizeof(OptTensor) == 64


def _f32_cuda(t, what):
    if not t.is_cuda or t.dtype != F32 or not t.is_contiguous():
        raise RuntimeError(f"pmoe_amd.optim: {what} must be a contiguous float32 tensor on the MI355X (cuda) device; "
                           f"got {t.dtype} on {t.device} (there is no CPU path)")
    return t.data_ptr()


class _Table:
    """Device copy of a pmoe_opt_tensor array + its chunk lists.  Tables are cached on the tuple of raw pointers they
    hold PLUS their element counts and the device (parameter / state storage is stable, and torch's caching allocator
    hands the gradient arena back at the same address step after step; the allocator also reuses addresses across
    models, so pointers alone would not identify a table), so the steady state uploads nothing; a new table goes up
    through pinned memory without blocking the host."""

    _cache = {}

    @classmethod
    def get(cls, key, device, build_rows, numels=()):
        key = (str(device), tuple(int(n) for n in numels)) + tuple(key)
        tab = cls._cache.get(key)
        if tab is None:
            if len(cls._cache) > 8:
                cls._cache.clear()
            tab = cls._cache[key] = cls(build_rows(), device)
        return tab

    def __init__(self, rows, device):
        self.n = len(rows)
        self.host = torch.from_numpy(rows.view(np.uint8).reshape(-1).copy()).pin_memory()     # kept alive with the table
        self.table = self.host.to(device, non_blocking=True)
        ct, ci = [], []
        for t, n in enumerate(rows["numel"]):
            k = (int(n) + CHUNK - 1) // CHUNK
            ct.extend([t] * k)
            ci.extend(range(k))
        self._chunks_host = torch.tensor([ct, ci], dtype=torch.int32).pin_memory()
        chunks = self._chunks_host.to(device, non_blocking=True)
        self.chunk_tensor, self.chunk_index = chunks[0], chunks[1]
        self.n_chunks = len(ct)

    def args(self):
        return (C.c_void_p(self.table.data_ptr()), C.c_void_p(self.chunk_tensor.data_ptr()),
                C.c_void_p(self.chunk_index.data_ptr()), self.n_chunks)


def _rows(n):
    return np.zeros(n, dtype=_ROW)


def _grad_table(params):
    ps = [p for p in params if p.grad is not None]
    if not ps:
        return None, ps
    ptrs = tuple(_f32_cuda(p.grad, "gradient") for p in ps)

    def build():
        rows = _rows(len(ps))
        rows["grad"] = ptrs
        rows["numel"] = [p.numel() for p in ps]
        return rows
    return _Table.get(("g",) + ptrs, ps[0].device, build, [p.numel() for p in ps]), ps


def clip_grad_norm_(parameters, max_norm, norm_type=2.0, scale=True):
    """``torch.nn.utils.clip_grad_norm_`` (L2 only): returns the total norm as a 0-d DEVICE tensor and scales the
    gradients in place by ``min(1, max_norm / (norm + 1e-6))``.  ``scale=False`` only measures (the reference's
    ``check_grad_norm``, utils/nn.py:10-19, without its per-tensor ``.item()`` syncs); the returned tensor carries
    ``.clip_state`` for ``FusedAdam.step(clip=...)``."""
    if float(norm_type) != 2.0:
        raise NotImplementedError("pmoe_amd.optim.clip_grad_norm_: only the L2 norm (the reference's default) is fused")
    if isinstance(parameters, torch.Tensor):
        parameters = [parameters]
    tab, ps = _grad_table(list(parameters))
    if tab is None:
        return torch.zeros((), dtype=F32, device="cuda")
    dev = ps[0].device
    partial = torch.empty(tab.n_chunks, dtype=F32, device=dev)
    norm = torch.empty(2, dtype=F32, device=dev)
    check(load().pmoe_mt_grad_norm(*tab.args(), float(max_norm), C.c_void_p(partial.data_ptr()), C.c_void_p(norm.data_ptr()),
                                   int(bool(scale)), stream_ptr()), "pmoe_mt_grad_norm")
    if scale:
        torch.autograd.graph.increment_version([p.grad for p in ps])
    total = norm[0]
    total.clip_state = norm
    return total


class FusedAdam(torch.optim.Optimizer):
    """``torch.optim.Adam`` semantics (same arguments, same per-parameter ``state`` keys ``step`` / ``exp_avg`` /
    ``exp_avg_sq`` / ``max_exp_avg_sq``, so ``state_dict()`` checkpoints interchange with the reference's optimizer,
    train_2.py:300-310), updated by one multi-tensor HIP launch per parameter group."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, amsgrad=False):
        if lr < 0 or eps < 0 or not (0 <= betas[0] < 1) or not (0 <= betas[1] < 1) or weight_decay < 0:
            raise ValueError("FusedAdam: invalid hyper-parameter")
        super().__init__(params, dict(lr=lr, betas=tuple(betas), eps=eps, weight_decay=weight_decay, amsgrad=amsgrad))

    @torch.no_grad()
    def step(self, closure=None, clip=None):
        """``clip``: the tensor returned by ``clip_grad_norm_(..., scale=False)`` -- its coefficient is applied to the
        gradients inside the update kernel (gradients stay unscaled in memory)."""
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        clip_state = getattr(clip, "clip_state", None) if clip is not None else None
        if clip is not None and clip_state is None:
            raise ValueError("FusedAdam.step(clip=...): pass the tensor returned by pmoe_amd.optim.clip_grad_norm_")
        for group in self.param_groups:
            ps = [p for p in group["params"] if p.grad is not None]
            if not ps:
                continue
            b1, b2 = group["betas"]
            ams = bool(group["amsgrad"])
            keys = ("exp_avg", "exp_avg_sq") + (("max_exp_avg_sq",) if ams else ())
            steps = []
            for p in ps:
                if p.grad.is_sparse:
                    raise RuntimeError("FusedAdam does not support sparse gradients")
                st = self.state[p]
                if not st:
                    st["step"] = torch.tensor(0.0, dtype=F32)      # torch keeps `step` as a CPU f32 scalar tensor
                    for k in keys:
                        st[k] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st["step"] += 1
                steps.append(float(st["step"]))
            cols = {"param": tuple(_f32_cuda(p, "parameter") for p in ps),
                    "grad": tuple(_f32_cuda(p.grad, "gradient") for p in ps)}
            for k in keys:
                cols[k] = tuple(_f32_cuda(self.state[p][k], k) for p in ps)
            uniform = min(steps) == max(steps)

            def build():
                rows = _rows(len(ps))
                for k, v in cols.items():
                    rows[k] = v
                rows["numel"] = [p.numel() for p in ps]
                rows["bc1"] = [1.0 - b1 ** k for k in steps]
                rows["bc2_sqrt"] = [(1.0 - b2 ** k) ** 0.5 for k in steps]
                return rows
            key = ("a",) + tuple(v for c in cols.values() for v in c) + (() if uniform else tuple(steps))
            tab = _Table.get(key, ps[0].device, build, [p.numel() for p in ps])
            # all tensors at the same step (the normal case): bias corrections travel as kernel arguments and the cached
            # table is reused; otherwise the per-tensor values of a freshly built table are used (argument < 0)
            bc1 = 1.0 - b1 ** steps[0] if uniform else -1.0
            bc2s = (1.0 - b2 ** steps[0]) ** 0.5 if uniform else -1.0
            check(load().pmoe_mt_adam(*tab.args(), float(group["lr"]), float(b1), float(b2), float(group["eps"]),
                                      float(group["weight_decay"]), int(ams), float(bc1), float(bc2s),
                                      C.c_void_p(clip_state.data_ptr()) if clip_state is not None else None, stream_ptr()),
                  "pmoe_mt_adam")
            # the kernel wrote through raw pointers: tell autograd (and the engine's packed-weight cache, which keys
            # on the version counters) that these tensors changed in place
            torch.autograd.graph.increment_version(ps)
        return loss


class FusedAveragedModel(torch.optim.swa_utils.AveragedModel):
    """``AveragedModel(model)`` (train_2.py:120,184) whose ``update_parameters`` is one multi-tensor launch."""

    @torch.no_grad()
    def update_parameters(self, model):
        mine, theirs = list(self.module.parameters()), list(model.parameters())
        if len(mine) != len(theirs):
            raise ValueError("FusedAveragedModel: parameter lists differ")
        for a, p in zip(mine, theirs):
            if a.shape != p.shape:
                raise ValueError("FusedAveragedModel: parameter shapes differ")
        pp = tuple(_f32_cuda(p.detach(), "parameter") for p in theirs)
        aa = tuple(_f32_cuda(a.detach(), "averaged parameter") for a in mine)

        def build():
            rows = _rows(len(mine))
            rows["param"], rows["swa"] = pp, aa
            rows["numel"] = [p.numel() for p in theirs]
            return rows
        tab = _Table.get(("s",) + pp + aa, mine[0].device, build, [p.numel() for p in theirs])
        # host-side mirror of n_averaged: no device->host sync per update when the buffer lives on the GPU
        n = self.__dict__.get("_n_host")
        if n is None or self.__dict__.get("_n_seen") != (id(self.n_averaged), self.n_averaged._version):
            n = int(self.n_averaged.item())      # first call, or the buffer was replaced / loaded from a checkpoint
        check(load().pmoe_mt_swa_update(*tab.args(), n, stream_ptr()), "pmoe_mt_swa_update")
        torch.autograd.graph.increment_version(mine)
        self.n_averaged += 1
        self.__dict__["_n_host"], self.__dict__["_n_seen"] = n + 1, (id(self.n_averaged), self.n_averaged._version)
