"""Host-side runtime plumbing: bf16 weight shadows, gradient buffers, the flat parameter arena and dropout seeds.

MI355X-first layout: all trainable parameters live in ONE flat fp32 arena (plus a flat fp32 gradient arena and a flat
bf16 shadow), so the per-step work on parameters is three streams over contiguous HBM -- zero the gradients, one
bucketed RCCL all-reduce over slices of the gradient arena, one fused Adam pass that also refreshes the bf16 shadow --
instead of ~350 small per-tensor launches.  ``nn.Parameter``s stay what the reference's code expects (named tensors in
``state_dict()``); they are simply views into the arena.
"""
from __future__ import annotations

import itertools
import weakref
from typing import Dict, Iterable, List, Optional

import torch

from . import ops

_ALIGN = 8  # elements: 16 B for the bf16 shadow, 32 B for fp32


class ParamArena:
    """Flatten a module's trainable parameters into contiguous data / grad / bf16-shadow buffers."""

    def __init__(self, params: Iterable[torch.nn.Parameter], adjacent: Iterable[tuple] = ()):
        """adjacent: tuples of parameters to lay out back to back (e.g. the query/key/value weights of an attention
        block, so that one GEMM can use them as a single [3D, D] matrix -- see fused_views)."""
        plist = [p for p in params if p.requires_grad]
        assert plist, "no trainable parameters"
        group_of = {}
        for grp in adjacent:
            grp = tuple(p for p in grp if p.requires_grad)
            if len(grp) > 1 and all(id(p) not in group_of for p in grp):
                for p in grp:
                    group_of[id(p)] = grp
        self.params: List[torch.nn.Parameter] = []
        seen = set()
        for p in plist:
            for q in group_of.get(id(p), (p,)):
                if id(q) not in seen:
                    seen.add(id(q))
                    self.params.append(q)
        dev = self.params[0].device
        assert dev.type == "cuda", "ParamArena needs device parameters"
        self.offsets: Dict[int, int] = {}
        off = 0
        for p in self.params:
            self.offsets[id(p)] = off
            off += (p.numel() + _ALIGN - 1) // _ALIGN * _ALIGN
        self.numel = off
        self.data = torch.zeros(off, device=dev, dtype=torch.float32)
        self.grad = torch.zeros(off, device=dev, dtype=torch.float32)
        self.shadow = torch.zeros(off, device=dev, dtype=torch.bfloat16)
        for p in self.params:
            o, n = self.offsets[id(p)], p.numel()
            self.data[o:o + n].copy_(p.data.reshape(-1).float())
            p.data = self.data[o:o + n].view(p.shape)
            p.grad = self.grad[o:o + n].view(p.shape)
            p._mmdti_arena = self
        self._version: Dict[int, int] = {}
        self.shadow16: Optional[torch.Tensor] = None      # fp16 shadow (fp16 forward-operand mode), built on first use
        self._epoch, self._epoch16 = 0, -1                # the fp16 shadow is valid while _epoch16 == _epoch
        self.refresh_shadow()
        self.adam_m: Optional[torch.Tensor] = None
        self.adam_v: Optional[torch.Tensor] = None
        self.step_count = 0

    def refresh_shadow(self):
        """Re-cast the whole fp32 arena into the bf16 shadow the GEMMs read."""
        ops.lib().mmdti_cast_f32_bf16(ops._stream(), self.data.data_ptr(), self.shadow.data_ptr(), self.numel, 0.0, 0, 0)
        for p in self.params:
            self._version[id(p)] = p._version
        self._epoch += 1

    def _fresh(self, p):
        """The shadow is refreshed by the constructor and by adam_step (which writes both through raw pointers).  Any
        OTHER in-place write to a parameter -- ``load_state_dict`` on a model already bound to the arena (the reference
        trains, then loads the best checkpoint into the same model, tasks/trainer.py:406-410), ``p.copy_()`` -- bumps the
        tensor's version counter: re-cast that parameter's slice before a GEMM reads it."""
        if p._version != self._version[id(p)]:
            o, n = self.offsets[id(p)], p.numel()
            ops.lib().mmdti_cast_f32_bf16(ops._stream(), self.data[o:o + n].data_ptr(), self.shadow[o:o + n].data_ptr(), n, 0.0, 0, 0)
            self._version[id(p)] = p._version
            self._epoch += 1

    def zero_grad(self):
        self.grad.zero_()
        for p in self.params:          # re-bind in case an optimizer dropped the views (set_to_none)
            if p.grad is None or p.grad.data_ptr() != self.grad.data_ptr() + 4 * self.offsets[id(p)]:
                o, n = self.offsets[id(p)], p.numel()
                p.grad = self.grad[o:o + n].view(p.shape)

    def bf16(self, p: torch.nn.Parameter) -> torch.Tensor:
        self._fresh(p)
        o, n = self.offsets[id(p)], p.numel()
        return self.shadow[o:o + n].view(p.shape)

    def _fresh16(self):
        """fp16 shadow of the whole arena (the weights of the fp16 forward-operand mode): one cast pass whenever the fp32 data
        has changed since the last one (an optimizer step, a reload)."""
        if self.shadow16 is None:
            self.shadow16 = torch.empty(self.numel, device=self.data.device, dtype=torch.float16)
        if self._epoch16 != self._epoch:
            ops.lib().mmdti_cast_f32_f16(ops._stream(), self.data.data_ptr(), self.shadow16.data_ptr(), self.numel, 0.0, 0, 0)
            self._epoch16 = self._epoch

    def f16(self, p: torch.nn.Parameter) -> torch.Tensor:
        self._fresh(p)
        self._fresh16()
        o, n = self.offsets[id(p)], p.numel()
        return self.shadow16[o:o + n].view(p.shape)

    def fused(self, plist):
        """(bf16 shadow, fp32 data, fp32 grad) views spanning parameters that sit back to back in the arena (rows
        concatenated), or None."""
        o0 = self.offsets.get(id(plist[0]))
        if o0 is None:
            return None
        o, cols = o0, plist[0].shape[1:]
        for p in plist:
            if self.offsets.get(id(p)) != o or p.shape[1:] != cols:
                return None
            o += p.numel()
        for p in plist:
            self._fresh(p)
        shape = (sum(p.shape[0] for p in plist),) + tuple(cols)
        w16 = None
        if ops.FWD_F16:
            self._fresh16()
            w16 = self.shadow16[o0:o].view(shape)
        # (bf16 shadow, fp32 data, fp32 gradient, forward-GEMM shadow: fp16 in the fp16 forward-operand mode, else the bf16 one)
        return self.shadow[o0:o].view(shape), self.data[o0:o].view(shape), self.grad[o0:o].view(shape), (w16 if w16 is not None else self.shadow[o0:o].view(shape))

    def adam_step(self, lr, betas=(0.9, 0.999), eps=1e-6, weight_decay=0.0, max_norm: Optional[float] = None, step_state=None):
        """torch.optim.Adam semantics (tasks/trainer.py:160) + optional global-norm clipping (:274), one fused pass.
        step_state: device-resident schedule (ops.step_state_advance) -- lr and the step count then live on the device."""
        if self.adam_m is None:
            self.adam_m = torch.zeros_like(self.data)
            self.adam_v = torch.zeros_like(self.data)
        self.step_count += 1
        scale = None
        if max_norm is not None:
            buf = torch.zeros(1 + 2048, device=self.data.device, dtype=torch.float32)      # the sum | per-workgroup partials (fixed-order fold)
            ss = buf[:1]
            ops.sumsq(self.grad, ss, buf[1:])
            # clip_grad_norm_: coef = max_norm / (norm + 1e-6), clamped to 1 (tiny scalar math; stays on device, no sync)
            scale = torch.clamp(max_norm / (ss.sqrt() + 1e-6), max=1.0)
        # (the Adam pass writes BOTH 16-bit shadows -- bf16 for the backward GEMMs, fp16 for the forward ones -- through raw pointers:
        #  a captured graph's replays keep them fresh too; the fp16 one exists once a forward GEMM has asked for it)
        ops.adam_step(self.data, self.grad, self.adam_m, self.adam_v, self.shadow, lr, betas[0], betas[1], eps, weight_decay,
                      self.step_count, scale, step_state, p_f16=self.shadow16 if self._epoch16 == self._epoch else None)
        self._epoch += 1
        if self.shadow16 is not None and self._epoch16 == self._epoch - 1:
            self._epoch16 = self._epoch


class _ShadowCache:
    """bf16 copies of parameters that are not arena-managed (unit tests, ad-hoc modules): re-cast when the parameter's
    version counter or storage changes."""

    def __init__(self):
        self._c: Dict[int, tuple] = {}

    def get(self, p: torch.Tensor, f16: bool = False) -> torch.Tensor:
        key = (id(p), f16)
        ent = self._c.get(key)
        ver = (p.data_ptr(), p._version, tuple(p.shape))
        # id() values are recycled once a parameter dies: the weak reference proves the entry belongs to THIS tensor
        if ent is None or ent[0]() is not p or ent[1] != ver:
            if len(self._c) > 4096:
                self._c = {k: e for k, e in self._c.items() if e[0]() is not None}
            src = p.detach().contiguous()
            ent = (weakref.ref(p), ver, ops.cast_act16(src) if f16 else ops.cast_bf16(src))
            self._c[key] = ent
        return ent[2]


_shadow_cache = _ShadowCache()


def wbf16(p: torch.Tensor) -> torch.Tensor:
    """bf16 view/copy of a weight for the MFMA GEMMs."""
    arena = getattr(p, "_mmdti_arena", None)
    if arena is not None:
        return arena.bf16(p)
    return _shadow_cache.get(p)


def wfwd(p: torch.Tensor) -> torch.Tensor:
    """The weight as the operand of a FORWARD GEMM: the bf16 shadow, or -- fp16 forward-operand mode (ops.FWD_F16) -- the fp16 one."""
    if not ops.FWD_F16:
        return wbf16(p)
    arena = getattr(p, "_mmdti_arena", None)
    if arena is not None:
        return arena.f16(p)
    return _shadow_cache.get(p, f16=True)


def fused_views(plist):
    """Views over parameters laid out back to back in one ParamArena (see ParamArena(adjacent=...)), else None."""
    arena = getattr(plist[0], "_mmdti_arena", None)
    if arena is None or any(getattr(p, "_mmdti_arena", None) is not arena or not p.requires_grad for p in plist):
        return None
    return arena.fused(plist)


def gbuf(p: torch.Tensor) -> Optional[torch.Tensor]:
    """fp32 gradient buffer of a parameter that kernels accumulate into (atomically); None if frozen."""
    if not p.requires_grad:
        return None
    if p.grad is None:
        p.grad = torch.zeros_like(p, memory_format=torch.contiguous_format)
    return p.grad


class DropoutState:
    """Counter-based dropout: every forward call of a module takes a fresh 64-bit seed; sites within the call are
    numbered, and the backward regenerates the masks from (seed, site)."""

    def __init__(self, seed: int = 0x5EED):
        self.base = seed
        self._calls = itertools.count(1)

    def next_seed(self) -> int:
        return (self.base * 0x9E3779B97F4A7C15 + next(self._calls) * 0xD1B54A32D192ED03) & 0xFFFFFFFFFFFFFFFF

    def reseed(self, seed: int):
        self.base = seed
        self._calls = itertools.count(1)


dropout_state = DropoutState()


# ---- gradient-ready notifications (consumed by parallel.ArenaReducer to overlap the all-reduce with backward) ----
_grad_ready_hooks: Dict[int, object] = {}


def add_grad_ready_hook(owner, fn):
    """Register `fn(params)`; one hook per owner (a FineTuner).  Each reducer ignores parameters outside its own arena,
    so several engines in one process do not disturb each other.  Held weakly: a dead owner's hook disappears."""
    ref = weakref.WeakMethod(fn) if hasattr(fn, "__self__") else (lambda f=fn: f)
    _grad_ready_hooks[id(owner)] = ref


def remove_grad_ready_hook(owner):
    _grad_ready_hooks.pop(id(owner), None)


def notify_grads_ready(params):
    """Called by the backward of a module once every gradient of `params` has been enqueued on the stream."""
    if not _grad_ready_hooks:
        return
    params = list(params)
    for key, ref in list(_grad_ready_hooks.items()):
        fn = ref()
        if fn is None:
            del _grad_ready_hooks[key]
        else:
            fn(params)
