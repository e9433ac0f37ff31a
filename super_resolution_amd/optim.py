"""Adam over one flat parameter buffer.

The reference trains with ``torch.optim.Adam(net.parameters(), lr=1e-4, betas=(0.9, 0.999))``
(/root/reference/esrgan/train.py:117-122).  The generator has 702 parameter tensors; even torch's multi-tensor Adam
spends milliseconds of host time per step walking them, which at the reference's patch size is a fifth of the
iteration.  ``FlatAdam`` is the same optimizer (same update, same ``state_dict`` layout, works under ``GradScaler``
and lr schedulers) that moves the parameters of each group into one contiguous buffer — every ``Parameter`` stays the
same object and becomes a view of it — so that the update is a single kernel launch (``sr355_adam_step``).  The
generator's backward hands out its gradients as slices of one buffer in the same order, which is used in place; any
other gradient layout is gathered first.
"""
from typing import List

import torch
from torch import Tensor

from . import ops


class _Group:
    __slots__ = ("params", "flat", "exp_avg", "exp_avg_sq", "offsets", "n", "step", "step_t", "scratch")


class FlatAdam(torch.optim.Optimizer):
    """``torch.optim.Adam`` (no amsgrad, no maximize) with one kernel launch per parameter group.

    Construct it after the module is on its device, as with any optimizer.  Parameters with ``requires_grad=False``
    are left alone; a trainable parameter whose ``.grad`` is ``None`` at ``step()`` while others have gradients
    is an error (torch's Adam would skip it; a flat update cannot).

    ``reduce_group``: a ``torch.distributed`` process group over which the gradients are averaged before every
    update, as ONE all-reduce of the flat gradient buffer per parameter group (data-parallel training of a module
    whose backward does not reduce its gradients itself, e.g. the discriminator: the replacement of
    ``nn.DataParallel(netD)`` at /root/reference/esrgan/train.py:108-111 with one process per GPU)."""

    def __init__(self, params, lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8, weight_decay: float = 0.0,
                 reduce_group=None, reduce_algorithm: str = "all_reduce"):
        self._reduce_group = reduce_group
        self._reduce_algorithm = reduce_algorithm      # "all_reduce" | "direct" (ddp.direct_mean_)
        self._pending = None
        if lr < 0 or eps < 0 or not 0 <= betas[0] < 1 or not 0 <= betas[1] < 1 or weight_decay < 0:
            raise ValueError("invalid Adam hyper-parameters")
        super().__init__(params, dict(lr=lr, betas=tuple(betas), eps=eps, weight_decay=weight_decay))
        self._groups: List[_Group] = [self._flatten(g) for g in self.param_groups]

    def add_param_group(self, param_group) -> None:
        """``torch.optim.Optimizer.add_param_group`` (fine-tuning: unfreeze a layer and hand it to the optimizer):
        the new group gets a flat buffer of its own."""
        super().add_param_group(param_group)
        if hasattr(self, "_groups"):          # the base constructor adds the initial groups through this method too
            self._groups.append(self._flatten(self.param_groups[-1]))

    # -- set-up ---------------------------------------------------------------------------
    def _flatten(self, group) -> _Group:
        ps = [p for p in group["params"] if p.requires_grad]
        fg = _Group()
        fg.params, fg.step, fg.scratch = ps, 0, None
        fg.step_t = torch.tensor(0.0)
        if not ps:
            fg.flat = fg.exp_avg = fg.exp_avg_sq = None
            fg.offsets, fg.n = [], 0
            return fg
        dev = ps[0].device
        for p in ps:
            if p.dtype != torch.float32 or p.device != dev or not p.is_cuda:
                raise RuntimeError("FlatAdam needs the fp32 parameters of a group on one ROCm device")
        sizes = [(p.numel() + 3) // 4 * 4 for p in ps]          # every slice 16-byte aligned
        fg.offsets, off = [], 0
        for n in sizes:
            fg.offsets.append(off)
            off += n
        fg.n = fg.offsets[-1] + ps[-1].numel()                  # elements the update covers (no tail padding)
        with torch.no_grad():
            fg.flat = torch.zeros(fg.n, dtype=torch.float32, device=dev)
            for p, o in zip(ps, fg.offsets):
                view = fg.flat[o:o + p.numel()].view(p.shape)
                view.copy_(p.data)
                p.data = view
        fg.exp_avg = torch.zeros_like(fg.flat)
        fg.exp_avg_sq = torch.zeros_like(fg.flat)
        for p, o in zip(ps, fg.offsets):      # torch.optim.Adam's per-parameter state, as views of the flat state
            self.state[p] = {"step": fg.step_t, "exp_avg": fg.exp_avg[o:o + p.numel()].view(p.shape),
                             "exp_avg_sq": fg.exp_avg_sq[o:o + p.numel()].view(p.shape)}
        return fg

    def state_dict(self):
        # torch.optim.Adam's layout; every parameter gets a step tensor of its own (torch's Adam increments each one it
        # is handed, so a shared one must not leak into it)
        sd = super().state_dict()
        sd["state"] = {k: {**v, "step": torch.tensor(float(v["step"]))} for k, v in sd["state"].items()}
        return sd

    def load_state_dict(self, state_dict) -> None:
        super().load_state_dict(state_dict)          # replaces the state tensors by copies: fold them back in
        for fg in self._groups:
            steps = set()
            for p, o in zip(fg.params, fg.offsets):
                st = self.state.get(p)
                if not st:
                    continue
                fg.exp_avg[o:o + p.numel()].view(p.shape).copy_(st["exp_avg"])
                fg.exp_avg_sq[o:o + p.numel()].view(p.shape).copy_(st["exp_avg_sq"])
                steps.add(int(float(st["step"])))
            if len(steps) > 1:
                raise RuntimeError("FlatAdam keeps one step count per group; the loaded state has several")
            fg.step = steps.pop() if steps else 0
            fg.step_t = torch.tensor(float(fg.step))
            for p, o in zip(fg.params, fg.offsets):
                self.state[p] = {"step": fg.step_t, "exp_avg": fg.exp_avg[o:o + p.numel()].view(p.shape),
                                 "exp_avg_sq": fg.exp_avg_sq[o:o + p.numel()].view(p.shape)}

    # -- the update -----------------------------------------------------------------------
    def _flat_grad(self, fg: _Group):
        grads = [p.grad for p in fg.params]
        missing = sum(g is None for g in grads)
        if missing == len(grads):
            return None
        if missing:
            raise RuntimeError(f"FlatAdam: {missing} of {len(grads)} trainable parameters have no gradient; freeze them "
                               "(requires_grad_(False)) before building the optimizer")
        # in place when the gradients already are consecutive 16-byte aligned slices of one buffer, in order
        base = grads[0].data_ptr()
        if base % 16 == 0 and all(g.data_ptr() == base + 4 * o and g.is_contiguous() and g.dtype == torch.float32
                                  for g, o in zip(grads, fg.offsets)):
            storage = grads[0].untyped_storage()
            first = (base - storage.data_ptr()) // 4
            if storage.data_ptr() + storage.nbytes() >= base + 4 * fg.n:
                return torch.empty(0, dtype=torch.float32, device=fg.flat.device).set_(storage, first, (fg.n,), (1,))
        if fg.scratch is None:
            fg.scratch = torch.zeros_like(fg.flat)
        views = [fg.scratch[o:o + p.numel()].view(p.shape) for p, o in zip(fg.params, fg.offsets)]
        torch._foreach_copy_(views, grads)
        return fg.scratch

    @staticmethod
    def _check_views(fg: _Group, every: bool) -> None:
        """The update writes ``fg.flat``; it reaches the module only while every parameter still is a view of it.
        ``module.to(...)``, ``.half()`` or assigning ``p.data`` after the optimizer was built break that silently."""
        base = fg.flat.data_ptr()
        idx = range(len(fg.params)) if every else (0, len(fg.params) - 1)
        for i in idx:
            if fg.params[i].data_ptr() != base + 4 * fg.offsets[i]:
                raise RuntimeError("FlatAdam: a parameter no longer aliases the optimizer's flat buffer (the module was "
                                   "moved, cast or had .data reassigned after the optimizer was built); rebuild the "
                                   "optimizer after such a change")

    @torch.no_grad()
    def begin_step(self) -> None:
        """Optional first half of ``step()`` for an optimizer with a ``reduce_group``: starts the all-reduce of the
        gradients and returns without waiting, so that work the caller enqueues between ``begin_step()`` and
        ``step()`` (anything that needs neither these gradients nor the updated parameters) runs while the
        collective is in flight.  ``step()`` then waits for it and applies the update.  A no-op without a group."""
        if self._reduce_group is None or self._pending is not None:
            return
        from .ddp import start_all_reduce_mean
        self._pending = []
        for fg in self._groups:
            grad = self._flat_grad(fg) if fg.flat is not None else None
            self._pending.append((grad, start_all_reduce_mean(grad, self._reduce_group, self._reduce_algorithm)
                                         if grad is not None else None))

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        pending, self._pending = self._pending, None
        for i, (group, fg) in enumerate(zip(self.param_groups, self._groups)):
            if fg.flat is None:
                continue
            if pending is not None and i < len(pending):
                grad, handle = pending[i]
                if handle is not None:
                    handle.wait()                                            # in place: .grad holds the mean after it
            else:
                grad = self._flat_grad(fg)
                if grad is not None and self._reduce_group is not None:
                    from .ddp import all_reduce_mean
                    all_reduce_mean(grad, self._reduce_group, self._reduce_algorithm)
            if grad is None:
                continue
            self._check_views(fg, every=fg.step % 128 == 0)          # all of them now and then, the ends every step
            fg.step += 1
            fg.step_t.fill_(float(fg.step))
            b1, b2 = group["betas"]
            ops.adam_step(fg.flat, grad, fg.exp_avg, fg.exp_avg_sq, float(group["lr"]), float(b1), float(b2),
                          float(group["eps"]), float(group["weight_decay"]), fg.step)
            torch.autograd.graph.increment_version(fg.params)       # the parameters changed under their views
        return loss
