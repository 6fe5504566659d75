"""``dpc_amd.optim.Adam``: the reference's ``optim.Adam(params, lr=args.lr, weight_decay=args.wd)`` (dpc/main.py:80-81) as ONE fused
kernel over the engine's flat arenas instead of torch's per-tensor update over 76 parameters.

    model = dpc_amd.model.DPC_RNN(...).to('cuda')
    optimizer = dpc_amd.optim.Adam(model.parameters(), lr=1e-3, weight_decay=1e-5)
    ...
    optimizer.zero_grad(); loss.backward(); optimizer.step()          # dpc/main.py:229-231, unchanged

The parameters of ``DPC_RNN`` are views of the engine's parameter arena and, after ``loss.backward()``, their ``.grad`` are views of
its gradient arena (dpc_amd/model.py), so ``step()`` is ``dpc_adam_dev`` over the arenas with torch.optim.Adam's arithmetic (L2
weight decay added to the gradient, bias-corrected moments; tests/test_engine_gpu.py holds it to the reference's own Adam step).
``state_dict()`` / ``load_state_dict()`` speak torch.optim.Adam's layout (dpc_amd/checkpoint.py), so checkpoints move both ways.
The engine is created by the module's first forward; a step before that has nothing to update and raises.
"""
from __future__ import annotations

import torch

from . import checkpoint as ckpt
from .engine import engine_of


class Adam(torch.optim.Optimizer):
    def __init__(self, params, lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8, weight_decay: float = 0.0, amsgrad: bool = False):
        if amsgrad:
            raise ValueError("amsgrad is not supported (the reference does not use it, dpc/main.py:80-81)")
        if tuple(betas) != (0.9, 0.999) or eps != 1e-8:
            raise ValueError("dpc_amd.optim.Adam runs the reference's configuration: betas (0.9, 0.999), eps 1e-8")
        super().__init__(params, dict(lr=lr, betas=tuple(betas), eps=eps, weight_decay=weight_decay))
        if len(self.param_groups) != 1:
            raise ValueError("one parameter group (the reference passes model.parameters(), dpc/main.py:74-78)")
        self._pending_state = None

    def _engine(self):
        ps = self.param_groups[0]["params"]
        eng = engine_of(ps[0]) if ps else None
        if eng is None:
            raise RuntimeError("dpc_amd.optim.Adam: the parameters are not backed by an engine yet -- run a forward of the "
                               "dpc_amd.model.DPC_RNN they belong to first (it moves them into the engine's arena)")
        names = list(eng.PRM.keys())
        if len(ps) != len(names) or any(p.data_ptr() != eng.PRM[k].data_ptr() for p, k in zip(ps, names)):
            raise RuntimeError("dpc_amd.optim.Adam updates ALL parameters of one DPC_RNN in registration order (model.parameters())")
        return eng, names

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        eng, names = self._engine()
        if self._pending_state is not None:
            ckpt.load_optimizer_state(eng, self._pending_state)
            self._pending_state = None
        g = self.param_groups[0]
        for p, k in zip(g["params"], names):
            if p.grad is None:
                raise RuntimeError(f"dpc_amd.optim.Adam: parameter {k} has no gradient (torch would skip it; the fused update covers "
                                   "the whole arena) -- call loss.backward() first")
            if p.grad.data_ptr() != eng.G[k].data_ptr():   # gradients that did not come from the engine's backward: bring them in
                eng.G[k].copy_(p.grad)
        eng.lr, eng.wd = float(g["lr"]), float(g["weight_decay"])
        eng.adam_step()
        eng.packed_for_step = -1
        return loss

    def state_dict(self):
        try:
            eng, _ = self._engine()
        except RuntimeError:
            return super().state_dict()   # nothing has run yet: torch's own (empty) state
        return ckpt.optimizer_state_dict(eng)

    def load_state_dict(self, state_dict):
        g0 = state_dict["param_groups"][0]
        self.param_groups[0]["lr"] = float(g0["lr"])
        self.param_groups[0]["weight_decay"] = float(g0.get("weight_decay", 0.0))
        try:
            eng, _ = self._engine()
        except RuntimeError:
            self._pending_state = state_dict   # the engine does not exist yet: applied at the first step
            return
        ckpt.load_optimizer_state(eng, state_dict)
