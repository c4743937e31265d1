"""hipGraph capture of a whole training step.

The reference's real shapes (stacks of 2-3 layers over 12/17/49 tokens) are launch- and host-bound: ~210 launches
for ~1.3 ms of GPU work per step.  Everything the library enqueues is capture-safe (no allocation, no
synchronisation, dropout seeds read from device memory at run time), so the whole step - zero_grad, forward, loss,
backward, optimizer - can be recorded once and replayed with a single launch.  PyTorch's ``torch.cuda.CUDAGraph``
is the capture vehicle (it owns the private memory pool); on ROCm it is a hipGraph.
"""
from __future__ import annotations

from typing import Callable, Dict

import torch


class GraphedTrainStep:
    """``step = GraphedTrainStep(model, optimizer, loss_fn, example_batch)``; ``loss = step(batch)``.

    ``loss_fn(model, batch) -> scalar loss`` must only use the tensors in ``batch`` (a dict); they are copied into
    static buffers before each replay.  The optimizer must be capturable (e.g. ``torch.optim.Adam(..., fused=True,
    capturable=True)``).

    Data parallelism: pass the wrapper as ``dp=``.  The step then is zero_grad, forward, loss (its (sum, count) all-reduce),
    backward (the bucketed gradient all-reduces issued from the layer hooks), ``dp.finish()``, optimizer - and ALL of it is
    recorded: RCCL collectives enqueued through torch's process group are capturable (they are stream operations on the
    group's communication stream, joined to the capture by events), so a replay re-runs the reductions with no Python and
    no host issue time on the critical path.  Every rank must construct (and later call) the step the same number of
    times.  A wrapped model WITHOUT ``dp=`` is refused: its collectives would be recorded but ``finish()`` would not run.
    """

    def __init__(self, model: torch.nn.Module, optimizer: torch.optim.Optimizer,
                 loss_fn: Callable[[torch.nn.Module, Dict[str, torch.Tensor]], torch.Tensor],
                 example_batch: Dict[str, torch.Tensor], warmup: int = 3, dp=None):
        self.model, self.optimizer, self.loss_fn, self.dp = model, optimizer, loss_fn, dp
        wrapped = any(getattr(m, "_grad_hook", None) is not None or callable(getattr(m, "global_mean", None))
                      for m in model.modules())
        if wrapped and dp is None:
            raise RuntimeError("GraphedTrainStep: the model is wrapped by dp.DataParallel (gradient hooks / loss reduction "
                               "are installed); pass the wrapper as dp= so that its finish() is part of the captured step")
        if dp is not None and dp.model is not model:
            raise ValueError("GraphedTrainStep: dp= wraps another model")
        self.static = {k: v.clone() for k, v in example_batch.items()}
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):  # warm-up off the capture: lazy initialisations, optimizer state, workspaces
            for _ in range(warmup):
                self._eager()
        torch.cuda.current_stream().wait_stream(side)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.static_loss = self._eager()

    def _eager(self) -> torch.Tensor:
        self.optimizer.zero_grad(set_to_none=True)
        loss = self.loss_fn(self.model, self.static)
        loss.backward()
        if self.dp is not None:
            self.dp.finish()
        self.optimizer.step()
        return loss

    def __call__(self, batch: Dict[str, torch.Tensor]) -> torch.Tensor:
        for k, v in batch.items():
            self.static[k].copy_(v, non_blocking=True)
        self.graph.replay()
        return self.static_loss
