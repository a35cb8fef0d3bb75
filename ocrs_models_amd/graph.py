"""A whole train step recorded into a hipGraph and replayed (the CDNA4 guide's "capture launch-bound inner loops in hipGraphs").

A detection step is ~190 kernel launches issued from Python through ctypes (~10-15 us of host time each).  At BASELINE configs[1]
(B = 32 x 1024^2, 13.7 ms of GPU work) the host stays ahead of the GPU; at small batches -- BASELINE configs[0], B = 2 x 512^2: ~0.5 ms
of GPU work -- the step is host-bound several times over.  ``GraphedTrainStep`` records forward + loss + backward + optimizer step once
(``torch.cuda.CUDAGraph`` = hipGraph on ROCm; the C ABI launches on the capturing stream, every workspace comes from the graph's private
pool, the loss's top-k selection and Adam's step count run on the device) and replays it with one host call per step.

    step = GraphedTrainStep(model, optimizer, loss_fn, example_input, example_target)   # optimizer: optim.Adam(..., capturable=True)
    loss = step(x, target)          # device scalar (static buffer: read it before the next call or clone it)

The constructor's `warmup` eager iterations are REAL optimizer steps on the example batch (they size the allocator pools and build the
pointer tables): pass a batch you want trained on, or snapshot / restore the model and optimizer state around the construction.

Restrictions of a capture: fixed input shapes; the optimizer must be capturable; the learning rate is baked at capture time (re-capture
after a scheduler step); no gradient bucketer (single GPU).
"""
from __future__ import annotations

import os

import torch


class GraphedTrainStep:
    def __init__(self, model, optimizer, loss_fn, example_input: torch.Tensor, example_target: torch.Tensor, warmup: int = 3):
        if not getattr(optimizer, "capturable", False):
            raise RuntimeError("GraphedTrainStep needs ocrs_models_amd.optim.Adam(..., capturable=True)")
        self.model, self.opt, self.loss_fn = model, optimizer, loss_fn
        self.x = example_input.clone()
        self.t = example_target.clone()
        prev = os.environ.get("OCRS_OVERLAP")
        os.environ["OCRS_OVERLAP"] = "0"  # (the side-stream overlap of the ConvTranspose weight gradients is a fork/join the capture does not need)
        try:
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(warmup):  # allocator / pointer-table / lazy-initialisation warm-up outside the capture
                    self._body()
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            self.graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph):
                self.loss = self._body()
            # the capture ran Adam.step()'s HOST bookkeeping (state["step"] += 1) but no kernel: take that count back so the host mirror
            # equals the device-side step count (checkpoints store the mirror; a resumed capturable run seeds the device count from it)
            for group in self.opt.param_groups:
                for p in group["params"]:
                    st = self.opt.state.get(p)
                    if st:
                        st["step"] -= 1
        finally:
            if prev is None:
                os.environ.pop("OCRS_OVERLAP", None)
            else:
                os.environ["OCRS_OVERLAP"] = prev

    def _body(self):
        self.opt.zero_grad(set_to_none=True)
        pred = self.model(self.x)
        loss = self.loss_fn(pred, self.t)
        loss.backward()
        self.opt.step()
        return loss.detach()

    def __call__(self, x: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
        self.x.copy_(x, non_blocking=True)
        self.t.copy_(target, non_blocking=True)
        self.graph.replay()
        for group in self.opt.param_groups:  # (host mirror of the device-side step count, for checkpoints)
            for p in group["params"]:
                st = self.opt.state.get(p)
                if st:
                    st["step"] += 1
        return self.loss
