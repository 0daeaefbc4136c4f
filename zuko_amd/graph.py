r"""HIP-graph replay of a flow call (small batches: the launches, not the arithmetic, are the time).

`LazyComposedTransform.forward` (zuko/lazy.py:119-128) chains T transforms; at BASELINE.json configs[0]'s batch of 4 096 rows a `log_prob` of the
conditional NSF(3, 5, transforms=3) is 3 + 1 launches of ~10 us of GPU work each behind ~0.3 ms of Python / ctypes / launch overhead.  The library
launches on torch's current stream, so `torch.cuda.graph` captures it: `capture` records ONE call on static input buffers and returns a callable
that copies new inputs in and replays the graph (measured: 0.36 ms eager -> 0.076 ms, bit-identical; bench.py: side_paths.nsf_cfg1_conditional).

`capture_step` does the same for a whole OPTIMISATION step — forward, backward through the one-node training paths (zuko_amd/train.py,
coupling_train.py), optimizer update — which the reference runs as the README loop (tests/test_flows.py:22-29): every weight gather, amax and split of a
training step is a launch on the current stream, so the step replays with the parameters it updates (the conditional NSF(3, 5) at 4 096 rows: 2.4 ms
eager -> 0.6 ms; MAF cfg3 at 2^14 rows: 4.6 -> 2.3 ms; RealNVP cfg4 is GPU-bound either way).
"""

from __future__ import annotations

import torch
from torch import Tensor

__all__ = ["capture", "capture_step"]


class CapturedCall:
    """`captured(x[, c])` -> the output of the captured call for these inputs (a view of the graph's static output buffer: valid until the next
    call; `.clone()` it to keep it).  Shapes, dtypes and devices are fixed at capture time, and so are the WEIGHTS: the fused kernels read weight
    streams gathered from the parameters when they last changed, and that gather happened before the capture (it is not part of the graph) — after
    an optimizer step, `load_state_dict` or any other parameter update, capture again."""

    def __init__(self, graph, x: Tensor, c: Tensor | None, out) -> None:
        self.graph, self.x, self.c, self.out = graph, x, c, out

    def __call__(self, x: Tensor, c: Tensor | None = None):
        if tuple(x.shape) != tuple(self.x.shape) or x.dtype != self.x.dtype or (c is None) != (self.c is None) or (c is not None and tuple(c.shape) != tuple(self.c.shape)):
            raise ValueError(f"zuko_amd.capture: the graph was recorded for x {tuple(self.x.shape)} {self.x.dtype}" + ("" if self.c is None else f", c {tuple(self.c.shape)}") +
                             f"; got x {tuple(x.shape)} {x.dtype}" + ("" if c is None else f", c {tuple(c.shape)}"))
        self.x.copy_(x)
        if c is not None:
            self.c.copy_(c)
        self.graph.replay()
        return self.out


def capture(flow, x: Tensor, c: Tensor | None = None, call: str = "log_prob", warmup: int = 3) -> CapturedCall:
    """Record `getattr(flow(c), call)(x)` — "log_prob" (default), or "transform" for z = flow(c).transform(x) — as a HIP graph on static copies of `x` / `c`
    (example inputs: only their shape / dtype / device matter) and return the replaying callable.  Inference only (runs under no_grad); the warm-up calls
    build plans, weight streams and any static-shape kernel before the capture, on a side stream as torch.cuda.graph requires."""
    if not x.is_cuda:
        raise ValueError("zuko_amd.capture: inputs must live on the GPU")
    xs = x.detach().clone()
    cs = None if c is None else c.detach().clone()

    def once():
        d = flow(cs) if cs is not None else flow()
        return d.log_prob(xs) if call == "log_prob" else getattr(d, call)(xs)

    with torch.no_grad():
        side = torch.cuda.Stream(device=x.device)
        side.wait_stream(torch.cuda.current_stream(x.device))
        with torch.cuda.stream(side):
            for _ in range(max(1, warmup)):
                once()
        torch.cuda.current_stream(x.device).wait_stream(side)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            out = once()
    return CapturedCall(graph, xs, cs, out)


class CapturedStep:
    """`captured(x[, c])` -> the loss of ONE optimisation step on these inputs (a 0-dim tensor in the graph's static memory: read it before the next
    call).  The step updates the flow's parameters and the optimizer's state in place, exactly as the eager loop would; shapes are fixed."""

    def __init__(self, graph, x: Tensor, c: Tensor | None, loss: Tensor) -> None:
        self.graph, self.x, self.c, self.loss = graph, x, c, loss

    def __call__(self, x: Tensor | None = None, c: Tensor | None = None) -> Tensor:
        if x is not None:
            if tuple(x.shape) != tuple(self.x.shape) or x.dtype != self.x.dtype:
                raise ValueError(f"zuko_amd.capture_step: the graph was recorded for x {tuple(self.x.shape)} {self.x.dtype}; got {tuple(x.shape)} {x.dtype}")
            self.x.copy_(x)
        if c is not None:
            if self.c is None or tuple(c.shape) != tuple(self.c.shape):
                raise ValueError("zuko_amd.capture_step: context does not match the recorded one")
            self.c.copy_(c)
        self.graph.replay()
        return self.loss


def capture_step(flow, optimizer, x: Tensor, c: Tensor | None = None, loss_fn=None, warmup: int = 3) -> CapturedStep:
    """Record one optimisation step — `loss = loss_fn(flow, x, c)` (default `-flow(c).log_prob(x).mean()`), `zero_grad`, `backward`, `optimizer.step()` —
    as a HIP graph on static copies of `x` / `c` and return the replaying callable.  The optimizer must be capturable (`torch.optim.Adam(..., capturable=True)`:
    its step counter lives on the device).  The `warmup` eager steps that precede the capture (torch.cuda.graph needs the kernels loaded, the plans built and
    the optimizer state allocated) ARE optimisation steps: the parameters move during them."""
    if not x.is_cuda:
        raise ValueError("zuko_amd.capture_step: inputs must live on the GPU")
    for group in optimizer.param_groups:
        if group.get("capturable") is False:
            raise ValueError("zuko_amd.capture_step: the optimizer must be built with capturable=True (its step counter has to live on the device)")
    xs = x.detach().clone()
    cs = None if c is None else c.detach().clone()
    loss_out = torch.zeros((), dtype=torch.float32, device=x.device)

    def once():
        loss = loss_fn(flow, xs, cs) if loss_fn is not None else -(flow(cs) if cs is not None else flow()).log_prob(xs).mean()
        optimizer.zero_grad(set_to_none=False)  # (the .grad tensors are part of the graph's memory: they are zeroed and accumulated in place)
        loss.backward()
        optimizer.step()
        loss_out.copy_(loss.detach())

    side = torch.cuda.Stream(device=x.device)
    side.wait_stream(torch.cuda.current_stream(x.device))
    with torch.cuda.stream(side):
        for _ in range(max(2, warmup)):
            once()
    torch.cuda.current_stream(x.device).wait_stream(side)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        once()
    return CapturedStep(graph, xs, cs, loss_out)
