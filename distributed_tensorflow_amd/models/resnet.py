"""ResNet-50 on the sharded parameter-server state (benchmark configuration "resnet50_sync": synthetic
ImageNet-shape data, sync mode, optimizer state sharded over the ranks).

The convolutions are library calls (MIOpen through PyTorch): the reference has no convolutional
model, and the hand-written hot path of this framework is the dense path; batch normalisation, the
residual additions and the ReLUs run in the fused kernels of ``ops/batchnorm.py``.  What this model
exercises is the parameter-server side at a realistic size (25.6 M parameters, 161 tensors):
all trainable tensors are views into ONE flat fp32 buffer, autograd writes gradients into the
matching flat gradient buffer, and a step is

    forward/backward -> reduce-scatter(flat grads) -> fused Adam/SGD on the owned slice -> all-gather

with optimizer state (Adam m, v) existing only for the slice a rank owns.  With several ranks the
flat gradient is aggregated in ~32 MB buckets while autograd is still running: a hook on every
parameter counts completed gradients per bucket, and complete buckets are handed to the engine
(``parallel/bucketed.py``) in descending bucket order, the same order on every rank.
"""
from __future__ import annotations

import os
import struct
from typing import List, Optional

import torch
import torch.nn as nn
import torch.nn.functional as F

from ..ops.batchnorm import FusedBatchNormAct
from ..ops.classifier import linear_xent
from ..ops.maxpool import max_pool2d
from ..parallel.sharded_state import ShardedState, VariableSpec
from .mlp import StepRecord


class Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 1, bias=False)
        self.bn1 = FusedBatchNormAct(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, stride=stride, padding=1, bias=False)
        self.bn2 = FusedBatchNormAct(planes)
        self.conv3 = nn.Conv2d(planes, planes * 4, 1, bias=False)
        self.bn3 = FusedBatchNormAct(planes * 4)
        self.downsample = downsample
        self.fork = False            # True: the output goes to a next block, as a (main branch, skip connection) pair

    def forward(self, x):
        # a preceding block hands its output over as two handles of one tensor, so that the gradients of the two uses
        # reach its last batch norm separately and are added inside that layer's backward kernel
        main, skip = x if isinstance(x, tuple) else (x, x)
        idt = skip if self.downsample is None else self.downsample(skip)
        out = self.bn1(self.conv1(main))                 # batch norm + ReLU: one operator (ops/batchnorm.py)
        out = self.bn2(self.conv2(out))
        return self.bn3(self.conv3(out), residual=idt, fork=self.fork)   # batch norm + residual + ReLU


class ResNet(nn.Module):
    def __init__(self, layers=(3, 4, 6, 3), num_classes=1000, width=64):
        super().__init__()
        self.inplanes = width
        self.conv1 = nn.Conv2d(3, width, 7, stride=2, padding=3, bias=False)
        self.bn1 = FusedBatchNormAct(width)
        self.layer1 = self._make(width, layers[0], 1)
        self.layer2 = self._make(width * 2, layers[1], 2)
        self.layer3 = self._make(width * 4, layers[2], 2)
        self.layer4 = self._make(width * 8, layers[3], 2)
        self.fc = nn.Linear(width * 8 * 4, num_classes)
        blocks = [b for layer in (self.layer1, self.layer2, self.layer3, self.layer4) for b in layer]
        for b in blocks[:-1]:
            b.fork = True
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")

    def _make(self, planes, blocks, stride):
        down = None
        if stride != 1 or self.inplanes != planes * 4:
            down = nn.Sequential(nn.Conv2d(self.inplanes, planes * 4, 1, stride=stride, bias=False),
                                 FusedBatchNormAct(planes * 4, relu=False))
        layers = [Bottleneck(self.inplanes, planes, stride, down)]
        self.inplanes = planes * 4
        layers += [Bottleneck(self.inplanes, planes) for _ in range(1, blocks)]
        return nn.Sequential(*layers)

    def forward(self, x, features_only: bool = False):
        """Logits, or with ``features_only`` the pooled features in front of ``fc``: the training step then runs
        classifier and loss as one node on the framework's GEMM and softmax cross-entropy kernels
        (``ops/classifier.linear_xent`` with ``self.fc``'s parameters)."""
        x = self.bn1(self.conv1(x))
        x = max_pool2d(x, 3, 2, 1)                       # own kernels on the GPU (ops/maxpool.py)
        x = self.layer4(self.layer3(self.layer2(self.layer1(x))))
        x = torch.flatten(F.adaptive_avg_pool2d(x, 1), 1)
        return x if features_only else self.fc(x)


def resnet50(num_classes=1000, width=64, layers=(3, 4, 6, 3)) -> ResNet:
    return ResNet(layers, num_classes, width)


conv1x1_calls = {"own_wgrad": 0}


def _own_conv1x1_wgrad(m: nn.Conv2d, x: torch.Tensor, gy: torch.Tensor, w: torch.Tensor):
    """DTF_OWN_CONV1X1_WGRAD=1: the weight gradient of a stride-1 1x1 convolution on channels-last bf16 tensors through
    the framework's own kernel (``ops.conv1x1_wgrad``, csrc/kernels/conv1x1.hip) instead of the library's.  Off by
    default: measured per network at batch 256 it needs 3.55 ms against the library's 2.62 ms (3.1-3.3 TB/s on the
    56 x 56 layers, where the library reaches 4.6; the 7 x 7 layers pay for their partial sums).  Returns None where it
    does not apply."""
    if os.environ.get("DTF_OWN_CONV1X1_WGRAD", "0") != "1":
        return None
    if (m.kernel_size != (1, 1) or m.stride != (1, 1) or m.padding != (0, 0) or m.groups != 1 or x.dtype != torch.bfloat16
            or gy.dtype != torch.bfloat16 or x.dim() != 4 or m.in_channels % 8 or m.out_channels % 8):
        return None
    if not (x.is_contiguous(memory_format=torch.channels_last) and gy.is_contiguous(memory_format=torch.channels_last)):
        return None
    from .. import ops
    n, cin, h, wd = x.shape
    x2 = x.permute(0, 2, 3, 1).reshape(n * h * wd, cin)              # views of the channels-last storage
    g2 = gy.permute(0, 2, 3, 1).reshape(n * h * wd, m.out_channels)
    gw = ops.conv1x1_wgrad(x2, g2).view(m.out_channels, cin, 1, 1)
    conv1x1_calls["own_wgrad"] += 1
    return gw.contiguous(memory_format=torch.channels_last) if w.is_contiguous(memory_format=torch.channels_last) else gw


class _ConvWgradAside(torch.autograd.Function):
    """A convolution whose backward pass computes the weight gradient on a second stream.

    In the backward pass of a ResNet the chain that the next layer waits for is (batch-norm backward -> data
    gradient of the convolution); the weight gradient is needed only when the optimizer runs.  Its kernels (and the
    library's zero-fill / cast helpers around them) go to a side stream, where they run beside the bandwidth-bound
    batch-norm kernels of the main chain and fill the gaps between its dependent launches.  The owner of the side
    stream joins it before it reads the gradients (``TorchModuleProgram._gradients_to_flat``).

    The weight gradient is NOT returned to autograd: the engine believes a node's outputs were produced on the stream
    the node ran on, and where it copies a returned gradient (strides that differ from the leaf's, a hook, a second
    reference) that copy would be a main-stream kernel racing the side-stream producer.  The gradient is stored into
    ``w.grad`` here, on the side stream, and autograd gets None for ``w``."""

    @staticmethod
    def forward(ctx, x, w, conv, side, stored=None):
        if x.dtype != w.dtype:
            x = x.to(w.dtype)                               # (what autocast does with the network's fp32 input)
        ctx.save_for_backward(x, w)
        ctx.conv, ctx.side, ctx.stored = conv, side, stored
        return conv._conv_forward(x, w, None)

    @staticmethod
    def backward(ctx, gy):
        x, w = ctx.saved_tensors
        m, side = ctx.conv, ctx.side
        args = (None, m.stride, m.padding, m.dilation, False, [0, 0], m.groups)
        gx = gw = None
        if ctx.needs_input_grad[1]:
            main = torch.cuda.current_stream(gy.device)
            side.wait_stream(main)                          # gy exists from here on
            with torch.cuda.stream(side):
                gw = _own_conv1x1_wgrad(m, x, gy, w)
                if gw is None:
                    gw = torch.ops.aten.convolution_backward(gy, x, w, *args, [False, True, False])[1]
                if w.grad is None:
                    w.grad = gw
                else:
                    w.grad.add_(gw)                         # a weight used by several convolutions of one step
            # their memory must not be handed out again on the main stream while the side stream still reads it
            gy.record_stream(side)
            x.record_stream(side)
        if ctx.needs_input_grad[0]:
            gx = torch.ops.aten.convolution_backward(gy, x, w, *args, [True, False, False])[0]
        if gw is not None and ctx.stored is not None:
            ctx.stored()                                    # the owner counts this gradient towards its bucket
        return gx, None, None, None, None


class TorchModuleProgram:
    """Step program around a PyTorch module whose parameters live in a ``ShardedState``.

    Has the interface the engines expect from ``MLPProgram`` (``run``, ``after_compute``,
    ``gs_inc``, ``fused``, ``read_log``) so that ``SyncReplicasEngine`` drives it unchanged.
    """

    def __init__(self, module: nn.Module, device, images: torch.Tensor, labels: torch.Tensor, batch: int,
                 world_size: int = 1, rank: int = 0, optimizer: str = "adam", lr: float = 1e-3, gs_inc: int = 1,
                 autocast: Optional[torch.dtype] = None, channels_last: bool = True, log_capacity: int = 1024,
                 bucket_bytes: Optional[int] = None):
        self.device = torch.device(device)
        self.module = module.to(self.device)
        if channels_last and images.dim() == 4:
            self.module = self.module.to(memory_format=torch.channels_last)
        named = [(n.replace(".", "/"), p) for n, p in self.module.named_parameters()]
        specs = [VariableSpec(n, tuple(p.shape)) for n, p in named]
        self.state = ShardedState(specs, self.device, world_size, rank, optimizer, lr)
        # move every parameter (and its gradient) into the flat buffers
        for n, p in named:
            flat = self.state.var(n)
            flat.copy_(p.detach().reshape(flat.shape))
            p.data = flat
            p.grad = self.state.grad(n)
        self.state.initialized = True
        # One worker on a GPU: autograd hands the gradient tensors over (``p.grad`` is None when the backward pass
        # starts, nothing is accumulated) and a few multi-tensor copies move them into the flat gradient buffer, where
        # the preassigned gradient views cost one add kernel per parameter and step.  Convolutions then read
        # persistent compute-precision copies of their weights, refreshed by one multi-tensor copy per step, instead of
        # casting every weight in the forward pass and its gradient back in the backward pass (autocast).  With the
        # gradient hooks of the bucketed aggregation (several workers) the gradients must be in the flat buffer when a
        # hook fires: that path keeps the preassigned views.  DTF_DIRECT_GRADS=0 selects it everywhere.
        self.direct_grads = self.device.type == "cuda" and os.environ.get("DTF_DIRECT_GRADS", "1") != "0"
        self._direct_now = False
        self._in_step = False                   # True while _step_body runs its forward and backward pass
        self._params = [p for _, p in named]
        self._grad_views = [self.state.grad(n) for n, _ in named]
        self._copies = {}                       # id(master parameter) -> compute-precision leaf the convolution reads
        # weight gradients of the convolutions on a second stream (DTF_WGRAD_STREAM=0: on the main stream)
        self._side = (torch.cuda.Stream(self.device)
                      if self.direct_grads and os.environ.get("DTF_WGRAD_STREAM", "1") != "0" else None)
        if self.direct_grads and autocast is not None:
            fmt = torch.channels_last if (channels_last and images.dim() == 4) else torch.contiguous_format
            for m in self.module.modules():
                if isinstance(m, nn.Conv2d) and m.bias is None and m.padding_mode == "zeros":
                    w = torch.empty_like(m.weight, dtype=autocast, memory_format=fmt).requires_grad_(True)
                    self._copies[id(m.weight)] = w
                    m.forward = self._conv_forward(m, w, m.weight)
        # the batch counters of all batch-norm layers advance by one multi-tensor add per step, not one launch per layer
        self._bn_counters = []
        for m in self.module.modules():
            if isinstance(m, FusedBatchNormAct) and m.num_batches_tracked is not None:
                m.count_batches = False
                self._bn_counters.append(m.num_batches_tracked)
        self.images, self.labels = images.to(self.device), labels.to(self.device)
        if channels_last and self.images.dim() == 4:
            self.images = self.images.contiguous(memory_format=torch.channels_last)
        self.B = batch
        self.n = self.images.shape[0]
        self.autocast = autocast
        self.gs_inc = gs_inc
        self.fused = False
        self.post = False
        # classifier and loss on the framework's kernels where the module exposes its features (DTF_FUSED_HEAD=0: operators)
        self.fused_head = isinstance(self.module, ResNet) and os.environ.get("DTF_FUSED_HEAD", "1") != "0"
        # DTF_MODULE_GRAPH=1, one rank on a GPU: the whole step (forward, backward, optimizer) is captured into a
        # hipGraph after three eager steps and replayed; the host then only copies the next batch into the graph's
        # input buffers.  Opt-in: the ResNet-50 step is bound by the GPU (26.81 ms replayed, 26.97 ms eager on one stream; with the weight
        # gradients on a second stream the eager step is the faster one, 25.9 against 26.7 ms).
        self.use_graphs = (self.device.type == "cuda" and world_size == 1 and os.environ.get("DTF_MODULE_GRAPH", "0") == "1")
        self._graphs = {}
        self._graph = None
        self._gx = self._gy = None
        self._graph_stream = None
        self.after_compute = None
        self.grad_ready = None                  # engine hook: called with a bucket index of grad_groups()
        self._groups = None
        self.bucket_bytes = bucket_bytes
        self.steps_enqueued = 0
        self.log_cap = log_capacity
        self.log = torch.zeros((log_capacity, 2), dtype=torch.float32, device=self.device)   # loss, correct
        self._last = torch.zeros(2, dtype=torch.float32, device=self.device)                  # ... of the step just run
        self._last_gs = torch.zeros(1, dtype=torch.int32, device=self.device)
        self.gs_log = torch.zeros(log_capacity, dtype=torch.int32, device=self.device)

    def _conv_forward(self, m: nn.Conv2d, w: torch.Tensor, master: torch.Tensor):
        stored = lambda: self._count_gradient(master)

        def forward(x):
            # the compute copies are refreshed at the start of a training step only: any other forward pass
            # (evaluation, a pass after load_values or a restore) reads the master weights
            if self._direct_now and self.module.training and torch.is_grad_enabled() and self._in_step:
                if self._side is not None:
                    return _ConvWgradAside.apply(x, w, m, self._side, stored)
                return m._conv_forward(x, w, None)
            return nn.Conv2d.forward(m, x)
        return forward

    def _gradients_to_flat(self) -> None:
        """After a backward pass in direct mode: every parameter's gradient (of its compute copy, where it has one)
        into its view of the flat gradient buffer, grouped by source dtype (multi-tensor copies)."""
        cur = torch.cuda.current_stream(self.device)
        if self._side is not None:
            cur.wait_stream(self._side)                     # the weight gradients are complete
        by_dtype = {}
        for p, view in zip(self._params, self._grad_views):
            leaf = self._copies.get(id(p), p)
            g = leaf.grad
            if g is None:
                view.zero_()                    # took no part in this step
                continue
            if self._side is not None and id(p) in self._copies:
                g.record_stream(cur)                        # allocated on the side stream, read here
            dst, src = by_dtype.setdefault(g.dtype, ([], []))
            dst.append(view)
            src.append(g)
        for dst, src in by_dtype.values():
            torch._foreach_copy_(dst, src)

    def grad_groups(self):
        """Gradient buckets (consecutive parameters, layout order) and the hooks that report them complete."""
        if self._groups is None:
            from ..parallel.bucketed import group_by_bytes
            if os.environ.get("DTF_WGRAD_STREAM_BUCKETED", "0") != "1":
                # With buckets the weight gradients stay on the main stream.  Measured on the ResNet-50 step at batch
                # 256 with a one-rank communicator (library algorithm search off): 30.6 ms, against 32.6 ms with the
                # second stream (a bucket's copy then queues behind every weight gradient that stream still has to
                # run, and the bucket's collective behind the copy) and 31.0 ms with gradients accumulated into
                # preassigned views; one worker without buckets: 29.6 ms with the second stream, 30.6 without.
                self._side = None
            self._groups = group_by_bytes(self.state, self.bucket_bytes) if self.bucket_bytes else group_by_bytes(self.state)
            bucket_of = {n: i for i, g in enumerate(self._groups) for n in g}
            self._pending = [0] * len(self._groups)
            self._complete = [False] * len(self._groups)
            self._next = len(self._groups) - 1
            self._bucket_of_param = {}
            self._bucket_members = [[] for _ in self._groups]       # (parameter, its view of the flat gradient buffer)
            views = {id(p): v for p, v in zip(self._params, self._grad_views)}
            for n, p in self.module.named_parameters():
                b = bucket_of[n.replace(".", "/")]
                self._bucket_of_param[id(p)] = b
                self._bucket_members[b].append((p, views[id(p)]))
                if self.direct_grads and self._side is not None and id(p) in self._copies:
                    # its gradient never passes through autograd's accumulation: the weight-gradient node of its
                    # convolution stores it on the side stream and reports it (_count_gradient)
                    continue
                if self.direct_grads and id(p) in self._copies:
                    self._copies[id(p)].register_post_accumulate_grad_hook(lambda _leaf, p=p: self._count_gradient(p))
                else:
                    p.register_post_accumulate_grad_hook(self._count_gradient)
        return self._groups

    def _count_gradient(self, param) -> None:
        """One more gradient of the step exists (enqueued on the stream that computes it).  A bucket whose gradients
        are all there is moved into the flat buffer (direct mode) and handed to the engine; buckets are released in
        descending order only: the collectives are issued in one order on every rank."""
        if self.grad_ready is None or self._groups is None:
            return
        b = self._bucket_of_param[id(param)]
        self._pending[b] -= 1
        if self._pending[b] == 0:
            self._complete[b] = True
            while self._next >= 0 and self._complete[self._next]:
                self._release_bucket(self._next)
                self._next -= 1

    def _release_bucket(self, b: int) -> None:
        if not self._direct_now:
            self.grad_ready(b)                              # the gradients were accumulated into the flat buffer's views
            return
        # Direct mode: the bucket's gradients are separate tensors - the convolutions' weight gradients computed on the
        # side stream, everything else on the main stream.  One multi-tensor copy per source dtype moves them into the
        # flat buffer ON THE SIDE STREAM (which first waits for the main stream's work enqueued so far), and the engine
        # is told there: its collective stream then waits for that copy, while the main stream - batch-norm backward and
        # data gradients of the layers below - never waits for a weight gradient.
        main = torch.cuda.current_stream(self.device)
        side = self._side if self._side is not None else main
        if side is not main:
            side.wait_stream(main)
        with torch.cuda.stream(side):
            by_dtype = {}
            for p, view in self._bucket_members[b]:
                g = self._copies.get(id(p), p).grad
                if g is None:
                    view.zero_()
                    continue
                if side is not main:
                    g.record_stream(side)
                dst, src = by_dtype.setdefault(g.dtype, ([], []))
                dst.append(view)
                src.append(g)
            for dst, src in by_dtype.values():
                torch._foreach_copy_(dst, src)
            self.grad_ready(b)

    def enqueue_step(self) -> None:
        s = self.steps_enqueued
        if self.grad_ready is not None:
            self._pending = [len(g) for g in self._groups]
            self._complete = [False] * len(self._groups)
            self._next = len(self._groups) - 1
        lo = (s * self.B) % max(1, self.n - self.B + 1)
        x, y = self.images[lo:lo + self.B], self.labels[lo:lo + self.B]
        if self._graph is not None:
            self._gx.copy_(x)
            self._gy.copy_(y)
            self._graph.replay()
        else:
            self._step_body(x, y)
        slot = s % self.log_cap
        self.log[slot].copy_(self._last)
        self.gs_log[slot:slot + 1].copy_(self._last_gs)
        self.steps_enqueued += 1

    def _step_body(self, x, y) -> None:
        """The device work of one step; touches no host state, so it can be captured and replayed."""
        # (gradient buckets registered but no engine listening: the hooks sit on the master parameters, keep their views)
        self._direct_now = direct = self.direct_grads and (self.grad_ready is not None or self._groups is None)
        if direct:
            if self._copies:
                leaves = list(self._copies.values())
                with torch.no_grad():
                    torch._foreach_copy_(leaves, [p for p in self._params if id(p) in self._copies])
                for w in leaves:
                    w.grad = None
            for p in self._params:
                p.grad = None
        else:
            for p, view in zip(self._params, self._grad_views):
                p.grad = view
            self.state.grads.zero_()
        if self._bn_counters:
            torch._foreach_add_(self._bn_counters, 1)
        self._in_step = True
        try:
            self._forward_backward(x, y)
        finally:
            self._in_step = False
        if direct and self.grad_ready is None:
            self._gradients_to_flat()
        elif direct and self._side is not None:
            torch.cuda.current_stream(self.device).wait_stream(self._side)     # the last buckets' copies
        self._last_gs.copy_(self.state.state[0:1])
        if self.after_compute is not None:
            self.after_compute()
        self.state.state[0:1].add_(self.gs_inc)
        self.state.state[1:2].add_(1)

    def _forward_backward(self, x, y) -> None:
        if self.fused_head:
            # classifier + loss + accuracy count: one autograd node on the framework's own kernels
            with torch.autocast(self.device.type, dtype=self.autocast, enabled=self.autocast is not None):
                feat = self.module(x, features_only=True)
            loss, hits, _ = linear_xent(feat, self.module.fc.weight, self.module.fc.bias, y)
            loss.backward()
            self._last[0:1].copy_(loss.detach().reshape(1))
            self._last[1:2].copy_(hits.reshape(1))
        else:
            with torch.autocast(self.device.type, dtype=self.autocast, enabled=self.autocast is not None):
                logits = self.module(x)
            loss = F.cross_entropy(logits.float(), y)
            loss.backward()
            self._last[0:1].copy_(loss.detach().reshape(1))
            self._last[1:2].copy_((logits.argmax(1) == y).sum().float().reshape(1))

    def _capture(self) -> None:
        """Capture ``_step_body`` on static input buffers.  Called after eager steps have run (MIOpen has chosen its
        algorithms, the allocator is warm).  Any failure leaves the program in eager mode."""
        import warnings
        self._gx = torch.empty_like(self.images[:self.B])
        self._gy = torch.empty_like(self.labels[:self.B])
        if self._gx.dim() == 4 and self.images.is_contiguous(memory_format=torch.channels_last):
            self._gx = self._gx.contiguous(memory_format=torch.channels_last)
        self._gx.copy_(self.images[:self.B])
        self._gy.copy_(self.labels[:self.B])
        saved = {k: v.clone() for k, v in self._snapshot().items()}
        stream = self._graph_stream = torch.cuda.Stream(self.device)
        try:
            stream.wait_stream(torch.cuda.current_stream(self.device))
            with torch.cuda.stream(stream):
                self._step_body(self._gx, self._gy)           # once on the capture stream (its library handles exist)
            torch.cuda.current_stream(self.device).wait_stream(stream)
            torch.cuda.synchronize(self.device)
            graph = torch.cuda.CUDAGraph()
            from .mlp import _no_gc
            with _no_gc():                                    # no destructor of an older graph inside the capture
                with torch.cuda.graph(graph, stream=stream):
                    self._step_body(self._gx, self._gy)
            self._graph = graph
        except Exception as e:                                # noqa: BLE001 - any capture problem means: stay eager
            warnings.warn(f"step capture failed ({type(e).__name__}: {e}); running the module eagerly")
            self.use_graphs = False
            self._graph = None
            torch.cuda.synchronize(self.device)
        # the rehearsal step on the capture stream trained on batch 0 a second time: undo it
        torch.cuda.synchronize(self.device)
        for k, v in self._snapshot().items():
            v.copy_(saved[k])

    def _snapshot(self):
        """Every tensor a step changes (parameters, optimizer state, counters, batch-norm buffers)."""
        st = self.state
        out = {"params": st.params, "state": st.state}
        for name in ("m", "v", "exp_avg", "exp_avg_sq"):
            t = getattr(st, name, None)
            if isinstance(t, torch.Tensor):
                out[name] = t
        for n, b in self.module.named_buffers():
            out["buf/" + n] = b
        return out

    def run(self, n_steps: int, graph_steps: int = 0) -> None:
        self.module.train()
        for _ in range(n_steps):
            if (self.use_graphs and self._graph is None and self.grad_ready is None and self.steps_enqueued >= 3):
                self._capture()
            self.enqueue_step()

    def read_log(self, first_local_step: int, count: int) -> List[StepRecord]:
        log, gs = self.log.cpu(), self.gs_log.cpu()
        out = []
        for s in range(first_local_step, first_local_step + count):
            i = s % self.log_cap
            out.append(StepRecord(local_step=s + 1, global_step=int(gs[i]), loss=float(log[i, 0]),
                                  train_correct=int(log[i, 1]), batch=self.B))
        return out
