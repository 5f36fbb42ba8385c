"""Wide MLP (4096-wide hidden layers, ~100 M parameters): the bandwidth/MFMA-bound configuration.

Same network family, loss and bookkeeping as the reference model (``mlp.py``), but the hidden
layers run on the tiled MFMA GEMM (``csrc/kernels/gemm_tiled.hip``) in bf16 with fp32
accumulation and fp32 master weights, or in exact fp32.  Every product is of the form
``A[M,K] . B[N,K]^T``; the transposed operand copies that the backward products need
(``H^T``, ``dZ^T``, ``W^T``) are written by the epilogue of the kernel that produces the tensor.

Per step:  gather+cast batch -> L x forward GEMM (+bias, ReLU) -> head (loss, accuracy, dz)
-> per layer, top down: weight-gradient GEMM (+bias gradient), backward GEMM (ReLU mask) -> small
kernel for the 10-class layer -> optimizer / collectives.  The compute copies of the weights (W and W^T in the
compute dtype) are rebuilt from the fp32 masters at the start of a step, except for the layers whose weight-gradient
epilogue applies the optimizer: that epilogue writes both copies itself.

A layer's gradients are final as soon as its weight-gradient product has been enqueued; ``grad_ready(l)``
then lets the engine aggregate that layer's bucket on a second stream while the layers below are still
being differentiated (``parallel/bucketed.py``).
"""
from __future__ import annotations

import os
from typing import Optional, Sequence

import torch

from .. import ops
from ..data.device_dataset import DeviceDataset
from ..parallel.sharded_state import ShardedState
from .mlp import MLPProgram


class WideMLPProgram(MLPProgram):
    _supports_mlp2 = False

    def __init__(self, state: ShardedState, dataset: DeviceDataset, batch: int, hidden: Sequence[int], classes: int = 10,
                 loss_mode: int = ops.LOSS_REFERENCE, dtype: torch.dtype = torch.bfloat16, gs_inc: int = 1,
                 log_capacity: int = 2048, use_graphs: Optional[bool] = None):
        if batch % 8 or any(h % 8 for h in hidden):
            raise ValueError("wide MLP: batch and hidden sizes must be multiples of 8 (16-byte GEMM staging)")
        self.cdtype = dtype
        super().__init__(state, dataset, batch, hidden, classes, loss_mode, fused_update=False,
                         post_update_accuracy=False, gs_inc=gs_inc, log_capacity=log_capacity, use_graphs=use_graphs)
        dev, B, T = self.device, batch, dtype
        d_in = dataset.images.shape[1]
        dims = [d_in] + list(hidden)
        self.dims = dims
        z = lambda *s: torch.zeros(*s, dtype=T, device=dev)
        self.xb, self.xb_t = z(B, d_in), z(d_in, B)
        self.h = [z(B, h) for h in hidden]                 # activations in the compute dtype
        self.h_t = [z(h, B) for h in hidden[:-1]]          # transposed copies feed the weight gradients
        self.hl32 = self.hact[-1]                          # fp32 copy of the last hidden layer for the head kernel
        self.hl_t = z(hidden[-1], B)
        self.dzc = [z(B, h) for h in hidden]
        self.dzc_t = [z(h, B) for h in hidden]
        # compute copies of the weights: W [in,out] for the backward product, W^T [out,in] for the forward one
        self.w = [None if T == torch.float32 else z(dims[i], dims[i + 1]) for i in range(len(hidden))]
        self.w_t = [z(dims[i + 1], dims[i]) for i in range(len(hidden))]
        self.refresh_compute_copies()
        self.grad_ready = None              # engine hook: called with a bucket index of grad_groups()
        self.folded = [False] * len(hidden)  # layers whose weight-gradient epilogue applies the optimizer itself
        # bucketed aggregation: the engine's side stream rebuilds a layer's compute copies right behind the bucket's
        # all-gather (use_side_refresh), so the next step starts without a pass over all the masters
        self._run_behind = None
        self._side_fresh = [False] * len(hidden)
        # weight-gradient products whose tiles would fill a fraction of the chip (first layer: 784 inputs) run split-K
        self.split_k = [None] * len(hidden)
        if self.is_cuda and T == torch.bfloat16 and os.environ.get("DTF_SPLIT_K", "1") != "0":
            for l in range(len(hidden)):
                s = ops.split_k_for(dims[l], dims[l + 1], B)
                if s > 1:
                    self.split_k[l] = ops.SplitK(s, dims[l], dims[l + 1], dev)
        # bf16 on the GPU: the output layer's kernels read the bf16 activation and its transpose as the last forward
        # product wrote them (16-wave kernels of csrc/kernels/head_wide.hip); otherwise the fp32 copy and the 4-wave
        # kernels shared with the reference model.  DTF_WIDE_HEAD=0 selects the latter for comparison.
        self.bf16_head = (self.is_cuda and T == torch.bfloat16 and self.R == self.B and os.environ.get("DTF_WIDE_HEAD", "1") != "0"
                          and ops.head_wide_bf16_takes(B, hidden[-1], classes) and self._head_writes_bf16())
        del self.schedule_tail_entries      # this program enqueues op by op; the engine then uses after_compute
        if self.is_cuda:
            self._args = [self._build_args(q) for q in range(self.P)]      # now with the compute-dtype buffers

    def grad_groups(self):
        """Gradient buckets in layout order: one per hidden layer, then the output layer."""
        return [[f"{n}_w", f"{n}_b"] for n in self.names] + [["sm_w", "sm_b"]]

    def _head_dz(self):
        """Where the head writes dz of the last hidden layer: with bf16 products straight into their operands (row-major
        and transposed copy), else as fp32 for the cast that follows."""
        if self._head_writes_bf16():
            return {"dz1": None, "dz1_bf": self.dzc[-1], "dz1_bf_t": self.dzc_t[-1]}
        return {"dz1": self.dz[-1]}

    def _head_writes_bf16(self) -> bool:
        # (the base constructor builds its arguments before the compute-dtype buffers exist; they are rebuilt below)
        return self.cdtype == torch.bfloat16 and hasattr(self, "dzc_t") and self.dzc_t[-1].stride(0) % 4 == 0

    def _build_args(self, q: int):
        st = self.state
        if getattr(self, "bf16_head", False):
            head = ops.build_head_wide_args(self.h[-1], self.hl_t, st.var("sm_w"), st.var("sm_b"), self.data.labels, self.B,
                                            self.loss_mode, rows=self.ridx[q], dz2=self.dz2, dz1_bf=self.dzc[-1],
                                            dz1_bf_t=self.dzc_t[-1], loss_part=self.loss_part,
                                            correct_part=self.correct_part, post_part=self.post_part)
            wgrad = ops.build_out_wgrad_wide_args(self.hl_t, self.dz2, st.grad("sm_w"), st.grad("sm_b"), self.B,
                                                  tail=self._tail(q))
            return {"head": head, "wgrad": wgrad, "items": None}
        head = ops.build_head_args(self.hact[-1], st.var("sm_w"), st.var("sm_b"), self.data.labels, self.R, self.B,
                                   self.loss_mode, rows=self.ridx[q], dz2=self.dz2, **self._head_dz(),
                                   loss_part=self.loss_part, correct_part=self.correct_part, post_part=self.post_part)
        items = [ops.WgradItem(self.hact[-1], self.dz2, self.B, n=self.C, dw=st.grad("sm_w"), db=st.grad("sm_b"),
                               w=st.var("sm_w"), b=st.var("sm_b"))]
        wgrad = ops.build_wgrad_args(items, lr=None, tail=self._tail(q))
        return {"head": head, "wgrad": wgrad, "items": items}

    def refresh_compute_copies(self, skip_folded: bool = False) -> None:
        """After an update of the fp32 master weights: rewrite W (compute dtype) and W^T."""
        for l in range(len(self.names)):
            if skip_folded and (self.folded[l] or self._side_fresh[l]):
                continue                    # kept current by the weight-gradient epilogue / behind the bucket of the previous step
            self._refresh_layer(l)

    def _refresh_layer(self, l: int) -> None:
        ops.gather_cast_transpose(self.state.var(f"{self.names[l]}_w"), None, self.w[l], self.w_t[l])

    def use_side_refresh(self, run_behind) -> None:
        """Bucketed aggregation (``parallel/bucketed.py``): ``run_behind(fn)`` enqueues fn on the engine's second
        stream behind the buckets released so far and behind the compute stream's work so far.  The step then rebuilds
        a layer's compute copies there, after the layer's all-gather and after the backward product that still reads
        the old copy, instead of rebuilding every layer's copies at the start of the next step.
        DTF_SIDE_REFRESH=0 keeps the pass at the start of the step."""
        if self.is_cuda and self.cdtype != torch.float32 and os.environ.get("DTF_SIDE_REFRESH", "1") != "0":
            self._run_behind = run_behind

    def enable_folded_optimizer(self):
        """One worker, bf16: the weight-gradient products that run on the 8-wave kernel apply the optimizer in their
        epilogue (master weights and moments read and written once, the gradient never stored, the compute copy
        refreshed in place).  Returns the flat ranges the engine still has to update (biases, the layers whose
        product is too small for that kernel, the output layer)."""
        st = self.state
        if st.world_size != 1 or self.cdtype != torch.bfloat16 or not self.is_cuda:
            raise ValueError("the folded optimizer needs one worker and the bf16 GPU pipeline")
        for l in range(len(self.names)):
            M, N, K = self.dims[l], self.dims[l + 1], self.B
            blocks = -(-M // 256) * -(-N // 256)
            self.folded[l] = (ops.gemm_pingpong_takes(M, N, K) and blocks >= 192 and (M % 256 == 0 or M >= 2048)
                              and (N % 256 == 0 or N >= 2048))
        self._graphs.clear()
        ranges, lo = [], 0
        for l, name in enumerate(self.names):
            if self.folded[l]:
                off = st.layout.offsets[f"{name}_w"]
                if off > lo:
                    ranges.append((lo, off))
                lo = off + st.var(f"{name}_w").numel()
                lo = (lo + 3) // 4 * 4
        if lo < st.layout.total:
            ranges.append((lo, st.layout.total))
        return ranges

    def run(self, n_steps: int, graph_steps: int = 64) -> None:
        # the host has rewritten the masters (initialise, restore, broadcast) since the copies were made: rebuild all
        # of them here, outside any captured step (steps skip the layers whose epilogue keeps the copies current)
        if self._started and getattr(self, "_values_seen", None) != self.state.values_version:
            self.refresh_compute_copies()
            self._values_seen = self.state.values_version
        super().run(n_steps, graph_steps)

    def start(self) -> None:
        super().start()
        self.refresh_compute_copies()       # the masters may have been initialised or restored since construction
        self._values_seen = self.state.values_version

    def _weight(self, l: int) -> torch.Tensor:
        return self.w[l] if self.w[l] is not None else self.state.var(f"{self.names[l]}_w")

    def enqueue_step(self) -> None:
        q = self.steps_enqueued & 1
        st, B, L = self.state, self.B, len(self.names)
        rows = self.ridx[q]
        # the master weights may have been written since the last step (optimizer, all-gather, an async pull)
        self.refresh_compute_copies(skip_folded=True)
        ops.gather_cast_transpose(self.data.images, rows, self.xb, self.xb_t, m=B)
        x = self.xb
        for l, name in enumerate(self.names):
            last = l == L - 1
            ops.gemm_nt(x, self.w_t[l], self.h[l], ops.EPI_FWD, bias=st.var(f"{name}_b"), relu=True,
                        out_t=self.hl_t if last else self.h_t[l],
                        out_f32=self.hl32 if last and not self.bf16_head else None, m=B)
            x = self.h[l]
        if self.bf16_head:
            ops.head_wide_bf16(self._args[q]["head"])
        elif self.is_cuda:
            ops.head(self.hl32, st.var("sm_w"), st.var("sm_b"), self.data.labels, self.R, B, args=self._args[q]["head"])
        else:
            ops.head(self.hl32, st.var("sm_w"), st.var("sm_b"), self.data.labels, self.R, B, self.loss_mode, rows=rows,
                     dz2=self.dz2, loss_part=self.loss_part, correct_part=self.correct_part,
                     post_part=self.post_part, **self._head_dz())
        if not self._head_writes_bf16():
            # dz of the last hidden layer: fp32 from the head -> compute dtype + transposed copy
            ops.gather_cast_transpose(self.dz[-1], None, self.dzc[-1], self.dzc_t[-1], m=B)
        # fp32 runs read the master weights in the backward product, which the bucket's all-gather rewrites:
        # there the bucket is released only after that product has been enqueued
        early = self.grad_ready is not None and self.w[0] is not None
        for l in range(L - 1, -1, -1):
            a_t = self.xb_t if l == 0 else self.h_t[l - 1]
            wname = f"{self.names[l]}_w"
            if self.folded[l]:
                # the epilogue below rewrites W_l: the backward product that reads it goes first
                if l > 0:
                    ops.gemm_nt(self.dzc[l], self._weight(l), self.dzc[l - 1], ops.EPI_BWD, mask=self.h[l - 1],
                                mask_t=self.h_t[l - 1], out_t=self.dzc_t[l - 1], m=B)
                adam = st.optimizer != "sgd"
                opt = ops.FoldedOptimizer(st.var(wname), st._view(st.m, wname) if adam else None,
                                          st._view(st.v, wname) if adam else None, self.w[l], st.lr, st.beta1, st.beta2,
                                          st.eps, st.state[2:3], 1, wt=self.w_t[l])
                ops.gemm_nt(a_t, self.dzc_t[l], st.grad(wname), ops.EPI_WGRAD, bias_out=st.grad(f"{self.names[l]}_b"),
                            opt=opt)            # refreshes W_l and W_l^T (compute dtype) itself
                continue
            ops.gemm_nt(a_t, self.dzc_t[l], st.grad(wname), ops.EPI_WGRAD,
                        bias_out=st.grad(f"{self.names[l]}_b"), split_k=self.split_k[l])
            if early:
                self.grad_ready(l)
            if l > 0:
                ops.gemm_nt(self.dzc[l], self._weight(l), self.dzc[l - 1], ops.EPI_BWD, mask=self.h[l - 1],
                            mask_t=self.h_t[l - 1], out_t=self.dzc_t[l - 1], m=B)
            if self.grad_ready is not None and not early:
                self.grad_ready(l)
            if self.grad_ready is not None and self._run_behind is not None:
                # W_l's last reader of this step has been enqueued: its copies may be rewritten once its bucket is back
                self._run_behind(lambda l=l: self._refresh_layer(l))
                self._side_fresh[l] = True
        if self.bf16_head:
            ops.out_wgrad_wide(self._args[q]["wgrad"])
        elif self.is_cuda:
            ops.wgrad_grouped(self._args[q]["items"], args=self._args[q]["wgrad"])
        else:
            st_ = self.state
            ops.wgrad_grouped([ops.WgradItem(self.hact[-1], self.dz2, self.B, n=self.C, dw=st_.grad("sm_w"),
                                             db=st_.grad("sm_b"), w=st_.var("sm_w"), b=st_.var("sm_b"))], lr=None)
            self._tail_cpu(q, init_only=False)
        if self.grad_ready is not None:
            self.grad_ready(L)
        if self.after_compute is not None:
            self.after_compute()
        self.steps_enqueued += 1

    def _count_correct(self, images, labels, rows=None, n=None, chunk: int = 8192) -> int:
        n = n if n is not None else images.shape[0]
        st, T = self.state, self.cdtype
        total = 0
        self.refresh_compute_copies()
        for lo in range(0, n, chunk):
            m = min(chunk, n - lo)
            mp = (m + 7) // 8 * 8
            buf = self._eval_buf.get(mp)
            if buf is None:
                nb = (m + 15) // 16
                buf = {"x": torch.zeros((mp, images.shape[1]), dtype=T, device=self.device),
                       "h": [torch.zeros((mp, h), dtype=T, device=self.device) for h in self.hidden],
                       "h32": torch.zeros((mp, self.hidden[-1]), dtype=torch.float32, device=self.device),
                       "lp": torch.zeros(nb, dtype=torch.float32, device=self.device),
                       "cp": torch.zeros(nb, dtype=torch.int32, device=self.device),
                       "pp": torch.zeros(nb, dtype=torch.int32, device=self.device)}
                self._eval_buf = {mp: buf}
            if rows is not None:
                ops.gather_cast_transpose(images, rows[lo:lo + m].contiguous(), buf["x"], None, m=m)
                lab, lrows = labels, rows[lo:lo + m].contiguous()
            else:
                ops.gather_cast_transpose(images[lo:lo + m], None, buf["x"], None, m=m)
                lab, lrows = labels[lo:lo + m], None
            x = buf["x"]
            for l, name in enumerate(self.names):
                last = l == len(self.names) - 1
                ops.gemm_nt(x, self.w_t[l], buf["h"][l], ops.EPI_FWD, bias=st.var(f"{name}_b"), relu=True,
                            out_f32=buf["h32"] if last else None, m=m)
                x = buf["h"][l]
            ops.head(buf["h32"], st.var("sm_w"), st.var("sm_b"), lab, m, 0, self.loss_mode, rows=lrows, loss_part=buf["lp"],
                     correct_part=buf["cp"], post_part=buf["pp"])
            total += int(buf["pp"].sum().item())
        return total
