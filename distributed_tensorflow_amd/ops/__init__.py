"""Op layer: every op has one HIP implementation (GPU tensors, gfx950 kernels in
``csrc/kernels``) and one plain PyTorch statement (CPU tensors, ``reference.py``).

GPU tensors always go to the HIP kernels; if the kernel library is missing the
call raises instead of falling back.  All GPU entry points write into
caller-provided output buffers so that a whole training step allocates nothing
and can be captured into a hipGraph.
"""
from __future__ import annotations

import ctypes as C
import os
import sys

import torch

from . import _lib, reference
from .reference import LOSS_REFERENCE, LOSS_STANDARD  # noqa: F401


def _f32c(t, name):
    if t.dtype != torch.float32 or not t.is_contiguous():
        raise ValueError(f"{name}: expected a contiguous float32 tensor, got {t.dtype} contiguous={t.is_contiguous()}")
    return t


def _i32c(t, name):
    if t.dtype != torch.int32 or not t.is_contiguous():
        raise ValueError(f"{name}: expected a contiguous int32 tensor")
    return t


def _same_device(*ts):
    dev = None
    for t in ts:
        if t is None:
            continue
        if dev is None:
            dev = t.device
        elif t.device != dev:
            raise ValueError(f"tensors on different devices: {dev} vs {t.device}")
    return dev


# --------------------------------------------------------------------------- linear
def linear_fwd(x, w, b=None, relu=False, rows=None, out=None, m=None, _build_only=False):
    """``out[:m] = act(x[rows[:m]] @ w + b)``; K1 of the reference graph (distributed.py:78-79)."""
    K, N = w.shape
    M = m if m is not None else (rows.shape[0] if rows is not None else x.shape[0])
    if x.shape[1] != K:
        raise ValueError(f"linear_fwd: x has {x.shape[1]} columns, w has {K} rows")
    if not x.is_cuda:
        y = reference.linear_fwd(x if rows is not None else x[:M], w, b, relu, rows[:M] if rows is not None else None)
        if out is None:
            return y
        out[:M].copy_(y)
        return out
    _same_device(x, w, b, rows, out)
    if out is None:
        out = torch.empty((M, N), dtype=torch.float32, device=x.device)
    if out.shape[0] < M or out.shape[1] != N:
        raise ValueError(f"linear_fwd: out {tuple(out.shape)} cannot hold [{M},{N}]")
    if rows is not None and rows.shape[0] < M:
        raise ValueError("linear_fwd: rows shorter than m")
    if rows is None and x.shape[0] < M:
        raise ValueError("linear_fwd: x has fewer rows than m")
    if b is not None and b.numel() != N:
        raise ValueError("linear_fwd: bias size mismatch")
    a = _lib.SmallGemmArgs(A=_f32c(x, "x").data_ptr(), a_rows=_lib.ptr(None if rows is None else _i32c(rows, "rows")),
                           B=_f32c(w, "w").data_ptr(), bias=_lib.ptr(None if b is None else _f32c(b, "b")), mask=None,
                           C=_f32c(out, "out").data_ptr(), M=M, N=N, K=K, lda=x.stride(0), ldb=w.stride(0),
                           ldc=out.stride(0), ldmask=0, relu=int(bool(relu)), b_transposed=0,
                           a_bytes=_lib.span_bytes(x), b_bytes=_lib.span_bytes(w))
    if _build_only:
        return a
    _lib.check(_lib.kernels().dtf_small_gemm(C.byref(a), _lib.stream_handle()), "linear_fwd")
    return out


def linear_fwd_call(x, w, b, relu, rows, out, m):
    """(launcher, validated argument struct) of a forward launch, for callers that replay it every step."""
    return _lib.kernels().dtf_small_gemm, linear_fwd(x, w, b, relu, rows, out, m, _build_only=True)


def linear_bwd_data_call(dy, w, act, out, m):
    return _lib.kernels().dtf_small_gemm, linear_bwd_data(dy, w, act, out, m, _build_only=True)


def linear_bwd_data(dy, w, act=None, out=None, m=None, _build_only=False):
    """``out[:m] = (dy[:m] @ w^T) * (act > 0)``; MatMul grad + ReluGrad (K7)."""
    K, N = w.shape          # w is the forward weight [in, out]; dy is [M, out]
    M = m if m is not None else dy.shape[0]
    if dy.shape[1] != N:
        raise ValueError("linear_bwd_data: dy/w shape mismatch")
    if not dy.is_cuda:
        r = reference.linear_bwd_data(dy[:M], w, None if act is None else act[:M])
        if out is None:
            return r
        out[:M].copy_(r)
        return out
    _same_device(dy, w, act, out)
    if out is None:
        out = torch.empty((M, K), dtype=torch.float32, device=dy.device)
    if out.shape[0] < M or out.shape[1] != K or dy.shape[0] < M:
        raise ValueError("linear_bwd_data: bad out/dy shape")
    if act is not None and (act.shape[0] < M or act.shape[1] != K):
        raise ValueError("linear_bwd_data: bad act shape")
    a = _lib.SmallGemmArgs(A=_f32c(dy, "dy").data_ptr(), a_rows=None, B=_f32c(w, "w").data_ptr(), bias=None,
                           mask=_lib.ptr(None if act is None else _f32c(act, "act")), C=_f32c(out, "out").data_ptr(),
                           M=M, N=K, K=N, lda=dy.stride(0), ldb=w.stride(0), ldc=out.stride(0),
                           ldmask=0 if act is None else act.stride(0), relu=0, b_transposed=1,
                           a_bytes=_lib.span_bytes(dy), b_bytes=_lib.span_bytes(w))
    if _build_only:
        return a
    _lib.check(_lib.kernels().dtf_small_gemm(C.byref(a), _lib.stream_handle()), "linear_bwd_data")
    return out


class WgradItem:
    """One ``dW = x[rows]^T @ dy``, ``db = colsum(dy)`` problem of a grouped launch."""

    def __init__(self, x, dy, kb, n=None, rows=None, dw=None, db=None, w=None, b=None):
        self.x, self.dy, self.kb, self.rows = x, dy, kb, rows
        self.n = n if n is not None else dy.shape[1]
        self.dw, self.db, self.w, self.b = dw, db, w, b


def make_tail(state, log=None, loss_part=None, correct_part=None, post_part=None, ridx_cur=None, ridx_next=None,
              perm=None, batch=0, n_train=0, gs_inc=1, enable=True):
    nparts = 0 if loss_part is None else loss_part.numel()
    return _lib.TailArgs(state=_lib.ptr(state), log=_lib.ptr(log), loss_part=_lib.ptr(loss_part),
                         correct_part=_lib.ptr(correct_part), post_part=_lib.ptr(post_part), ridx_cur=_lib.ptr(ridx_cur),
                         ridx_next=_lib.ptr(ridx_next), perm=_lib.ptr(perm), nparts=nparts,
                         log_cap=0 if log is None else log.shape[0], B=batch, n_train=n_train, gs_inc=gs_inc,
                         enable=int(enable), inv_b=(1.0 / batch) if batch else 0.0, pad_=0)


def build_wgrad_args(items, lr=None, tail=None):
    """Pack a grouped weight-gradient launch.  ``lr`` given => parameters are updated in the
    epilogue (theta -= lr*grad, world size 1 + SGD); otherwise gradients are stored."""
    if not 0 < len(items) <= _lib.MAX_WGRAD_PROBLEMS:
        raise ValueError("wgrad_grouped: between 1 and 8 problems per launch")
    args = _lib.WgradArgs()
    tiles = 0
    fused = lr is not None
    for i, it in enumerate(items):
        mo = it.x.shape[1]
        n = it.n
        tgt_w = it.w if fused else it.dw
        tgt_b = it.b if fused else it.db
        if tgt_w is None or tgt_b is None:
            raise ValueError("wgrad_grouped: missing output tensors")
        if tuple(tgt_w.shape) != (mo, n) or tgt_b.numel() != n:
            raise ValueError(f"wgrad_grouped: output shape {tuple(tgt_w.shape)} != ({mo},{n})")
        if it.dy.shape[0] < it.kb or it.dy.shape[1] < n:
            raise ValueError("wgrad_grouped: dy too small")
        if it.rows is None and it.x.shape[0] < it.kb:
            raise ValueError("wgrad_grouped: x too small")
        if it.rows is not None and it.rows.numel() < it.kb:
            raise ValueError("wgrad_grouped: rows too short")
        p = args.prob[i]
        p.X = _f32c(it.x, "x").data_ptr()
        p.x_rows = _lib.ptr(None if it.rows is None else _i32c(it.rows, "rows"))
        p.dY = _f32c(it.dy, "dy").data_ptr()
        p.dW = _lib.ptr(it.dw)
        p.db = _lib.ptr(it.db)
        p.W = _lib.ptr(it.w)
        p.b = _lib.ptr(it.b)
        p.Mo, p.N, p.Kb = mo, n, it.kb
        p.ldx, p.ldy, p.ldw = it.x.stride(0), it.dy.stride(0), tgt_w.stride(0)
        p.tile_start = tiles
        p.tiles_n = (n + 15) // 16
        p.x_bytes, p.y_bytes = _lib.span_bytes(it.x), _lib.span_bytes(it.dy)
        tiles += p.tiles_n * ((mo + 15) // 16)
    args.nprob = len(items)
    args.total_tiles = tiles
    args.mode = 1 if fused else 0
    args.lr = float(lr) if fused else 0.0
    if tail is not None:
        args.tail = tail
    else:
        args.tail.enable = 0
    return args


def wgrad_grouped(items, lr=None, tail=None, args=None):
    """All weight/bias gradients of the step in one launch (K6, K8, K9; optionally K10+K11 fused)."""
    dev = items[0].x.device if items else None
    if dev is not None and dev.type != "cuda":
        for it in items:
            dw, db = reference.wgrad(it.x if it.rows is not None else it.x[:it.kb], it.dy[:it.kb, :it.n],
                                     None if it.rows is None else it.rows[:it.kb])
            if lr is not None:
                it.w.sub_(lr * dw)
                it.b.sub_(lr * db)
            else:
                it.dw.copy_(dw)
                it.db.copy_(db)
        return None
    if args is None:
        args = build_wgrad_args(items, lr, tail)
    _lib.check(_lib.kernels().dtf_wgrad_grouped(C.byref(args), _lib.stream_handle()), "wgrad_grouped")
    return args


# --------------------------------------------------------------------------- head
def build_head_args(hact, w2, b2, labels, m_total, m_train, loss_mode, rows=None, dz2=None, dz1=None, probs=None,
                    loss_part=None, correct_part=None, post_part=None, dz1_bf=None, dz1_bf_t=None):
    hd, c = w2.shape
    if c > 16:
        raise ValueError("head: at most 16 classes")
    nblk = (m_total + 15) // 16
    for name, t in (("loss_part", loss_part), ("correct_part", correct_part), ("post_part", post_part)):
        if t is not None and t.numel() < nblk:
            raise ValueError(f"head: {name} needs {nblk} entries")
    if hact.shape[0] < m_total or hact.shape[1] != hd:
        raise ValueError("head: hact shape mismatch")
    if rows is not None and rows.numel() < m_total:
        raise ValueError("head: rows too short")
    if rows is None and labels.shape[0] < m_total:
        raise ValueError("head: labels too short")
    if labels.shape[1] != c:
        raise ValueError("head: labels width mismatch")
    if m_train > 0:
        if dz2 is not None and (dz2.shape[0] < m_train or dz2.shape[1] != 16 or not dz2.is_contiguous()):
            raise ValueError("head: dz2 must be [>=m_train, 16] contiguous")
        if dz1 is not None and (dz1.shape[0] < m_train or dz1.shape[1] != hd):
            raise ValueError("head: dz1 shape mismatch")
    if probs is not None and (probs.shape[0] < m_total or probs.shape[1] != c or not probs.is_contiguous()):
        raise ValueError("head: probs shape mismatch")
    if dz1_bf is not None and (dz1_bf.dtype != torch.bfloat16 or dz1_bf.shape[0] < m_train or dz1_bf.shape[1] != hd
                               or dz1_bf.stride(1) != 1):
        raise ValueError("head: dz1_bf must be bf16 [>=m_train, hidden], row-major")
    if dz1_bf_t is not None and (dz1_bf is None or dz1_bf_t.dtype != torch.bfloat16 or dz1_bf_t.shape[0] != hd
                                 or dz1_bf_t.shape[1] < m_train or dz1_bf_t.stride(1) != 1):
        raise ValueError("head: dz1_bf_t must be bf16 [hidden, >=m_train], row-major, next to dz1_bf")
    return _lib.HeadArgs(Hact=_f32c(hact, "hact").data_ptr(), W2=_f32c(w2, "w2").data_ptr(), b2=_f32c(b2, "b2").data_ptr(),
                         labels=_f32c(labels, "labels").data_ptr(),
                         rows=_lib.ptr(None if rows is None else _i32c(rows, "rows")), dz2=_lib.ptr(dz2),
                         dz1=_lib.ptr(dz1), probs=_lib.ptr(probs), loss_part=_lib.ptr(loss_part),
                         correct_part=_lib.ptr(correct_part), post_part=_lib.ptr(post_part), M_total=m_total,
                         M_train=m_train, Hd=hd, C=c, ldh=hact.stride(0), ld_dz1=0 if dz1 is None else dz1.stride(0),
                         loss_mode=loss_mode, inv_b=(1.0 / m_train) if m_train else 0.0,
                         h_bytes=_lib.span_bytes(hact), pad_=0, dz1_bf=_lib.ptr(dz1_bf), dz1_bf_t=_lib.ptr(dz1_bf_t),
                         ld_bf=0 if dz1_bf is None else dz1_bf.stride(0), ld_bf_t=0 if dz1_bf_t is None else dz1_bf_t.stride(0))


def head(hact, w2, b2, labels, m_total, m_train, loss_mode=LOSS_REFERENCE, rows=None, dz2=None, dz1=None, probs=None,
         loss_part=None, correct_part=None, post_part=None, args=None, dz1_bf=None, dz1_bf_t=None):
    """Fused last layer: logits, softmax, (double-)softmax cross-entropy, accuracy counts,
    dz2 and dz1 (K2-K5 + K7 of the reference graph, distributed.py:81-87)."""
    if not hact.is_cuda:
        r = reference.head(hact[:m_total], w2, b2, labels, m_train, loss_mode, rows,
                           want_dz1=dz1 is not None or dz1_bf is not None)
        if probs is not None:
            probs[:m_total].copy_(r["probs"])
        if m_train > 0:
            if dz2 is not None:
                dz2[:m_train].zero_()
                dz2[:m_train, :w2.shape[1]].copy_(r["dz2"])
            if dz1 is not None:
                dz1[:m_train].copy_(r["dz1"])
            if dz1_bf is not None:
                dz1_bf[:m_train].copy_(r["dz1"])
                if dz1_bf_t is not None:
                    dz1_bf_t[:, :m_train].copy_(dz1_bf[:m_train].t())
        for t, v in ((loss_part, r["loss_sum"]), (correct_part, r["correct"]), (post_part, r["post_correct"])):
            if t is not None:
                t.zero_()
                t[0] = v
        return r
    if args is None and w2.shape[1] > 16:
        return _head_wide(hact, w2, b2, labels, m_total, m_train, loss_mode, rows, dz2, dz1, probs, loss_part,
                          correct_part, post_part, dz1_bf, dz1_bf_t)
    if args is None:
        args = build_head_args(hact, w2, b2, labels, m_total, m_train, loss_mode, rows, dz2, dz1, probs, loss_part,
                               correct_part, post_part, dz1_bf, dz1_bf_t)
    _lib.check(_lib.kernels().dtf_head(C.byref(args), _lib.stream_handle()), "head")
    return args


def head_wide_bf16_takes(m: int, hidden: int, classes: int) -> bool:
    """Shapes of the 16-wave output-layer kernels (``csrc/kernels/head_wide.hip``)."""
    return m % 8 == 0 and hidden % 8 == 0 and 0 < classes <= 16


def build_head_wide_args(h, h_t, w2, b2, labels, m, loss_mode, rows=None, dz2=None, dz1_bf=None, dz1_bf_t=None,
                         loss_part=None, correct_part=None, post_part=None):
    """Arguments of ``head_wide_bf16``: ``h`` bf16 [>=m, hidden] and its transpose ``h_t`` [hidden, >=m] as the forward
    product wrote them; every one of the ``m`` rows is a training row."""
    hd, c = w2.shape
    if not head_wide_bf16_takes(m, hd, c):
        raise ValueError("head_wide_bf16: m and hidden must be multiples of 8, at most 16 classes")
    for name, t, shp in (("h", h, (m, hd)), ("h_t", h_t, (hd, m))):
        if t.dtype != torch.bfloat16 or t.shape[0] < shp[0] or t.shape[1] < shp[1] or t.stride(1) != 1 or t.stride(0) % 8:
            raise ValueError(f"head_wide_bf16: {name} must be bf16 [>={shp[0]}, >={shp[1]}], row-major, rows 16-byte aligned")
    if h.shape[1] != hd or h_t.shape[0] != hd:
        raise ValueError("head_wide_bf16: activation / weight shape mismatch")
    nblk = (m + 15) // 16
    for name, t in (("loss_part", loss_part), ("correct_part", correct_part), ("post_part", post_part)):
        if t is not None and t.numel() < nblk:
            raise ValueError(f"head_wide_bf16: {name} needs {nblk} entries")
    if rows is not None and rows.numel() < m:
        raise ValueError("head_wide_bf16: rows too short")
    if rows is None and labels.shape[0] < m:
        raise ValueError("head_wide_bf16: labels too short")
    if labels.shape[1] != c:
        raise ValueError("head_wide_bf16: labels width mismatch")
    if dz2 is not None and (dz2.dtype != torch.float32 or dz2.shape[0] < m or dz2.shape[1] != 16 or not dz2.is_contiguous()):
        raise ValueError("head_wide_bf16: dz2 must be float32 [>=m, 16] contiguous")
    if dz1_bf is not None and (dz1_bf.dtype != torch.bfloat16 or dz1_bf.shape[0] < m or dz1_bf.shape[1] != hd
                               or dz1_bf.stride(1) != 1):
        raise ValueError("head_wide_bf16: dz1_bf must be bf16 [>=m, hidden], row-major")
    if dz1_bf_t is not None and (dz1_bf is None or dz1_bf_t.dtype != torch.bfloat16 or dz1_bf_t.shape[0] != hd
                                 or dz1_bf_t.shape[1] < m or dz1_bf_t.stride(1) != 1):
        raise ValueError("head_wide_bf16: dz1_bf_t must be bf16 [hidden, >=m], row-major, next to dz1_bf")
    _same_device(h, h_t, w2, b2, labels, rows, dz2, dz1_bf, dz1_bf_t, loss_part, correct_part, post_part)
    return _lib.HeadWideArgs(H=h.data_ptr(), Ht=h_t.data_ptr(), W2=_f32c(w2, "w2").data_ptr(), b2=_f32c(b2, "b2").data_ptr(),
                             labels=_f32c(labels, "labels").data_ptr(),
                             rows=_lib.ptr(None if rows is None else _i32c(rows, "rows")), dz2=_lib.ptr(dz2),
                             dz1_bf=_lib.ptr(dz1_bf), dz1_bf_t=_lib.ptr(dz1_bf_t), loss_part=_lib.ptr(loss_part),
                             correct_part=_lib.ptr(correct_part), post_part=_lib.ptr(post_part), M=m, Hd=hd, C=c,
                             ldh=h.stride(0), ldht=h_t.stride(0), ld_bf=0 if dz1_bf is None else dz1_bf.stride(0),
                             ld_bf_t=0 if dz1_bf_t is None else dz1_bf_t.stride(0), loss_mode=loss_mode, inv_b=1.0 / m,
                             pad_=0)


def head_wide_bf16(args):
    """Output layer of a wide bf16 network on the GPU: logits, (double-)softmax cross-entropy, accuracy count, dz2 and
    the masked dz of the last hidden layer (bf16, row-major + transposed), reading the bf16 activation and its
    transpose (distributed.py:81-87 of the reference)."""
    _lib.check(_lib.kernels().dtf_head_wide(C.byref(args), _lib.stream_handle()), "head_wide_bf16")
    return args


def build_out_wgrad_wide_args(h_t, dz2, dw, db, batch, tail=None):
    hd, c = dw.shape
    if h_t.dtype != torch.bfloat16 or h_t.shape[0] != hd or h_t.shape[1] < batch or h_t.stride(1) != 1 or h_t.stride(0) % 8:
        raise ValueError("out_wgrad_wide: h_t must be bf16 [hidden, >=batch], row-major, rows 16-byte aligned")
    if batch % 8 or not 0 < c <= 16:
        raise ValueError("out_wgrad_wide: batch must be a multiple of 8, at most 16 classes")
    if dz2.dtype != torch.float32 or dz2.shape[0] < batch or dz2.shape[1] != 16 or not dz2.is_contiguous():
        raise ValueError("out_wgrad_wide: dz2 must be float32 [>=batch, 16] contiguous")
    if dw.dtype != torch.float32 or dw.stride(1) != 1 or db.dtype != torch.float32 or db.numel() != c:
        raise ValueError("out_wgrad_wide: dw [hidden, classes] and db [classes] must be float32")
    _same_device(h_t, dz2, dw, db)
    args = _lib.OutWgradWideArgs(Ht=h_t.data_ptr(), dz2=dz2.data_ptr(), dW=dw.data_ptr(), db=db.data_ptr(), Hd=hd, C=c,
                                 B=batch, ldht=h_t.stride(0), ldw=dw.stride(0), pad_=0)
    if tail is not None:
        args.tail = tail
    return args


def out_wgrad_wide(args):
    """``dw = h^T dz2``, ``db = colsum(dz2)`` of the output layer from the transposed bf16 activation, and the step
    tail when the arguments carry one."""
    _lib.check(_lib.kernels().dtf_out_wgrad_wide(C.byref(args), _lib.stream_handle()), "out_wgrad_wide")
    return args


def _head_wide(hact, w2, b2, labels, m_total, m_train, loss_mode, rows, dz2, dz1, probs, loss_part, correct_part, post_part,
               dz1_bf, dz1_bf_t):
    """The last layer with more than 16 classes (beyond the one-workgroup head kernel): logits by the small GEMM,
    loss / gradient / hit counts by the general softmax cross-entropy kernel (training rows with gradient, the
    accuracy-only rows without), dz1 by the masked backward GEMM.  Same outputs as the head kernel; the partial
    arrays receive their sums in entry 0."""
    if probs is not None or dz1_bf is not None or dz1_bf_t is not None:
        raise ValueError("head: probabilities / bf16 gradients are only produced for at most 16 classes")
    c = w2.shape[1]
    z = linear_fwd(hact, w2, b2, relu=False, m=m_total)
    for t in (loss_part, correct_part, post_part):
        if t is not None:
            t.zero_()
    if m_train > 0:
        if dz2 is not None and (dz2.shape[0] < m_train or dz2.shape[1] < c):
            raise ValueError("head: dz2 must be [>=m_train, >=C]")
        g = dz2[:m_train, :c] if dz2 is not None else None
        r = softmax_xent(z[:m_train], labels if rows is not None else labels[:m_train], loss_mode,
                         grad_scale=1.0 / m_train, want_grad=g is not None, dlogits=g, label_rows=rows)
        if loss_part is not None:
            loss_part[0:1].copy_(r.loss_sum.reshape(1))
        if correct_part is not None:
            correct_part[0:1].copy_(r.correct.reshape(1))
        if dz1 is not None:
            linear_bwd_data(g, w2, act=hact, out=dz1, m=m_train)
    if m_total > m_train and post_part is not None:
        r = softmax_xent(z[m_train:m_total], labels if rows is not None else labels[m_train:m_total], loss_mode,
                         want_grad=False, label_rows=None if rows is None else rows[m_train:m_total])
        post_part[0:1].copy_(r.correct.reshape(1))
    return None


class XentResult:
    """Outputs of :func:`softmax_xent`: per-workgroup partial loss sums and hit counts (device tensors, folded without
    a host synchronisation by ``loss_sum`` / ``correct``), optional per-row losses and gradients."""

    def __init__(self, loss_part, correct_part, loss_rows, dlogits, dlogits_bf, args=None):
        self.loss_part, self.correct_part, self.loss_rows = loss_part, correct_part, loss_rows
        self.dlogits, self.dlogits_bf, self.args = dlogits, dlogits_bf, args

    @property
    def loss_sum(self) -> torch.Tensor:
        return self.loss_part.sum()

    @property
    def correct(self) -> torch.Tensor:
        return self.correct_part.sum()


def softmax_xent(logits, labels, loss_mode=LOSS_STANDARD, grad_scale=None, want_grad=True, dlogits=None, dlogits_bf=None,
                 loss_rows=None, loss_part=None, correct_part=None, label_rows=None, _build_only=False) -> XentResult:
    """Fused softmax cross-entropy for ANY class count (K3-K5 of the reference graph beyond the 16-class head
    kernel; ``csrc/kernels/softmax_xent.hip``): loss, gradient wrt the logits (scaled by ``grad_scale``, default
    ``1/B``) and the argmax hit count in one pass.  ``labels``: float ``[B, C]`` (one-hot / soft) or int64 ``[B]``
    class indices.  ``loss_mode``: ``LOSS_STANDARD`` or ``LOSS_REFERENCE`` (softmax applied twice, distributed.py:81+86).
    ``dlogits_bf``: additionally (or only) the gradient as bf16, the operand of bf16 backward products."""
    if logits.dim() != 2 or logits.dtype != torch.float32:
        raise ValueError("softmax_xent: logits must be float32 [B, C]")
    b, c = logits.shape
    if label_rows is not None and (label_rows.dtype != torch.int32 or label_rows.numel() < b):
        raise ValueError("softmax_xent: label_rows must be int32 [>=B]")
    nlab = b if label_rows is None else labels.shape[0]
    if labels.dim() == 2:
        if tuple(labels.shape) != (nlab, c) or labels.dtype != torch.float32:
            raise ValueError("softmax_xent: label rows must be float32 [B, C]")
    elif labels.dim() != 1 or labels.numel() != nlab or labels.dtype != torch.int64:
        raise ValueError("softmax_xent: class indices must be int64 [B]")
    scale = (1.0 / b) if grad_scale is None else float(grad_scale)
    dev = logits.device
    if not logits.is_cuda:
        if label_rows is not None:
            labels = labels.index_select(0, label_rows[:b].long())
        r = reference.softmax_xent(logits, labels, loss_mode, scale)
        lp = torch.zeros(1, dtype=torch.float32) if loss_part is None else loss_part.zero_()
        cp = torch.zeros(1, dtype=torch.int32) if correct_part is None else correct_part.zero_()
        lp[0], cp[0] = r["loss_rows"].sum(), r["correct"]
        if loss_rows is not None:
            loss_rows.copy_(r["loss_rows"])
        if want_grad and dlogits is None and dlogits_bf is None:
            dlogits = torch.empty_like(logits)
        if dlogits is not None:
            dlogits.copy_(r["dlogits"])
        if dlogits_bf is not None:
            dlogits_bf.copy_(r["dlogits"])
        return XentResult(lp, cp, r["loss_rows"] if loss_rows is None else loss_rows, dlogits, dlogits_bf)
    lib = _lib.kernels()
    nblk = lib.dtf_softmax_xent_blocks(b)
    if loss_part is None:
        loss_part = torch.empty(nblk, dtype=torch.float32, device=dev)
    if correct_part is None:
        correct_part = torch.empty(nblk, dtype=torch.int32, device=dev)
    if loss_part.numel() < nblk or correct_part.numel() < nblk:
        raise ValueError(f"softmax_xent: partial buffers need {nblk} entries")
    if want_grad and dlogits is None and dlogits_bf is None:
        dlogits = torch.empty_like(logits)
    for name, t, dt in (("dlogits", dlogits, torch.float32), ("dlogits_bf", dlogits_bf, torch.bfloat16)):
        if t is not None and (t.dtype != dt or t.shape[0] < b or t.shape[1] != c or t.stride(1) != 1):
            raise ValueError(f"softmax_xent: {name} must be {dt} [>=B, C], row-major")
    if logits.stride(1) != 1 or (labels.dim() == 2 and labels.stride(1) != 1):
        raise ValueError("softmax_xent: operands must be row-major")
    _same_device(logits, labels, dlogits, dlogits_bf, loss_rows, loss_part, correct_part, label_rows)
    args = _lib.XentArgs(logits=logits.data_ptr(), labels_f=labels.data_ptr() if labels.dim() == 2 else None,
                         labels_i=labels.data_ptr() if labels.dim() == 1 else None, dlogits=_lib.ptr(dlogits),
                         dlogits_bf=_lib.ptr(dlogits_bf), loss_rows=_lib.ptr(loss_rows), loss_part=loss_part.data_ptr(),
                         correct_part=correct_part.data_ptr(), B=b, C=c, ld_logits=logits.stride(0),
                         ld_labels=labels.stride(0) if labels.dim() == 2 else 0,
                         ld_dlogits=0 if dlogits is None else dlogits.stride(0),
                         ld_dlogits_bf=0 if dlogits_bf is None else dlogits_bf.stride(0), loss_mode=int(loss_mode),
                         grad_scale=scale, label_rows=_lib.ptr(label_rows))
    if _build_only:
        return XentResult(loss_part[:nblk], correct_part[:nblk], loss_rows, dlogits, dlogits_bf, args)
    _lib.check(lib.dtf_softmax_xent(C.byref(args), _lib.stream_handle()), "softmax_xent")
    return XentResult(loss_part[:nblk], correct_part[:nblk], loss_rows, dlogits, dlogits_bf, args)


def mlp2_eligible(in_dim, hidden, classes, batch) -> bool:
    """Shapes the two-launch step of ``csrc/kernels/mlp2_fused.hip`` covers (the reference network does)."""
    return len(hidden) == 1 and hidden[0] <= 128 and classes <= 16 and batch <= 128 and in_dim % 4 == 0


MLP2_SLOTS = 3          # rotation period of the per-step buffers of the two-launch step
MLP2_LABG_ROWS = 128
MLP2_ROWS = 128          # rows per slot of the gathered-batch buffer


def build_mlp2_args(x, labels, xg, labg, ridx, hact, zpart, w2_snap, ctr_snap, w1, b1, w2, b2, wlog, batch, rows_total,
                    loss_mode, slot, tail, lr=None, dw1=None, db1=None, dw2=None, db2=None, peer=None, async_push=None,
                    shard=None):
    """Argument pack of the two-launch step (K1-K11 of the reference graph in two kernels) for buffer slot
    ``slot``.  ``lr`` given: SGD applied in the epilogue; otherwise the four gradients are stored.  ``peer``
    (an ``_lib.Mlp2Peer`` whose own slot holds the four gradient tensors) together with ``lr``: the backward
    launch exchanges gradient tiles with the peers and applies ``theta -= lr * sum over ranks``.  ``async_push``
    (an ``_lib.Mlp2Async``) together with ``lr``: the backward launch applies ``lr * gradient`` to the mapped master
    slices and keeps the result as the local replica (async mode).  ``shard`` (an ``_lib.Mlp2Shard``) together with
    ``lr``: sharded exchange, every workgroup's tiles are summed and updated by their owner rank and redistributed."""
    in_dim, hd = w1.shape
    c = w2.shape[1]
    if not mlp2_eligible(in_dim, [hd], c, batch):
        raise ValueError("mlp2: shape outside the fused step's range")
    if rows_total not in (batch, 2 * batch):
        raise ValueError("mlp2: rows_total must be B or 2B")
    if not 0 <= slot < MLP2_SLOTS:
        raise ValueError("mlp2: slot out of range")
    r16 = (rows_total + 15) // 16
    if x.shape[1] != in_dim or tuple(w2.shape) != (hd, c) or b1.numel() != hd or b2.numel() != c:
        raise ValueError("mlp2: parameter shapes do not chain")
    if labels.shape[1] != c or labels.shape[0] != x.shape[0]:
        raise ValueError("mlp2: labels do not match the dataset")
    if hact.shape[0] < max(rows_total, MLP2_ROWS) or hact.shape[1] != hd:
        raise ValueError("mlp2: hact needs max(rows_total, 128) rows")
    if tuple(xg.shape) != (MLP2_SLOTS, MLP2_ROWS, in_dim) or tuple(labg.shape) != (MLP2_SLOTS, 16, MLP2_LABG_ROWS):
        raise ValueError("mlp2: gathered-batch buffers have the wrong shape")
    if zpart.numel() != 16 * r16 * 16 * 8 or w2_snap.numel() < hd * c or ctr_snap.numel() < 2:
        raise ValueError("mlp2: scratch buffers have the wrong size")
    if wlog.dim() != 2 or wlog.shape[1] != _lib.WLOG_INTS:
        raise ValueError("mlp2: wide log must be [capacity, 64]")
    if ridx.dim() != 2 or ridx.shape[0] != MLP2_SLOTS or ridx.shape[1] < 2 * batch:
        raise ValueError("mlp2: row index table must be [3, >= 2B]")
    fused = lr is not None and peer is None and async_push is None and shard is None
    if sum(x is not None for x in (peer, async_push, shard)) > 1:
        raise ValueError("mlp2: peer exchange, sharded exchange and async push are different modes")
    if shard is not None and lr is None:
        raise ValueError("mlp2: the sharded exchange applies the update itself and needs lr (already divided by R)")
    if async_push is not None and lr is None:
        raise ValueError("mlp2: the async push applies the update itself and needs lr")
    if peer is not None and lr is None:
        raise ValueError("mlp2: the peer exchange applies the update itself and needs lr (already divided by R)")
    if not fused and async_push is None and shard is None and any(t is None for t in (dw1, db1, dw2, db2)):
        raise ValueError("mlp2: gradient outputs missing")
    for name, t, shp in (("dw1", dw1, (in_dim, hd)), ("dw2", dw2, (hd, c))):
        if t is not None and tuple(t.shape) != shp:
            raise ValueError(f"mlp2: {name} shape mismatch")
    _same_device(x, ridx, labels, xg, labg, hact, zpart, w2_snap, ctr_snap, w1, b1, w2, b2, wlog, dw1, db1, dw2, db2)
    for name, t in (("w1", w1), ("b1", b1), ("w2", w2), ("b2", b2), ("dw1", dw1), ("db1", db1), ("dw2", dw2), ("db2", db2),
                    ("x", x), ("labels", labels), ("xg", xg), ("labg", labg), ("hact", hact), ("zpart", zpart),
                    ("w2_snap", w2_snap)):
        if t is not None:
            _f32c(t, name)
    for name, t in (("ridx", ridx), ("ctr_snap", ctr_snap), ("wlog", wlog)):
        _i32c(t, name)
    return _lib.Mlp2Args(X=x.data_ptr(), labels=labels.data_ptr(), Xg=xg.data_ptr(), labg=labg.data_ptr(),
                         ridx=ridx.data_ptr(), H=hact.data_ptr(), zpart=zpart.data_ptr(), W2s=w2_snap.data_ptr(),
                         ctr_snap=ctr_snap.data_ptr(), W1=w1.data_ptr(), b1=b1.data_ptr(), W2=w2.data_ptr(),
                         b2=b2.data_ptr(), dW1=_lib.ptr(dw1), db1=_lib.ptr(db1), dW2=_lib.ptr(dw2), db2=_lib.ptr(db2),
                         wlog=wlog.data_ptr(), In=in_dim, Hd=hd, C=c, B=batch, R=rows_total, ldx=x.stride(0),
                         ldh=hact.stride(0), ridx_len=ridx.shape[1], h_rows=hact.shape[0], q=slot, loss_mode=loss_mode,
                         mode=4 if shard is not None else 3 if async_push is not None else (2 if peer is not None else (1 if fused else 0)),
                         shard=shard if shard is not None else _lib.Mlp2Shard(),
                         async_=async_push if async_push is not None else _lib.Mlp2Async(),
                         log_cap=wlog.shape[0], lr=float(lr) if lr is not None else 0.0, inv_b=1.0 / batch,
                         x_bytes=_lib.span_bytes(x), tail=tail, peer=peer if peer is not None else _lib.Mlp2Peer())


def mlp2_fwd(args):
    _lib.check(_lib.kernels().dtf_mlp2_fwd(C.byref(args), _lib.stream_handle()), "mlp2_fwd")


def mlp2_bwd(args):
    _lib.check(_lib.kernels().dtf_mlp2_bwd(C.byref(args), _lib.stream_handle()), "mlp2_bwd")


def pipe_tiles(hd, batch) -> int:
    """Tiles per side of the pipelined step's operand-layout matrices: 8 (the reference shape's class: hidden <= 127,
    batch <= 128) or 16 (hidden <= 256, batch <= 256); ``pipe_td_of`` in ``csrc/kernels/mlp_pipe.hip``."""
    return 8 if hd <= 127 and batch <= 128 else 16


def pipe_eligible(in_dim, hidden, classes, batch) -> bool:
    """Shapes the one-launch pipelined step of ``csrc/kernels/mlp_pipe.hip`` covers: one hidden layer of at most 256
    units, batches of at most 256 rows, at most 16 classes (--hidden_units / --batch_size of
    /root/reference/distributed.py:11-13 are free integers; its defaults are 100 and 100)."""
    if not (len(hidden) == 1 and hidden[0] <= 256 and classes <= 16 and batch <= 256 and in_dim % 4 == 0):
        return False
    chunks = 16 * (4 if pipe_tiles(hidden[0], batch) == 8 else 5)     # split-K of the Q products over 16 waves
    return (in_dim + 16) // 16 + (batch + 15) // 16 <= chunks and (in_dim + 16) // 16 <= 64


class PipeBuffers:
    """Working buffers of the pipelined step (two generations each, see ``DtfPipeArgs``)."""

    def __init__(self, in_dim, hd, device, batch=128):
        f32, i32 = torch.float32, torch.int32
        self.HP = (hd + 1 + 15) // 16 * 16
        self.KAp = (in_dim + 1 + 15) // 16 * 16
        self.td = td = pipe_tiles(hd, batch)
        rows, mat = 16 * td, 256 * td * td
        z = lambda *shape: torch.zeros(shape, dtype=f32, device=device)
        self.xa = z(_lib.PIPE_SLOTS, rows, self.KAp)
        self.labg = z(_lib.PIPE_SLOTS, 16, rows)
        self.wm = z(2, self.KAp, self.HP)
        self.w2m = z(2, self.HP, 16)
        # matrices that are only read as MFMA operands are stored per 16x16 tile in fragment order (mlp_pipe.hip)
        self.q = z(2, mat)
        self.z1 = z(2, mat)
        self.labi = torch.zeros((_lib.PIPE_SLOTS, rows), dtype=i32, device=device)
        self.gg = z(2, 3, mat)
        self.d1 = z(2, mat)
        self.p2 = z(2, td, self.HP, 16)
        self.ctr = torch.zeros((2, 2), dtype=i32, device=device)


def build_pipe_args(x, labels, bufs, ridx, w1, b1, w2, b2, wlog, batch, rows_total, loss_mode, slot, tail, lr, roles,
                    flush=False, prime=False):
    """Argument pack of one launch of the pipelined step: ``roles`` all five step roles (a step at slot ``slot``),
    only Q (the prime launch, ``slot`` = the slot before the run's first step) or W | F with ``flush`` (the flush
    launch, ``slot`` = the slot after the run's last step)."""
    in_dim, hd = w1.shape
    c = w2.shape[1]
    if not pipe_eligible(in_dim, [hd], c, batch):
        raise ValueError("pipelined step: shape outside its range")
    if rows_total not in (batch, 2 * batch):
        raise ValueError("pipelined step: rows_total must be B or 2B")
    if not 0 <= slot < _lib.PIPE_SLOTS:
        raise ValueError("pipelined step: slot out of range")
    if x.shape[1] != in_dim or tuple(w2.shape) != (hd, c) or b1.numel() != hd or b2.numel() != c:
        raise ValueError("pipelined step: parameter shapes do not chain")
    if labels.shape[1] != c or labels.shape[0] != x.shape[0]:
        raise ValueError("pipelined step: labels do not match the dataset")
    if (bufs.HP != (hd + 1 + 15) // 16 * 16 or bufs.KAp != (in_dim + 1 + 15) // 16 * 16
            or bufs.td != pipe_tiles(hd, batch)):
        raise ValueError("pipelined step: working buffers were sized for another network or batch")
    if wlog.dim() != 2 or wlog.shape[1] != _lib.WLOG_INTS:
        raise ValueError("pipelined step: wide log must be [capacity, 64]")
    if ridx.dim() != 2 or ridx.shape[0] != _lib.PIPE_SLOTS or ridx.shape[1] < 2 * batch:
        raise ValueError("pipelined step: row index table must be [8, >= 2B]")
    ts = (x, labels, bufs.xa, bufs.labg, bufs.wm, bufs.w2m, bufs.q, bufs.z1, bufs.gg, bufs.d1, bufs.p2, w1, b1, w2, b2)
    _same_device(*ts, ridx, wlog, bufs.ctr, bufs.labi)
    for i, t in enumerate(ts):
        _f32c(t, f"pipe operand {i}")
    for name, t in (("ridx", ridx), ("ctr", bufs.ctr), ("wlog", wlog), ("labi", bufs.labi)):
        _i32c(t, name)
    return _lib.PipeArgs(X=x.data_ptr(), labels=labels.data_ptr(), Xa=bufs.xa.data_ptr(), labg=bufs.labg.data_ptr(),
                         ridx=ridx.data_ptr(), Wm=bufs.wm.data_ptr(), W2m=bufs.w2m.data_ptr(), Q=bufs.q.data_ptr(),
                         Z1=bufs.z1.data_ptr(), labi=bufs.labi.data_ptr(), GG=bufs.gg.data_ptr(), D1=bufs.d1.data_ptr(), P2=bufs.p2.data_ptr(), ctr=bufs.ctr.data_ptr(),
                         W1=w1.data_ptr(), b1=b1.data_ptr(), W2=w2.data_ptr(), b2=b2.data_ptr(), wlog=wlog.data_ptr(),
                         In=in_dim, Hd=hd, C=c, B=batch, R=rows_total, HP=bufs.HP, KAp=bufs.KAp, ldxa=bufs.KAp,
                         ldx=x.stride(0), ridx_len=ridx.shape[1], q=slot, loss_mode=loss_mode, log_cap=wlog.shape[0],
                         roles=roles, flush=int(flush), prime=int(prime), td=bufs.td, lr=float(lr), inv_b=1.0 / batch, tail=tail)


def pipe_step(args):
    _lib.check(_lib.kernels().dtf_pipe_step(C.byref(args), _lib.stream_handle()), "pipe_step")


def pipe_prime(args):
    _lib.check(_lib.kernels().dtf_pipe_prime(C.byref(args), _lib.stream_handle()), "pipe_prime")


def pipe_flush(args):
    _lib.check(_lib.kernels().dtf_pipe_flush(C.byref(args), _lib.stream_handle()), "pipe_flush")


def pipem_eligible(in_dim, hidden, classes, batch, world) -> bool:
    """Shapes and worlds the N-worker pipelined step of ``csrc/kernels/mlp_pipe_multi.hip`` covers: the one-worker
    shapes, at most 8 workers, and the Gram strips of one row block of every worker within the LDS of one CU."""
    if not (len(hidden) == 1 and hidden[0] <= 127 and classes <= 16 and batch <= 128 and in_dim % 4 == 0
            and (in_dim + 16) // 16 <= 64 and 1 <= world <= 8):
        return False
    nrb = (batch + 15) // 16
    nrb_t = 2 if nrb <= 2 else 4 if nrb <= 4 else 7 if nrb <= 7 else 8
    return 9216 * 4 + 2 * world * nrb_t * 1024 <= 160 * 1024


class PipeMultiBuffers:
    """Working buffers of the N-worker pipelined step: every worker's gathered batches, the Gram matrices against
    every worker's batches; dz1 and the output-layer shares live in the exchange regions (``DtfPipeMultiArgs``)."""

    def __init__(self, in_dim, hd, world, device):
        f32, i32 = torch.float32, torch.int32
        self.world = world
        self.HP = (hd + 1 + 15) // 16 * 16
        self.KAp = (in_dim + 1 + 15) // 16 * 16
        z = lambda *shape: torch.zeros(shape, dtype=f32, device=device)
        self.xa = z(_lib.PIPE_SLOTS, world, 128, self.KAp)
        self.labg = z(_lib.PIPE_SLOTS, 16, 128)
        self.wm = z(2, self.KAp, self.HP)
        self.w2m = z(2, self.HP, 16)
        self.q = z(2, 16384)
        self.z1 = z(2, 16384)
        self.labi = torch.zeros((_lib.PIPE_SLOTS, 128), dtype=i32, device=device)
        self.gg = z(2, 4, world, 16384)
        self.ctr = torch.zeros((2, 2), dtype=i32, device=device)


def build_pipem_args(x, labels, bufs, ridx, w1, b1, w2, b2, wlog, batch, rows_total, loss_mode, slot, tail, lr_over_n, roles,
                     regions, perms, err, rank, world, timeout_s, flush=False, prime=False):
    """Argument pack of one launch of the N-worker pipelined step.  ``regions``: device addresses of every rank's
    exchange region as mapped here; ``perms``: one ``[2, n]`` int32 tensor per rank (the row order of its shuffle) or
    None; ``err``: int32 device tensor whose first word the kernels set when a wait ran into ``timeout_s``;
    ``lr_over_n``: the learning rate divided by the number of workers."""
    in_dim, hd = w1.shape
    c = w2.shape[1]
    if not pipem_eligible(in_dim, [hd], c, batch, world):
        raise ValueError("N-worker pipelined step: shape or world outside its range")
    if rows_total not in (batch, 2 * batch):
        raise ValueError("N-worker pipelined step: rows_total must be B or 2B")
    if not 0 <= slot < _lib.PIPE_SLOTS or not 0 <= rank < world:
        raise ValueError("N-worker pipelined step: slot or rank out of range")
    if x.shape[1] != in_dim or tuple(w2.shape) != (hd, c) or b1.numel() != hd or b2.numel() != c:
        raise ValueError("N-worker pipelined step: parameter shapes do not chain")
    if labels.shape[1] != c or labels.shape[0] != x.shape[0]:
        raise ValueError("N-worker pipelined step: labels do not match the dataset")
    if bufs.world != world or bufs.HP != (hd + 1 + 15) // 16 * 16 or bufs.KAp != (in_dim + 1 + 15) // 16 * 16:
        raise ValueError("N-worker pipelined step: working buffers were sized for another network or world")
    if wlog.dim() != 2 or wlog.shape[1] != _lib.WLOG_INTS:
        raise ValueError("N-worker pipelined step: wide log must be [capacity, 64]")
    if ridx.dim() != 3 or ridx.shape[0] != _lib.PIPE_SLOTS or ridx.shape[1] != world or ridx.shape[2] < 2 * batch:
        raise ValueError("N-worker pipelined step: row index table must be [8, world, >= 2B]")
    if len(regions) != world or any(not r for r in regions):
        raise ValueError("N-worker pipelined step: one mapped region per rank")
    if len(perms) != world or any((t is None) != (perms[0] is None) for t in perms):
        raise ValueError("N-worker pipelined step: one permutation buffer per rank, or none at all")
    ts = (x, labels, bufs.xa, bufs.labg, bufs.wm, bufs.w2m, bufs.q, bufs.z1, bufs.gg, w1, b1, w2, b2)
    _same_device(*ts, ridx, wlog, bufs.ctr, bufs.labi, err, *[t for t in perms if t is not None])
    for i, t in enumerate(ts):
        _f32c(t, f"pipe operand {i}")
    for name, t in (("ridx", ridx), ("ctr", bufs.ctr), ("wlog", wlog), ("labi", bufs.labi), ("err", err)):
        _i32c(t, name)
    n_train = x.shape[0]
    for t in perms:
        if t is not None:
            _i32c(t, "perm")
            if tuple(t.shape) != (2, n_train):
                raise ValueError("N-worker pipelined step: a permutation buffer must be [2, rows of the data set]")
    base = _lib.PipeArgs(X=x.data_ptr(), labels=labels.data_ptr(), Xa=bufs.xa.data_ptr(), labg=bufs.labg.data_ptr(),
                         ridx=ridx.data_ptr(), Wm=bufs.wm.data_ptr(), W2m=bufs.w2m.data_ptr(), Q=bufs.q.data_ptr(),
                         Z1=bufs.z1.data_ptr(), labi=bufs.labi.data_ptr(), GG=bufs.gg.data_ptr(), D1=None, P2=None,
                         ctr=bufs.ctr.data_ptr(), W1=w1.data_ptr(), b1=b1.data_ptr(), W2=w2.data_ptr(), b2=b2.data_ptr(),
                         wlog=wlog.data_ptr(), In=in_dim, Hd=hd, C=c, B=batch, R=rows_total, HP=bufs.HP, KAp=bufs.KAp,
                         ldxa=bufs.KAp, ldx=x.stride(0), ridx_len=ridx.shape[2], q=slot, loss_mode=loss_mode,
                         log_cap=wlog.shape[0], roles=roles, flush=int(flush), prime=int(prime), td=8, lr=float(lr_over_n),
                         inv_b=1.0 / batch, tail=tail)
    out = _lib.PipeMultiArgs(b=base, err=err.data_ptr(), rank=rank, world=world,
                             timeout_ticks=int(min(4.0e9, max(1.0, timeout_s * 1e8))),
                             # ranks that share one GPU (rehearsals) would all put their row workgroups on XCD 0
                             # (DTF_PIPEM_PLAIN_ORDER=1: plain block order everywhere, for comparison)
                             plain_order=1 if ((world > 1 and os.environ.get("DTF_SHARE_GPU") == "1")
                                               or os.environ.get("DTF_PIPEM_PLAIN_ORDER") == "1") else 0)
    for r in range(world):
        out.region[r] = regions[r]
        out.perm[r] = None if perms[r] is None else perms[r].data_ptr()
    out._keep = (perms, err)
    return out


def pipem_step(args):
    _lib.check(_lib.kernels().dtf_pipem_step(C.byref(args), _lib.stream_handle()), "pipem_step")


def pipem_prime(args):
    _lib.check(_lib.kernels().dtf_pipem_prime(C.byref(args), _lib.stream_handle()), "pipem_prime")


def pipem_flush(args):
    _lib.check(_lib.kernels().dtf_pipem_flush(C.byref(args), _lib.stream_handle()), "pipem_flush")


def step_tail(tail, init_only=False):
    _lib.check(_lib.kernels().dtf_step_tail(C.byref(tail), int(init_only), _lib.stream_handle()), "step_tail")


# --------------------------------------------------------------------------- optimizers / init
def sgd_apply(param, grad, lr, grad_scale=1.0):
    """``param -= lr * grad_scale * grad`` over a flat shard (K10; ApplyGradientDescent, distributed.py:89)."""
    if param.numel() != grad.numel():
        raise ValueError("sgd_apply: size mismatch")
    if not param.is_cuda:
        return reference.sgd_apply(param, grad, lr, grad_scale)
    _same_device(param, grad)
    _lib.check(_lib.kernels().dtf_sgd_apply(_f32c(param, "param").data_ptr(), _f32c(grad, "grad").data_ptr(), param.numel(),
                                            lr, grad_scale, _lib.stream_handle()), "sgd_apply")
    return param


def adam_apply(param, grad, m, v, lr, beta1=0.9, beta2=0.999, eps=1e-8, grad_scale=1.0, step=None, step_tensor=None,
               step_offset=0):
    """Fused Adam over a flat shard.  The step number is ``step`` (host int) or
    ``step_tensor[0] + step_offset`` read on the device (hipGraph-replayable)."""
    n = param.numel()
    if not (grad.numel() == m.numel() == v.numel() == n):
        raise ValueError("adam_apply: size mismatch")
    if not param.is_cuda:
        t = step if step is not None else int(step_tensor[0].item()) + step_offset
        return reference.adam_apply(param, grad, m, v, lr, beta1, beta2, eps, grad_scale, t)
    _same_device(param, grad, m, v, step_tensor)
    if step_tensor is None:
        sp, off = None, int(step)
    else:
        sp, off = _i32c(step_tensor, "step_tensor").data_ptr(), step_offset
    _lib.check(_lib.kernels().dtf_adam_apply(_f32c(param, "param").data_ptr(), _f32c(grad, "grad").data_ptr(),
                                             _f32c(m, "m").data_ptr(), _f32c(v, "v").data_ptr(), n, lr, beta1, beta2, eps,
                                             grad_scale, sp, off, _lib.stream_handle()), "adam_apply")
    return param


def optimizer_ranges(params, grads, ranges, lr, m=None, v=None, beta1=0.9, beta2=0.999, eps=1e-8, grad_scale=1.0,
                     step_tensor=None, step_offset=0):
    """SGD (``m`` and ``v`` None) or Adam over the element ranges ``[(lo, hi), ...]`` of one flat buffer, one launch
    on the GPU; the same arithmetic as ``sgd_apply`` / ``adam_apply``."""
    if not params.is_cuda or len(ranges) > _lib.MAX_OPT_RANGES or any((lo | hi) & 3 for lo, hi in ranges):
        for lo, hi in ranges:
            if m is None:
                sgd_apply(params[lo:hi], grads[lo:hi], lr, grad_scale)
            else:
                adam_apply(params[lo:hi], grads[lo:hi], m[lo:hi], v[lo:hi], lr, beta1, beta2, eps, grad_scale,
                           step_tensor=step_tensor, step_offset=step_offset)
        return params
    _same_device(params, grads, m, v, step_tensor)
    a = _lib.OptRangesArgs(p=_f32c(params, "params").data_ptr(), g=_f32c(grads, "grads").data_ptr(), m=_lib.ptr(m),
                           v=_lib.ptr(v), step_ptr=None if step_tensor is None else _i32c(step_tensor, "step").data_ptr(),
                           nranges=len(ranges), step_offset=step_offset, lr=lr, b1=beta1, b2=beta2, eps=eps,
                           gscale=grad_scale, pad_=0)
    for i, (lo, hi) in enumerate(ranges):
        if not 0 <= lo < hi <= params.numel():
            raise ValueError("optimizer_ranges: range outside the buffer")
        a.lo[i], a.hi[i] = lo, hi
    lib = _lib.kernels()
    lib.dtf_opt_ranges.restype = C.c_int
    lib.dtf_opt_ranges.argtypes = [C.POINTER(_lib.OptRangesArgs), C.c_void_p]
    _lib.check(lib.dtf_opt_ranges(C.byref(a), _lib.stream_handle()), "optimizer_ranges")
    return params


def accumulate(dst, src, alpha=1.0, beta=1.0):
    """``dst = alpha*dst + beta*src`` (gradient accumulation across rounds)."""
    if dst.numel() != src.numel():
        raise ValueError("accumulate: size mismatch")
    if not dst.is_cuda:
        dst.mul_(alpha).add_(src, alpha=beta)
        return dst
    _lib.check(_lib.kernels().dtf_accumulate(_f32c(dst, "dst").data_ptr(), _f32c(src, "src").data_ptr(), dst.numel(), alpha,
                                             beta, _lib.stream_handle()), "accumulate")
    return dst


def upload_from_pinned(dst, src):
    """``dst`` (device tensor) = ``src`` (page-locked host tensor of the same dtype and size) copied by a kernel that
    reads the host memory directly, on the current stream; the caller keeps ``src`` alive and unchanged until the
    stream has passed the copy."""
    if not dst.is_cuda or src.is_cuda or not src.is_pinned():
        raise ValueError("upload_from_pinned: dst on the GPU, src in page-locked host memory")
    if dst.dtype != src.dtype or dst.numel() != src.numel() or dst.element_size() != 4:
        raise ValueError("upload_from_pinned: same 4-byte dtype and element count on both sides")
    if not dst.is_contiguous() or not src.is_contiguous():
        raise ValueError("upload_from_pinned: contiguous tensors")
    _lib.check(_lib.kernels().dtf_upload_words(dst.data_ptr(), src.data_ptr(), dst.numel(), _lib.stream_handle()),
               "upload_from_pinned")
    return dst


def h2d_verified(dst, src, what: str = "tensor", attempts: int = 3):
    """``dst`` (tensor on the GPU, any view) = ``src`` (host tensor of the same shape; converted to ``dst``'s dtype),
    checked by reading it back, and written again when the device does not hold what was sent.

    Set-up paths only (dataset, restored or injected parameters): it waits for the device.  The data travels by
    ``upload_from_pinned`` (a kernel reading page-locked host memory) when ``dst`` is contiguous 4-byte data and the
    kernel library is there, otherwise by PyTorch's copy.  Why the read-back: it is what showed that device memory
    which had been an exported uncached region (``csrc/kernels/p2p_allreduce.hip``), was freed and then handed out
    again by the allocator can hold 128-byte lines that stay zero whatever is written to them; those regions are
    pooled now, and this check stays as the tripwire for anything of that kind."""
    if not dst.is_cuda:
        dst.copy_(src)
        return dst
    host = src.detach().to(dtype=dst.dtype, device="cpu").contiguous()
    if tuple(host.shape) != tuple(dst.shape):
        raise ValueError(f"h2d_verified: {what}: shape {tuple(host.shape)} != {tuple(dst.shape)}")
    by_kernel = dst.is_contiguous() and dst.element_size() == 4 and dst.numel() > 0 and _lib.kernels_available()
    staged = host.pin_memory() if by_kernel else host
    as_bits = (lambda t: t.view(torch.int32)) if dst.element_size() == 4 and dst.is_contiguous() else (lambda t: t)
    with torch.cuda.device(dst.device):
        for attempt in range(attempts):
            if by_kernel:
                upload_from_pinned(dst, staged)
            else:
                dst.copy_(staged)
            torch.cuda.current_stream().synchronize()
            back = dst.detach().cpu()
            if torch.equal(as_bits(back), as_bits(host)):
                return dst
            diff = (as_bits(back) != as_bits(host)).reshape(-1)
            where = diff.nonzero().flatten()
            sample = []
            if dst.element_size() == 4 and dst.is_contiguous():
                hb, db = as_bits(host).reshape(-1), as_bits(back).reshape(-1)
                sample = [f"[{int(i)}] sent {int(hb[i]) & 0xffffffff:08x} holds {int(db[i]) & 0xffffffff:08x}" for i in where[:6]]
            idx = where.tolist()
            runs, start, prev = [], idx[0], idx[0]
            for i in idx[1:] + [None]:
                if i is None or i != prev + 1:
                    runs.append((start, prev))
                    start = i
                prev = i
            sample.append("runs " + " ".join(f"{a}-{b}" for a, b in runs[:24]) + (" ..." if len(runs) > 24 else ""))
            sys.stderr.write(f"[h2d] {what}: {int(diff.sum())} of {host.numel()} elements at device address "
                             f"{dst.data_ptr():#x} differ from what was uploaded (attempt {attempt + 1} of {attempts}; "
                             f"first at element {int(where[0])}, last at {int(where[-1])}; {'; '.join(sample)})\n")
    raise RuntimeError(f"h2d_verified: {what}: the device does not hold the uploaded data after {attempts} attempts")


def truncated_normal_(out, stddev, seed, stream_id=0):
    """Fill ``out`` with N(0, stddev) truncated at two standard deviations (distributed.py:67-72)."""
    if not out.is_cuda:
        g = torch.Generator(device="cpu")
        g.manual_seed((int(seed) * 1000003 + int(stream_id)) % (2 ** 63))
        out.copy_(reference.truncated_normal(tuple(out.shape), stddev, g))
        return out
    _lib.check(_lib.kernels().dtf_truncated_normal(_f32c(out, "out").data_ptr(), out.numel(), stddev, int(seed),
                                                   int(stream_id), _lib.stream_handle()), "truncated_normal")
    return out


# --------------------------------------------------------------------------- tiled GEMM (wide layers)
EPI_FWD, EPI_BWD, EPI_WGRAD = 0, 1, 2


def _dtype_code(t):
    if t.dtype == torch.float32:
        return 0
    if t.dtype == torch.bfloat16:
        return 1
    raise ValueError(f"tiled GEMM supports float32 and bfloat16, got {t.dtype}")


class FoldedOptimizer:
    """What the weight-gradient epilogue needs to apply the optimizer itself (one worker, no aggregation):
    master weights ``p`` (and Adam moments ``m``, ``v``; both None = SGD) shaped like the gradient, the compute-dtype
    copy ``w`` to refresh (and its transpose ``wt`` [N, >=M], the forward product's operand), the device step counter
    (``step_tensor[0] + step_offset`` = number of this update)."""

    def __init__(self, p, m=None, v=None, w=None, lr=0.01, beta1=0.9, beta2=0.999, eps=1e-8, step_tensor=None,
                 step_offset=0, wt=None):
        self.p, self.m, self.v, self.w, self.wt = p, m, v, w, wt
        self.lr, self.beta1, self.beta2, self.eps = lr, beta1, beta2, eps
        self.step_tensor, self.step_offset = step_tensor, step_offset


def gemm_pingpong_takes(M: int, N: int, K: int) -> bool:
    """Shapes the 8-wave kernel accepts (multiples of 8, K >= 256); mirrors ``dtf_gemm_pingpong_takes``."""
    return M % 8 == 0 and N % 8 == 0 and K % 8 == 0 and K >= 256


class SplitK:
    """Workspace of a split-K weight-gradient product: ``splits`` partial results [splits, M, N] fp32 (and partial
    bias gradients [splits, N]), summed in split order by a second launch."""

    def __init__(self, splits: int, m: int, n: int, device):
        self.splits = int(splits)
        self.ws = torch.empty((self.splits, m, n), dtype=torch.float32, device=device)
        self.bias_ws = torch.empty((self.splits, n), dtype=torch.float32, device=device)


def split_k_for(M: int, N: int, K: int) -> int:
    """How many K ranges a weight-gradient product should be cut into so that its 256x256 tiles fill the 256 CUs
    (1 = leave it alone): 784 x 4096 is 64 tiles -> 4."""
    tiles = -(-M // 256) * -(-N // 256)
    if not gemm_pingpong_takes(M, N, K) or tiles >= 128:
        return 1
    s = min(16, 256 // tiles)
    while s > 1 and K // s < 1024:
        s -= 1
    return s


def gemm_nt(a, b, out, epilogue=EPI_FWD, bias=None, relu=False, mask=None, out_t=None, out_f32=None, bias_out=None,
            m=None, mask_t=None, opt=None, split_k=None):
    """``out[:m] = epilogue(a[:m] @ b^T)`` with ``a`` [M,K] and ``b`` [N,K], both k-contiguous.

    forward: ``+bias``, ReLU, optional transposed copy ``out_t`` [N,M] and fp32 copy ``out_f32``;
    backward: zero where ``mask <= 0`` (``mask_t``: the same mask transposed, [N, >=M], read with wide loads
    when given), optional ``out_t``;
    weight gradient: ``out`` is fp32, ``bias_out[n] = sum_k b[n,k]``; with ``opt`` (a ``FoldedOptimizer``) the
    gradient is not stored: the epilogue updates ``opt.p`` (and moments) and refreshes ``opt.w`` (bf16, 8-wave
    kernel only; the call fails if the shape is outside that kernel's range); with ``split_k`` (a ``SplitK``) the
    product runs as that many K ranges per tile and a summing pass (bf16, 8-wave kernel, ``out`` contiguous).
    """
    M = m if m is not None else a.shape[0]
    N, K = b.shape
    if a.shape[1] != K:
        raise ValueError(f"gemm_nt: a is [*, {a.shape[1]}], b is [*, {K}]")
    if a.dtype != b.dtype:
        raise ValueError("gemm_nt: operand dtypes differ")
    if not a.is_cuda:
        c = a[:M].float() @ b.float().t()
        if epilogue == EPI_FWD:
            if bias is not None:
                c = c + bias
            if relu:
                c = torch.relu(c)
        elif epilogue == EPI_BWD:
            c = torch.where(mask[:M].float() > 0, c, torch.zeros_like(c))
        else:
            if bias_out is not None:
                bias_out.copy_(b.float().sum(1))
            if opt is not None:
                g = c.contiguous()
                if opt.m is None:
                    sgd_apply(opt.p, g, opt.lr)
                else:
                    adam_apply(opt.p, g, opt.m, opt.v, opt.lr, opt.beta1, opt.beta2, opt.eps, 1.0,
                               step_tensor=opt.step_tensor, step_offset=opt.step_offset)
                if opt.w is not None:
                    opt.w.copy_(opt.p.to(opt.w.dtype))
                if opt.wt is not None:
                    opt.wt[:, :M].copy_(opt.p.t().to(opt.wt.dtype))
                return out
        out[:M].copy_(c.to(out.dtype))
        if out_t is not None:
            out_t[:, :M].copy_(c.t().to(out_t.dtype))
        if out_f32 is not None:
            out_f32[:M].copy_(c)
        return out
    _same_device(a, b, out, bias, mask, out_t, out_f32, bias_out)
    want_out = torch.float32 if epilogue == EPI_WGRAD else a.dtype
    if out.dtype != want_out or out.shape[0] < M or out.shape[1] != N:
        raise ValueError(f"gemm_nt: out must be {want_out} [>={M},{N}], got {out.dtype} {tuple(out.shape)}")
    for name, t in (("a", a), ("b", b), ("out", out)):
        if t.stride(-1) != 1:
            raise ValueError(f"gemm_nt: {name} must be row-major")
    if out_t is not None and (out_t.dtype != a.dtype or out_t.shape[0] != N or out_t.shape[1] < M):
        raise ValueError("gemm_nt: out_t must be [N, >=M] of the operand dtype")
    if out_f32 is not None and (out_f32.dtype != torch.float32 or out_f32.shape[0] < M or out_f32.shape[1] != N):
        raise ValueError("gemm_nt: out_f32 shape/dtype mismatch")
    if mask is not None and (mask.dtype != a.dtype or mask.shape[0] < M or mask.shape[1] != N):
        raise ValueError("gemm_nt: mask must be [>=M, N] of the operand dtype")
    if mask_t is not None and (mask is None or mask_t.dtype != a.dtype or mask_t.shape[0] != N or mask_t.shape[1] < M
                               or mask_t.stride(1) != 1):
        raise ValueError("gemm_nt: mask_t must be the transposed mask [N, >=M] of the operand dtype")
    if bias is not None and (bias.dtype != torch.float32 or bias.numel() != N):
        raise ValueError("gemm_nt: bias must be float32 [N]")
    if bias_out is not None and (bias_out.dtype != torch.float32 or bias_out.numel() != N):
        raise ValueError("gemm_nt: bias_out must be float32 [N]")
    args = _lib.GemmArgs(A=a.data_ptr(), B=b.data_ptr(), C=out.data_ptr(), Ct=_lib.ptr(out_t), C32=_lib.ptr(out_f32),
                         bias=_lib.ptr(bias), mask=_lib.ptr(mask), bias_out=_lib.ptr(bias_out), M=M, N=N, K=K,
                         lda=a.stride(0), ldb=b.stride(0), ldc=out.stride(0), ldct=0 if out_t is None else out_t.stride(0),
                         ldc32=0 if out_f32 is None else out_f32.stride(0), ldmask=0 if mask is None else mask.stride(0),
                         relu=int(bool(relu)), epilogue=epilogue, dtype=_dtype_code(a), a_bytes=_lib.span_bytes(a),
                         b_bytes=_lib.span_bytes(b), maskT=_lib.ptr(mask_t), ldmaskt=0 if mask_t is None else mask_t.stride(0),
                         pad_=0)
    if split_k is not None and split_k.splits > 1:
        if epilogue != EPI_WGRAD or opt is not None or a.dtype != torch.bfloat16 or M != a.shape[0]:
            raise ValueError("gemm_nt: split-K belongs to a plain bf16 weight-gradient product")
        if tuple(split_k.ws.shape) != (split_k.splits, M, N) or tuple(split_k.bias_ws.shape) != (split_k.splits, N) \
                or out.stride(0) != N:
            raise ValueError("gemm_nt: split-K workspace must be [splits, M, N] / [splits, N] and out contiguous")
        _same_device(a, split_k.ws, split_k.bias_ws)
        args.ksplit, args.split_ws, args.split_bias_ws = split_k.splits, split_k.ws.data_ptr(), split_k.bias_ws.data_ptr()
    if opt is not None:
        if epilogue != EPI_WGRAD or a.dtype != torch.bfloat16 or M != a.shape[0]:
            raise ValueError("gemm_nt: the folded optimizer belongs to a bf16 weight-gradient product")
        for name, t in (("p", opt.p), ("m", opt.m), ("v", opt.v)):
            if t is not None and (t.dtype != torch.float32 or tuple(t.shape) != (M, N) or t.stride(0) != out.stride(0)
                                  or t.stride(1) != 1):
                raise ValueError(f"gemm_nt: opt.{name} must be float32 [{M},{N}] laid out like the gradient")
        if (opt.m is None) != (opt.v is None):
            raise ValueError("gemm_nt: Adam needs both moments")
        if opt.w is not None and (opt.w.dtype != a.dtype or tuple(opt.w.shape) != (M, N) or opt.w.stride(0) != out.stride(0)):
            raise ValueError("gemm_nt: opt.w must be the compute-dtype copy, laid out like the gradient")
        if opt.m is not None and opt.step_tensor is None:
            raise ValueError("gemm_nt: Adam reads its step number from a device counter")
        if opt.wt is not None and (opt.wt.dtype != a.dtype or opt.wt.shape[0] != N or opt.wt.shape[1] < M
                                   or opt.wt.stride(1) != 1):
            raise ValueError("gemm_nt: opt.wt must be the transposed compute-dtype copy [N, >=M]")
        _same_device(a, opt.p, opt.m, opt.v, opt.w, opt.wt, opt.step_tensor)
        args.opt_p, args.opt_m, args.opt_v = opt.p.data_ptr(), _lib.ptr(opt.m), _lib.ptr(opt.v)
        args.opt_w, args.opt_wt = _lib.ptr(opt.w), _lib.ptr(opt.wt)
        args.ldwt = 0 if opt.wt is None else opt.wt.stride(0)
        args.opt_step = None if opt.step_tensor is None else _i32c(opt.step_tensor, "step_tensor").data_ptr()
        args.opt_step_offset = int(opt.step_offset)
        args.opt_lr, args.opt_b1, args.opt_b2, args.opt_eps = float(opt.lr), float(opt.beta1), float(opt.beta2), float(opt.eps)
    _lib.check(_lib.kernels().dtf_gemm_tiled(C.byref(args), _lib.stream_handle()), "gemm_nt")
    return out


def transpose16(src, dst_t):
    """``dst_t[c, r] = src[r, c]`` for 2-byte elements (bf16): the transposed compute copy of weights."""
    M, N = src.shape
    if src.dtype != dst_t.dtype or src.element_size() != 2 or tuple(dst_t.shape[:1]) != (N,) or dst_t.shape[1] < M:
        raise ValueError("transpose16: dst_t must be [N, >=M] of the same 2-byte dtype")
    if not src.is_cuda:
        dst_t[:, :M].copy_(src.t())
        return dst_t
    _same_device(src, dst_t)
    if src.stride(1) != 1 or dst_t.stride(1) != 1:
        raise ValueError("transpose16: operands must be row-major")
    lib = _lib.kernels()
    lib.dtf_transpose16.restype = C.c_int
    lib.dtf_transpose16.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
    _lib.check(lib.dtf_transpose16(src.data_ptr(), dst_t.data_ptr(), M, N, src.stride(0), dst_t.stride(0), _lib.stream_handle()),
               "transpose16")
    return dst_t


def gemm_set_modes(big_tile: int = -1, pingpong: int = -1) -> None:
    """Kernel selection of ``gemm_nt`` (tests, benchmarks): 0 = never, 1 = whenever the shape allows,
    2 = by problem size, -1 = unchanged.  ``big_tile``: 256x256 blocks with 4 waves; ``pingpong``: the 8-wave
    256x256 kernel with direct-to-LDS staging (``csrc/kernels/gemm_pingpong.hip``)."""
    lib = _lib.kernels()
    lib.dtf_gemm_set_modes.restype = None
    lib.dtf_gemm_set_modes.argtypes = [_lib.C.c_int, _lib.C.c_int]
    lib.dtf_gemm_set_modes(big_tile, pingpong)


def gather_cast_transpose(src, rows, dst=None, dst_t=None, m=None):
    """``dst[i] = src[rows[i]]`` (cast to dst's dtype) and ``dst_t = dst^T``; the batch gather of the wide pipeline."""
    M = m if m is not None else (rows.numel() if rows is not None else src.shape[0])
    N = src.shape[1]
    ref = dst if dst is not None else dst_t
    if not src.is_cuda:
        x = src.index_select(0, rows[:M].long()) if rows is not None else src[:M]
        if dst is not None:
            dst[:M].copy_(x.to(dst.dtype))
        if dst_t is not None:
            dst_t[:, :M].copy_(x.t().to(dst_t.dtype))
        return ref
    if dst is not None and dst_t is not None and dst.dtype != dst_t.dtype:
        raise ValueError("gather_cast_transpose: dst and dst_t dtypes differ")
    _lib.check(_lib.kernels().dtf_gather_cast_transpose(
        _f32c(src, "src").data_ptr(), _lib.ptr(None if rows is None else _i32c(rows, "rows")), _lib.ptr(dst), _lib.ptr(dst_t),
        M, N, src.stride(0), 0 if dst is None else dst.stride(0), 0 if dst_t is None else dst_t.stride(0), _dtype_code(ref),
        _lib.stream_handle()), "gather_cast_transpose")
    return ref


_conv1x1_ws = {}


def conv1x1_wgrad(x2d, dy2d, out=None):
    """Weight gradient of a stride-1 1x1 convolution on channels-last activations (``csrc/kernels/conv1x1.hip``):
    ``dW[co, ci] = sum_m dy2d[m, co] * x2d[m, ci]`` with ``x2d`` = the input as ``[N*H*W, Cin]`` and ``dy2d`` = the output
    gradient as ``[N*H*W, Cout]``, both bf16, contiguous.  Returns ``[Cout, Cin]`` bf16 (fp32 accumulation; the
    reduction is split over workgroups and the partial sums are added in a fixed order).  CPU tensors: the PyTorch
    statement."""
    if x2d.dim() != 2 or dy2d.dim() != 2 or x2d.shape[0] != dy2d.shape[0]:
        raise ValueError("conv1x1_wgrad: x [M, Cin] and dy [M, Cout] with the same M")
    M, cin = x2d.shape
    cout = dy2d.shape[1]
    if not x2d.is_cuda:
        dw = (dy2d.float().t() @ x2d.float()).to(torch.bfloat16)
        if out is not None:
            out.copy_(dw)
            return out
        return dw
    if x2d.dtype != torch.bfloat16 or dy2d.dtype != torch.bfloat16 or not x2d.is_contiguous() or not dy2d.is_contiguous():
        raise ValueError("conv1x1_wgrad: contiguous bf16 operands")
    if cin % 8 or cout % 8:
        raise ValueError("conv1x1_wgrad: channel counts must be multiples of 8")
    lib = _lib.kernels()
    if not getattr(lib, "_conv1x1_declared", False):
        lib.dtf_conv1x1_wgrad_ws_floats.restype = C.c_longlong
        lib.dtf_conv1x1_wgrad_ws_floats.argtypes = [C.c_longlong, C.c_int, C.c_int]
        lib.dtf_conv1x1_wgrad.restype = C.c_int
        lib.dtf_conv1x1_wgrad.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_longlong, C.c_int, C.c_int, C.c_void_p]
        lib._conv1x1_declared = True
    need = int(lib.dtf_conv1x1_wgrad_ws_floats(M, cin, cout))
    key = (x2d.device, torch.cuda.current_stream(x2d.device).cuda_stream)
    ws = _conv1x1_ws.get(key)
    if ws is None or ws.numel() < need:
        ws = _conv1x1_ws[key] = torch.empty(max(need, 1 << 22), dtype=torch.float32, device=x2d.device)   # one per stream
    if out is None:
        out = torch.empty((cout, cin), dtype=torch.bfloat16, device=x2d.device)
    elif out.dtype != torch.bfloat16 or not out.is_contiguous() or out.numel() != cout * cin:
        raise ValueError("conv1x1_wgrad: out must be contiguous bf16 with Cout * Cin elements")
    _lib.check(lib.dtf_conv1x1_wgrad(x2d.data_ptr(), dy2d.data_ptr(), ws.data_ptr(), out.data_ptr(), M, cin, cout,
                                     _lib.stream_handle()), "conv1x1_wgrad")
    return out
