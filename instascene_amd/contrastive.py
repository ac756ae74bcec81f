"""``contrastive_loss`` — host-side mirror of the reference's
``utils/contrastive_utils.contrastive_loss`` (utils/contrastive_utils.py:18-73): same signature and
semantics, with the whole computation — label filtering, prototypes, concentration, similarity, loss
and backward — inside the HIP library (``iso_contrastive_forward/backward``: exact-fp32 MFMA products,
atomic-free deterministic reductions, no device->host synchronisation; the reference syncs three times
per call through ``torch.unique``)."""
from __future__ import annotations

import ctypes

import torch

from . import _hot
from ._lib import check, lib


def _p(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def _stream():
    return _hot.stream_ptr()


class _ProtoNCE(torch.autograd.Function):
    @staticmethod
    def forward(ctx, features, labels, predef_u, K, temp_lambda, consider_negative, min_pixnum):
        L = lib()
        feats = features.contiguous().float()
        N, F = feats.shape
        if labels.dtype not in (torch.int64, torch.int32):
            labels = labels.to(torch.int64)
        labels = labels.contiguous()
        pre = predef_u.contiguous().float() if predef_u is not None else None
        nbytes = L.iso_contrastive_scratch_bytes(N, F, K)
        state = torch.empty(nbytes, dtype=torch.uint8, device=feats.device)
        loss = torch.empty(1, dtype=torch.float32, device=feats.device)
        with _hot.on_device(feats.device):
            check(L.iso_contrastive_forward(N, F, K, _p(feats), _p(labels), int(labels.dtype == torch.int64), _p(pre),
                                            int(bool(consider_negative)), int(min_pixnum), float(temp_lambda), _p(loss),
                                            _p(state), nbytes, _stream()), "iso_contrastive_forward")
        ctx.save_for_backward(state)
        ctx.dims = (N, F, K, nbytes, pre is not None)
        return loss[0]

    @staticmethod
    def backward(ctx, grad_loss):
        L = lib()
        (state,) = ctx.saved_tensors
        N, F, K, nbytes, has_pre = ctx.dims
        g = grad_loss.reshape(1).contiguous().float()
        out = torch.empty((N, F), dtype=torch.float32, device=state.device)
        with _hot.on_device(state.device):
            check(L.iso_contrastive_backward(N, F, K, int(has_pre), _p(g), _p(out), _p(state), nbytes, _stream()),
                  "iso_contrastive_backward")
        return out, None, None, None, None, None, None


def contrastive_loss(features, masks, predef_u_list=None, min_pixnum=0, temp_lambda=1000, consider_negative=False,
                     num_labels=None):
    """Drop-in for ``utils.contrastive_utils.contrastive_loss`` (reference :18-73).

    features [N,F] float (CUDA), masks [N] integer labels; returns the summed ProtoNCE loss.
    ``num_labels`` (extension, optional): an upper bound on ``masks.max() + 1``.  When neither it nor
    ``predef_u_list`` is given the bound is read from ``masks`` (one host sync, still fewer than the reference)."""
    if not features.is_cuda:
        raise RuntimeError("contrastive_loss: features must be a CUDA tensor (the HIP library is the only backend)")
    if features.shape[0] == 0:
        return features.sum() * 0.0
    if predef_u_list is not None:
        K = int(predef_u_list.shape[0])
    elif num_labels is not None:
        K = int(num_labels)
    else:
        K = int(masks.max().item()) + 1
    K = max(K, 1)
    return _ProtoNCE.apply(features, masks, predef_u_list, K, float(temp_lambda), bool(consider_negative), int(min_pixnum))


class _ProtoNCEBatch(torch.autograd.Function):
    """``sum_b weights[b] * contrastive_loss(features[b], labels[b], predefs[b])`` for up to four problems of the same
    shape, evaluated by ONE sequence of launches (``iso_contrastive_forward_batch``)."""

    @staticmethod
    def forward(ctx, K, temp_lambda, consider_negative, min_pixnum, weights, labels, predefs, stacked, *features):
        L = lib()
        # ``stacked`` = k > 1: features[0] is [k N, F], the first k problems stacked (a trainer's sampled rows of one render);
        # its gradient comes back as ONE tensor instead of k that autograd would have to concatenate
        first = features[0].contiguous().float()
        feats = ([first[j * (first.shape[0] // stacked):(j + 1) * (first.shape[0] // stacked)] for j in range(stacked)]
                 if stacked > 1 else [first]) + [f.contiguous().float() for f in features[1:]]
        nb = len(feats)
        N, F = feats[0].shape
        labs = [l.contiguous() if l.dtype == torch.int64 else l.to(torch.int64).contiguous() for l in labels]
        pres = [p.contiguous().float() if p is not None else None for p in predefs]
        dev = feats[0].device
        one = L.iso_contrastive_scratch_bytes(N, F, K)
        state = torch.empty(one * nb, dtype=torch.uint8, device=dev)
        loss = torch.empty(nb + 1, dtype=torch.float32, device=dev)
        ptrs = lambda ts: (ctypes.c_void_p * nb)(*[None if t is None else t.data_ptr() for t in ts])
        w = (ctypes.c_float * nb)(*[float(x) for x in weights])
        with _hot.on_device(dev):
            check(L.iso_contrastive_forward_batch(nb, N, F, K, ptrs(feats), ptrs(labs), 1, ptrs(pres),
                                                  int(bool(consider_negative)), int(min_pixnum), float(temp_lambda), w,
                                                  _p(loss), ctypes.c_void_p(loss.data_ptr() + 4 * nb), _p(state), one * nb,
                                                  _stream()), "iso_contrastive_forward_batch")
        ctx.save_for_backward(state)
        ctx.dims = (nb, N, F, K, one * nb, [p is not None for p in pres], [float(x) for x in weights], int(stacked))
        ctx.mark_non_differentiable(loss)
        ctx.set_materialize_grads(False)      # (no zero-filled gradient for the non-differentiable second output)
        return loss[nb], loss

    @staticmethod
    def backward(ctx, grad_total, _grad_parts):
        L = lib()
        (state,) = ctx.saved_tensors
        nb, N, F, K, nbytes, has_pre, weights, stacked = ctx.dims
        n_in = nb - (stacked - 1 if stacked > 1 else 0)
        if grad_total is None:
            return (None,) * (8 + n_in)
        g = grad_total.reshape(1).contiguous().float()
        if stacked > 1:
            head = torch.empty((stacked * N, F), dtype=torch.float32, device=state.device)
            parts = [head[j * N:(j + 1) * N] for j in range(stacked)]
            rest = [torch.empty((N, F), dtype=torch.float32, device=state.device) for _ in range(nb - stacked)]
            outs, ret = parts + rest, [head] + rest
        else:
            outs = [torch.empty((N, F), dtype=torch.float32, device=state.device) for _ in range(nb)]
            ret = outs
        flags = (ctypes.c_int * nb)(*[int(h) for h in has_pre])
        w = (ctypes.c_float * nb)(*weights)
        optr = (ctypes.c_void_p * nb)(*[o.data_ptr() for o in outs])
        with _hot.on_device(state.device):
            check(L.iso_contrastive_backward_batch(nb, N, F, K, flags, _p(g), w, optr, _p(state), nbytes, _stream()),
                  "iso_contrastive_backward_batch")
        return (None,) * 8 + tuple(ret)


def contrastive_loss_batch(features, masks, predef_u_lists, weights, num_labels, min_pixnum=0, temp_lambda=1000,
                           consider_negative=False, stacked: int = 0):
    """``sum_b weights[b] * contrastive_loss(features[b], masks[b], predef_u_lists[b], num_labels=num_labels)`` (weighted
    losses added in order) for 1..4 problems whose ``[N,F]`` shapes agree and whose prototypes, where predefined, have
    ``num_labels`` rows; one sequence of launches instead of one per loss.  Returns ``(total, weighted_parts[nb+1])``."""
    stacked = int(stacked) if stacked and int(stacked) > 1 else 0
    nb = len(masks)
    if not (1 <= nb <= 4) or len(features) != nb - max(0, stacked - 1) or len(predef_u_lists) != nb or len(weights) != nb:
        raise ValueError("contrastive_loss_batch: 1..4 problems, one mask / prototype entry / weight each")
    if stacked and (features[0].dim() != 2 or features[0].shape[0] % stacked):
        raise ValueError("contrastive_loss_batch: features[0] must stack `stacked` problems of equal size")
    # (``stacked`` = k: features[0] is [k N, F] and holds the first k problems one after the other)
    shape = (features[0].shape[0] // stacked, features[0].shape[1]) if stacked else tuple(features[0].shape)
    K = max(int(num_labels), 1)
    for j, f in enumerate(features):
        if not f.is_cuda:
            raise RuntimeError("contrastive_loss_batch: features must be CUDA tensors (the HIP library is the only backend)")
        if (tuple(f.shape) != shape and not (stacked and j == 0)) or shape[0] == 0:
            raise ValueError("contrastive_loss_batch: all problems need the same non-empty [N,F] shape")
    for p in predef_u_lists:
        if p is not None and int(p.shape[0]) != K:
            raise ValueError("contrastive_loss_batch: predefined prototypes must have num_labels rows")
    return _ProtoNCEBatch.apply(K, float(temp_lambda), bool(consider_negative), int(min_pixnum), tuple(weights),
                                tuple(masks), tuple(predef_u_lists), stacked, *features)


class _RowNorm(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, eps):
        L = lib()
        xc = x.contiguous().float()
        out = torch.empty_like(xc)
        with _hot.on_device(xc.device):
            check(L.iso_rownorm(xc.shape[0], xc.shape[1], float(eps), 0, _p(xc), None, _p(out), _stream()), "iso_rownorm")
        ctx.save_for_backward(xc)
        ctx.eps = float(eps)
        return out

    @staticmethod
    def backward(ctx, dy):
        L = lib()
        (xc,) = ctx.saved_tensors
        dyc = dy.contiguous().float()
        out = torch.empty_like(xc)
        with _hot.on_device(xc.device):
            check(L.iso_rownorm(xc.shape[0], xc.shape[1], ctx.eps, 1, _p(xc), _p(dyc), _p(out), _stream()), "iso_rownorm")
        return out, None


class _RowNorm2(torch.autograd.Function):
    """y = rownorm(x, eps1), z = rownorm(y, eps2) in one pass each way (``iso_rownorm2``)."""

    @staticmethod
    def forward(ctx, x, eps1, eps2):
        L = lib()
        xc = x.contiguous().float()
        y, z = torch.empty_like(xc), torch.empty_like(xc)
        with _hot.on_device(xc.device):
            check(L.iso_rownorm2(xc.shape[0], xc.shape[1], float(eps1), float(eps2), 0, _p(xc), None, None, _p(y), _p(z),
                                 _stream()), "iso_rownorm2")
        ctx.save_for_backward(xc)
        ctx.eps = (float(eps1), float(eps2))
        ctx.set_materialize_grads(False)
        return y, z

    @staticmethod
    def backward(ctx, gy, gz):
        (xc,) = ctx.saved_tensors
        if gy is None and gz is None:
            return None, None, None
        L = lib()
        gy = None if gy is None else gy.contiguous().float()
        gz = None if gz is None else gz.contiguous().float()
        out = torch.empty_like(xc)
        with _hot.on_device(xc.device):
            check(L.iso_rownorm2(xc.shape[0], xc.shape[1], ctx.eps[0], ctx.eps[1], 1, _p(xc), _p(gy), _p(gz), _p(out), None,
                                 _stream()), "iso_rownorm2")
        return out, None, None


class _RowNorm2Given(torch.autograd.Function):
    """``y, z`` of :class:`_RowNorm2` when they were already produced (by the fused optimiser pass): the forward only
    attaches them to the graph, the backward is the same single kernel."""

    @staticmethod
    def forward(ctx, x, y, z, eps1, eps2):
        ctx.save_for_backward(x.detach())
        ctx.eps = (float(eps1), float(eps2))
        ctx.set_materialize_grads(False)
        return y.view_as(y), z.view_as(z)

    @staticmethod
    def backward(ctx, gy, gz):
        g = _RowNorm2.backward(ctx, gy, gz)
        return g[0], None, None, None, None


class _GatherRows(torch.autograd.Function):
    """``table[idx]``; inside a ``rasterizer.DeferredFeatureRows`` block the first backward keeps its gradient sparse
    (``sink.row_grads``) instead of building a dense ``[P,F]`` tensor by sort + scatter."""

    @staticmethod
    def forward(ctx, table, idx):
        ctx.save_for_backward(idx)
        ctx.shape = table.shape
        src = getattr(table, _ROW_SOURCE_ATTR, None)
        ctx.placeholder = src is not None
        if src is None:
            return table.index_select(0, idx)
        # the table is a placeholder for normalize(x) that was never stored (FeatureAdam.store_y = False): gather the rows
        # from x and normalise them on the way (iso_gather_rownorm; the bits of the stored table's rows)
        x, eps = src
        ix = idx.contiguous().to(torch.int64)
        out = torch.empty((ix.shape[0], x.shape[1]), dtype=torch.float32, device=x.device)
        with _hot.on_device(x.device):
            check(lib().iso_gather_rownorm(ix.shape[0], x.shape[1], x.shape[0], float(eps), _p(x), _p(ix), _p(out), _stream()),
                  "iso_gather_rownorm")
        return out

    @staticmethod
    def backward(ctx, g):
        from . import rasterizer as _rz
        (idx,) = ctx.saved_tensors
        sink = _rz._ROWS_SINK
        n = idx.shape[0]
        if sink is not None and sink.row_grads is None and g.is_cuda:
            # merged per distinct row right away (iso_rows_compact), early in the backward: by the time the per-Gaussian
            # tail runs the table is ready and the small kernel does not queue up behind the next view's binning
            sink.row_grads = compact_row_grads(idx, g, ctx.shape[0])
            return None, None
        if ctx.placeholder:
            raise RuntimeError("gather_rows: the table is a placeholder (FeatureAdam.store_y = False); its gradient can only "
                               "be collected by a rasterizer.DeferredFeatureRows block, once per step")
        dense = torch.zeros(ctx.shape, dtype=g.dtype, device=g.device)
        dense.index_put_((idx,), g, accumulate=True)
        return dense, None


_ROW_SOURCE_ATTR = "_isr_row_source"


class _SlotTable:
    """A persistent ``slot[P]`` table (int32) per (device, P, stream).  ``isr_feature_rows_step`` resets every entry it reads,
    so after a pass over all P rows the table is all -1 again and the next ``iso_rows_compact`` needs no 4 P-byte fill (6 MB, ~20 us
    on the step's main chain at C3).  ``dirty``: filled and not yet walked completely - the next fill then clears it the slow way."""
    __slots__ = ("slot", "dirty", "covered")

    def __init__(self, P, device):
        self.slot = torch.full((P,), -1, dtype=torch.int32, device=device)
        self.dirty, self.covered = False, 0


_SLOT_TABLES = {}
ROWS_COMPACT_MAX = 65536         # iso_rows_compact's limit (csrc/iso_contrastive.hip: ROWS_SPLIT_MAX)


def _slot_table(P: int, device) -> _SlotTable:
    key = (device.index, int(P), _hot.raw_stream(device))
    t = _SLOT_TABLES.get(key)
    if t is None:
        if len(_SLOT_TABLES) > 8:
            _SLOT_TABLES.clear()
        t = _SLOT_TABLES[key] = _SlotTable(P, device)
    return t


def compact_row_grads(idx: torch.Tensor, vals: torch.Tensor, P: int):
    """``(slot[P] int32, merged[n,F])``: the sparse gradient ``(idx, vals)`` of a row gather with repeated rows summed in
    index order (``iso_rows_compact``); ``slot[row]`` = position of the row's sum in ``merged`` or -1.  ``slot`` is the
    persistent table of this (device, P, stream) - valid until the next call; ``isr_feature_rows_step`` consumes it."""
    idx = idx.contiguous().to(torch.int64)
    vals = vals.contiguous().float()
    table = _slot_table(P, vals.device)
    if idx.shape[0] > ROWS_COMPACT_MAX:
        # beyond the kernels' 65 536 samples (the reference's default sample_batchsize is 32 768): one entry per distinct row by
        # torch (unique + index_add_: the sum of a repeated row is in atomic order, not in index order - the only difference)
        ok = (idx >= 0) & (idx < P)
        rows, inverse = torch.unique(idx[ok], return_inverse=True)
        merged = torch.zeros((max(int(rows.shape[0]), 1), vals.shape[1]), dtype=torch.float32, device=vals.device)
        merged.index_add_(0, inverse, vals[ok])
        if table.dirty:
            table.slot.fill_(-1)
        table.slot[rows] = torch.arange(rows.shape[0], dtype=torch.int32, device=vals.device)
        table.dirty, table.covered = rows.shape[0] > 0, 0
        return table.slot, merged
    merged = torch.empty_like(vals)
    chain = torch.empty(max(1, idx.shape[0]), dtype=torch.int32, device=vals.device)       # scratch of the call
    with _hot.on_device(vals.device):
        check(lib().iso_rows_compact(idx.shape[0], vals.shape[1], P, _p(idx), _p(vals), _p(table.slot), _p(merged), _p(chain),
                                     0 if table.dirty else 1, _stream()), "iso_rows_compact")
    table.dirty, table.covered = idx.shape[0] > 0, 0
    return table.slot, merged


def _slot_consumed(slot: torch.Tensor, P: int, n_rows: int):
    """``isr_feature_rows_step`` walked ``n_rows`` more rows with this slot table (it reset what it read)."""
    for t in _SLOT_TABLES.values():
        if t.slot is slot:
            t.covered += int(n_rows)
            if t.covered >= P:
                t.dirty = False
            return


def gather_rows(table: torch.Tensor, idx: torch.Tensor) -> torch.Tensor:
    """``table[idx]`` for a ``[P,F]`` table and int64 row indices (train_semantic.py:183-190)."""
    return _GatherRows.apply(table, idx)


class FeatureAdam:
    """Adam on ONE ``[P,F]`` parameter with the arithmetic of ``torch.optim.Adam(lr, betas, eps)`` (the reference's
    optimiser for ``_seg_feature``, scene/gaussian_model.py:223-249), whose step also emits the parameter's two chained
    row normalisations (``iso_adam_rownorm2``) so that the next forward does not re-read it.  ``state_dict`` /
    ``load_state_dict`` use torch's key names."""

    def __init__(self, param: torch.nn.Parameter, lr: float, betas=(0.9, 0.999), eps: float = 1e-8, norm_eps=(1e-6, 1e-9)):
        if not param.is_cuda or param.dim() != 2 or (param.shape[1] & 3) or param.shape[1] > 256:
            raise ValueError("FeatureAdam needs a CUDA [P,F] parameter with F % 4 == 0 and F <= 256")
        self.param, self.lr, self.betas, self.eps, self.norm_eps = param, float(lr), betas, float(eps), norm_eps
        self._zero1 = None
        self.exp_avg = torch.zeros_like(param, memory_format=torch.contiguous_format)
        self.exp_avg_sq = torch.zeros_like(param, memory_format=torch.contiguous_format)
        self.step_count = 0
        self.normalized = None          # (version of param, y, z) written by the last step
        # leaf mode (``step_rows``): normalized_chain() hands out y and z as LEAVES; the caller finishes the chain rule,
        # the optimiser step and the next normalisation in one pass over the rows (isr_feature_rows_step)
        self.leaf_mode = False
        self.leaves = None
        self._slot = None
        # False: the one-pass tail does not write y = normalize(param) (one of its nine [P,F] streams); leaf mode then hands
        # out a placeholder whose rows gather_rows computes from the parameter.  For callers that read y only through
        # gather_rows (SegTrainer: the 3-D loss' 8 192 rows).
        self.store_y = True
        # False (needs store_y = False; FAST mode's per-block blend): the one-pass tail does not write z = normalize(normalize(param))
        # either - the table the next forward blends, another of its [P,F] streams - but only its two factors per row (q1, q2) [P,2];
        # leaf mode hands out an (unwritten) placeholder for z that carries (param, factors), and the rasterizer's forward passes
        # the RAW table plus the factors to the blend, which applies them to the rows it stages (isr_forward_render_scaled: the same
        # two multiplies in the same order - the same bits).  Anything else that needs z's values materialises them
        # (``materialize_scaled_rows``).
        self.store_z = True
        self._zscale = None
        self._zplace = None

    def step(self):
        p = self.param
        if p.grad is None:
            return
        L = lib()
        g = p.grad.contiguous().float()
        y, z = torch.empty_like(p.data), torch.empty_like(p.data)
        self.step_count += 1
        with _hot.on_device(p.device):
            check(L.iso_adam_rownorm2(p.shape[0], p.shape[1], self.lr, float(self.betas[0]), float(self.betas[1]), self.eps,
                                      self.step_count, float(self.norm_eps[0]), float(self.norm_eps[1]), _p(p.data), _p(g),
                                      _p(self.exp_avg), _p(self.exp_avg_sq), _p(y), _p(z), _stream()), "iso_adam_rownorm2")
        torch.autograd.graph.increment_version(p)       # the kernel wrote through the raw pointer: tell autograd
        self.normalized = (p._version, y, z)

    def step_rows(self, rows=None, grad_only: bool = False, row_grads=None, dense=None):
        """Leaf mode: finish the step from the gradients that reached the leaves of ``normalized_chain()`` — ``y.grad``
        (3-D loss), ``z.grad`` (any dense rasterizer gradient) — and the partial rows ``rows`` a
        ``rasterizer.DeferredFeatureRows`` block collected: reduction, chain rule through both normalisations, Adam and
        the next normalisations in ONE kernel (``isr_feature_rows_step``).  ``grad_only``: stop at ``param.grad`` (for an
        all-reduce; ``step()`` then completes).  ``row_grads``: sparse gradient on ``y`` rows — ``(idx int64, [n,F])``, or
        the ``(slot int32 [P], merged [n,F])`` pair ``compact_row_grads`` makes of it (``gather_rows`` does so itself)."""
        tail = self.begin_tail(rows, row_grads, dense)
        if tail is None:
            return
        P = self.param.shape[0]
        if grad_only:
            self.tail_gradient(tail, 0, P)
        else:
            self.tail_update(tail)

    # -- the tail in pieces (a data-parallel trainer walks the table in row ranges: see SegTrainer) ------------------
    def begin_tail(self, rows=None, row_grads=None, dense=None):
        """Collect what the leaves received; returns an opaque state for ``tail_gradient`` / ``tail_update`` (None when
        nothing carries a gradient).  ``dense``: a ``[P,F]`` gradient on ``z`` collected outside autograd
        (``DeferredFeatureRows(collect_dense=True).dense``)."""
        p = self.param
        if self.leaves is None:
            raise RuntimeError("step_rows: normalized_chain() was not called in leaf mode")
        y_leaf, z_leaf = self.leaves
        if y_leaf.numel() == 1 and y_leaf.grad is not None:
            raise RuntimeError("the values of the placeholder for the normalised feature were used directly; with "
                               "FeatureAdam.store_y = False only gather_rows may read it")
        gy = None if y_leaf.grad is None else y_leaf.grad.contiguous().float()
        gz = None if z_leaf.grad is None else z_leaf.grad.contiguous().float()
        if dense is not None:
            gz = dense if gz is None else gz + dense
        self.leaves = None
        if rows is None and gy is None and gz is None and row_grads is None:
            return None
        if rows is not None and (rows.P != p.shape[0] or rows.F != p.shape[1]):
            raise ValueError("step_rows: rows of a different model")
        slot = merged = None
        if row_grads is not None:
            a, b = row_grads
            # already merged (slot table + merged rows, from gather_rows' backward) or raw (indices, rows)
            slot, merged = (a, b) if a.dtype == torch.int32 else compact_row_grads(a, b, p.shape[0])
        return (rows, gy, gz, slot, merged)

    def _rows_kernel(self, tail, r0, n, grad_out, y, z, zscale=None):
        rows, gy, gz, slot, merged = tail
        p = self.param
        P, F = p.shape
        with _hot.on_device(p.device):
            check(lib().isr_feature_rows_step_scaled(P, int(r0), int(n), rows.R if rows is not None else 0, F,
                                                     _p(rows.geom) if rows is not None else None,
                                                     _p(rows.scratch) if rows is not None else None, _p(gz), _p(gy), _p(slot),
                                                     _p(merged), float(self.norm_eps[0]), float(self.norm_eps[1]), _p(p.data),
                                                     _p(grad_out), self.lr, float(self.betas[0]), float(self.betas[1]), self.eps,
                                                     max(1, self.step_count), _p(self.exp_avg), _p(self.exp_avg_sq), _p(y), _p(z),
                                                     _p(zscale), _stream()), "isr_feature_rows_step")
        if slot is not None:
            _slot_consumed(slot, P, n)

    def tail_gradient(self, tail, r0: int, r1: int):
        """``param.grad[r0:r1]`` = dL/dparam of those rows (allocated on first use)."""
        p = self.param
        if p.grad is None:
            p.grad = torch.empty_like(p.data)
        self._rows_kernel(tail, r0, r1 - r0, p.grad, None, None)

    def tail_update(self, tail):
        """Reduction + chain rule + Adam + next normalisations of every row in one pass."""
        p = self.param
        y = torch.empty_like(p.data) if self.store_y else None
        self.step_count += 1
        if self._scaled_rows_wanted():
            zs = self._scale_buffer()
            self._rows_kernel(tail, 0, p.shape[0], None, y, None, zs)
            z = _ScaledRows(zs)
        else:
            z = torch.empty_like(p.data)
            self._rows_kernel(tail, 0, p.shape[0], None, y, z)
        torch.autograd.graph.increment_version(p)
        self.normalized = (p._version, y, z)

    def _scaled_rows_wanted(self) -> bool:
        return (not self.store_z) and (not self.store_y) and self.param.is_cuda

    def _scale_buffer(self):
        p = self.param
        if self._zscale is None or self._zscale.shape[0] != p.shape[0] or self._zscale.device != p.device:
            self._zscale = torch.empty((p.shape[0], 2), dtype=torch.float32, device=p.device)
        return self._zscale

    def begin_step(self):
        """Start an Adam step applied in row ranges (``step_range``), finished by ``end_step``."""
        p = self.param
        self.step_count += 1
        self._pending_yz = (torch.empty_like(p.data) if self.store_y else None, torch.empty_like(p.data))

    def step_range(self, r0: int, r1: int, grad_rows=None):
        """Adam + the two normalisations on rows ``[r0, r1)`` from ``param.grad`` (``iso_adam_rownorm2`` on the slice), or
        from ``grad_rows [r1 - r0, F]`` when the summed gradient of those rows lives elsewhere (a reduce-scatter's shard)."""
        p = self.param
        if r1 <= r0:
            return
        F = p.shape[1]
        y, z = self._pending_yz
        at = lambda t: None if t is None else ctypes.c_void_p(t.data_ptr() + 4 * F * int(r0))
        g = at(p.grad) if grad_rows is None else _p(grad_rows.contiguous())
        with _hot.on_device(p.device):
            check(lib().iso_adam_rownorm2(int(r1 - r0), F, self.lr, float(self.betas[0]), float(self.betas[1]), self.eps,
                                          self.step_count, float(self.norm_eps[0]), float(self.norm_eps[1]), at(p.data),
                                          g, at(self.exp_avg), at(self.exp_avg_sq), at(y), at(z), _stream()),
                  "iso_adam_rownorm2")

    def renormalize_range(self, r0: int, r1: int):
        """The two normalisations of rows ``[r0, r1)`` of the CURRENT parameter into the pending outputs (``iso_rownorm2``):
        for rows whose Adam step ran on another rank and arrived by all-gather (SegTrainer's sharded tail)."""
        p = self.param
        if r1 <= r0:
            return
        F = p.shape[1]
        y, z = self._pending_yz
        at = lambda t: None if t is None else ctypes.c_void_p(t.data_ptr() + 4 * F * int(r0))
        with _hot.on_device(p.device):
            check(lib().iso_rownorm2(int(r1 - r0), F, float(self.norm_eps[0]), float(self.norm_eps[1]), 0, at(p.data), None, None,
                                     at(y), at(z), _stream()), "iso_rownorm2")

    def end_step(self):
        p = self.param
        y, z = self._pending_yz
        self._pending_yz = None
        torch.autograd.graph.increment_version(p)
        self.normalized = (p._version, y, z)

    def zero_grad(self, set_to_none: bool = True):
        if set_to_none:
            self.param.grad = None
        elif self.param.grad is not None:
            self.param.grad.zero_()

    def normalized_chain(self):
        """``row_normalize_chain(param, *norm_eps)`` — from the last step's output when the parameter is unchanged since."""
        p = self.param
        if self.leaf_mode and torch.is_grad_enabled():
            if self.normalized is None or self.normalized[0] != p._version:
                if self._scaled_rows_wanted():
                    # a table no step has touched yet: its factors alone (isr_row_scales: the tail kernel's own expressions)
                    zs = self._scale_buffer()
                    with _hot.on_device(p.device):
                        check(lib().isr_row_scales(p.shape[0], p.shape[1], float(self.norm_eps[0]), float(self.norm_eps[1]),
                                                   _p(p.data), _p(zs), _stream()), "isr_row_scales")
                    self.normalized = (p._version, None, _ScaledRows(zs))
                else:
                    with torch.no_grad():
                        y0, z0 = _RowNorm2.apply(p.detach(), *self.norm_eps)
                    self.normalized = (p._version, y0, z0)
            if isinstance(self.normalized[2], _ScaledRows):
                # z is not stored: an unwritten [P,F] placeholder (allocated once, never read) that knows how its rows are made
                if self._zplace is None or self._zplace.shape != p.shape or self._zplace.device != p.device:
                    self._zplace = torch.empty_like(p.data)
                z = self._zplace.detach().requires_grad_(True)
                setattr(z, SCALED_ROWS_ATTR, (p.detach(), self.normalized[2].scale))
            else:
                z = self.normalized[2].detach().requires_grad_(True)
            if self.normalized[1] is not None:
                y = y_leaf = self.normalized[1].detach().requires_grad_(True)
            else:
                # y was not stored (store_y = False): a zero-stride placeholder of the right shape that knows where its rows
                # come from (gather_rows reads them from the parameter); nothing else may read its values
                if self._zero1 is None or self._zero1.device != p.device or self._zero1.dtype != p.dtype:
                    self._zero1 = torch.zeros(1, dtype=p.dtype, device=p.device)
                y_leaf = self._zero1.detach().requires_grad_(True)       # (a fresh leaf without a fill kernel per step)
                y = y_leaf.expand(p.shape[0], p.shape[1])
                setattr(y, _ROW_SOURCE_ATTR, (p.detach(), float(self.norm_eps[0])))
            setattr(y, _MEMO_ATTR, (float(self.norm_eps[1]), z, y._version))
            self.leaves = (y_leaf, z)
            return y
        if self.normalized is not None and self.normalized[0] == p._version and self.normalized[1] is not None:
            y, z = _RowNorm2Given.apply(p, self.normalized[1], self.normalized[2], *self.norm_eps)
            setattr(y, _MEMO_ATTR, (float(self.norm_eps[1]), z, y._version))
            return y
        return row_normalize_chain(p, *self.norm_eps)

    def state_dict(self):
        return {"step": self.step_count, "exp_avg": self.exp_avg, "exp_avg_sq": self.exp_avg_sq, "lr": self.lr,
                "betas": self.betas, "eps": self.eps}

    def load_state_dict(self, sd):
        self.step_count = int(sd["step"])
        self.exp_avg.copy_(sd["exp_avg"])
        self.exp_avg_sq.copy_(sd["exp_avg_sq"])
        self.normalized = None


_MEMO_ATTR = "_isr_renormalized"
SCALED_ROWS_ATTR = "_isr_scaled_rows"       # on a [P,F] placeholder: (raw table x [P,F], factors [P,2]); its rows are (x q1) q2


class _ScaledRows:
    """FeatureAdam.normalized[2] when z is not stored (store_z = False): the two factors per row."""
    __slots__ = ("scale",)

    def __init__(self, scale):
        self.scale = scale


def materialize_scaled_rows(t):
    """The values of a tensor that may be a scaled-rows placeholder (FeatureAdam.store_z = False): (x q1) q2, the same two
    multiplies in the same order as the kernels that would have stored them."""
    src = getattr(t, SCALED_ROWS_ATTR, None)
    if src is None:
        return t
    x, qs = src
    return (x * qs[:, 0:1]) * qs[:, 1:2]


def row_normalize(x: torch.Tensor, eps: float) -> torch.Tensor:
    """``x / (x.norm(dim=1, keepdim=True) + eps)`` for a [N,F] CUDA tensor in one streaming HIP kernel each way
    (scene/gaussian_model.py:122-125 with eps 1e-6; gaussian_renderer/__init__.py:61-62 with eps 1e-9).
    A tensor produced by :func:`row_normalize_chain` already carries its re-normalisation and returns it."""
    memo = getattr(x, _MEMO_ATTR, None)
    if memo is not None and memo[0] == float(eps) and memo[2] == x._version:
        return memo[1]
    if not x.is_cuda or x.dim() != 2:
        return x / (x.norm(dim=-1, keepdim=True) + eps)
    return _RowNorm.apply(x, eps)


def row_normalize_chain(x: torch.Tensor, eps1: float, eps2: float) -> torch.Tensor:
    """``y = row_normalize(x, eps1)`` computed together with ``z = row_normalize(y, eps2)``: the reference normalises
    the feature in the model getter and again in ``render()``; a model getter that returns this ``y`` makes the
    second call free (it is looked up on the tensor), and the backward of both is one kernel."""
    if not x.is_cuda or x.dim() != 2 or (x.shape[1] & 3) != 0 or x.shape[1] > 256:
        return row_normalize(x, eps1)
    y, z = _RowNorm2.apply(x, eps1, eps2)
    setattr(y, _MEMO_ATTR, (float(eps2), z, y._version))
    return y
