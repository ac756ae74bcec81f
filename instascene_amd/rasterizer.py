"""Host-side mirror of the reference's ``diff_surfel_rasterization`` package
(submodules/diff-surfel-rasterization/diff_surfel_rasterization/__init__.py):
``GaussianRasterizationSettings`` (:179-191), ``GaussianRasterizer`` (:194-248),
the autograd function (:49-176) and the three ``_C`` entry points
(rasterize_points.h:18-74) — same names, argument order, return tuples and
exception types — on top of the C-ABI HIP library.

PyTorch is used only for device memory, the current stream and autograd
plumbing; all arithmetic happens in ``libinstascene_hip.so``.
"""
from __future__ import annotations

import collections
import contextlib
import ctypes
import weakref
import os
import sys
from typing import NamedTuple, Optional

import torch
import torch.nn as nn

from . import _hot, _lib, arena
from ._lib import GRAD_EXTRA, GRAD_GEOMETRY, MODE_EXACT, MODE_FAST, MODE_FEATURE_ONLY, MODE_PREBINNED, check, lib

# The shipped default is "fast_reflists": FAST arithmetic on the REFERENCE's tile rectangles, so that radii, tiles_touched, point_list,
# ranges and num_rendered are the reference's bit for bit (north-star: "bit-exact tile/sort indices").  ISR_MODE=fast selects the
# shorter lists described below (same output bits, ~2 % faster step, integer state = subsequences of the reference's).
_ENV_MODE = os.environ.get("ISR_MODE", "fast_reflists").lower()
_CONFIG = {
    # arithmetic mode of the per-pixel loops: FAST (the default) or "exact" (ISR_MODE=exact: op-for-op IEEE in the reference's
    # operation order, bit-identical to the CPU oracle, 1.9x slower blend).  FAST (csrc/isr_fast_pair.hpp) keeps EXACT's own two
    # roundings where they dominate (k.z, l.z of the ray-splat intersection), uses fused multiply-adds, v_rcp_f32 and v_exp_f32
    # elsewhere (the reference's nvcc build contracts to FMA too but keeps IEEE division and libdevice expf: FAST goes beyond
    # that in rcp / exp2 only), and re-evaluates with EXACT's instruction sequence every pair that lies within the rounding noise
    # of a decision (alpha = 1/255, rho3d = rho2d, depth = near): those decisions are EXACT's by construction, checked on the
    # device for every pair by the STATS build.  Measured (tests/test_gpu_fuzz.py, profiles/r04_fuzz_340_scenes.jsonl, 340 scenes):
    # 1 of 2.3 M pixels differs from the oracle in its last / median contributor or beyond 1e-4 (a T < 1e-4 stop; the full-size
    # C3 view: 4 of 2 073 600), no gradient row beyond 1e-3.  Forward and backward take identical per-pixel decisions.
    "mode": MODE_EXACT if _ENV_MODE == "exact" else MODE_FAST,
    # opt-in (ISR_MODE=fast / set_mode("fast")): FAST bins a splat only into the tiles its alpha >= 1/255 box reaches (ISR_PREPARE_TIGHT_RECTS) - the box the blend and
    # backward kernels already apply per 8x8 block (k_pack_hits' masks), so they walk exactly the same (block, splat) pairs in the
    # same order as with the reference's square rectangles: image, allmap channels 0-5, feature map, radii, tracer pairs and the
    # dense geometry gradients are the same bits (tests/test_gpu_rasterizer.py::test_tight_rectangles_change_no_output_bit, every
    # fuzz scene, and at full C3 / C2 size in test_gpu_fullsize.py); the distortion channel is evaluated relative to the depth of
    # the tile's first list entry and moves in its last bits, and the backward kernels that scan over chunks of 64 list positions
    # associate their products differently (1e-7 of the maximum).  Tile lists are order-preserving SUBSEQUENCES of the
    # reference's (8-19 % fewer instances to count, scatter, sort and pack at C2 / C3), num_rendered is their total.  "fast_reflists" (or ISR_TIGHT_RECTS=0) keeps the reference's
    # rectangles: tiles_touched, point_list and ranges then equal the reference's bit for bit, as they always do in EXACT mode.
    "tight_rects": _ENV_MODE in ("fast", "fast_tight") and os.environ.get("ISR_TIGHT_RECTS", "1") != "0",
    # produce the (gaussian, pixel) tracer list like the reference does on every forward
    "tracer": os.environ.get("ISR_TRACER", "1") != "0",
    # size the binning workspace from the previous view's instance count (+25 %) instead of a blocking
    # device->host read of R in the middle of every forward (the reference always blocks, rasterizer_impl.cu:287).
    # The true R is read back asynchronously and verified before the backward / the next forward.
    "async_binning": os.environ.get("ISR_ASYNC_BINNING", "0") == "1",
}
_ASYNC_GROWTH, _ASYNC_SLACK = 1.25, 65536
_R_ESTIMATE = {}      # (device, P, W, H, tight, view matrices' storage) -> verified instance count of THAT view
_PENDING = {}         # address of a forward's geometry buffer -> (pinned int64 tensor, event, capacity, estimate key)


LAST_NUM_RENDERED = 0   # instance count of the most recent forward (reporting only)
# A prefetched chain's key scatter (k_scatter) is the one kernel of the chain that suffers beside an HBM-saturating neighbour; with
# the gate on, prefetch_geometry() has the library record an event right behind it (isr_forward_bin_event) and leaves it here: the
# trainer makes its per-Gaussian tail wait for it (SegTrainer.gate_tail).  A small ring of events, recorded again and again.
_SCATTER_GATE = [False]
LAST_SCATTER_EVENT = None
_SCATTER_EVENTS = {}


def set_scatter_gate(on: bool):
    global LAST_SCATTER_EVENT
    _SCATTER_GATE[0] = bool(on)
    if not on:
        LAST_SCATTER_EVENT = None


def _next_scatter_event(dev):
    ring = _SCATTER_EVENTS.get(dev.index)
    if ring is None:
        evs = []
        for _ in range(8):
            e = torch.cuda.Event()
            e.record()                      # (materialises the hipEvent_t behind the object: its handle is what the library records)
            evs.append(e)
        ring = _SCATTER_EVENTS[dev.index] = [evs, 0]
    ring[1] = (ring[1] + 1) % len(ring[0])
    return ring[0][ring[1]]

_SCALED_ROWS_ATTR = "_isr_scaled_rows"       # = contrastive.SCALED_ROWS_ATTR (a feature table handed over raw + two factors per row)
_LATE_READBACK = os.environ.get("ISR_LATE_READBACK", "1") not in ("0",)    # prefetch_geometry: the count read-back behind the binning
_FWD_WAVE = os.environ.get("ISR_FWD_WAVE", "1") not in ("0",)      # the library's per-block FAST blend is the one in use


def set_mode(mode: str):
    """"exact" | "fast_reflists" (the shipped default: FAST arithmetic, the reference's tile lists) | "fast" (= "fast_tight": FAST
    arithmetic on the shorter lists, see _CONFIG)."""
    _CONFIG["mode"] = {"exact": MODE_EXACT, "fast": MODE_FAST, "fast_tight": MODE_FAST, "fast_reflists": MODE_FAST}[mode]
    _CONFIG["tight_rects"] = mode in ("fast", "fast_tight")
    ext = sys.modules.get("instascene_amd._C_hip")          # the compiled boundary follows, when it is loaded
    if ext is not None:
        ext.set_mode("fast" if mode == "fast_tight" else mode)


def get_mode() -> str:
    if _CONFIG["mode"] != MODE_FAST:
        return "exact"
    return "fast" if _CONFIG["tight_rects"] else "fast_reflists"


def set_tracer(enabled: bool):
    _CONFIG["tracer"] = bool(enabled)


def set_async_binning(enabled: bool):
    _CONFIG["async_binning"] = bool(enabled)


class _ViewState:
    """Geometry pass + binning of one view, kept for the next render of the same view with the same inputs."""
    __slots__ = ("radii", "geom", "img", "R", "binning", "nbytes", "busy", "refs", "__weakref__")

    def __init__(self, radii, geom, img, R, binning, refs=()):
        self.radii, self.geom, self.img, self.R, self.binning, self.refs = radii, geom, img, R, binning, refs
        self.nbytes = sum(t.numel() * t.element_size() for t in (radii, geom, img, binning))
        self.busy = 0           # forwards whose backward has not run yet: they still need this img / binning state

    def release(self):
        self.busy = max(0, self.busy - 1)


_VIEW_CACHE = collections.OrderedDict()       # geometry signature -> _ViewState (LRU)
VIEW_CACHE_HITS = 0


def set_view_cache(gigabytes: float):
    """Opt-in (default 0 = off).  While the geometry, SH and camera tensors of a render are unchanged (same storage, same
    version counters — e.g. the frozen-geometry feature training of train_semantic.py), the projection, the tile lists
    and their depth order are the same every time that view comes round.  With a budget, the forward keeps that state
    (~35 bytes per Gaussian + 16 per tile instance + 20 per pixel; 0.2 GB per 1080p view of a 1.5 M scene — a few hundred
    views fit the 288 GB of an MI355X) and the next render of the view starts at the blend kernel.  An entry is not reused
    while a backward that needs it is outstanding; any in-place update of an input invalidates it."""
    _CONFIG["view_cache_bytes"] = int(max(0.0, float(gigabytes)) * (1 << 30))
    if _CONFIG["view_cache_bytes"] == 0:
        _VIEW_CACHE.clear()


def _view_cache_put(sig, state):
    _VIEW_CACHE[sig] = state
    total = sum(v.nbytes for v in _VIEW_CACHE.values())
    for k in list(_VIEW_CACHE.keys()):
        if total <= _CONFIG["view_cache_bytes"] or k == sig:
            continue
        if _VIEW_CACHE[k].busy:
            continue
        total -= _VIEW_CACHE.pop(k).nbytes


class BinningOverflow(RuntimeError):
    """A forward that sized its binning workspace from an estimate turned out to need more: its outputs are truncated.
    The estimate of that view has been corrected; re-run the forward (``SegTrainer.step`` does)."""


_PINNED = {"ring": None, "next": 0}


def _pinned_slot():
    """A pinned int64 for the asynchronous read-back of an instance count: a ring of 64 slots allocated once (a fresh
    ``pin_memory()`` per forward goes through the host allocator every step); a slot comes round again long after its
    forward has been verified (at most a few forwards are ever outstanding)."""
    if _PINNED["ring"] is None:
        _PINNED["ring"] = torch.empty(64, dtype=torch.int64).pin_memory()
    i = _PINNED["next"]
    _PINNED["next"] = (i + 1) % 64
    return _PINNED["ring"][i:i + 1]


def _view_id(viewmatrix, projmatrix):
    """Identity of a camera for the size estimate: the storage of its two matrices (cameras are long-lived objects)."""
    return (viewmatrix.data_ptr() if viewmatrix is not None else 0, projmatrix.data_ptr() if projmatrix is not None else 0)


def _verify_entry(owner, pend):
    pinned, event, capacity, ekey = pend
    event.synchronize()
    R = int(pinned.item())
    _R_ESTIMATE[ekey] = R                 # the true count of THIS view: the next capacity is 1.25 R + 64 k
    if R > capacity:
        raise BinningOverflow(
            f"rasterizer: {R} tile instances exceeded the async binning capacity {capacity} estimated for this view; the "
            "outputs of that forward are truncated. The estimate has been corrected: re-run the forward.")


_OVERFLOWED = {}      # geometry buffer address -> message: forwards found truncated before anybody asked about them


def _verify_pending(owner):
    """Check the asynchronously read instance count of the forward whose geometry buffer is ``owner`` against the
    capacity it ran with (blocks until that forward's geometry pass has finished)."""
    msg = _OVERFLOWED.pop(owner, None)
    if msg is not None:
        raise BinningOverflow(msg)
    pend = _PENDING.pop(owner, None)
    if pend is not None:
        _verify_entry(owner, pend)


def _reap_pending():
    """Counts whose copy has completed are checked without blocking, so that entries of forwards whose backward never
    runs do not pile up; a truncated one is remembered until somebody asks about that forward."""
    for owner in [o for o, p in _PENDING.items() if p[1].query()]:
        pend = _PENDING.pop(owner)
        try:
            _verify_entry(owner, pend)
        except BinningOverflow as e:
            if len(_OVERFLOWED) > 64:
                _OVERFLOWED.clear()
            _OVERFLOWED[owner] = str(e)


def _ptr(t: Optional[torch.Tensor]):
    if t is None or t.numel() == 0:
        return None
    return ctypes.c_void_p(t.data_ptr())


def _stream():
    return _hot.stream_ptr()


def _f32c(t: Optional[torch.Tensor], name: str):
    if t is None or t.numel() == 0:
        return None
    if not t.is_cuda:
        raise RuntimeError(f"{name} must be a CUDA tensor")     # reference: CHECK_INPUT, rasterize_points.cu:27-28
    if t.dtype != torch.float32:
        t = t.float()
    return t.contiguous()


_TIGHT_RECTS = 0x100  # ISR_PREPARE_TIGHT_RECTS
_PREFETCH_LIMIT = int(float(os.environ.get("ISR_PREFETCH_MAX_GB", "8")) * (1 << 30))
_PREFETCHED = {}      # signature -> _Prefetched: geometry pass + binning issued ahead of its forward
PREFETCH_HITS = 0     # forwards that found their geometry pass already issued (statistics / tests)


def _tight(mode, tight=None) -> bool:
    return mode == MODE_FAST and bool(_CONFIG["tight_rects"] if tight is None else tight)


def _geometry_signature(mode, tight, P, W, H, degree, M, scale_modifier, tan_fovx, tan_fovy, prefiltered, tensors):
    """Identity of everything the geometry pass reads: scalars + (address, version, shape) of the CALLER's tensors.  An
    address can be recycled once its tensor is freed, so a cache entry additionally holds weak references to those
    tensors (:func:`_refs`) and is only honoured while every one of them is still the very same object."""
    return (int(mode), bool(tight), P, W, H, int(degree), M, float(scale_modifier), float(tan_fovx), float(tan_fovy),
            bool(prefiltered)) + tuple(
        None if (t is None or t.numel() == 0) else (t.data_ptr(), t._version, tuple(t.shape)) for t in tensors)


def _refs(tensors):
    return tuple(None if (t is None or t.numel() == 0) else weakref.ref(t) for t in tensors)


def _same_objects(refs, tensors) -> bool:
    for r, t in zip(refs, tensors):
        if r is None:
            if not (t is None or t.numel() == 0):
                return False
        elif r() is not t:
            return False
    return True


class _Prefetched:
    __slots__ = ("sig", "refs", "radii", "geom", "img", "R", "binning", "done", "inputs")

    def __init__(self, sig, refs, radii, geom, img, R, binning, done, inputs):
        self.sig, self.refs, self.radii, self.geom, self.img, self.R = sig, refs, radii, geom, img, R
        self.binning, self.done, self.inputs = binning, done, inputs


def _round_capacity(r: int) -> int:
    """Capacities in steps of 1/32 of their magnitude: the views of a scene then ask the caching allocator for a handful of
    distinct workspace sizes instead of one per view, and its pools stop growing (a hipMalloc of hundreds of MB in the middle
    of a training loop) after the first few steps rather than after the first visit of every view."""
    if r <= 0:
        return r
    g = 1 << max(0, r.bit_length() - 6)
    return (r + g - 1) // g * g


_CLASSES = [4096]


def _size_class(n: int) -> int:
    """The smallest member >= n of a geometric sequence of sizes (x 1.25 per step).  Workspaces whose size follows a
    quantity that DRIFTS - the tile-instance count of a view while the geometry trains, the number of Gaussians under
    density control - are allocated in these classes (the library is told the true count, the buffer is merely larger):
    the caching allocator then sees a new size only every +25 %, i.e. a handful of times per training run, instead of a new
    size at every step of the drift (round 2's 3 000-iteration train.py soak: reserved memory 1.29 -> 2.91 GB, 24 -> 30
    device allocations, because every 3 % step of R was a size the allocator had never served)."""
    n = int(n)
    while _CLASSES[-1] < n:
        _CLASSES.append((int(_CLASSES[-1] * 1.25) + 255) // 256 * 256)
    import bisect
    return _CLASSES[bisect.bisect_left(_CLASSES, n)]


def _workspace(nbytes_of, count, dev):
    """A byte workspace for ``count`` units, sized for the count's class (``nbytes_of(count_class)`` bytes)."""
    return arena.empty(nbytes_of(_size_class(max(1, int(count)))), torch.uint8, dev)


def _prepare(L, mode, key, view, st, P, degree, M, W, H, means3D, sh, colors, opacity, scales, scale_modifier, rotations,
             transMat_precomp, viewmatrix, projmatrix, campos, tan_fovx, tan_fovy, prefiltered, radii, geom, img, tight=None,
             readback=True):
    """K1 + tile scan (``isr_forward_prepare``) into ``radii / geom / img``; returns ``(R, is_capacity)``: the instance
    count, or - with async binning and a verified count of THIS view ``(key, view)`` from an earlier forward - the
    capacity the binning workspace is sized with (the true count is read back asynchronously and verified before the
    backward, or at once when autograd is off)."""
    global LAST_NUM_RENDERED
    num_rendered = ctypes.c_int64(0)
    ekey = key + (bool(_tight(mode, tight)),) + tuple(view)
    use_async = _CONFIG["async_binning"] and ekey in _R_ESTIMATE
    if _PENDING:
        _reap_pending()
    check(L.isr_forward_prepare(P, int(degree), M, W, H, _ptr(means3D), _ptr(sh), _ptr(colors), _ptr(opacity),
                                _ptr(scales), float(scale_modifier), _ptr(rotations), _ptr(transMat_precomp),
                                _ptr(viewmatrix), _ptr(projmatrix), _ptr(campos), float(tan_fovx), float(tan_fovy),
                                int(bool(prefiltered)) | (_TIGHT_RECTS if _tight(mode, tight) else 0), _ptr(radii),
                                _ptr(geom), _ptr(img),
                                None if use_async else ctypes.byref(num_rendered), st), "isr_forward_prepare")
    if use_async:
        R = _round_capacity(int(_R_ESTIMATE[ekey] * _ASYNC_GROWTH) + _ASYNC_SLACK)      # capacity, not the count
        if readback:
            _issue_readback(geom, R, ekey)
        else:
            # (prefetch_geometry issues it BEHIND the binning: the copy and its event are a blit kernel and a barrier packet -
            # ~10 us between the tile scan and k_scatter on a chain every microsecond of which shows in the step)
            _PENDING.pop(geom.data_ptr(), None)
        _OVERFLOWED.pop(geom.data_ptr(), None)      # a recycled address
        LAST_NUM_RENDERED = _R_ESTIMATE[ekey]
    else:
        # (this forward was sized exactly: whatever an earlier forward left under the same - recycled - buffer address is stale)
        _PENDING.pop(geom.data_ptr(), None)
        _OVERFLOWED.pop(geom.data_ptr(), None)
        R = int(num_rendered.value)
        _R_ESTIMATE[ekey] = R
        if len(_R_ESTIMATE) > 4096:
            for k in list(_R_ESTIMATE)[:1024]:
                del _R_ESTIMATE[k]
        LAST_NUM_RENDERED = R
    return R, use_async


def _issue_readback(geom, capacity, ekey):
    """Asynchronous read-back of a forward's true instance count (header[0] of its geometry state) on the current stream; the
    count is verified against ``capacity`` before the backward (``_verify_entry``)."""
    pinned = _pinned_slot()
    pinned.copy_(geom[:8].view(torch.int64), non_blocking=True)   # header[0] = R
    ev = torch.cuda.Event()
    ev.record()
    _PENDING[geom.data_ptr()] = (pinned, ev, capacity, ekey)


def prefetch_geometry(means3D, colors, opacity, scales, rotations, scale_modifier, transMat_precomp, viewmatrix,
                      projmatrix, tan_fovx, tan_fovy, image_height, image_width, sh, degree, campos, prefiltered,
                      stream=None, after=None) -> bool:
    """Issue the geometry pass (K1, tile counts, scan) and the binning (key scatter, tile sort) of a view NOW, for a
    ``rasterize_gaussians`` call that will follow with exactly these inputs (the same tensor objects, unmodified).  It
    reads no feature, so a trainer can overlap it with the rest of the current step (small, latency-bound loss kernels,
    the bandwidth-bound per-Gaussian tail, the all-reduce of the gradient): ``stream`` = a side ``torch.cuda.Stream`` to
    issue it on (default: the current stream), ``after`` = an event that stream waits for first (default: everything
    enqueued on the current stream so far).  The forward that consumes the entry waits for its completion event.  Needs
    async binning and a verified instance count of this view from an earlier forward; returns False (and does nothing)
    otherwise.  An entry that is never consumed is simply dropped."""
    L = lib()
    dev = means3D.device
    P, H, W = means3D.shape[0], int(image_height), int(image_width)
    key = (dev.index, P, W, H)
    mode = _CONFIG["mode"]
    view = _view_id(viewmatrix, projmatrix)
    if P == 0 or not _CONFIG["async_binning"] or (key + (bool(_tight(mode)),) + view) not in _R_ESTIMATE:
        return False
    originals = (means3D, sh, colors, opacity, scales, rotations, transMat_precomp, viewmatrix, projmatrix, campos)
    M = sh.shape[1] if (sh is not None and sh.dim() == 3) else 0
    sig = _geometry_signature(mode, _tight(mode), P, W, H, degree, M, scale_modifier, tan_fovx, tan_fovy, prefiltered, originals)
    kept = _VIEW_CACHE.get(sig) if _CONFIG.get("view_cache_bytes", 0) > 0 else None
    if kept is not None and kept.busy == 0 and _same_objects(kept.refs, originals):
        return False                      # the view cache already holds this view's state
    waiting = _PREFETCHED.get(sig)
    if waiting is not None and _same_objects(waiting.refs, originals):
        return True                       # already issued for exactly these inputs and not consumed yet
    # fp32 / contiguous forms (made on the caller's stream; a non-contiguous reference Camera matrix gets a temporary)
    means3D = _f32c(means3D, "means3D")
    colors, opacity = _f32c(colors, "colors"), _f32c(opacity, "opacity")
    scales, rotations = _f32c(scales, "scales"), _f32c(rotations, "rotations")
    transMat_precomp = _f32c(transMat_precomp, "transMat_precomp")
    viewmatrix, projmatrix = _f32c(viewmatrix, "viewmatrix"), _f32c(projmatrix, "projmatrix")
    sh, campos = _f32c(sh, "sh"), _f32c(campos, "campos")
    inputs = (means3D, sh, colors, opacity, scales, rotations, transMat_precomp, viewmatrix, projmatrix, campos)
    with _hot.on_device(dev):
        done = None
        side = stream if (stream is not None and stream.cuda_stream != _hot.raw_stream()) else None
        if side is not None:
            if after is None or after is False:
                # `after=False` used to mean "start at once"; temporaries made above on the caller's stream must be
                # complete before the side stream reads them, so the side stream always waits at least for those
                made_temporary = any(a is not b for a, b in zip(inputs, originals))
                if after is None or made_temporary:
                    after = torch.cuda.Event()
                    after.record()
            if after is not None and after is not False:
                side.wait_event(after)
            for t in inputs:              # read by kernels on the side stream: the allocator must not recycle them earlier
                if t is not None:
                    t.record_stream(side)
        with torch.cuda.stream(side) if side is not None else contextlib.nullcontext():
            radii = arena.empty((P,), torch.int32, dev)
            geom = _workspace(L.isr_geom_bytes, P, dev)
            img = arena.empty(L.isr_image_bytes(W, H), torch.uint8, dev)
            st = _stream()
            late_rb = side is not None and _LATE_READBACK
            R, is_cap = _prepare(L, mode, key, view, st, P, degree, M, W, H, means3D, sh, colors, opacity, scales, scale_modifier,
                                 rotations, transMat_precomp, viewmatrix, projmatrix, campos, tan_fovx, tan_fovy, prefiltered,
                                 radii, geom, img, readback=not late_rb)
            binning = _workspace(lambda c: L.isr_binning_bytes(c, W, H), R, dev)
            sev = _next_scatter_event(dev) if (side is not None and _SCATTER_GATE[0]) else None
            check(L.isr_forward_bin_event(P, W, H, _ptr(geom), _ptr(binning), R, _ptr(img),
                                          ctypes.c_void_p(sev.cuda_event) if sev is not None else None, st), "isr_forward_bin")
            if sev is not None:
                global LAST_SCATTER_EVENT
                LAST_SCATTER_EVENT = sev
            if late_rb and is_cap:
                _issue_readback(geom, R, key + (bool(_tight(mode)),) + tuple(view))
            if side is not None:
                done = torch.cuda.Event()
                done.record()
    # the entry keeps the converted inputs alive until it is consumed or dropped
    _PREFETCHED[sig] = _Prefetched(sig, _refs(originals), radii, geom, img, R, binning, done, inputs)
    # entries nobody came for: bounded by count AND by bytes (an entry pins radii + geom + img + binning - ~0.4 GB per view at
    # C3, ~1 GB at C5; ISR_PREFETCH_MAX_GB, default 8); the newest entry always stays
    limit = _PREFETCH_LIMIT
    nbytes = lambda e: sum(t.numel() * t.element_size() for t in (e.radii, e.geom, e.img, e.binning))
    total = sum(nbytes(e) for e in _PREFETCHED.values())
    while len(_PREFETCHED) > 1 and (len(_PREFETCHED) > 12 or total > limit):
        old = _PREFETCHED.pop(next(iter(_PREFETCHED)))
        total -= nbytes(old)
        _PENDING.pop(old.geom.data_ptr(), None)
    return True


def rasterize_gaussians(bg, means3D, colors, opacity, scales, rotations, scale_modifier, transMat_precomp,
                        extra_attrs, attr_degree, viewmatrix, projmatrix, tan_fovx, tan_fovy, image_height,
                        image_width, sh, degree, campos, prefiltered, debug, *, tracer=None, mode=None, tight=None,
                        _state_out=None, _verify_at_backward=False, feature_only=False):
    """Equivalent of ``_C.rasterize_gaussians`` (rasterize_points.cu:39-151).

    Returns ``(num_rendered, out_color, out_others, radii, out_extra, geomBuffer, binningBuffer, imgBuffer,
    gau_related_pixels, gau_pixel_indices)``.  ``gau_related_pixels`` holds ``H*W*10`` rows (a pixel has at most 9
    entries with weight > 0.1) instead of the reference's ``H*W*100`` and is not pre-filled;
    ``gau_pixel_indices`` is the reference's *last valid index* (count - 1) so that
    ``gau_related_pixels[:gau_pixel_indices + 1]`` is the list, as in the reference wrapper (:106).

    ``feature_only`` (opt-in extension, FAST mode with a feature channel): only ``out_extra`` and the state of a
    feature-only backward are produced; ``out_color`` / ``out_others`` come back empty and there is no tracer list."""
    L = lib()
    if means3D.dim() != 2 or means3D.shape[1] != 3:
        raise RuntimeError("means3D must have dimensions (num_points, 3)")
    dev = means3D.device
    P, H, W, F = means3D.shape[0], int(image_height), int(image_width), int(attr_degree)
    mode = _CONFIG["mode"] if mode is None else mode
    tight = _tight(mode, tight)
    tracer = _CONFIG["tracer"] if tracer is None else tracer
    feature_only = bool(feature_only) and mode == MODE_FAST and int(attr_degree) > 0
    if feature_only:
        tracer = False
    call_args = (bg, means3D, colors, opacity, scales, rotations, scale_modifier, transMat_precomp, extra_attrs, attr_degree,
                 viewmatrix, projmatrix, tan_fovx, tan_fovy, image_height, image_width, sh, degree, campos, prefiltered, debug)
    originals = (means3D, sh, colors, opacity, scales, rotations, transMat_precomp, viewmatrix, projmatrix, campos)
    view = _view_id(viewmatrix, projmatrix)
    means3D = _f32c(means3D, "means3D")
    bg = _f32c(bg, "background")
    colors, opacity = _f32c(colors, "colors"), _f32c(opacity, "opacity")
    scales, rotations = _f32c(scales, "scales"), _f32c(rotations, "rotations")
    transMat_precomp = _f32c(transMat_precomp, "transMat_precomp")
    viewmatrix, projmatrix = _f32c(viewmatrix, "viewmatrix"), _f32c(projmatrix, "projmatrix")
    sh, campos = _f32c(sh, "sh"), _f32c(campos, "campos")
    xscale = None
    scaled_src = getattr(extra_attrs, _SCALED_ROWS_ATTR, None) if (F > 0 and extra_attrs is not None) else None
    if scaled_src is not None:
        # a placeholder for normalize(normalize(x)) that was never stored (contrastive.FeatureAdam.store_z = False): the per-block
        # FAST blend takes the raw table and the two factors per row; every other kernel gets the values materialised
        if mode == MODE_FAST and _FWD_WAVE and P <= (1 << 26) and scaled_src[0].is_cuda:
            extra_attrs, xscale = scaled_src[0], _f32c(scaled_src[1], "extra_row_scale")
        else:
            extra_attrs = (scaled_src[0] * scaled_src[1][:, 0:1]) * scaled_src[1][:, 1:2]
    extra = _f32c(extra_attrs.to(dev) if (extra_attrs is not None and extra_attrs.numel() and not extra_attrs.is_cuda)
                  else extra_attrs, "extra_attrs") if F > 0 else None

    # outputs and state come out of the library's arena (arena.py): fresh tensors for the caller, but the memory behind them
    # survives the reference driver's per-iteration torch.cuda.empty_cache()
    out_color = arena.empty((0 if feature_only else 3, H, W), torch.float32, dev)
    out_others = arena.empty((0 if feature_only else 7, H, W), torch.float32, dev)
    out_extra = arena.empty((F, H, W), torch.float32, dev) if F > 0 else torch.empty(0, device=dev)
    M = sh.shape[1] if (sh is not None and sh.dim() == 3) else 0
    if P == 0:
        radii = torch.zeros((P,), dtype=torch.int32, device=dev)
        geom = torch.empty(L.isr_geom_bytes(P), dtype=torch.uint8, device=dev)
        img = torch.empty(L.isr_image_bytes(W, H), dtype=torch.uint8, device=dev)
        out_color.zero_()
        out_color += bg.view(3, 1, 1)
        out_others.zero_()
        if F > 0:
            out_extra.zero_()
        return (0, out_color, out_others, radii, out_extra, geom, torch.empty(0, dtype=torch.uint8, device=dev), img,
                torch.empty((0, 2), dtype=torch.int32, device=dev), torch.full((1,), -1, dtype=torch.int32, device=dev))
    with _hot.on_device(dev):
        st = _stream()
        key = (dev.index, P, W, H)
        sig = _geometry_signature(mode, tight, P, W, H, degree, M, scale_modifier, tan_fovx, tan_fovy, prefiltered, originals)
        ahead = _PREFETCHED.pop(sig, None)
        if ahead is not None and not _same_objects(ahead.refs, originals):
            _PENDING.pop(ahead.geom.data_ptr(), None)
            ahead = None                  # a recycled address: not the tensors that geometry pass read
        prebinned = 0
        sized_by_estimate = False
        kept = _VIEW_CACHE.get(sig) if _CONFIG.get("view_cache_bytes", 0) > 0 else None
        if kept is not None and not _same_objects(kept.refs, originals):
            _VIEW_CACHE.pop(sig, None)
            kept = None
        if kept is not None and kept.busy == 0:
            radii, geom, img, R, binning = kept.radii, kept.geom, kept.img, kept.R, kept.binning
            prebinned = MODE_PREBINNED
            _VIEW_CACHE.move_to_end(sig)
            global VIEW_CACHE_HITS
            VIEW_CACHE_HITS += 1
            if _state_out is not None:
                _state_out.append(kept)
        elif ahead is not None:
            # the geometry pass of this view was issued by prefetch_geometry()
            radii, geom, img, R, binning, done = ahead.radii, ahead.geom, ahead.img, ahead.R, ahead.binning, ahead.done
            if done is not None:                # ... on a side stream: order this stream behind it, keep its buffers alive
                cur = torch.cuda.current_stream()
                cur.wait_event(done)
                for t in (radii, geom, img, binning):
                    arena.used_on(t, cur)
            prebinned = MODE_PREBINNED
            sized_by_estimate = geom.data_ptr() in _PENDING or geom.data_ptr() in _OVERFLOWED
            global PREFETCH_HITS
            PREFETCH_HITS += 1
        else:
            radii = arena.empty((P,), torch.int32, dev)      # K1 writes every entry
            geom = _workspace(L.isr_geom_bytes, P, dev)
            img = arena.empty(L.isr_image_bytes(W, H), torch.uint8, dev)
            R, sized_by_estimate = _prepare(L, mode, key, view, st, P, degree, M, W, H, means3D, sh, colors, opacity, scales,
                                            scale_modifier, rotations, transMat_precomp, viewmatrix, projmatrix, campos,
                                            tan_fovx, tan_fovy, prefiltered, radii, geom, img, tight=tight)
            binning = _workspace(lambda c: L.isr_binning_bytes(c, W, H), R, dev)
        if tracer:
            grp = arena.empty((H * W * 10, 2), torch.int32, dev)
            gcount = torch.empty((1,), dtype=torch.int32, device=dev)
        else:
            grp, gcount = None, None
        check(L.isr_forward_render_scaled(P, F, W, H, int(mode) | prebinned | (MODE_FEATURE_ONLY if feature_only else 0), _ptr(bg), _ptr(colors), _ptr(transMat_precomp), _ptr(extra),
                                          _ptr(xscale), _ptr(geom), _ptr(binning), R, _ptr(img), _ptr(out_color), _ptr(out_others),
                                          _ptr(out_extra), _ptr(grp), H * W * 10 if tracer else 0, _ptr(gcount), st),
              "isr_forward_render")
    if sized_by_estimate and not _verify_at_backward:
        # no backward will follow to verify the estimate (an eval / no_grad render, or a direct call of this function):
        # check it now, and if the view needed more than the estimate allowed, render it again with the exact, blocking
        # sizing instead of returning a truncated image
        try:
            _verify_pending(geom.data_ptr())
        except BinningOverflow:
            was = _CONFIG["async_binning"]
            _CONFIG["async_binning"] = False
            try:
                return rasterize_gaussians(*call_args, tracer=tracer, mode=mode, tight=tight, _state_out=_state_out,
                                           feature_only=feature_only)
            finally:
                _CONFIG["async_binning"] = was
    if _CONFIG.get("view_cache_bytes", 0) > 0 and kept is None and _state_out is not None and not sized_by_estimate:
        st_new = _ViewState(radii, geom, img, R, binning, _refs(originals))
        if st_new.nbytes <= _CONFIG["view_cache_bytes"]:
            _view_cache_put(sig, st_new)
            _state_out.append(st_new)
    if tracer:
        gidx = gcount               # the library counts from -1: already the last valid index
    else:
        grp = torch.empty((0, 2), dtype=torch.int32, device=dev)
        gidx = torch.full((1,), -1, dtype=torch.int32, device=dev)
    return R, out_color, out_others, radii, out_extra, geom, binning, img, grp, gidx


def rasterize_gaussians_backward(bg, means3D, radii, colors, scales, rotations, extra_attrs, scale_modifier,
                                 transMat_precomp, viewmatrix, projmatrix, tan_fovx, tan_fovy, dL_dout_color,
                                 dL_dout_others, dL_dout_extra, sh, degree, campos, geomBuffer, R, binningBuffer,
                                 imageBuffer, debug, *, grad_mask=GRAD_EXTRA | GRAD_GEOMETRY, mode=None):
    """Equivalent of ``_C.rasterize_gaussians_backward`` (rasterize_points.cu:153-262).  Returns
    ``(dL_dmeans2D[P,3], dL_dcolors[P,3], dL_dopacity[P,1], dL_dmeans3D[P,3], dL_dtransMat[P,9], dL_dsh[P,M,3],
    dL_dscales[P,2], dL_drotations[P,4], dL_dextra[P,F])``; gradients not selected by ``grad_mask`` are None."""
    L = lib()
    dev = means3D.device
    P = means3D.shape[0]
    _ref = next(t for t in (dL_dout_color, dL_dout_others, dL_dout_extra) if t is not None)
    H, W = _ref.shape[1], _ref.shape[2]
    mode = _CONFIG["mode"] if mode is None else mode
    means3D = _f32c(means3D, "means3D")
    colors, scales, rotations = _f32c(colors, "colors"), _f32c(scales, "scales"), _f32c(rotations, "rotations")
    transMat_precomp = _f32c(transMat_precomp, "transMat_precomp")
    sh, campos, bg = _f32c(sh, "sh"), _f32c(campos, "campos"), _f32c(bg, "background")
    viewmatrix, projmatrix = _f32c(viewmatrix, "viewmatrix"), _f32c(projmatrix, "projmatrix")
    F = extra_attrs.shape[1] if (extra_attrs is not None and extra_attrs.numel()) else 0
    extra = _f32c(extra_attrs, "extra_attrs") if F else None
    M = sh.shape[1] if (sh is not None and sh.dim() == 3) else 0
    dC = _f32c(dL_dout_color, "dL_dout_color") if dL_dout_color is not None else None
    dO = _f32c(dL_dout_others, "dL_dout_others") if dL_dout_others is not None else None
    dE = _f32c(dL_dout_extra, "dL_dout_extra") if (F and dL_dout_extra is not None) else None
    if F == 0:
        grad_mask &= ~GRAD_EXTRA
    geomg = bool(grad_mask & GRAD_GEOMETRY)

    def new(*shape):
        return arena.empty(shape, torch.float32, dev)

    g2 = new(P, 3) if geomg else None
    gn = new(P, 3) if geomg else None
    go = new(P, 1) if geomg else None
    gc = new(P, 3) if geomg else None
    g3 = new(P, 3) if geomg else None
    gt = new(P, 9) if geomg else None
    gsh = new(P, M, 3) if geomg else None
    gs = new(P, 2) if geomg else None
    gr = new(P, 4) if geomg else None
    ge = new(P, F) if (grad_mask & GRAD_EXTRA) else None
    if P == 0 or grad_mask == 0:
        return g2, gc, go, g3, gt, gsh, gs, gr, (ge if F else torch.empty(0, device=dev))
    _verify_pending(geomBuffer.data_ptr())
    scratch = _workspace(lambda c: L.isr_backward_scratch_bytes(c, F, grad_mask), R, dev)
    nbytes = scratch.numel()
    with _hot.on_device(dev):
        check(L.isr_backward(P, int(degree), M, int(R), F, W, H, int(mode), grad_mask, _ptr(bg), _ptr(means3D), _ptr(sh),
                             _ptr(colors), _ptr(scales), float(scale_modifier), _ptr(rotations), _ptr(transMat_precomp),
                             _ptr(extra), _ptr(viewmatrix), _ptr(projmatrix), _ptr(campos), float(tan_fovx),
                             float(tan_fovy), _ptr(radii), _ptr(geomBuffer), _ptr(binningBuffer), _ptr(imageBuffer),
                             _ptr(dC), _ptr(dO), _ptr(dE), _ptr(g2), _ptr(gn), _ptr(go), _ptr(gc), _ptr(g3), _ptr(gt),
                             _ptr(gsh), _ptr(gs), _ptr(gr), _ptr(ge), _ptr(scratch), nbytes, _stream()),
              "isr_backward")
    if ge is None:
        ge = torch.empty(0, device=dev)
    return g2, gc, go, g3, gt, gsh, gs, gr, ge


def sample_extra(out_extra: torch.Tensor, pixels: torch.Tensor) -> torch.Tensor:
    """``out_extra.reshape(F, -1)[:, pixels].T`` as one gather kernel (``isr_sample_extra``); ``pixels`` int64 = y*W + x."""
    F, H, W = out_extra.shape
    pix = pixels.contiguous().to(torch.int64)
    out = torch.empty((pix.shape[0], F), dtype=torch.float32, device=out_extra.device)
    with _hot.on_device(out_extra.device):
        check(lib().isr_sample_extra(F, W, H, pix.shape[0], _ptr(out_extra), _ptr(pix), _ptr(out), _stream()), "isr_sample_extra")
    return out


class FeatureRows:
    """Per-(tile, Gaussian) partial rows of dL/dextra left by a sampled backward that stopped before the per-Gaussian
    reduction; finished by ``isr_feature_rows_step`` (``contrastive.FeatureAdam.step_rows``)."""

    def __init__(self, scratch, geom, R, P, F):
        self.scratch, self.geom, self.R, self.P, self.F = scratch, geom, int(R), int(P), int(F)


class DeferredFeatureRows:
    """``with DeferredFeatureRows() as sink: loss.backward()`` — the first backward of a render whose only gradient is the
    one of its sampled features hands over its :class:`FeatureRows` in ``sink.rows`` and reports NO gradient for
    ``extra_attrs`` (the caller owns the rest of the chain); any further such backward inside the block takes the normal
    path and its dense ``[P,F]`` gradient reaches ``extra_attrs`` through autograd as usual.  Likewise the first backward
    of a ``contrastive.gather_rows`` leaves its sparse gradient in ``sink.row_grads`` instead of scattering it into a dense
    ``[P,F]`` tensor.  ``collect_dense=True`` (a trainer whose renders inside the block all differentiate the SAME
    ``extra_attrs`` tensor, and which adds ``sink.dense`` to that tensor's gradient itself): the further sampled backwards add
    their rows into ONE ``[P,F]`` tensor ``sink.dense`` - zero-filled once, each view touching only the rows its samples reach
    - and report no gradient, instead of a dense ``[P,F]`` reduction per render plus autograd's sum of them."""

    def __init__(self, collect_dense: bool = False):
        self.rows: Optional[FeatureRows] = None
        self.row_grads = None          # (indices, [n,F] gradient) of the first contrastive.gather_rows backward
        self.collect_dense = bool(collect_dense)
        self.dense = None              # [P,F] sum of the non-deferred sampled backwards (collect_dense)

    def __enter__(self):
        global _ROWS_SINK
        self._prev, _ROWS_SINK = _ROWS_SINK, self
        return self

    def __exit__(self, *exc):
        global _ROWS_SINK
        _ROWS_SINK = self._prev
        return False


_ROWS_SINK: Optional[DeferredFeatureRows] = None


def rasterize_gaussians_backward_sampled(P, F, W, H, R, pixels, dL_dsampled, transMat_precomp, geomBuffer, binningBuffer,
                                         imageBuffer, *, accumulate_into=None, mode=None, rows_only=False):
    """dL/dextra ``[P,F]`` from the gradient of the features SAMPLED at ``pixels`` (``isr_backward_sampled``): the dense
    ``[F,H,W]`` gradient map is never built.  ``accumulate_into``: an existing dL/dextra to add to.  ``rows_only``: stop
    before the per-Gaussian reduction and return the :class:`FeatureRows`."""
    L = lib()
    dev = geomBuffer.device
    mode = _CONFIG["mode"] if mode is None else mode
    _verify_pending(geomBuffer.data_ptr())
    pix = pixels.contiguous().to(torch.int64)
    g = dL_dsampled.contiguous().float()
    n = pix.shape[0]
    if rows_only:
        out = None
    else:
        out = accumulate_into if accumulate_into is not None else arena.empty((P, F), torch.float32, dev)
    scratch = _workspace(lambda c: L.isr_backward_sampled_scratch_bytes(c, F, n, W, H), R, dev)
    nbytes = scratch.numel()
    with _hot.on_device(dev):
        check(L.isr_backward_sampled(P, int(R), F, W, H, int(mode), n, _ptr(pix), _ptr(g), _ptr(_f32c(transMat_precomp, "transMat_precomp")),
                                     _ptr(geomBuffer), _ptr(binningBuffer), _ptr(imageBuffer), _ptr(out),
                                     1 if accumulate_into is not None else 0, _ptr(scratch), nbytes, _stream()),
              "isr_backward_sampled")
    if rows_only:
        return FeatureRows(scratch, geomBuffer, R, P, F)
    return out


def mark_visible(means3D, viewmatrix, projmatrix):
    """Equivalent of ``_C.mark_visible`` (rasterize_points.cu:264-283)."""
    L = lib()
    P = means3D.shape[0]
    present = torch.zeros((P,), dtype=torch.bool, device=means3D.device)
    if P:
        means3D = _f32c(means3D, "means3D")
        with _hot.on_device(means3D.device):
            check(L.isr_mark_visible(P, _ptr(means3D), _ptr(_f32c(viewmatrix, "viewmatrix")),
                                     _ptr(_f32c(projmatrix, "projmatrix")), _ptr(present), _stream()), "isr_mark_visible")
    return present


def debug_state(P, W, H, R, geom, binning, img):
    """Integer/forward state as numpy arrays (parity tests only)."""
    import numpy as np
    L = lib()
    N, T = W * H, ((W + 15) // 16) * ((H + 15) // 16)
    out = dict(tiles_touched=np.zeros(P, np.uint32), point_list=np.zeros(max(R, 1), np.uint32),
               ranges=np.zeros((T, 2), np.uint32), n_contrib=np.zeros((2, N), np.uint32),
               final_T=np.zeros((3, N), np.float32), records=np.zeros((P, 20), np.float32))
    vp = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    check(L.isr_debug_state(P, W, H, R, _ptr(geom), _ptr(binning) if R > 0 else None, _ptr(img), vp(out["tiles_touched"]),
                            vp(out["point_list"]), vp(out["ranges"]), vp(out["n_contrib"]), vp(out["final_T"]),
                            vp(out["records"]), _stream()), "isr_debug_state")
    out["point_list"] = out["point_list"][:R]
    return out


def _attach_count(buf, last_index):
    buf._isr_last_index = last_index
    return buf


def slice_tracer(buf):
    """Valid rows of a tracer buffer returned with lazy slicing."""
    idx = getattr(buf, "_isr_last_index", None)
    return buf if idx is None else buf[:(idx + 1)]


class GaussianRasterizationSettings(NamedTuple):
    image_height: int
    image_width: int
    tanfovx: float
    tanfovy: float
    bg: torch.Tensor
    scale_modifier: float
    viewmatrix: torch.Tensor
    projmatrix: torch.Tensor
    sh_degree: int
    campos: torch.Tensor
    prefiltered: bool
    debug: bool


def _cpu_snapshot(args):
    """``cpu_deep_copy_tuple`` of the reference wrapper (diff_surfel_rasterization/__init__.py:17-19): tensors copied to
    the host BEFORE the call, so that a fault cannot corrupt what gets dumped."""
    return tuple(a.detach().cpu().clone() if isinstance(a, torch.Tensor) else a for a in args)


class _debug_call:
    """``raster_settings.debug`` (reference __init__.py:93-101,150-158; CHECK_CUDA, auxiliary.h:297-304): while the call
    runs the library synchronises after every kernel and reports which one faulted (``isr_set_debug``); if it raises, the
    argument snapshot taken beforehand goes to ``snapshot_fw.dump`` / ``snapshot_bw.dump`` in the working directory
    (``torch.save``, as the reference does) and the exception propagates."""

    def __init__(self, args, which):
        self.snapshot, self.which = _cpu_snapshot(args), which

    def __enter__(self):
        self.was = lib().isr_set_debug(1, int(os.environ.get("ISR_DEBUG_FAULT_AFTER", "0")))
        return self

    def __exit__(self, exc_type, exc, tb):
        lib().isr_set_debug(self.was, 0)
        if exc_type is not None:
            name = "snapshot_fw.dump" if self.which == "forward" else "snapshot_bw.dump"
            torch.save(self.snapshot, name)
            print(f"\nAn error occured in {self.which}. Please forward {name} for debugging.")
        return False


class _Token:
    """Dies with the autograd context that holds it (see weakref.finalize in _RasterizeGaussians.forward)."""
    __slots__ = ("__weakref__",)


class _RasterizeGaussians(torch.autograd.Function):
    @staticmethod
    def forward(ctx, means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, extra_attrs,
                raster_settings, sample_pixels=None, lazy_tracer=False, feature_only=False, defer_rows=True):
        rs = raster_settings
        ctx.defer_rows = bool(defer_rows)
        if feature_only and any(ctx.needs_input_grad[i] for i in (0, 1, 2, 3, 4, 5, 6, 7)):
            raise Exception("feature_only forward: only extra_attrs may require grad")
        attr_degree = extra_attrs.shape[1] if extra_attrs.shape[0] != 0 else 0
        kept_state = []
        args = (rs.bg, means3D, colors_precomp, opacities, scales, rotations, rs.scale_modifier, cov3Ds_precomp,
                extra_attrs, attr_degree, rs.viewmatrix, rs.projmatrix, rs.tanfovx, rs.tanfovy, rs.image_height,
                rs.image_width, sh, rs.sh_degree, rs.campos, rs.prefiltered, rs.debug)
        kw = dict(_state_out=kept_state,
                  _verify_at_backward=any(ctx.needs_input_grad),     # (grad mode is off inside Function.forward: ask the context)
                  feature_only=feature_only)
        if rs.debug:
            with _debug_call(args, "forward"):
                res = rasterize_gaussians(*args, **kw)
        else:
            res = rasterize_gaussians(*args, **kw)
        (num_rendered, color, depth, radii, extra, geomBuffer, binningBuffer, imgBuffer, gau_related_pixels,
         gau_pixel_indices) = res
        if kept_state and any(ctx.needs_input_grad):
            # the cached view state now backs a pending backward: not reusable until that has run (or the graph is freed)
            kept_state[0].busy += 1
            ctx.view_token = _Token()
            weakref.finalize(ctx.view_token, kept_state[0].release)
        if gau_related_pixels.shape[0] and not lazy_tracer:
            gau_related_pixels = gau_related_pixels[:(gau_pixel_indices + 1)]   # same slicing as the reference (:106)
        elif gau_related_pixels.shape[0]:
            # render() wraps the pair in a lazily sliced dict entry: slicing needs the count on the host (a sync)
            gau_related_pixels = _attach_count(gau_related_pixels, gau_pixel_indices)
        ctx.raster_settings = rs
        ctx.num_rendered = num_rendered
        ctx.mode = _CONFIG["mode"]
        # extension: features read at sampled pixels only — their gradient never becomes a dense [F,H,W] map
        has_samples = sample_pixels is not None and attr_degree > 0
        sampled = sample_extra(extra, sample_pixels) if has_samples else torch.empty(0, device=color.device)
        ctx.sample_pixels = sample_pixels.detach() if has_samples else None
        # (a scaled-rows placeholder, contrastive.FeatureAdam.store_z = False: the attribute does not survive save_for_backward)
        ctx.scaled_src = getattr(extra_attrs, _SCALED_ROWS_ATTR, None)
        ctx.save_for_backward(colors_precomp, means3D, scales, rotations, cov3Ds_precomp, radii, extra_attrs, sh,
                              geomBuffer, binningBuffer, imgBuffer)
        ctx.mark_non_differentiable(radii, gau_related_pixels)
        ctx.set_materialize_grads(False)     # unused outputs arrive as None (= zeros) instead of dense zero tensors
        return color, radii, depth, extra, gau_related_pixels, sampled

    @staticmethod
    def backward(ctx, grad_out_color, grad_radii, grad_depth, grad_out_extra, grad_gau_related_pixels, grad_sampled=None):
        rs = ctx.raster_settings
        (colors_precomp, means3D, scales, rotations, cov3Ds_precomp, radii, extra_attrs, sh, geomBuffer, binningBuffer,
         imgBuffer) = ctx.saved_tensors
        # needs_input_grad order: means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3D, extra
        need = ctx.needs_input_grad
        mask = 0
        if any(need[i] for i in (0, 1, 2, 3, 4, 5, 6, 7)):
            mask |= GRAD_GEOMETRY
        if need[8] and extra_attrs.numel():
            mask |= GRAD_EXTRA
        if ctx.sample_pixels is None:
            grad_sampled = None
        if grad_sampled is not None and (mask & GRAD_GEOMETRY):
            # geometry gradients through the feature need the dense map gradient: scatter the samples into it
            Fm, Hm, Wm = extra_attrs.shape[1], rs.image_height, rs.image_width
            dense = torch.zeros((Fm, Hm * Wm), dtype=torch.float32, device=means3D.device) if grad_out_extra is None \
                else grad_out_extra.reshape(Fm, -1).clone()
            dense.index_add_(1, ctx.sample_pixels.to(torch.int64), grad_sampled.t().contiguous().float())
            grad_out_extra, grad_sampled = dense.reshape(Fm, Hm, Wm), None
        if mask == 0 or (grad_out_color is None and grad_depth is None and grad_out_extra is None and grad_sampled is None):
            return (None,) * 14
        if grad_sampled is not None and grad_out_color is None and grad_depth is None and grad_out_extra is None:
            # the common case of feature training: only sampled features carry gradient
            sink = _ROWS_SINK
            if (sink is not None and sink.rows is None and ctx.defer_rows and extra_attrs.shape[1] % 4 == 0
                    and extra_attrs.shape[1] <= 256):
                sink.rows = rasterize_gaussians_backward_sampled(
                    means3D.shape[0], extra_attrs.shape[1], rs.image_width, rs.image_height, ctx.num_rendered,
                    ctx.sample_pixels, grad_sampled, cov3Ds_precomp, geomBuffer, binningBuffer, imgBuffer, mode=ctx.mode,
                    rows_only=True)
                return (None,) * 14
            if sink is not None and sink.collect_dense:
                if sink.dense is None:
                    sink.dense = torch.zeros((means3D.shape[0], extra_attrs.shape[1]), dtype=torch.float32, device=means3D.device)
                rasterize_gaussians_backward_sampled(means3D.shape[0], extra_attrs.shape[1], rs.image_width, rs.image_height,
                                                     ctx.num_rendered, ctx.sample_pixels, grad_sampled, cov3Ds_precomp,
                                                     geomBuffer, binningBuffer, imgBuffer, mode=ctx.mode,
                                                     accumulate_into=sink.dense)
                return (None,) * 14
            ge = rasterize_gaussians_backward_sampled(means3D.shape[0], extra_attrs.shape[1], rs.image_width, rs.image_height,
                                                      ctx.num_rendered, ctx.sample_pixels, grad_sampled, cov3Ds_precomp,
                                                      geomBuffer, binningBuffer, imgBuffer, mode=ctx.mode)
            return (None,) * 8 + (ge, None, None, None, None, None)
        if getattr(ctx, "scaled_src", None) is not None:          # the dense backward reads the feature's values: make them
            extra_attrs = (ctx.scaled_src[0] * ctx.scaled_src[1][:, 0:1]) * ctx.scaled_src[1][:, 1:2]
        bargs = (rs.bg, means3D, radii, colors_precomp, scales, rotations, extra_attrs, rs.scale_modifier, cov3Ds_precomp,
                 rs.viewmatrix, rs.projmatrix, rs.tanfovx, rs.tanfovy, grad_out_color, grad_depth, grad_out_extra, sh,
                 rs.sh_degree, rs.campos, geomBuffer, ctx.num_rendered, binningBuffer, imgBuffer, rs.debug)
        if rs.debug:
            with _debug_call(bargs, "backward"):
                grads = rasterize_gaussians_backward(*bargs, grad_mask=mask, mode=ctx.mode)
        else:
            grads = rasterize_gaussians_backward(*bargs, grad_mask=mask, mode=ctx.mode)
        (grad_means2D, grad_colors_precomp, grad_opacities, grad_means3D, grad_cov3Ds_precomp, grad_sh, grad_scales,
         grad_rotations, grad_extra_attrs) = grads
        if grad_sampled is not None and grad_extra_attrs is not None and grad_extra_attrs.numel():
            grad_extra_attrs = rasterize_gaussians_backward_sampled(
                means3D.shape[0], extra_attrs.shape[1], rs.image_width, rs.image_height, ctx.num_rendered, ctx.sample_pixels,
                grad_sampled, cov3Ds_precomp, geomBuffer, binningBuffer, imgBuffer, accumulate_into=grad_extra_attrs,
                mode=ctx.mode)

        def pick(i, g, ref):
            if not need[i] or g is None or ref.numel() == 0:
                return None
            return g

        return (pick(0, grad_means3D, means3D), grad_means2D if need[1] else None, pick(2, grad_sh, sh),
                pick(3, grad_colors_precomp, colors_precomp), grad_opacities if need[4] else None,
                pick(5, grad_scales, scales), pick(6, grad_rotations, rotations),
                pick(7, grad_cov3Ds_precomp, cov3Ds_precomp),
                grad_extra_attrs if (need[8] and extra_attrs.numel()) else None, None, None, None, None, None)


def rasterize_gaussians_autograd(means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp,
                                 extra_attrs, raster_settings, sample_pixels=None, lazy_tracer=False, feature_only=False,
                                 defer_rows=True):
    out = _RasterizeGaussians.apply(means3D, means2D, sh, colors_precomp, opacities, scales, rotations,
                                    cov3Ds_precomp, extra_attrs, raster_settings, sample_pixels, lazy_tracer, feature_only,
                                    defer_rows)
    return out if sample_pixels is not None else out[:5]


class GaussianRasterizer(nn.Module):
    def __init__(self, raster_settings):
        super().__init__()
        self.raster_settings = raster_settings

    def markVisible(self, positions):
        with torch.no_grad():
            rs = self.raster_settings
            return mark_visible(positions, rs.viewmatrix, rs.projmatrix)

    def prefetch(self, means3D, opacities, shs=None, colors_precomp=None, scales=None, rotations=None,
                 cov3D_precomp=None, stream=None, after=None) -> bool:
        """Issue the geometry pass of the ``forward`` call that will follow with these inputs (extension; see
        :func:`prefetch_geometry`)."""
        rs = self.raster_settings
        with torch.no_grad():
            return prefetch_geometry(means3D, colors_precomp, opacities, scales, rotations, rs.scale_modifier,
                                     cov3D_precomp, rs.viewmatrix, rs.projmatrix, rs.tanfovx, rs.tanfovy, rs.image_height,
                                     rs.image_width, shs, rs.sh_degree, rs.campos, rs.prefiltered, stream=stream, after=after)

    def forward(self, means3D, means2D, opacities, shs=None, colors_precomp=None, scales=None, rotations=None,
                cov3D_precomp=None, extra_attrs=None, sample_pixels=None, lazy_tracer=False, feature_only=False,
                defer_rows=True):
        """Reference signature (:210-248).  Extensions: ``sample_pixels`` (int64 ``y*W + x``, may repeat) appends a sixth
        result, the feature map read at those pixels ``[n, F]``; its gradient is propagated without a dense map.
        ``lazy_tracer``: return the whole tracer buffer with its count attached (``slice_tracer``) instead of slicing it
        here, which needs the count on the host, i.e. a device sync (``render()`` slices on first access).
        ``feature_only``: see :func:`rasterize_gaussians` (colour / allmap / tracer come back empty).  ``defer_rows=False``:
        this render's sampled backward never hands its partial rows to a ``DeferredFeatureRows`` block (a trainer that renders
        several views inside one block names the ONE render whose rows its fused tail consumes)."""
        rs = self.raster_settings
        if (shs is None and colors_precomp is None) or (shs is not None and colors_precomp is not None):
            raise Exception('Please provide excatly one of either SHs or precomputed colors!')
        if ((scales is None or rotations is None) and cov3D_precomp is None) or (
                (scales is not None or rotations is not None) and cov3D_precomp is not None):
            raise Exception('Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!')
        dev = means3D.device
        empty = lambda: torch.empty(0, dtype=torch.float32, device=dev)
        shs = empty() if shs is None else shs
        colors_precomp = empty() if colors_precomp is None else colors_precomp
        scales = empty() if scales is None else scales
        rotations = empty() if rotations is None else rotations
        cov3D_precomp = empty() if cov3D_precomp is None else cov3D_precomp
        extra_attrs = empty() if extra_attrs is None else extra_attrs
        return rasterize_gaussians_autograd(means3D, means2D, shs, colors_precomp, opacities, scales, rotations,
                                            cov3D_precomp, extra_attrs, rs, sample_pixels, lazy_tracer, feature_only, defer_rows)
