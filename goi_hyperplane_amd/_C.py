"""`_C`: the four functions the reference's pybind module exports
(submodules/diff-gaussian-rasterization/ext.cpp:15-20), re-implemented over the C ABI of
libgoi_raster.so.  Positional signatures, return tuples, the "empty tensor = absent" convention and
the error behaviour mirror rasterize_points.cu:35-123 (forward), :125-211 (trace), :213-306
(backward), :308-327 (mark_visible).  Tensors stay torch tensors on this side; only raw device
pointers, sizes and the current HIP stream cross into the library.
"""
from __future__ import annotations

import collections
import ctypes as C
import os
import threading
import warnings
import weakref

import torch

from . import _lib
from ._lib import ALLOC_FN, GoiRasterScene


# ---- which binding crosses into libgoi_raster.so ----------------------------------------------------------------------
# "compiled": goi_hyperplane_amd/lib/_goi_C.so, the pybind/torch C++ binding a maintainer of the reference would build in
# place of rasterize_points.cu (csrc/torch_binding.cpp; host C++ only) -- tensor allocation, checks and the C-ABI call
# happen in C++, one Python->C++ transition per operator call.  "ctypes": this file's own ctypes calls.  Same library,
# same kernels, bit-identical results; the compiled one costs less host time per call and is the default when it has
# been built (GOI_BINDING=ctypes|compiled, set_binding()).
_EXT = {"mod": None, "tried": False, "want": os.environ.get("GOI_BINDING", "compiled").strip().lower()}
_EMPTY = torch.Tensor([])


def _ext():
    if _EXT.get("error") is not None:
        raise _EXT["error"]  # (a stale binding fails EVERY call the same way -- never a quiet switch to ctypes)
    if not _EXT["tried"]:
        _EXT["tried"] = True
        path = os.path.join(os.path.dirname(_lib.LIB_PATH), "_goi_C.so")
        if _EXT["want"] == "compiled" and os.path.exists(path):
            import importlib.util
            _lib.load()  # same libgoi_raster.so instance (tickets, options and last_error are shared)
            spec = importlib.util.spec_from_file_location("_goi_C", path)
            mod = importlib.util.module_from_spec(spec)
            spec.loader.exec_module(mod)
            # abi_version(): the header the binding was COMPILED against; library_abi_version(): what the library it is
            # linked with reports -- a stale _goi_C.so next to a newer libgoi_raster.so fails the first test, and one so old
            # that it has no library_abi_version at all is a mismatch too (not an AttributeError)
            built = getattr(mod, "abi_version", lambda: None)()
            linked = getattr(mod, "library_abi_version", lambda: None)()
            if built != _lib.ABI_VERSION or linked != _lib.ABI_VERSION:
                _EXT["error"] = ImportError(f"{path}: built against ABI {built} (library: {linked}), expected "
                                            f"{_lib.ABI_VERSION}; rebuild (python -m goi_hyperplane_amd.build)")
                raise _EXT["error"]
            _EXT["mod"] = mod
    return _EXT["mod"]


def set_binding(name: str) -> None:
    """"compiled" (needs goi_hyperplane_amd/lib/_goi_C.so) or "ctypes"."""
    if name not in ("compiled", "ctypes"):
        raise ValueError("binding must be 'compiled' or 'ctypes'")
    _EXT.update(want=name, tried=False, mod=None, error=None)
    if name == "compiled" and _ext() is None:
        raise ImportError("the compiled binding has not been built (python -m goi_hyperplane_amd.build)")


def binding() -> str:
    return "compiled" if _ext() is not None else "ctypes"


def _e(t):
    """None -> the reference's marker for an absent tensor argument (an empty tensor)"""
    return _EMPTY if t is None else t


def _ptr(t):
    """Device pointer of a tensor, or None for the reference's 'empty tensor means absent'."""
    if t is None or t.numel() == 0:
        return None
    return C.c_void_p(t.data_ptr())


def _prep(t, name, dev, dtype=torch.float32):
    """contiguous + checked view of an input (the reference calls .contiguous() and assumes fp32)."""
    if t is None or t.numel() == 0:
        return None
    if t.dtype != dtype:
        raise TypeError(f"{name} must be {dtype}, got {t.dtype}")
    if t.device != dev:
        raise ValueError(f"{name} is on {t.device}, expected {dev}")
    return t.contiguous()


def _stream(dev):
    return C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)


def _check_device(means3D):
    if means3D.ndimension() != 2 or means3D.size(1) != 3:
        raise RuntimeError("means3D must have dimensions (num_points, 3)")  # rasterize_points.cu:58-60
    if not means3D.is_cuda:
        raise RuntimeError(
            "goi_hyperplane_amd: tensors must live on a ROCm GPU (cuda device); there is no CPU fallback in this "
            "package")
    return means3D.device


class _BinningAllocator:
    """The allocation callback handed to the library (reference: resizeFunctional,
    rasterize_points.cu:27-33): allocates the binning workspace as a uint8 torch tensor."""

    def __init__(self, dev):
        self.dev = dev
        self.tensor = torch.empty(0, dtype=torch.uint8, device=dev)
        self.error = None
        self.cb = ALLOC_FN(self._alloc)

    def _alloc(self, _user, nbytes):
        try:
            # num_rendered changes with every view; rounding the request up to 16 MiB steps lets the caching allocator
            # hand back the same block instead of growing a new one (a fresh hipMalloc inside a training step)
            step = 16 << 20
            self.tensor = torch.empty((int(nbytes) + step - 1) // step * step, dtype=torch.uint8, device=self.dev)
            return self.tensor.data_ptr()
        except Exception as ex:  # never let an exception cross the C boundary
            self.error = ex
            return None


_SCRATCH = {}


def _backward_scratch(nbytes: int, dev) -> torch.Tensor:
    """Grow-only backward scratch per (device, stream): the buffer is 4*R*129 bytes (4 GB at the headline
    size) and R changes with every view, so going through the caching allocator each step leaves it
    hunting for a block of a new size -- an occasional 4 GB hipMalloc inside a training step.  The
    scratch is dead when goi_raster_backward returns (same stream), so one buffer serves all calls."""
    key = (dev.index, torch.cuda.current_stream(dev).cuda_stream)
    buf = _SCRATCH.get(key)
    if buf is None or buf.numel() < nbytes:
        _SCRATCH.pop(key, None)
        buf = None
        buf = torch.empty(int(nbytes * 1.25) + 256, dtype=torch.uint8, device=dev)
        _SCRATCH[key] = buf
    return buf


def release_scratch(device=None) -> int:
    """Frees the grow-only backward scratch buffers (all devices, or one): they are dead between calls, hold
    ~1.25 x the largest view's 4*R*129 bytes per (device, stream) and are invisible to torch.cuda.empty_cache()
    while referenced here.  Returns the number of bytes released to the caching allocator."""
    freed = 0
    for key in [k for k in _SCRATCH if device is None or k[0] == torch.device(device).index]:
        freed += _SCRATCH.pop(key).numel()
    if _EXT["mod"] is not None:
        freed += int(_EXT["mod"].release_scratch())  # (the compiled binding keeps its own; all devices)
    return freed


# ---- speculative forward ---------------------------------------------------------------------------------------------
# The reference's forward blocks on num_rendered (cuda_rasterizer/rasterizer_impl.cu:285) to size the binning workspace;
# a 1.6 ms training step cannot afford a synchronous host hop (every host hiccup idles the GPU one for one).  In
# speculative mode (the default) the forward is enqueued whole against a binning buffer sized from the counts seen so
# far (headroom x the high-water mark), the count comes back through an asynchronous ticket, and `num_rendered` is
# returned as a LazyCount that only waits when somebody reads it.  See include/goi_raster.h, goi_raster_forward_async.
#
#   * the first frame on a device (no history), P == 0, debug=True, sort_variant != 1 and mode "exact" take the exact,
#     synchronous path;
#   * a frame whose count fits its capacity is bit-identical to the exact path's (outputs, lists, gradients);
#   * OVERFLOW (count > capacity): the frame was rendered from a truncated instance list.  Whoever reads the count
#     (int(num_rendered), .resolve()) BEFORE consuming the outputs gets the frame redone in place -- bit-identical to the
#     exact path.  If nobody asks, the DEVICE still knows: emit sets the frame's "truncated" word and every backward
#     kernel of such a frame writes ZERO gradients (csrc/common.h COUNTER_OVF), so the view trains nothing -- a skipped
#     view, not a wrong one -- and FusedAdam.step(skip_if=truncated_flag()) leaves parameters and moments untouched.
#     The host finds out at a later poll: a RasterOverflowWarning (GOI_OVERFLOW=raise: a RasterOverflowError) names
#     the skipped view and the capacity is raised.
#   * the first `min_history` frames of a scene on a device are exact (they teach the capacity policy); a frame whose
#     Gaussian count or image size (in tiles) differs from the previous frame's by more than a factor of two starts over;
#   * at most `max_ahead` frames stay unresolved per device; the oldest is waited for beyond that (flow control: the
#     host may run that far ahead of the GPU, which is what absorbs a host stall).  A pending frame pins no device
#     memory: the LazyCount refers to the frame's tensors weakly (a frame whose outputs have died has no consumer
#     left to repair for).
class RasterOverflowWarning(UserWarning):
    pass


class RasterOverflowError(RuntimeError):
    pass


def _env_forward_mode():
    v = os.environ.get("GOI_FORWARD", "speculative").strip().lower()
    if v not in ("speculative", "exact"):
        raise ValueError(f"GOI_FORWARD={v!r}: expected speculative or exact")
    return v


def _env_inference_mode():
    v = os.environ.get("GOI_FORWARD_INFERENCE", "exact").strip().lower()
    if v not in ("speculative", "exact"):
        raise ValueError(f"GOI_FORWARD_INFERENCE={v!r}: expected speculative or exact")
    return v


# A frame nobody can backpropagate through (torch.no_grad(), or no input requires a gradient: a GUI frame, an evaluation
# render) is rendered for its IMAGE, and the caller usually hands that image to the host right away: such frames take the
# exact forward by default -- a truncated picture is not worth the host round trip it saves.  The autograd wrapper
# (rasterizer.rasterize_gaussians) says which kind a frame is; the raw operator called directly counts as a training frame.
_CALL = threading.local()

_FWD = {"mode": _env_forward_mode(), "inference": _env_inference_mode(),
        "headroom": float(os.environ.get("GOI_BINNING_HEADROOM", "2.0")),
        "capacity": None, "on_overflow": os.environ.get("GOI_OVERFLOW", "warn").strip().lower(),
        "max_ahead": int(os.environ.get("GOI_MAX_AHEAD", "64")),
        "min_history": int(os.environ.get("GOI_SPECULATE_AFTER", "3")),
        "depth_cut": os.environ.get("GOI_DEPTH_CUT", "0").strip().lower() in ("1", "on", "true", "yes")}
_SPEC = {}  # device index -> {"high_water": int, "P": int, "pending": deque of LazyCount}
_SPEC_LOCK = threading.RLock()
SPECULATION_STATS = {"exact_frames": 0, "speculative_frames": 0, "overflows": 0, "redone": 0, "waits": 0, "cached_frames": 0,
                     "skipped_views": 0, "cut_frames": 0, "cut_failures": 0, "cut_overflows": 0}
_MIN_CAPACITY = 1 << 16
_KEEP_WORKSPACES = 2  # pending frames per device whose workspaces stay alive for a possible redo


def set_forward_mode(speculative=None, headroom=None, capacity="keep", on_overflow=None, max_ahead=None,
                     inference_speculative=None, min_history=None, depth_cut=None):
    """speculative: True / False (exact, the reference's synchronous forward) for frames a backward may follow.
    inference_speculative: the same for frames rendered without autograd (default False: exact).  headroom: capacity = headroom x the
    largest num_rendered seen on the device.  capacity: an int forces that capacity for every frame (tests), None returns
    to the policy.  on_overflow: "warn" | "raise" for overflows found after the fact.  max_ahead: unresolved frames
    allowed per device."""
    if speculative is not None:
        _FWD["mode"] = "speculative" if speculative else "exact"
    if inference_speculative is not None:
        _FWD["inference"] = "speculative" if inference_speculative else "exact"
    if headroom is not None:
        if not headroom >= 1.0:
            raise ValueError("headroom must be >= 1")
        _FWD["headroom"] = float(headroom)
    if capacity != "keep":
        _FWD["capacity"] = None if capacity is None else max(1, int(capacity))
    if on_overflow is not None:
        if on_overflow not in ("warn", "raise"):
            raise ValueError("on_overflow must be 'warn' or 'raise'")
        _FWD["on_overflow"] = on_overflow
    if max_ahead is not None:
        _FWD["max_ahead"] = max(1, int(max_ahead))
    if min_history is not None:  # counts seen on a device (for the current scene) before frames are speculative
        _FWD["min_history"] = max(1, int(min_history))
    if depth_cut is not None:  # speculative depth cut-off of the tile lists (see _depth_cut_for): False also forgets what was learnt
        _FWD["depth_cut"] = bool(depth_cut)
        if not depth_cut:
            forget_depth_cuts()


def _spec_state(dev):
    st = _SPEC.get(dev.index)
    if st is None:
        st = _SPEC[dev.index] = {"high_water": 0, "P": 0, "seen": 0, "pending": collections.deque()}
    return st


def _other_scene(st, P, tiles):
    """A frame that is not comparable with the ones the capacity policy has learnt from: the Gaussian count OR the image
    (in tiles: num_rendered scales with it) changed by more than a factor of two.  tiles = 0: unknown (not checked)."""
    if P > 2 * st["P"] or 2 * P < st["P"]:
        return True
    t0 = st.get("tiles", 0)
    return bool(tiles and t0 and (tiles > 2 * t0 or 2 * tiles < t0))


def _tiles(H, W):
    return ((int(W) + 15) // 16) * ((int(H) + 15) // 16)


def _note_count(dev, P, n, tiles=0, cut=False):
    st = _spec_state(dev)
    if _other_scene(st, P, tiles):  # another scene / another image size: forget what the previous one needed
        st["high_water"] = 0
        st["high_water_cut"] = 0
        st["seen"] = 0
    st["P"] = P
    if tiles:
        st["tiles"] = tiles
    if cut:  # a frame whose lists were cut at its camera's learnt depths holds far fewer instances: its own high-water mark
        st["high_water_cut"] = max(st.get("high_water_cut", 0), int(n))
        return
    st["high_water"] = max(st["high_water"], int(n))
    st["seen"] = st.get("seen", 0) + 1


def _pick_capacity(dev, P, debug, prefiltered, tiles=0, cut=False):
    """Instances to size a speculative frame for, or None for an exact frame.  (prefiltered=True promises something the
    kernel checks and the reference traps on: that error must surface in THIS call, so such frames stay exact.)"""
    if (P == 0 or debug or prefiltered or _FWD["mode"] != "speculative"
            or _lib.OPTIONS.get("sort_variant", 1) != 1):
        return None
    if getattr(_CALL, "inference", False) and _FWD["inference"] != "speculative" and _FWD["capacity"] is None:
        return None  # an image-only frame: exact unless asked otherwise (a forced capacity, as in the tests, overrides)
    if _FWD["capacity"] is not None:
        return _FWD["capacity"]
    st = _spec_state(dev)
    if st["high_water"] <= 0 or _other_scene(st, P, tiles) or st.get("seen", 0) < _FWD["min_history"]:
        return None  # nothing (or too little) to go by yet: this frame is exact and teaches the policy
    if cut and st.get("high_water_cut", 0) > 0:  # (the first cut frames of a scene are sized like uncut ones)
        return max(_MIN_CAPACITY, int(_FWD["headroom"] * st["high_water_cut"]) + 4096)
    return max(_MIN_CAPACITY, int(_FWD["headroom"] * st["high_water"]) + 4096)


def poll_counts(dev=None, wait=False):
    """Resolve what can be resolved without waiting (wait=True: everything) on one device or all.  Called at the start
    of every forward; an overflow found here is reported per set_forward_mode(on_overflow=...)."""
    with _SPEC_LOCK:
        for idx, st in list(_SPEC.items()):
            if dev is not None and torch.device(dev).index != idx:
                continue
            pend = st["pending"]
            while pend:
                if not pend[0]._resolve(wait=wait or len(pend) > _FWD["max_ahead"], lazy=True):
                    break


class LazyCount:
    """`num_rendered` of a speculative frame: behaves like the int the reference returns, but the value only crosses to
    the host when it is read.  Reading it (int(), comparisons, arithmetic, .resolve()) waits for the frame's counters
    and, if the frame overflowed its capacity, redoes it in place first."""

    __slots__ = ("dev", "ticket", "capacity", "layout", "binning", "overflowed", "redone", "_n", "_redo", "_stream",
                 "_error", "P", "tiles", "_hold", "cut_key", "cam_key", "cut_failed", "_full_args", "__weakref__")

    def __init__(self, dev, ticket, capacity, binning, stream, redo, P, workspaces=None):
        self.dev, self.ticket, self.capacity, self.layout, self.binning = dev, ticket, capacity, capacity, binning
        self.overflowed = self.redone = False
        self._n, self._redo, self._stream, self._error, self.P = None, redo, stream, None, P
        self.tiles = 0  # (set by the caller: the image size is part of what the capacity policy compares)
        self.cam_key = None      # key of this frame's camera in the depth-cut registry (its count is reported back there)
        self.cut_key = None      # ... set iff the frame's lists were built with the camera's learnt depth cut
        self.cut_failed = False  # ... and the cut turned out too tight for this frame
        self._full_args = None   # the operator's arguments: what rendering the frame AGAIN without a cut needs
        # The frame's workspaces (geometry / image state, radii) are needed for a redo but belong to nobody once the
        # operator has returned under no_grad: the newest few pending frames of a device keep them alive (a caller who
        # reads the count does so right after the forward), older ones let go (a pending frame must not pin memory).
        self._hold = workspaces
        pend = _spec_state(dev)["pending"]
        pend.append(self)
        if len(pend) > _KEEP_WORKSPACES:
            pend[-1 - _KEEP_WORKSPACES]._hold = None
            pend[-1 - _KEEP_WORKSPACES]._full_args = None

    @property
    def resolved(self):
        return self._n is not None or self._error is not None

    def _resolve(self, wait, lazy):
        with _SPEC_LOCK:  # (several host threads may poll the same pending frame)
            return self._resolve_locked(wait, lazy)

    def _resolve_locked(self, wait, lazy):
        if self._error is not None:
            if lazy:
                return True
            raise self._error
        if self._n is not None:
            return True
        lib = _lib.load()
        n = C.c_int(0)
        fl = C.c_uint(0)
        if wait:
            SPECULATION_STATS["waits"] += 1
        r = lib.goi_raster_ticket_result2(self.ticket, 1 if wait else 0, C.byref(n), C.byref(fl))
        if r == 0:
            return False
        self.ticket = None
        pend = _spec_state(self.dev)["pending"]
        for idx, item in enumerate(pend):  # by IDENTITY: deque.remove() compares with ==, which on a LazyCount means
            if item is self:                # "resolve and compare the counts"
                del pend[idx]
                break
        if r < 0:
            self._redo = self._hold = None
            self._error = RuntimeError(_lib.last_error())
            raise self._error
        self._n = int(n.value)
        _note_count(self.dev, self.P, self._n, getattr(self, "tiles", 0), cut=self.cut_key is not None)
        if self.cam_key is not None:
            ce = _DEPTH_CUTS["entries"].get(self.cam_key)
            if ce is not None:
                ce["n"] = self._n  # (the camera's most recent count: sizes its next frame)
        if self.cut_key is not None and ((fl.value & 4) or self._n > self.capacity):
            # The depth cut this frame's lists were built with was TOO TIGHT: some pixel reached the end of a cut list
            # unsaturated, or looked at an entry beyond its tile's cut.  The device has already made the frame harmless (zero
            # gradients); the camera forgets its cut.  A cut frame that OVERFLOWED its (cut-sized) capacity is treated the same
            # way: the redo of an overflowed frame re-bins from the cut geometry workspace and blends WITHOUT the cut check, so
            # a too-tight cut would come back as "exact"  -- such a frame is rendered again whole and uncut instead.
            # The two signals are kept apart: the device's flag says the CUT was too tight -- the camera forgets it;
            # a count above the capacity alone says the CAPACITY guess was short -- the cut may be fine and is kept (the
            # capacity policy has just learnt the count: the camera's next frame fits), only this frame is rendered again.
            self.cut_failed = True
            if fl.value & 4:
                SPECULATION_STATS["cut_failures"] += 1
                _DEPTH_CUTS["entries"].pop(self.cut_key, None)
            else:
                SPECULATION_STATS["cut_overflows"] += 1
            if lazy:
                self._redo = self._hold = self._full_args = None
                SPECULATION_STATS["skipped_views"] += 1
                if not _DEPTH_CUTS.get("warned"):
                    _DEPTH_CUTS["warned"] = True
                    warnings.warn("goi_hyperplane_amd: a frame's speculative depth cut-off was too tight (the scene has moved since "
                                  "its camera was rendered last) and nobody read num_rendered before using the frame: its "
                                  "backward produced ZERO gradients (the view was skipped); the camera re-learns its cut on its "
                                  "next visit.  GOI_DEPTH_CUT=0 / set_forward_mode(depth_cut=False) turns the cut-off off. "
                                  "(Further occurrences are only counted: speculation_stats()['cut_failures'].)",
                                  RasterOverflowWarning, stacklevel=3)
            else:
                try:
                    self._render_again_uncut()
                finally:
                    self._redo = self._hold = self._full_args = None
            return True
        if self._n > self.capacity:
            self.overflowed = True
            SPECULATION_STATS["overflows"] += 1
            if lazy:
                # found after the fact: whatever consumed the outputs is already enqueued.  Nothing was trained on the
                # truncated frame: its backward wrote zero gradients on the device (COUNTER_OVF) -- a skipped view.
                self._redo = self._hold = None
                SPECULATION_STATS["skipped_views"] += 1
                msg = (f"goi_hyperplane_amd: a speculative forward overflowed its binning capacity (num_rendered = "
                       f"{self._n} > {self.capacity}) and nobody read num_rendered before using the frame: its image "
                       f"was that of a truncated instance list and its backward produced ZERO gradients (the view was "
                       f"skipped, nothing was trained on it). The capacity has been raised; use GOI_BINNING_HEADROOM / "
                       f"set_forward_mode(headroom=...) or GOI_FORWARD=exact to avoid skipped views.")
                if _FWD["on_overflow"] == "raise":
                    raise RasterOverflowError(msg)
                warnings.warn(msg, RasterOverflowWarning, stacklevel=3)
            else:
                try:
                    self._redo_frame()
                finally:
                    self._redo = self._hold = None
        self._redo = self._hold = None
        return True

    def _redo_frame(self):
        lib = _lib.load()
        make_scene, refs = self._redo
        live = [r() for r in refs]
        if all(t is None for t in live[3:7]):
            return  # every output has been released: nobody can consume the truncated frame, nothing to repair
        if any(t is None for t in live):
            raise RasterOverflowError(
                f"a speculative forward overflowed its binning capacity (num_rendered = {self._n} > {self.capacity}) and "
                "cannot be redone: its workspaces have already been released (the count was read too late -- the last "
                f"{_KEEP_WORKSPACES} frames of a device keep theirs)")
        sc, _keep = make_scene()
        geom, img, radii, outs = live[0], live[1], live[2], live[3:7]
        stream = torch.cuda.ExternalStream(self._stream, device=self.dev)
        with torch.cuda.device(self.dev), torch.cuda.stream(stream):
            step = 16 << 20
            need = int(lib.goi_raster_binning_bytes(self._n))
            binning = torch.empty((need + step - 1) // step * step, dtype=torch.uint8, device=self.dev)
            r = lib.goi_raster_forward_redo(C.byref(sc), self._n, _ptr(geom), _ptr(img), _ptr(binning),
                                            *[_ptr(o) for o in outs], _ptr(radii), C.c_void_p(self._stream))
        if r < 0:
            raise RuntimeError(_lib.last_error())
        self.binning, self.layout, self.redone = binning, self._n, True
        SPECULATION_STATS["redone"] += 1

    def _render_again_uncut(self):
        """The frame once more, whole and exact (goi_raster_forward), into the same outputs and workspaces."""
        lib = _lib.load()
        if self._full_args is None or self._redo is None:
            raise RasterOverflowError("a frame's depth cut-off was too tight and the frame cannot be rendered again: its inputs "
                                      f"have been released (the count was read too late -- the last {_KEEP_WORKSPACES} frames of "
                                      "a device keep theirs)")
        _mk, refs = self._redo
        live = [r() for r in refs]
        if all(t is None for t in live[3:7]):
            return
        if any(t is None for t in live):
            raise RasterOverflowError("a frame's depth cut-off was too tight and the frame cannot be rendered again: its "
                                      "workspaces have already been released")
        geom, img, radii, outs = live[0], live[1], live[2], live[3:7]
        (background, means3D, colors, semantics, opacity, scales, rotations, scale_modifier, cov3D_precomp, viewmatrix,
         projmatrix, tan_fovx, tan_fovy, H, W, sh, degree, campos, prefiltered, debug) = self._full_args
        dev = self.dev
        stream = torch.cuda.ExternalStream(self._stream, device=dev)
        with torch.cuda.device(dev), torch.cuda.stream(stream):
            ten = dict(bg=_prep(background, "background", dev), means3D=_prep(means3D, "means3D", dev), sh=_prep(sh, "sh", dev),
                       colors=_prep(colors, "colors_precomp", dev), semantics=_prep(semantics, "semantics", dev),
                       opacity=_prep(opacity, "opacities", dev), scales=_prep(scales, "scales", dev),
                       rotations=_prep(rotations, "rotations", dev), cov3D=_prep(cov3D_precomp, "cov3D_precomp", dev),
                       viewmatrix=_prep(viewmatrix, "viewmatrix", dev), projmatrix=_prep(projmatrix, "projmatrix", dev),
                       campos=_prep(campos, "campos", dev))
            sc = _scene(self.P, int(semantics.size(1)), int(H), int(W), ten["bg"], ten["means3D"], ten["sh"], ten["colors"],
                        ten["semantics"], ten["opacity"], ten["scales"], ten["rotations"], scale_modifier, ten["cov3D"],
                        ten["viewmatrix"], ten["projmatrix"], tan_fovx, tan_fovy, degree, ten["campos"], prefiltered, False)
            alloc = _BinningAllocator(dev)
            n = lib.goi_raster_forward(C.byref(sc), _ptr(geom), _ptr(img), alloc.cb, None, *[_ptr(o) for o in outs], _ptr(radii),
                                       C.c_void_p(self._stream))
        if alloc.error is not None:
            raise alloc.error
        if n < 0:
            raise RuntimeError(_lib.last_error())
        self.binning, self.layout, self.redone, self._n = alloc.tensor, int(n), True, int(n)
        SPECULATION_STATS["redone"] += 1
        _note_count(dev, self.P, int(n), getattr(self, "tiles", 0))

    def resolve(self) -> int:
        """Waits for the count; an overflowed frame is redone in place (exact outputs from here on)."""
        self._resolve(wait=True, lazy=False)
        return self._n

    def ready(self) -> bool:
        return self._resolve(wait=False, lazy=False)

    __int__ = __index__ = lambda self: self.resolve()
    __bool__ = lambda self: self.resolve() != 0
    __eq__ = lambda self, o: self.resolve() == o
    __ne__ = lambda self, o: self.resolve() != o
    __lt__ = lambda self, o: self.resolve() < o
    __le__ = lambda self, o: self.resolve() <= o
    __gt__ = lambda self, o: self.resolve() > o
    __ge__ = lambda self, o: self.resolve() >= o
    __add__ = __radd__ = lambda self, o: self.resolve() + o
    __sub__ = lambda self, o: self.resolve() - o
    __rsub__ = lambda self, o: o - self.resolve()
    __mul__ = __rmul__ = lambda self, o: self.resolve() * o
    __truediv__ = lambda self, o: self.resolve() / o
    __floordiv__ = lambda self, o: self.resolve() // o
    __hash__ = lambda self: id(self)
    __float__ = lambda self: float(self.resolve())

    def __repr__(self):
        return str(self._n) if self._n is not None else f"<LazyCount pending, capacity {self.capacity}>"

    __str__ = __repr__
    __format__ = lambda self, spec: format(self.resolve(), spec)


def _flag_view(geom, P):
    if geom is None or P <= 0 or geom.numel() == 0:
        return None
    ptr = _lib.load().goi_raster_truncated_flag(C.c_void_p(geom.data_ptr()), P)
    if not ptr:
        return None
    off = int(ptr) - int(geom.data_ptr())
    return geom[off:off + 4].view(torch.int32)


def _note_frame(geom, P):
    """Remembers the geometry workspace of this thread's most recent frame (truncated_flag).  A STRONG reference, on purpose:
    the natural caller asks for the flag after loss.backward() has released the frame's graph (opt.step(skip_if=...)), when
    nothing else keeps the workspace alive; a 4-byte copy instead would cost every frame a kernel.  What it pins is ONE
    geometry workspace (~105 bytes per Gaussian) until the thread's next frame replaces it -- release_last_frame() lets go
    earlier.  With a truncation window open (accumulate_truncation) the frame's "truncated" word is also OR-ed into the
    window's accumulator on the device."""
    _CALL.last_frame = (geom if isinstance(geom, torch.Tensor) else None, int(P))
    acc = getattr(_CALL, "trunc_acc", None)
    if acc is not None:
        f = _flag_view(geom, int(P))
        if f is not None and f.device == acc.device:
            acc.bitwise_or_(f)  # (enqueued behind the frame on the same stream: nothing waits)


def accumulate_truncation(device=None):
    """Opens (device given) or closes (None) a TRUNCATION WINDOW on this thread: from now on every forward ORs its frame's
    "truncated" word into one int32[1] device tensor, which truncated_flag(accumulated=True) hands out -- the flag to pass
    to FusedAdam.step(skip_if=...) when SEVERAL views are accumulated into one optimiser step (truncated_flag() alone
    reports the last frame only and would miss a truncated earlier view).  Costs one 4-byte device operation per frame
    while open.  reset_truncation() starts the next window."""
    _CALL.trunc_acc = None if device is None else torch.zeros(1, dtype=torch.int32, device=device)


def release_last_frame():
    """Drops this thread's reference to its most recent frame's geometry workspace (truncated_flag() returns None until the
    next forward)."""
    _CALL.last_frame = None


def reset_truncation():
    acc = getattr(_CALL, "trunc_acc", None)
    if acc is not None:
        acc.zero_()


def truncated_flag(accumulated: bool = False):
    """int32[1] device tensor: the "truncated" word of this thread's most recent forward (a view into its geometry
    workspace), non-zero iff that frame's instance list did not fit the capacity of a speculative forward (or its lists
    could not be sorted) -- in which case its backward writes zero gradients on the device.
    FusedAdam.step(skip_if=truncated_flag()) then leaves parameters and moments untouched for that view: nothing on the host
    ever waits.  NOTE that the skip is the CALLER's to ask for: a plain step() after a truncated frame sees zero gradients,
    which still moves every parameter by its momentum and decays both moments.
    accumulated=True: the OR over every frame since the window was opened / reset (accumulate_truncation), for optimiser
    steps that accumulate several views.  None if there was no frame (or P == 0)."""
    if accumulated:
        return getattr(_CALL, "trunc_acc", None)
    last = getattr(_CALL, "last_frame", None)
    if last is None or last[0] is None:
        return None
    return _flag_view(last[0], last[1])


# ---- speculative depth cut-off of the tile lists (include/goi_raster.h, goi_raster_forward_async_cut) ----------------------------
# On an opaque scene three quarters of the (tile, Gaussian) instances lie behind their tile's saturation front: emitted, sorted
# and never looked at.  A training loop renders the same cameras epoch after epoch, so every speculative training frame LEARNS, per
# tile, the depth up to which its list was worth listing (with a margin of a quarter + 64 list positions), and the next frame of
# the same camera lists nothing deeper.  A frame whose pixels all saturate inside their cut lists is bit-identical to the uncut
# one.  If the scene has moved so far that a pixel wants more than its cut list holds, the DEVICE notices (the forward blend raises
# the frame's flag), the frame back-propagates zeros like a truncated one, and the host -- at its next poll, or at once if the
# count is read before the frame is used -- renders it again without a cut / counts a skipped view and forgets the camera's cut.
# Keyed like the geometry cache: by the identity and version of the camera's three tensors (kept alive by the entry), the image
# and the Gaussian count.  OPT-IN (GOI_DEPTH_CUT=1 / set_forward_mode(depth_cut=True)): measured on the headline workload it lists
# 1.90 M instead of 4.10 M instances per view but returns only 8 us of the 1451 us step (emit is bound by its
# per-Gaussian gathers, not by the instances it writes; the per-tile depth lookups cost preprocess what the sorts gain).
_DEPTH_CUTS = {"entries": collections.OrderedDict(), "max_entries": int(os.environ.get("GOI_DEPTH_CUT_CAMERAS", "4096"))}


def forget_depth_cuts() -> None:
    _DEPTH_CUTS["entries"].clear()


def depth_cut_stats() -> dict:
    return {"cameras": len(_DEPTH_CUTS["entries"]), "cut_frames": SPECULATION_STATS["cut_frames"],
            "cut_failures": SPECULATION_STATS["cut_failures"]}


def _depth_cut_for(dev, P, H, W, viewmatrix, projmatrix, campos, tan_fovx, tan_fovy, scale_modifier):
    """-> (key, zcut_in or None, zcut_out, n_prev) for a speculative frame, or (None, None, None, 0) when the cut-off does not
    apply.  n_prev: num_rendered of this camera's most recent frame whose count has arrived (0: none yet) -- what a cut frame's
    binning capacity is sized from: a cut frame holds no more instances than the camera's last frame did, and the counts of
    OTHER cameras (or of another scene of the same size) say nothing about it."""
    if not _FWD["depth_cut"] or _lib.OPTIONS.get("cull_variant", 2) != 2:
        return None, None, None, 0
    key = (dev.index, torch.cuda.current_stream(dev).cuda_stream, int(P), int(H), int(W), float(tan_fovx), float(tan_fovy),
           float(scale_modifier), _ident(viewmatrix), _ident(projmatrix), _ident(campos))
    ents = _DEPTH_CUTS["entries"]
    e = ents.get(key)
    z_in = None
    if e is not None:
        ents.move_to_end(key)
        z_in = e["z"]
    n_prev = e.get("n", 0) if e is not None else 0
    # (+inf = "no cut": the array is registered as the camera's cut before the frame is enqueued, and a launch that raises
    # must not leave uninitialised memory behind as the next frame's zcut_in; the frame's first kernel clears it for learning)
    z_out = torch.full((_tiles(H, W),), float("inf"), dtype=torch.float32, device=dev)
    # what this frame learns is the camera's cut from now on: the next frame of the camera runs behind this one on the same
    # stream.  (A too-tight frame learns +inf for the tiles that failed: the array is conservative whatever became of the frame.)
    ents[key] = {"z": z_out, "keyed": (viewmatrix, projmatrix, campos), "n": n_prev}
    while len(ents) > _DEPTH_CUTS["max_entries"]:
        ents.popitem(last=False)
    return key, z_in, z_out, n_prev


def _layout_of(R):
    """(instances the binning buffer of this frame was laid out for, that buffer or None) for the R a caller hands to
    the backward: a plain int (exact frame) or the LazyCount of a speculative one -- which is NOT resolved here."""
    if isinstance(R, LazyCount):
        if not R.resolved:
            R._resolve(wait=False, lazy=True)  # free look: keeps the statistics and the capacity policy current
        return R.layout, R.binning
    return int(R), None


SCRATCH_STATS = {"sized_by_count": 0, "sized_by_capacity": 0}  # full backwards of speculative frames, by how their row scratch was laid out


def _scratch_instances(R, layout: int) -> int:
    """Instances the backward's row scratch is laid out for (goi_raster_backward3).  A speculative frame's workspaces are sized
    for its CAPACITY; if its count has reached the host by now (the free look of _layout_of: the loss usually sits between the
    forward and this call) the scratch -- 129 bytes per instance and quadrant, the largest workspace of a step -- only has to
    hold the COUNT.  0 = as the binning layout."""
    if isinstance(R, LazyCount) and not R.redone:
        if R.resolved and R._error is None and R._n is not None and 0 < R._n <= layout and not R.overflowed and not R.cut_failed:
            SCRATCH_STATS["sized_by_count"] += 1
            return max(int(R._n), 1)
        SCRATCH_STATS["sized_by_capacity"] += 1
    return 0


def _scene(P, S, H, W, bg, means3D, sh, colors, semantics, opacity, scales, rotations, scale_modifier, cov3D,
           viewmatrix, projmatrix, tan_fovx, tan_fovy, degree, campos, prefiltered, debug):
    M = 0 if (sh is None or sh.numel() == 0) else int(sh.size(1))
    return GoiRasterScene(
        int(P), int(degree), M, int(S), int(W), int(H), _ptr(bg), _ptr(means3D), _ptr(sh), _ptr(colors),
        _ptr(semantics), _ptr(opacity), _ptr(scales), float(scale_modifier), _ptr(rotations), _ptr(cov3D),
        _ptr(viewmatrix), _ptr(projmatrix), _ptr(campos), float(tan_fovx), float(tan_fovy), int(bool(prefiltered)),
        int(bool(debug)))


# ---- opt-in geometry cache ---------------------------------------------------------------------------
# The reference's semantic stage trains ONLY the semantic features (arguments/__init__.py:85-90): positions, covariances,
# opacities and SH colours are frozen, so for a camera that was rendered before, everything in front of the blend
# (preprocess, depth sort, scan, emit, tile sort: 0.35 of the 0.63 ms forward at the headline size) reproduces what is still
# sitting in that frame's workspaces.  With the cache on, such a frame runs goi_raster_forward_reblend alone.  It is OPT-IN
# because the premise cannot be verified here: the activated tensors the rasterizer is handed (exp(scaling), sigmoid(opacity),
# ...) are new objects on every call.  A frame is eligible only if no geometry input requires a gradient; the key holds the
# camera tensors' and the positions' identity and version (an optimizer step, a densification or a new camera object miss),
# and the entry keeps those tensors alive so that their addresses cannot be recycled under the key.
_GEOM_CACHE = {"max_bytes": int(float(os.environ.get("GOI_GEOMETRY_CACHE_GB", "0")) * (1 << 30)), "bytes": 0,
               "entries": collections.OrderedDict(), "hits": 0, "misses": 0, "evictions": 0}
_GEOM_LOCK = threading.RLock()


def set_geometry_cache(max_bytes) -> None:
    """Enables (max_bytes > 0), resizes or disables and empties (0 / None) the per-camera geometry cache.  About
    goi_raster_geom_bytes(P) + goi_raster_binning_bytes(capacity) per cached camera (~270 MB at 1 M Gaussians, 1600x1056)."""
    with _GEOM_LOCK:
        _GEOM_CACHE["max_bytes"] = int(max_bytes or 0)
        _geom_evict()


def geometry_cache_stats() -> dict:
    with _GEOM_LOCK:
        return {k: (len(v) if k == "entries" else v) for k, v in _GEOM_CACHE.items()}


def _geom_evict():
    c = _GEOM_CACHE
    while c["entries"] and c["bytes"] > c["max_bytes"]:
        _k, e = c["entries"].popitem(last=False)  # least recently used
        c["bytes"] -= e["bytes"]
        c["evictions"] += 1


def _ident(t):
    return None if (t is None or not isinstance(t, torch.Tensor) or t.numel() == 0) else (t.data_ptr(), t._version, tuple(t.shape))


def _geom_key(dev, P, H, W, means3D, scale_modifier, viewmatrix, projmatrix, tan_fovx, tan_fovy, degree, campos, prefiltered,
              colors, cov3D):
    # (precomputed colours / covariances are the caller's own tensors: their identity is meaningful; the SH coefficients and
    # the activated scales / rotations / opacities are rebuilt by the reference's model on every call and cannot be keyed)
    return (dev.index, P, H, W, float(scale_modifier), float(tan_fovx), float(tan_fovy), int(degree), bool(prefiltered),
            _ident(means3D), _ident(viewmatrix), _ident(projmatrix), _ident(campos), _ident(colors), _ident(cov3D),
            tuple(sorted(_lib.OPTIONS.items())))


def rasterize_gaussians(background, means3D, colors, semantics, opacity, scales, rotations, scale_modifier,
                        cov3D_precomp, viewmatrix, projmatrix, tan_fovx, tan_fovy, image_height, image_width, sh,
                        degree, campos, prefiltered, debug):
    """-> (num_rendered, color[3,H,W], semantic[S,H,W], depth[1,H,W], alpha[1,H,W], radii[P] i32,
    geomBuffer u8, binningBuffer u8, imgBuffer u8)"""
    args = (background, means3D, colors, semantics, opacity, scales, rotations, scale_modifier, cov3D_precomp, viewmatrix,
            projmatrix, tan_fovx, tan_fovy, image_height, image_width, sh, degree, campos, prefiltered, debug)
    P = int(means3D.size(0)) if isinstance(means3D, torch.Tensor) and means3D.ndimension() == 2 else 0
    if not (_GEOM_CACHE["max_bytes"] > 0 and getattr(_CALL, "geometry_frozen", False) and P > 0 and not debug
            and isinstance(semantics, torch.Tensor) and semantics.ndimension() == 2 and semantics.size(0) == P):
        return _rasterize_gaussians_frame(*args)
    dev = _check_device(means3D)
    H, W = int(image_height), int(image_width)
    key = _geom_key(dev, P, H, W, means3D, scale_modifier, viewmatrix, projmatrix, tan_fovx, tan_fovy, degree, campos,
                    prefiltered, colors, cov3D_precomp)
    with _GEOM_LOCK:
        e = _GEOM_CACHE["entries"].get(key)
        if e is not None:
            R = e["R"]
            drop = False
            if isinstance(R, LazyCount):
                if not R.resolved:
                    R._resolve(wait=False, lazy=True)
                usable = R.resolved and R._error is None
                if usable and R.overflowed and not R.redone:
                    # the cached frame was TRUNCATED and never repaired (the overflow was only found lazily): its binning
                    # workspace holds a cut-off tile list.  Never reblend over it: drop the entry and render the frame again
                    # (the capacity policy has been raised by the resolve above)
                    usable, drop = False, True
                elif R.resolved and R._error is not None:
                    drop = True
            else:
                usable = True
            if usable:
                _GEOM_CACHE["entries"].move_to_end(key)
                _GEOM_CACHE["hits"] += 1
            else:
                if drop:
                    _GEOM_CACHE["bytes"] -= e["bytes"]
                    del _GEOM_CACHE["entries"][key]
                e = None
    if e is not None:
        return _reblend(e, background, semantics, P, H, W, dev)
    res = _rasterize_gaussians_frame(*args)
    R, _c, _s, _d, _a, radii, geom, binning, img = res
    with _GEOM_LOCK:
        _GEOM_CACHE["misses"] += 1
        # The key identifies tensors by (address, version, shape): that is only sound while the tensors are ALIVE -- a freed
        # camera matrix or `xyz[mask]` temporary hands its address (version 0 again) to the next frame's tensors.  The entry
        # therefore holds strong references to every keyed tensor for as long as it lives (and accounts for their bytes).
        keyed = tuple(t for t in (means3D, viewmatrix, projmatrix, campos, colors, cov3D_precomp)
                      if isinstance(t, torch.Tensor) and t.numel() > 0)
        nbytes = (geom.numel() + img.numel() + radii.numel() * 4 + (binning.numel() if isinstance(binning, torch.Tensor) else 0)
                  + sum(t.numel() * t.element_size() for t in keyed))
        old = _GEOM_CACHE["entries"].pop(key, None)
        if old is not None:
            _GEOM_CACHE["bytes"] -= old["bytes"]
        if nbytes <= _GEOM_CACHE["max_bytes"]:
            _GEOM_CACHE["entries"][key] = dict(R=R, radii=radii, geom=geom, binning=binning, img=img, bytes=nbytes,
                                               stream=torch.cuda.current_stream(dev), keyed=keyed)
            _GEOM_CACHE["bytes"] += nbytes
            _geom_evict()
    return res


def _reblend(e, background, semantics, P, H, W, dev):
    """A cached camera: the blend alone, over the cached frame's workspaces (goi_raster_forward_reblend)."""
    lib = _lib.load()
    S = int(semantics.size(1))
    R = e["R"]
    layout, bin_override = _layout_of(R)
    binning = bin_override if bin_override is not None else e["binning"]
    f32 = dict(dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        cur = torch.cuda.current_stream(dev)
        if e["stream"] != cur:
            cur.wait_stream(e["stream"])  # the frame that filled the workspaces ran on another stream
        out_color = torch.empty((3, H, W), **f32)
        out_sem = torch.empty((S, H, W), **f32)
        out_depth = torch.empty((1, H, W), **f32)
        out_alpha = torch.empty((1, H, W), **f32)
        img = torch.empty_like(e["img"])
        bg_c, sem_c = _prep(background, "background", dev), _prep(semantics, "semantics", dev)
        sc = _scene(P, S, H, W, bg_c, None, None, None, sem_c, None, None, None, 1.0, None, None, None, 0.0, 0.0, 0, None,
                    False, False)
        r = lib.goi_raster_forward_reblend(C.byref(sc), int(layout), _ptr(e["geom"]), _ptr(binning), _ptr(e["img"]), _ptr(img),
                                           _ptr(out_color), _ptr(out_sem), _ptr(out_depth), _ptr(out_alpha), _stream(dev))
    if r < 0:
        raise RuntimeError(_lib.last_error())
    _note_frame(e["geom"], P)
    SPECULATION_STATS["cached_frames"] += 1
    return R, out_color, out_sem, out_depth, out_alpha, e["radii"], e["geom"], binning, img


def _rasterize_gaussians_frame(background, means3D, colors, semantics, opacity, scales, rotations, scale_modifier,
                               cov3D_precomp, viewmatrix, projmatrix, tan_fovx, tan_fovy, image_height, image_width, sh,
                               degree, campos, prefiltered, debug):
    lib = _lib.load()
    dev = _check_device(means3D)
    P, H, W = int(means3D.size(0)), int(image_height), int(image_width)
    if semantics is None or semantics.numel() == 0:
        if P > 0:
            raise RuntimeError("semantics [P,S] is required (the reference dereferences it unconditionally, "
                               "cuda_rasterizer/forward.cu:363)")
        S = 10
    else:
        if semantics.ndimension() != 2 or semantics.size(0) != P:
            raise RuntimeError("semantics must have dimensions (num_points, S)")
        S = int(semantics.size(1))
    if not (1 <= S <= 32):
        raise RuntimeError(f"unsupported number of semantic channels S={S} (1..32)")
    ext = _ext()
    if ext is not None:
        poll_counts(dev)
        args = (background, means3D, _e(colors), _e(semantics), _e(opacity), _e(scales), _e(rotations),
                float(scale_modifier), _e(cov3D_precomp), viewmatrix, projmatrix, float(tan_fovx), float(tan_fovy), H, W,
                _e(sh), int(degree), campos, bool(prefiltered), bool(debug))
        cap = _pick_capacity(dev, P, debug, prefiltered, _tiles(H, W))
        if cap is None:
            res = ext.rasterize_gaussians(*args)
            _note_frame(res[6], P)
            if P > 0:
                SPECULATION_STATS["exact_frames"] += 1
                _note_count(dev, P, res[0], _tiles(H, W))
            return res
        cut_key, z_in, z_out, n_prev = _depth_cut_for(dev, P, H, W, viewmatrix, projmatrix, campos, tan_fovx, tan_fovy,
                                                      scale_modifier)
        if z_in is not None and n_prev > 0 and _FWD["capacity"] is None:
            cap = min(cap, max(_MIN_CAPACITY, int(_FWD["headroom"] * n_prev) + 4096))
        ticket, out_color, out_sem, out_depth, out_alpha, radii, geom, binning, img = ext.rasterize_gaussians_async(
            *args, cap, z_in, z_out)
        _note_frame(geom, P)
        SPECULATION_STATS["speculative_frames"] += 1
        refs = [weakref.ref(t) for t in (geom, img, radii, out_color, out_sem, out_depth, out_alpha)]

        def make_scene(bg=background, sem=semantics):  # only what a redo reads (include/goi_raster.h)
            bg_c, sem_c = bg.contiguous(), sem.contiguous()
            return _scene(P, S, H, W, bg_c, None, None, None, sem_c, None, None, None, 1.0, None, None, None, 0.0, 0.0, 0,
                          None, False, False), (bg_c, sem_c)
        n = LazyCount(dev, ticket, cap, binning, torch.cuda.current_stream(dev).cuda_stream, (make_scene, refs), P,
                      workspaces=(geom, img, radii))
        n.tiles = _tiles(H, W)
        n.cam_key = cut_key
        if z_in is not None:
            n.cut_key = cut_key
            n._full_args = (background, means3D, colors, semantics, opacity, scales, rotations, scale_modifier, cov3D_precomp,
                            viewmatrix, projmatrix, tan_fovx, tan_fovy, H, W, sh, degree, campos, prefiltered, debug)
            SPECULATION_STATS["cut_frames"] += 1
        return n, out_color, out_sem, out_depth, out_alpha, radii, geom, binning, img
    f32 = dict(dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        out_color = torch.empty((3, H, W), **f32)
        out_sem = torch.empty((S, H, W), **f32)
        out_depth = torch.empty((1, H, W), **f32)
        out_alpha = torch.empty((1, H, W), **f32)
        radii = torch.empty((P,), dtype=torch.int32, device=dev)
        ten = dict(bg=_prep(background, "background", dev), means3D=_prep(means3D, "means3D", dev),
                   sh=_prep(sh, "sh", dev), colors=_prep(colors, "colors_precomp", dev),
                   semantics=_prep(semantics, "semantics", dev), opacity=_prep(opacity, "opacities", dev),
                   scales=_prep(scales, "scales", dev), rotations=_prep(rotations, "rotations", dev),
                   cov3D=_prep(cov3D_precomp, "cov3D_precomp", dev), viewmatrix=_prep(viewmatrix, "viewmatrix", dev),
                   projmatrix=_prep(projmatrix, "projmatrix", dev), campos=_prep(campos, "campos", dev))
        geom = torch.empty(lib.goi_raster_geom_bytes(P) if P > 0 else 0, dtype=torch.uint8, device=dev)
        img = torch.empty(lib.goi_raster_image_bytes(W, H) if P > 0 else 0, dtype=torch.uint8, device=dev)
        _note_frame(geom, P)
        sc = _scene(P, S, H, W, ten["bg"], ten["means3D"], ten["sh"], ten["colors"], ten["semantics"], ten["opacity"],
                    ten["scales"], ten["rotations"], scale_modifier, ten["cov3D"], ten["viewmatrix"],
                    ten["projmatrix"], tan_fovx, tan_fovy, degree, ten["campos"], prefiltered, debug)
        poll_counts(dev)
        cap = _pick_capacity(dev, P, debug, prefiltered, _tiles(H, W))
        if cap is not None:
            # speculative frame: everything is enqueued now, the count arrives through the ticket
            cut_key, z_in, z_out, n_prev = _depth_cut_for(dev, P, H, W, viewmatrix, projmatrix, campos, tan_fovx, tan_fovy,
                                                          scale_modifier)
            if z_in is not None and n_prev > 0 and _FWD["capacity"] is None:
                cap = min(cap, max(_MIN_CAPACITY, int(_FWD["headroom"] * n_prev) + 4096))
            step = 16 << 20
            binning = torch.empty((int(lib.goi_raster_binning_bytes(cap)) + step - 1) // step * step, dtype=torch.uint8,
                                  device=dev)
            stream = torch.cuda.current_stream(dev).cuda_stream
            outs = (out_color, out_sem, out_depth, out_alpha)
            ticket = lib.goi_raster_forward_async_cut(C.byref(sc), _ptr(geom), _ptr(img), _ptr(binning), cap,
                                                      *[_ptr(o) for o in outs], _ptr(radii), _ptr(z_in), _ptr(z_out),
                                                      C.c_void_p(stream))
            if ticket < 0:
                raise RuntimeError(_lib.last_error())
            SPECULATION_STATS["speculative_frames"] += 1
            # what a redo needs: workspaces and outputs (weakly: dead outputs have no consumer left to repair for) and the
            # two inputs the back half of the frame reads, the semantic rows and the background (strongly: they may be
            # temporaries of the caller, e.g. pc.get_semantics under a mask)
            refs = [weakref.ref(t) for t in (geom, img, radii) + outs]
            keep = (ten["semantics"], ten["bg"])
            n = LazyCount(dev, ticket, cap, binning, stream, (lambda: (sc, keep), refs), P, workspaces=(geom, img, radii))
            n.tiles = _tiles(H, W)
            n.cam_key = cut_key
            if z_in is not None:
                n.cut_key = cut_key
                n._full_args = (background, means3D, colors, semantics, opacity, scales, rotations, scale_modifier,
                                cov3D_precomp, viewmatrix, projmatrix, tan_fovx, tan_fovy, H, W, sh, degree, campos, prefiltered,
                                debug)
                SPECULATION_STATS["cut_frames"] += 1
            return n, out_color, out_sem, out_depth, out_alpha, radii, geom, binning, img
        alloc = _BinningAllocator(dev)
        n = lib.goi_raster_forward(C.byref(sc), _ptr(geom), _ptr(img), alloc.cb, None, _ptr(out_color), _ptr(out_sem),
                                   _ptr(out_depth), _ptr(out_alpha), _ptr(radii), _stream(dev))
        if alloc.error is not None:
            raise alloc.error
        if n < 0:
            raise RuntimeError(_lib.last_error())
        if P > 0:
            SPECULATION_STATS["exact_frames"] += 1
            _note_count(dev, P, n, _tiles(H, W))
    return n, out_color, out_sem, out_depth, out_alpha, radii, geom, alloc.tensor, img


def _backward_impl(background, means3D, radii, colors, semantics, scales, rotations, scale_modifier,
                                 cov3D_precomp, viewmatrix, projmatrix, tan_fovx, tan_fovy, dL_dout_color,
                                 dL_dout_semantic, dL_dout_depth, dL_dout_alpha, sh, degree, campos, geomBuffer, R,
                                 binningBuffer, imageBuffer, alphas, debug, sh_factored=False, accumulate_into=None):
    """-> (dL_dmeans2D[P,3], dL_dcolors[P,3], dL_dsemantics[P,S], dL_dopacity[P,1], dL_dmeans3D[P,3],
    dL_dcov3D[P,6], dL_dsh[P,M,3], dL_dscales[P,3], dL_drotations[P,4])

    accumulate_into (compiled binding only): the dL_dmeans3D an earlier call of the same batch returned -- this view's gradients
    are ADDED to that call's tensors (goi_raster_backward3, GOI_BACKWARD_ACCUMULATE) and the same tensors are returned.

    sh_factored (not in the reference's pybind module; FACTORED mode of goi_raster_backward): dL_dsh is not formed
    (returned as None) and dL_dcolors is the clamp-masked colour gradient g, the factor of
    dL/dSH[k] = basis_k(view direction) * g -- see sh_grad_from_views and dist.allreduce_gradients_sh_factored."""
    lib = _lib.load()
    dev = _check_device(means3D)
    ext = _ext()
    if ext is not None:
        R_layout, lazy_binning = _layout_of(R)
        return ext.backward_ex(background, means3D, radii, _e(colors), semantics, _e(scales), _e(rotations),
                               float(scale_modifier), _e(cov3D_precomp), viewmatrix, projmatrix, float(tan_fovx),
                               float(tan_fovy), dL_dout_color, dL_dout_semantic, dL_dout_depth, dL_dout_alpha, _e(sh),
                               int(degree), campos, geomBuffer, R_layout,
                               binningBuffer if lazy_binning is None else lazy_binning, imageBuffer, alphas, bool(debug),
                               bool(sh_factored), _scratch_instances(R, R_layout), accumulate_into)
    if accumulate_into is not None:
        raise RuntimeError("accumulate_into needs the compiled binding (python -m goi_hyperplane_amd.build)")
    P = int(means3D.size(0))
    # the reference reads H, W off dL_dout_color (rasterize_points.cu:243-244); here any upstream gradient may be
    # None (an output the loss does not use), so the sizes come from tensors that always exist
    H, W = int(alphas.size(-2)), int(alphas.size(-1))
    S = int(semantics.size(1))
    M = 0 if (sh is None or sh.numel() == 0) else int(sh.size(1))
    sh_factored = bool(sh_factored) and M > 0
    f32 = dict(dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        # every element is written by the library (atomically accumulated ones are zeroed there).
        # The gradients of the six Gaussian parameter tensors are views of ONE flat buffer (256-byte
        # aligned sections): autograd hands these views to the leaves as .grad, so the data-parallel
        # exchange can be a single all-reduce over the buffer instead of one per tensor (dist.py).
        sections = (("means3D", (P, 3)), ("sh", (P, 0 if sh_factored else M, 3)), ("semantics", (P, S)),
                    ("opacity", (P, 1)), ("scales", (P, 3)), ("rotations", (P, 4)))
        offs, total = {}, 0
        for name, shape in sections:
            offs[name] = total
            n = 1
            for d in shape:
                n *= d
            total += (n + 63) // 64 * 64
        flat = torch.empty((total,), **f32)

        def view(name, shape):
            n = 1
            for d in shape:
                n *= d
            return flat[offs[name]:offs[name] + n].view(shape)
        dL_dmeans3D = view("means3D", (P, 3))
        dL_dsh = None if sh_factored else view("sh", (P, M, 3))
        dL_dsemantics = view("semantics", (P, S))
        dL_dopacity = view("opacity", (P, 1))
        dL_dscales = view("scales", (P, 3))
        dL_drotations = view("rotations", (P, 4))
        del flat
        dL_dmeans2D = torch.empty((P, 3), **f32)
        dL_dcolors = torch.empty((P, 3), **f32)
        dL_ddepths = torch.empty((P, 1), **f32)
        dL_dconic = torch.empty((P, 2, 2), **f32)
        dL_dcov3D = torch.empty((P, 6), **f32)
        if P != 0:
            ten = dict(bg=_prep(background, "background", dev), means3D=_prep(means3D, "means3D", dev),
                       sh=_prep(sh, "sh", dev), colors=_prep(colors, "colors_precomp", dev),
                       semantics=_prep(semantics, "semantics", dev), scales=_prep(scales, "scales", dev),
                       rotations=_prep(rotations, "rotations", dev), cov3D=_prep(cov3D_precomp, "cov3D_precomp", dev),
                       viewmatrix=_prep(viewmatrix, "viewmatrix", dev), projmatrix=_prep(projmatrix, "projmatrix", dev),
                       campos=_prep(campos, "campos", dev), radii=_prep(radii, "radii", dev, torch.int32),
                       alphas=_prep(alphas, "alphas", dev), g_c=_prep(dL_dout_color, "dL_dout_color", dev),
                       g_s=_prep(dL_dout_semantic, "dL_dout_semantic", dev),
                       g_d=_prep(dL_dout_depth, "dL_dout_depth", dev), g_a=_prep(dL_dout_alpha, "dL_dout_alpha", dev))
            sc = _scene(P, S, H, W, ten["bg"], ten["means3D"], ten["sh"], ten["colors"], ten["semantics"], None,
                        ten["scales"], ten["rotations"], scale_modifier, ten["cov3D"], ten["viewmatrix"],
                        ten["projmatrix"], tan_fovx, tan_fovy, degree, ten["campos"], False, debug)
            R_layout, lazy_binning = _layout_of(R)
            if lazy_binning is not None:
                binningBuffer = lazy_binning  # (a redone frame has a new buffer)
            R_scratch = _scratch_instances(R, R_layout)
            scratch = _backward_scratch(lib.goi_raster_backward_scratch_bytes(R_scratch or R_layout, S), dev)
            r = lib.goi_raster_backward3(
                C.byref(sc), R_layout, R_scratch, 0, _ptr(geomBuffer), _ptr(binningBuffer), _ptr(imageBuffer), _ptr(ten["radii"]),
                _ptr(ten["alphas"]), _ptr(ten["g_c"]), _ptr(ten["g_s"]), _ptr(ten["g_d"]), _ptr(ten["g_a"]),
                _ptr(dL_dmeans2D), _ptr(dL_dconic), _ptr(dL_dopacity), _ptr(dL_dcolors), _ptr(dL_dsemantics),
                _ptr(dL_ddepths), _ptr(dL_dmeans3D), _ptr(dL_dcov3D), _ptr(dL_dsh), _ptr(dL_dscales),
                _ptr(dL_drotations), _ptr(scratch), None, _stream(dev))
            if r < 0:
                raise RuntimeError(_lib.last_error())
    return (dL_dmeans2D, dL_dcolors, dL_dsemantics, dL_dopacity, dL_dmeans3D, dL_dcov3D, dL_dsh, dL_dscales,
            dL_drotations)


def rasterize_gaussians_backward(background, means3D, radii, colors, semantics, scales, rotations, scale_modifier,
                                 cov3D_precomp, viewmatrix, projmatrix, tan_fovx, tan_fovy, dL_dout_color,
                                 dL_dout_semantic, dL_dout_depth, dL_dout_alpha, sh, degree, campos, geomBuffer, R,
                                 binningBuffer, imageBuffer, alphas, debug):
    """-> (dL_dmeans2D[P,3], dL_dcolors[P,3], dL_dsemantics[P,S], dL_dopacity[P,1], dL_dmeans3D[P,3],
    dL_dcov3D[P,6], dL_dsh[P,M,3], dL_dscales[P,3], dL_drotations[P,4]) -- the reference's RasterizeGaussiansBackwardCUDA
    (rasterize_points.cu:213-306), same argument list."""
    return _backward_impl(background, means3D, radii, colors, semantics, scales, rotations, scale_modifier, cov3D_precomp, viewmatrix, projmatrix, tan_fovx, tan_fovy, dL_dout_color, dL_dout_semantic, dL_dout_depth, dL_dout_alpha, sh, degree, campos, geomBuffer, R, binningBuffer, imageBuffer, alphas, debug)


def rasterize_gaussians_backward_accumulate(accumulate_into, *args):
    """rasterize_gaussians_backward(*args) (not in the reference's pybind module) whose gradients are ADDED, on the device, to the
    tensors an earlier call of the same batch returned: accumulate_into = that call's dL_dmeans3D (None: a plain call, the
    batch's first).  Returns the same nine tensors.  goi_raster_backward3 / GOI_BACKWARD_ACCUMULATE; compiled binding only."""
    return _backward_impl(*args, accumulate_into=accumulate_into)


def rasterize_gaussians_backward_sh_factored(background, means3D, radii, colors, semantics, scales, rotations, scale_modifier,
                                 cov3D_precomp, viewmatrix, projmatrix, tan_fovx, tan_fovy, dL_dout_color,
                                 dL_dout_semantic, dL_dout_depth, dL_dout_alpha, sh, degree, campos, geomBuffer, R,
                                 binningBuffer, imageBuffer, alphas, debug):
    """Same arguments and result tuple as rasterize_gaussians_backward (not in the reference's pybind module), in the
    FACTORED mode of goi_raster_backward: dL_dsh is None and dL_dcolors is the clamp-masked colour gradient."""
    return _backward_impl(background, means3D, radii, colors, semantics, scales, rotations, scale_modifier, cov3D_precomp, viewmatrix, projmatrix, tan_fovx, tan_fovy, dL_dout_color, dL_dout_semantic, dL_dout_depth, dL_dout_alpha, sh, degree, campos, geomBuffer, R, binningBuffer, imageBuffer, alphas, debug, sh_factored=True)


def sh_grad_from_views(means3D, campos, gcol, degree, M):
    """dL/dSH [P,M,3] of V views from means3D [P,3], the camera centres campos [V,3] and the clamp-masked colour
    gradients gcol [V,P,3] that rasterize_gaussians_backward(sh_factored=True) returns as dL_dcolors
    (goi_raster_sh_grad_from_views): sum over the views, in index order, of basis(view direction) x gcol."""
    lib = _lib.load()
    dev = _check_device(means3D)
    P, V = int(means3D.size(0)), int(campos.size(0))
    if tuple(gcol.shape) != (V, P, 3):
        raise ValueError(f"gcol must be [V={V}, P={P}, 3], got {tuple(gcol.shape)}")
    with torch.cuda.device(dev):
        out = torch.empty((P, int(M), 3), dtype=torch.float32, device=dev)
        if P and V:
            m, c, g = _prep(means3D, "means3D", dev), _prep(campos, "campos", dev), _prep(gcol, "gcol", dev)
            if lib.goi_raster_sh_grad_from_views(P, int(degree), int(M), V, _ptr(m), _ptr(c), _ptr(g), _ptr(out), _stream(dev)) < 0:
                raise RuntimeError(_lib.last_error())
        elif P:
            out.zero_()
    return out


def rasterize_gaussians_backward_semantics(background, means3D, radii, semantics, viewmatrix, projmatrix, tan_fovx,
                                           tan_fovy, dL_dout_semantic, campos, geomBuffer, R, binningBuffer,
                                           imageBuffer, alphas, sh_degree=0, debug=False):
    """-> dL_dsemantics[P,S] only (not in the reference's pybind module): the feature-gradient-only
    backward behind goi_raster_backward_semantics, for training runs that optimise only the semantic
    features (the reference's default, arguments/__init__.py:85-90).  Bit-identical to the
    dL_dsemantics of rasterize_gaussians_backward."""
    lib = _lib.load()
    dev = _check_device(means3D)
    ext = _ext()
    if ext is not None:
        R_layout, lazy_binning = _layout_of(R)
        return ext.backward_semantics(background, means3D, radii, semantics, viewmatrix, projmatrix, float(tan_fovx),
                                      float(tan_fovy), dL_dout_semantic, campos, geomBuffer, R_layout,
                                      binningBuffer if lazy_binning is None else lazy_binning, imageBuffer, alphas,
                                      int(sh_degree), bool(debug))
    P = int(means3D.size(0))
    S = int(dL_dout_semantic.size(0))
    H, W = int(dL_dout_semantic.size(1)), int(dL_dout_semantic.size(2))
    with torch.cuda.device(dev):
        dL_dsemantics = torch.empty((P, S), dtype=torch.float32, device=dev)
        if P != 0:
            ten = dict(bg=_prep(background, "background", dev), means3D=_prep(means3D, "means3D", dev),
                       semantics=_prep(semantics, "semantics", dev), viewmatrix=_prep(viewmatrix, "viewmatrix", dev),
                       projmatrix=_prep(projmatrix, "projmatrix", dev), campos=_prep(campos, "campos", dev),
                       radii=_prep(radii, "radii", dev, torch.int32), alphas=_prep(alphas, "alphas", dev),
                       g_s=_prep(dL_dout_semantic, "dL_dout_semantic", dev))
            # the geometry inputs are not read by this path (everything it needs is in the forward's
            # workspaces); the scene only has to pass validation
            sc = _scene(P, S, H, W, ten["bg"], ten["means3D"], None, ten["means3D"], ten["semantics"], None,
                        None, None, 1.0, ten["means3D"], ten["viewmatrix"], ten["projmatrix"], tan_fovx, tan_fovy,
                        sh_degree, ten["campos"], False, debug)
            R_layout, lazy_binning = _layout_of(R)
            if lazy_binning is not None:
                binningBuffer = lazy_binning
            scratch = _backward_scratch(lib.goi_raster_backward_scratch_bytes(R_layout, S), dev)
            r = lib.goi_raster_backward_semantics(C.byref(sc), R_layout, _ptr(geomBuffer), _ptr(binningBuffer),
                                                  _ptr(imageBuffer), _ptr(ten["radii"]), _ptr(ten["alphas"]),
                                                  _ptr(ten["g_s"]), _ptr(dL_dsemantics), _ptr(scratch), _stream(dev))
            if r < 0:
                raise RuntimeError(_lib.last_error())
    return dL_dsemantics


def rasterize_gaussians_trace(background, means3D, colors, img_sem, opacity, scales, rotations, scale_modifier,
                              cov3D_precomp, viewmatrix, projmatrix, tan_fovx, tan_fovy, image_height, image_width, sh,
                              degree, campos, prefiltered, debug):
    """-> (num_rendered, color[3,H,W], gau_sem[P,S], num_gsem[P] i32, geomBuffer, binningBuffer, imgBuffer)"""
    lib = _lib.load()
    dev = _check_device(means3D)
    P, H, W = int(means3D.size(0)), int(image_height), int(image_width)
    if img_sem is None or img_sem.numel() == 0:
        raise RuntimeError("img_sem [S,H,W] is required")
    ext = _ext()
    if ext is not None:
        res = ext.rasterize_gaussians_trace(background, means3D, _e(colors), img_sem, _e(opacity), _e(scales),
                                            _e(rotations), float(scale_modifier), _e(cov3D_precomp), viewmatrix,
                                            projmatrix, float(tan_fovx), float(tan_fovy), H, W, _e(sh), int(degree),
                                            campos, bool(prefiltered), bool(debug))
        if P > 0:
            _note_count(dev, P, res[0], _tiles(H, W))
        return res
    S = int(img_sem.size(0))
    f32 = dict(dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        out_color = torch.empty((3, H, W), **f32)
        gau_sem = torch.zeros((P, S), **f32)
        num_gsem = torch.zeros((P,), dtype=torch.int32, device=dev)
        radii = torch.empty((P,), dtype=torch.int32, device=dev)
        ten = dict(bg=_prep(background, "background", dev), means3D=_prep(means3D, "means3D", dev),
                   sh=_prep(sh, "sh", dev), colors=_prep(colors, "colors_precomp", dev),
                   img=_prep(img_sem, "img_sem", dev), opacity=_prep(opacity, "opacities", dev),
                   scales=_prep(scales, "scales", dev), rotations=_prep(rotations, "rotations", dev),
                   cov3D=_prep(cov3D_precomp, "cov3D_precomp", dev), viewmatrix=_prep(viewmatrix, "viewmatrix", dev),
                   projmatrix=_prep(projmatrix, "projmatrix", dev), campos=_prep(campos, "campos", dev))
        geom = torch.empty(lib.goi_raster_geom_bytes(P) if P > 0 else 0, dtype=torch.uint8, device=dev)
        img = torch.empty(lib.goi_raster_image_bytes(W, H) if P > 0 else 0, dtype=torch.uint8, device=dev)
        alloc = _BinningAllocator(dev)
        sc = _scene(P, S, H, W, ten["bg"], ten["means3D"], ten["sh"], ten["colors"], None, ten["opacity"],
                    ten["scales"], ten["rotations"], scale_modifier, ten["cov3D"], ten["viewmatrix"],
                    ten["projmatrix"], tan_fovx, tan_fovy, degree, ten["campos"], prefiltered, debug)
        n = lib.goi_raster_trace(C.byref(sc), _ptr(ten["img"]), _ptr(geom), _ptr(img), alloc.cb, None, _ptr(out_color),
                                 _ptr(gau_sem), _ptr(num_gsem), _ptr(radii), _stream(dev))
        if alloc.error is not None:
            raise alloc.error
        if n < 0:
            raise RuntimeError(_lib.last_error())
        if P > 0:
            _note_count(dev, P, n, _tiles(H, W))
    return n, out_color, gau_sem, num_gsem, geom, alloc.tensor, img


def mark_visible(means3D, viewmatrix, projmatrix):
    """-> bool[P]: view-space z > 0.2 (cuda_rasterizer/auxiliary.h:139-164)."""
    lib = _lib.load()
    dev = _check_device(means3D)
    ext = _ext()
    if ext is not None:
        return ext.mark_visible(means3D, viewmatrix, projmatrix)
    P = int(means3D.size(0))
    with torch.cuda.device(dev):
        present = torch.zeros((P,), dtype=torch.bool, device=dev)
        if P != 0:
            m = _prep(means3D, "means3D", dev)
            v = _prep(viewmatrix, "viewmatrix", dev)
            p = _prep(projmatrix, "projmatrix", dev)
            if lib.goi_raster_mark_visible(P, _ptr(m), _ptr(v), _ptr(p), C.c_void_p(present.data_ptr()),
                                           _stream(dev)) < 0:
                raise RuntimeError(_lib.last_error())
    return present


BLEND_STATS_FIELDS = ("quadrants", "rounds", "positions", "forward_pairs", "member_pairs", "live_lanes", "pairs_8x4",
                      "pairs_4x4", "pairs_2x2", "dead_member_pairs", "pixels", "sum_n_contrib")


def blend_stats(P, W, H, R, geomBuffer, binningBuffer, imgBuffer) -> dict:
    """Lane utilisation of the blend kernels for the frame whose forward filled the three workspaces
    (goi_raster_blend_stats; diagnostic).  Returns the raw counters by name plus the derived ratios:
    lane_utilisation_backward = live lanes / (64 x member pairs), lane_utilisation_forward = live lanes / (64 x the pairs
    the forward evaluates), and what a split of the wave into 8x4 / 4x4 / 2x2 blocks would reach at best (live lanes /
    (block size x (block, Gaussian) pairs): perfect balance between the blocks of a wave assumed)."""
    lib = _lib.load()
    dev = geomBuffer.device
    R, lazy_binning = _layout_of(R)
    if lazy_binning is not None:
        binningBuffer = lazy_binning
    with torch.cuda.device(dev):
        out = torch.zeros(16, dtype=torch.int64, device=dev)
        if lib.goi_raster_blend_stats(P, W, H, R, _ptr(geomBuffer), _ptr(binningBuffer), _ptr(imgBuffer), _ptr(out),
                                      _stream(dev)) < 0:
            raise RuntimeError(_lib.last_error())
        v = [int(x) for x in out.cpu().tolist()]
    d = dict(zip(BLEND_STATS_FIELDS, v))
    live = float(d["live_lanes"])
    d["lane_utilisation_backward"] = live / max(64.0 * d["member_pairs"], 1.0)
    d["lane_utilisation_forward"] = live / max(64.0 * d["forward_pairs"], 1.0)
    d["lane_utilisation_8x4_blocks"] = live / max(32.0 * d["pairs_8x4"], 1.0)
    d["lane_utilisation_4x4_blocks"] = live / max(16.0 * d["pairs_4x4"], 1.0)
    d["lane_utilisation_2x2_blocks"] = live / max(4.0 * d["pairs_2x2"], 1.0)
    d["live_lanes_per_member_pair"] = live / max(float(d["member_pairs"]), 1.0)
    d["contributions_per_pixel"] = live / max(float(d["pixels"]), 1.0)
    return d


def debug_views(P, W, H, R, geomBuffer, binningBuffer, imgBuffer):
    """Tests only: decoded copies of the opaque workspaces as a dict of tensors."""
    lib = _lib.load()
    dev = geomBuffer.device
    n_true = int(R)  # (resolves -- and, after an overflow, redoes -- a speculative frame)
    R, lazy_binning = _layout_of(R)
    if lazy_binning is not None:
        binningBuffer = lazy_binning
    T = ((W + 15) // 16) * ((H + 15) // 16)
    with torch.cuda.device(dev):
        out = dict(depths=torch.zeros(P, device=dev), means2D=torch.zeros(P, 2, device=dev),
                   conic_opacity=torch.zeros(P, 4, device=dev), rgb=torch.zeros(P, 3, device=dev),
                   tiles_touched=torch.zeros(P, dtype=torch.int32, device=dev),
                   point_list=torch.zeros(max(R, 0), dtype=torch.int32, device=dev),
                   ranges=torch.zeros(T, 2, dtype=torch.int32, device=dev),
                   n_contrib=torch.zeros(H * W, dtype=torch.int32, device=dev))
        r = lib.goi_raster_debug_views(P, W, H, R, _ptr(geomBuffer), _ptr(binningBuffer), _ptr(imgBuffer),
                                       *[_ptr(out[k]) for k in ("depths", "means2D", "conic_opacity", "rgb",
                                                                "tiles_touched", "point_list", "ranges", "n_contrib")],
                                       _stream(dev))
        if r < 0:
            raise RuntimeError(_lib.last_error())
        torch.cuda.synchronize(dev)
    out["point_list"] = out["point_list"][:n_true]  # (the workspace of a speculative frame holds `capacity` slots)
    return out
