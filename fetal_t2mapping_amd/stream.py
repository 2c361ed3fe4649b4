"""Streaming many subjects through one GPU (BASELINE.json config 5: 32 x 256^3 x 8 TE).

The reference handles subjects one after another inside ``process_t2maps`` (run_t2mapping.py:358).
Here the host->HBM copy of subject s+1 and the HBM->host copy of subject s-1 overlap the fit of
subject s: pinned staging buffers, one copy stream per direction, HIP events for the hand-offs.
Results are identical to calling ``fit_volume`` per subject (voxels are independent).
"""
from __future__ import annotations

import ctypes as C
import os
import sys
from concurrent.futures import ThreadPoolExecutor
from typing import Iterable, Iterator, Optional, Tuple

import numpy as np

from . import _abi
from ._lib import check, require_gpu
from .t2map import T2Maps, make_config


_COPY_THREADS = 8
_pool = None


def _parallel_copy(dst: np.ndarray, src: np.ndarray) -> None:
    """dst[:] = src for large flat float32 arrays, split over a few threads: a single-threaded
    memcpy into the pinned staging buffer (about 10 GB/s) would otherwise cost more than the PCIe
    transfer and the fit together (numpy releases the GIL while copying)."""
    global _pool
    n = dst.size
    if n < (1 << 22):
        dst[:] = src
        return
    if _pool is None:
        _pool = ThreadPoolExecutor(_COPY_THREADS)
    step = -(-n // _COPY_THREADS)
    list(_pool.map(lambda i: dst.__setitem__(slice(i, min(i + step, n)), src[i:min(i + step, n)]), range(0, n, step)))


class _Slot:
    def __init__(self, dev):
        self.dev = dev
        self.cap_e = self.cap_n = 0
        self.meta = None
        self.d2h_queued = False
        self.tl = None

    def ensure(self, n_te: int, n: int, need_h_in: bool):
        import torch

        if n_te * n > self.cap_e or n > self.cap_n:
            self.cap_e, self.cap_n = max(self.cap_e, n_te * n), max(self.cap_n, n)
            self.h_in = None
            self.h_mask = torch.empty(self.cap_n, dtype=torch.uint8).pin_memory()
            self.h_out = torch.empty(4 * self.cap_n, dtype=torch.float32).pin_memory()
            self.d_in = torch.empty(self.cap_e, dtype=torch.float32, device=self.dev)
            self.d_mask = torch.empty(self.cap_n, dtype=torch.uint8, device=self.dev)
            self.d_out = torch.empty(4 * self.cap_n, dtype=torch.float32, device=self.dev)
            self.ev_h2d, self.ev_fit, self.ev_d2h = (torch.cuda.Event() for _ in range(3))
        if need_h_in and self.h_in is None:  # only inputs that are not already pinned get staged
            self.h_in = torch.empty(self.cap_e, dtype=torch.float32).pin_memory()


# Page-locking a 256^3 x 8 TE staging set takes longer than fitting several subjects, so the
# slots (pinned + device buffers, events) and the two copy streams outlive a fit_subjects() call.
_contexts = {}


def _context(dev, depth: int):
    import torch

    key = (dev.index, depth)
    if key not in _contexts:
        _contexts[key] = ([_Slot(dev) for _ in range(depth)], torch.cuda.Stream(dev), torch.cuda.Stream(dev))
    return _contexts[key]


def release(device=None) -> None:
    """Drop the cached staging buffers (all devices, or one)."""
    for key in [k for k in _contexts if device is None or k[0] == device]:
        del _contexts[key]


def fit_subjects(subjects: Iterable[Tuple[np.ndarray, Optional[np.ndarray]]], TEeffs, fit, fit_params, prior=True,
                 norm=False, *, solver="lbfgsb", precision="f64", device=0, depth=2, copy_out=True) -> Iterator[T2Maps]:
    """Yield the maps of each ``(echoes (nTE,Z,Y,X) float32, mask (Z,Y,X) or None)`` in order.

    ``echoes`` may be a numpy array (copied into a pinned staging buffer by a few threads) or an
    already pinned, contiguous float32 CPU torch tensor (DMA'd from where it lies: the decode stage
    can write straight into pinned memory).  ``copy_out=False`` yields views of the pinned output
    buffer instead of copies; they stay valid until ``depth`` further subjects have been yielded.
    """
    import torch

    lib = require_gpu()
    cfg = make_config(fit, fit_params, TEeffs, prior, norm, solver, precision)
    dev = torch.device("cuda", device)
    with torch.cuda.device(dev):
        compute = torch.cuda.current_stream()
        slots, s_in, s_out = _context(dev, depth)
        if any(slot.meta is not None for slot in slots):  # an abandoned earlier generator may have left work in flight
            torch.cuda.synchronize(dev)
            for slot in slots:
                slot.meta, slot.d2h_queued, slot.tl = None, False, None

        def queue_d2h(slot):
            if slot.meta is None or slot.d2h_queued:
                return
            n = slot.meta[1]
            with torch.cuda.stream(s_out):
                s_out.wait_event(slot.ev_fit)
                if slot.tl is not None:
                    slot.tl[4].record(s_out)
                slot.h_out[: 4 * n].copy_(slot.d_out[: 4 * n], non_blocking=True)
                slot.ev_d2h.record(s_out)
                if slot.tl is not None:
                    slot.tl[5].record(s_out)
            slot.d2h_queued = True

        def finish(slot):
            queue_d2h(slot)
            slot.ev_d2h.synchronize()
            shape, n = slot.meta
            if copy_out:
                out = np.empty((4, n), np.float32)
                _parallel_copy(out.reshape(-1), slot.h_out[: 4 * n].numpy())
            else:
                out = slot.h_out[: 4 * n].numpy().reshape(4, n)
            slot.meta = None
            return T2Maps(*(out[j].reshape(shape) for j in range(4)))

        timeline = [] if os.environ.get("T2FIT_STREAM_TIMELINE") else None  # diagnostic: per-stage HIP-event times
        n_sub = 0
        prev = None
        for s, (echoes, mask) in enumerate(subjects):
            n_sub = s + 1
            slot = slots[s % depth]
            if slot.meta is not None:
                yield finish(slot)  # also frees the slot's pinned buffers for reuse
            n_te = echoes.shape[0]
            if n_te != cfg.n_te:
                raise ValueError("every subject must have the configured number of echoes")
            shape = tuple(echoes.shape[1:])
            n = int(np.prod(shape))
            direct = (isinstance(echoes, torch.Tensor) and echoes.dtype == torch.float32 and echoes.is_contiguous()
                      and not echoes.is_cuda and echoes.is_pinned())
            slot.ensure(n_te, n, not direct)
            src = echoes.reshape(-1) if direct else slot.h_in[: n_te * n]
            if not direct:
                _parallel_copy(slot.h_in[: n_te * n].numpy(), np.ascontiguousarray(echoes, np.float32).reshape(-1))
            if mask is None:
                slot.h_mask[:n].numpy()[:] = 1
            else:
                mk = np.asarray(mask).reshape(-1)
                # the kernels test mask != 0 themselves: one-byte masks are staged as they are
                _parallel_copy(slot.h_mask[:n].numpy(), mk.view(np.uint8) if mk.dtype.itemsize == 1 else (mk != 0).view(np.uint8))
            tl = None
            if timeline is not None:
                tl = [torch.cuda.Event(enable_timing=True) for _ in range(6)]
                timeline.append(tl)
                tl[0].record(s_in)
            with torch.cuda.stream(s_in):
                slot.d_in[: n_te * n].copy_(src, non_blocking=True)
                slot.d_mask[:n].copy_(slot.h_mask[:n], non_blocking=True)
                slot.ev_h2d.record(s_in)
                if tl is not None:
                    tl[1].record(s_in)
            # The two copy directions share one engine queue on this platform and a queued copy that waits for a
            # kernel blocks everything behind it: the device->host copy of the PREVIOUS subject (which waits for
            # its fit) is therefore queued only now, after this subject's host->device copy.
            if prev is not None:
                queue_d2h(prev)
            compute.wait_event(slot.ev_h2d)
            if tl is not None:
                tl[2].record(compute)
            maps = _abi.T2FitMaps()
            base = slot.d_out.data_ptr()
            maps.t2, maps.k, maps.sigma, maps.res = (base + 4 * n * j for j in range(4))
            check(lib.t2fit_volume_dev(C.byref(cfg), slot.d_in.data_ptr(), _abi.LAYOUT_TE_MAJOR, slot.d_mask.data_ptr(),
                                       n, C.byref(maps), C.c_void_p(compute.cuda_stream)))
            slot.ev_fit.record(compute)
            if tl is not None:
                tl[3].record(compute)
            slot.meta, slot.d2h_queued, slot.tl = (shape, n), False, tl
            prev = slot
        if prev is not None:
            queue_d2h(prev)
        for k in range(n_sub - min(depth, n_sub), n_sub):  # drain in submission order
            slot = slots[k % depth]
            if slot.meta is not None:
                yield finish(slot)
        if timeline:
            torch.cuda.synchronize()
            t0 = timeline[0][0]
            for i, tl in enumerate(timeline):  # ms since the first copy was queued: h2d, fit, d2h as [start, end]
                print(f"[t2fit stream] subject {i}: " + "  ".join(
                    f"{name} {t0.elapsed_time(tl[a]):7.2f}-{t0.elapsed_time(tl[b]):7.2f}" for name, a, b in
                    (("h2d", 0, 1), ("fit", 2, 3), ("d2h", 4, 5))), file=sys.stderr)
