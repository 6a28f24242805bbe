"""Host -> device staging of client state dicts.

Each arriving client's tensors are packed (native multi-threaded memcpy, fa_pack_host) into a
pinned host slot laid out as one row of the :class:`~appfl_amd.layout.Layout`, and the slot is
copied to row k of a device slab on a dedicated copy stream, so the packing of client k+1
overlaps the PCIe transfer of client k.  Stands where the reference materialises per-tensor CPU
tensors and hands them to the aggregator (agent/server.py:505-525, sync_scheduler.py:46).
"""

from __future__ import annotations

import collections
import contextlib
import ctypes as C
import weakref
from typing import Dict, List, Optional

import torch

from . import _native as N
from .layout import Layout

import os

# threads of the native pack / host copy; APPFL_AMD_PACK_THREADS overrides (a server that shares
# its cores with gRPC workers may want fewer)
# (default: 8, or half the CPUs this process may run on where those are fewer than 16 -- the
# pack workers spin briefly between windows, which needs cores of their own)
def _default_pack_threads() -> int:
    try:
        cpus = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        cpus = os.cpu_count() or 1
    return max(1, min(8, cpus // 2))


PACK_THREADS = max(1, min(64, int(os.environ.get("APPFL_AMD_PACK_THREADS", "0")) or _default_pack_threads()))
UPLOAD_WINDOW = 32 << 20


class ClientSlab:
    """K client rows in device memory.  With `device` = cpu (multi-rank rehearsals of the sharded
    aggregator on a machine without a GPU) the slab is host memory and every copy is immediate;
    nothing is computed here either way."""

    def __init__(self, layout: Layout, device: torch.device, slots: int = 3):
        self.layout = layout
        self.device = torch.device(device)
        self.on_gpu = self.device.type == "cuda"
        self.row_bytes = layout.row_bytes
        self.capacity = 0
        self.slab: Optional[torch.Tensor] = None
        self.nslots = slots
        self.pinned = torch.empty((slots, self.row_bytes), dtype=torch.uint8, pin_memory=self.on_gpu)
        self.copy_stream = torch.cuda.Stream(device=device) if self.on_gpu else None
        self.slot_free = [None] * slots   # event recorded after the slot's last H2D copy
        self.next_slot = 0
        self.last_copy: Optional[torch.cuda.Event] = None
        self.row_event: Dict[int, "torch.cuda.Event"] = {}   # row -> event after its last upload
        # row -> [(byte_lo, byte_hi, event after that window's transfer)] of its last upload
        self.row_windows: Dict[int, List] = {}
        self.has_device_sources = False   # the last upload read device tensors of the caller

    def reserve(self, rows: int, preserve: int = 0) -> None:
        """Room for `rows` client rows; the first `preserve` rows keep their contents when the
        slab has to grow (clients staged on arrival, before the round's size is known)."""
        if rows <= self.capacity:
            return
        if preserve:
            rows = max(rows, 2 * self.capacity)
        need = rows * self.row_bytes
        if self.on_gpu:
            free, _ = torch.cuda.mem_get_info(self.device)
            have = 0 if (self.slab is None or preserve) else self.slab.numel()
            if need - have > free:
                raise MemoryError(
                    f"staging {rows} client updates needs {need / 2**30:.1f} GiB of device memory, "
                    f"{free / 2**30:.1f} GiB free")
            # kernels of earlier rounds may still read the old slab on the compute stream
            torch.cuda.current_stream(self.device).synchronize()
            self.copy_stream.synchronize()
        old = self.slab if preserve else None
        self.slab = None
        self.slab = torch.empty((rows, self.row_bytes), dtype=torch.uint8, device=self.device)
        if old is not None:
            self.slab[:preserve].copy_(old[:preserve])
            if self.on_gpu:
                torch.cuda.current_stream(self.device).synchronize()
        self.capacity = rows

    def _on_copy_stream(self):
        return torch.cuda.stream(self.copy_stream) if self.on_gpu else contextlib.nullcontext()

    def _copy_event(self):
        """Event after everything enqueued on the copy stream so far (None on a CPU slab)."""
        if not self.on_gpu:
            return None
        ev = torch.cuda.Event()
        ev.record(self.copy_stream)
        return ev

    def row_ptr(self, k: int) -> int:
        return self.slab.data_ptr() + k * self.row_bytes

    def row_view(self, k: int, byte_offset: int, numel: int, dtype: torch.dtype) -> torch.Tensor:
        item = torch.empty((), dtype=dtype).element_size()
        return self.slab[k, byte_offset:byte_offset + numel * item].view(dtype)

    def begin_round(self) -> None:
        """Order this round's uploads after the kernels that read the slab last round."""
        if self.on_gpu:
            self.copy_stream.wait_stream(torch.cuda.current_stream(self.device))

    def upload(self, k: int, state: Dict[str, torch.Tensor], fill_missing_from=None,
               on_window=None, window: Optional[int] = None) -> None:
        """Pack `state` into row k (`on_window(a, b, event)`, if given, is called after the
        transfer of each window [a, b) of the row has been enqueued; only valid when every entry
        of `state` is a host tensor).  Entries of the layout that `state` lacks and that take the
        weighted step are filled from `fill_missing_from` (the global flat vectors by key, or the
        fp32 one alone), which makes
        their difference zero -- the reference skips such a client for that entry
        (fedavg_aggregator.py:360-361); a missing plain-mean entry raises KeyError (:289).
        `window`: bytes per pack / transfer window (default UPLOAD_WINDOW); the events of the
        windows are kept in `row_windows[k]` when every source is a host tensor."""
        lay = self.layout
        host_src: List[int] = []
        host_off: List[int] = []
        host_len: List[int] = []
        keep = {}          # name -> converted tensor; must outlive the memcpy / the queued copies
        dev_items = []
        missing = []
        for e in lay.entries:          # a windowed layout lists one piece per (entry, window)
            t = keep.get(e.name)
            if t is None:
                t = state.get(e.name)
                if t is None:
                    if e.cls not in ("step", "step_f64"):
                        raise KeyError(e.name)
                    missing.append(e)
                    continue
                full = e.numel if e.full_numel < 0 else e.full_numel
                if t.numel() != full:
                    raise ValueError(f"{e.name}: client tensor has {t.numel()} elements, "
                                     f"global state has {full}")
                if t.dtype != e.dtype:
                    t = t.to(e.dtype)
                if not t.is_contiguous():
                    t = t.contiguous()
                keep[e.name] = t
            if t.is_cuda:
                dev_items.append((e, t))
                continue
            if e.numel:
                host_src.append(t.data_ptr() + e.src_offset * t.element_size())
                host_off.append(e.byte_offset)
                host_len.append(e.numel * t.element_size())
        slot = self.next_slot
        self.next_slot = (slot + 1) % self.nslots
        if self.slot_free[slot] is not None:
            self.slot_free[slot].synchronize()
        T = len(host_src)
        if self.on_gpu and (dev_items or missing):
            # device-side sources may still be being written on the caller's stream
            self.copy_stream.wait_stream(torch.cuda.current_stream(self.device))
        if T:
            # pack and transfer in windows of the row, so the PCIe copy of one window runs while
            # the next is being packed (a single large client would otherwise pay both in turn)
            srcs, offs, lens = ((C.c_void_p * T)(*host_src), (C.c_size_t * T)(*host_off),
                                (C.c_size_t * T)(*host_len))
            pinned, lib = self.pinned[slot], N.load()
            step = max(1 << 20, int(window)) if window else UPLOAD_WINDOW
            record = window is not None and self.on_gpu and not dev_items and not missing
            wins = []
            for a in range(0, self.row_bytes, step):
                b = min(self.row_bytes, a + step)
                rc = lib.fa_pack_host_window(C.c_void_p(pinned.data_ptr()), srcs, offs, lens, T,
                                             a, b, PACK_THREADS)
                N.check(rc, "fa_pack_host_window")
                with self._on_copy_stream():
                    self.slab[k, a:b].copy_(pinned[a:b], non_blocking=True)
                    landed = self._copy_event() if (on_window is not None or record) else None
                if record:
                    wins.append((a, b, landed))
                if on_window is not None:
                    # lets the caller start work on bytes [a, b) of the row once `landed` fires
                    on_window(a, b, landed)
        with self._on_copy_stream():
            for e, t in dev_items:
                src = t.view(-1)[e.src_offset:e.src_offset + e.numel]
                self.row_view(k, e.byte_offset, e.numel, e.dtype).copy_(src, non_blocking=True)
                if self.on_gpu:
                    # the source (possibly a temporary made above on the caller's stream) must not
                    # be handed out again by the allocator before this queued copy has run
                    t.record_stream(self.copy_stream)
            for e in missing:
                off = lay.flat_offset(e)
                src = (fill_missing_from[lay.flat_key(e.cls)] if isinstance(fill_missing_from, dict)
                       else fill_missing_from)
                self.row_view(k, e.byte_offset, e.numel, e.dtype).copy_(
                    src[off:off + e.numel], non_blocking=True)
            ev = self._copy_event()
        self.slot_free[slot] = ev
        self.last_copy = ev
        self.row_windows.pop(k, None)
        if ev is not None:
            self.row_event[k] = ev
            if T and record and len(wins) > 1:
                self.row_windows[k] = wins
        self.has_device_sources = bool(dev_items)
        del keep

    def wait_uploads(self) -> None:
        """Blocks the host until every upload enqueued so far has landed (a caller that hands in
        device tensors may overwrite them once this returns)."""
        if self.last_copy is not None:
            self.last_copy.synchronize()

    def ready_prefix(self, rows) -> int:
        """How many leading rows of `rows` (row indices in client order) have landed already.
        A kernel launched now may read those; the others are behind `last_copy`."""
        n = 0
        for r in rows:
            ev = self.row_event.get(r)
            if ev is not None and not ev.query():
                break
            n += 1
        return n

    def end_round(self) -> None:
        """Make the compute stream wait for every upload of this round."""
        if self.last_copy is not None:
            torch.cuda.current_stream(self.device).wait_event(self.last_copy)


class _PinnedPool:
    """Pinned host buffers handed out as the memory of result tensors.  A tensor made with
    `torch.frombuffer(lease)` keeps `lease` (a numpy view of the pinned buffer) referenced from
    its storage, so the finalizer of `lease` fires exactly when the last tensor sharing that
    storage -- views included -- has died; only then does the buffer go back on the free list."""

    MAX_BYTES = 8 << 30            # pinned memory held by live + free leases; beyond it: no lease

    def __init__(self):
        self._free = collections.deque()
        self._held = 0

    def lease(self, nbytes: int):
        """A numpy view of `nbytes` bytes of a pinned buffer (smallest free one that fits, else a
        new one), or None when the pool is exhausted."""
        best = None
        for _ in range(len(self._free)):
            buf = self._free.popleft()
            if buf.numel() >= nbytes and (best is None or buf.numel() < best.numel()):
                if best is not None:
                    self._free.append(best)
                best = buf
            else:
                self._free.append(buf)
        if best is None:
            size = max(nbytes, 1 << 20)
            if self._held + size > self.MAX_BYTES:
                # give back the free buffers that were too small before giving up
                while self._free:
                    self._held -= self._free.popleft().numel()
                if self._held + size > self.MAX_BYTES:
                    return None
            try:
                best = torch.empty(size, dtype=torch.uint8, pin_memory=True)
            except RuntimeError:       # the host refuses to lock more pages: no lease
                return None
            self._held += size
        lease = best.numpy()[:nbytes]
        weakref.finalize(lease, self._free.append, best)
        return lease


class HostDownloader:
    """Device -> host copies of flat state vectors into tensors nobody else holds.

    Preferred: the result tensor's memory IS a pinned buffer leased from a pool, so the D2H
    transfer (PCIe rate) lands in place and nothing is copied on the host; the buffer returns to
    the pool when the caller drops the last tensor that views it.  When the pool is exhausted
    (a caller keeping many results alive): pinned D2H transfers of DOWNLOAD_CHUNK bytes into a
    reusable staging buffer, each followed -- while the next one is still in flight -- by a
    multi-threaded copy into a fresh pageable tensor first touched through huge pages."""

    DOWNLOAD_CHUNK = 32 << 20

    def __init__(self, pool: Optional[_PinnedPool] = None):
        self._pinned: Dict[torch.dtype, torch.Tensor] = {}
        self._pool = pool if pool is not None else _PinnedPool()

    def fetch(self, flat: torch.Tensor, n: int, wait: bool = True) -> torch.Tensor:
        """`wait=False` only enqueues the transfer: the caller does host work meanwhile (building
        the per-entry views of the result takes a quarter of the transfer's time for a ResNet)
        and must call :meth:`wait` for `flat`'s device before reading the tensor."""
        if n == 0:
            return torch.empty(0, dtype=flat.dtype)
        if not flat.is_cuda:           # CPU rehearsal of the sharded aggregator: already host memory
            return flat[:n].clone()
        item = flat.element_size()
        lease = self._pool.lease(n * item) if self._pool is not None else None
        if lease is not None:
            out = torch.frombuffer(lease, dtype=flat.dtype, count=n)
            del lease
            out.copy_(flat[:n], non_blocking=True)
            if wait:
                self.wait(flat.device)
            return out
        return self._fetch_pageable(flat, n)

    @staticmethod
    def wait(device) -> None:
        """Blocks until the transfers enqueued by `fetch(..., wait=False)` have landed."""
        if torch.device(device).type == "cuda":
            torch.cuda.current_stream(device).synchronize()

    def fetch_parts(self, parts: List[torch.Tensor]) -> torch.Tensor:
        """One fresh host tensor holding the concatenation of `parts` (1-D device tensors of one
        dtype, e.g. the blocks of a gathered model): every part is its own D2H copy into its
        place in one leased pinned buffer."""
        parts = [p for p in parts if p.numel()]
        if not parts:
            return torch.empty(0, dtype=torch.float32)
        n = sum(p.numel() for p in parts)
        if not parts[0].is_cuda:
            return torch.cat([p.reshape(-1) for p in parts]).clone()
        out = self.lease(parts[0].dtype, n)
        if out is None:
            return self._fetch_pageable(torch.cat(parts), n)
        pos = 0
        for p in parts:
            out[pos:pos + p.numel()].copy_(p, non_blocking=True)
            pos += p.numel()
        torch.cuda.current_stream(parts[0].device).synchronize()
        return out

    def lease(self, dtype: torch.dtype, n: int) -> Optional[torch.Tensor]:
        """An n-element result tensor on a leased pinned buffer for a caller that fills it
        itself (piecewise, on its own stream); None when the pool is exhausted."""
        if n == 0 or self._pool is None:
            return None
        lease = self._pool.lease(n * torch.empty((), dtype=dtype).element_size())
        if lease is None:
            return None
        return torch.frombuffer(lease, dtype=dtype, count=n)

    def _fetch_pageable(self, flat: torch.Tensor, n: int) -> torch.Tensor:
        buf = self._pinned.get(flat.dtype)
        if buf is None or buf.numel() < n:
            buf = torch.empty(n, dtype=flat.dtype, pin_memory=True)
            self._pinned[flat.dtype] = buf
        item = flat.element_size()
        per = max(1, self.DOWNLOAD_CHUNK // item)
        stream = torch.cuda.current_stream(flat.device)
        landed = []
        for a in range(0, n, per):
            b = min(n, a + per)
            buf[a:b].copy_(flat[a:b], non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(stream)
            landed.append((a, b, ev))
        out = torch.empty(n, dtype=flat.dtype)
        lib = N.load()
        for a, b, ev in landed:
            ev.synchronize()
            rc = lib.fa_host_copy_fresh(C.c_void_p(out.data_ptr() + a * item),
                                        C.c_void_p(buf.data_ptr() + a * item),
                                        (b - a) * item, PACK_THREADS)
            N.check(rc, "fa_host_copy_fresh")
        return out
