"""State dict <-> flat device vectors.

The aggregation kernels work on flat vectors, one per arithmetic class:

* ``step``     fp32 entries that take the weighted step (and the server optimizer)
* ``mean_f32`` fp32 entries that take the plain mean (non-parameters when a model was given)
* ``step_f64`` / ``mean_f64`` the same two classes for float64 entries
* ``mean_i64`` / ``mean_i32`` integer entries (plain mean with truncating division)

(classification: algorithm/aggregator/fedavg_aggregator.py:333-340 of the reference.)

bfloat16 / float16 entries take the two fp32 classes and stay 16 bits wide: in the rows and in
the flat vector (still keyed ``"f32"``) their elements are 2 bytes, and the kernels of
:mod:`appfl_amd.half_ops` round every operation to the 16-bit type.  All entries of the two
classes must share one type (`Layout.float_dtype`): a state dict that mixes fp32 with 16-bit
tensors (or the two 16-bit types) is refused.

A client's state dict is packed into one ROW of bytes with the segments in that order::

    [ step fp32 ... | mean fp32 ... | pad to 16 | step fp64 ... | mean fp64 ... | int64 ... | int32 ... | pad to 256 ]

so the fp32 part of every row is one contiguous, 16-byte aligned vector whose first ``n_step``
elements are the weighted class (likewise the fp64 part), and the global state uses the same
offsets.
"""

from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, Iterable, List, Optional, Sequence, Set, Tuple

import torch

STEP, MEAN_F32, MEAN_I64, MEAN_I32 = "step", "mean_f32", "mean_i64", "mean_i32"
STEP_F64, MEAN_F64 = "step_f64", "mean_f64"
CLASSES = (STEP, MEAN_F32, STEP_F64, MEAN_F64, MEAN_I64, MEAN_I32)      # order inside a row
STEP_CLASSES = (STEP, STEP_F64)
_ITEMSIZE = {STEP: 4, MEAN_F32: 4, STEP_F64: 8, MEAN_F64: 8, MEAN_I64: 8, MEAN_I32: 4}
_DTYPE = {STEP: torch.float32, MEAN_F32: torch.float32, STEP_F64: torch.float64,
          MEAN_F64: torch.float64, MEAN_I64: torch.int64, MEAN_I32: torch.int32}
ROW_ALIGN = 256
HALF_DTYPES = (torch.bfloat16, torch.float16)


def _round_up(x: int, a: int) -> int:
    return (x + a - 1) // a * a


@dataclass(frozen=True)
class Entry:
    name: str
    shape: Tuple[int, ...]
    dtype: torch.dtype
    numel: int
    cls: str
    offset: int        # element offset inside its class segment
    byte_offset: int   # byte offset inside a row
    src_offset: int = 0    # first element of the client's tensor this entry covers
    full_numel: int = -1   # elements of the client's whole tensor (-1: numel, the entry is whole)


def classify(dtype: torch.dtype, is_parameter: bool) -> str:
    """fedavg_aggregator.py:333-340: integers and non-parameters bypass the weighted step."""
    if dtype == torch.int64:
        return MEAN_I64
    if dtype == torch.int32:
        return MEAN_I32
    if dtype == torch.float64:
        return STEP_F64 if is_parameter else MEAN_F64
    if dtype not in (torch.float32,) + HALF_DTYPES:
        raise NotImplementedError(
            f"state-dict entries of dtype {dtype} are not supported by the MI355X aggregation "
            "kernels (fp32, bf16, fp16, fp64, int64 and int32 are); there is no CPU fallback")
    return STEP if is_parameter else MEAN_F32


class Layout:
    """Offsets of every entry of the global state inside the flat vectors and inside a row."""

    def __init__(self, names_shapes_dtypes: Sequence[Tuple[str, Tuple[int, ...], torch.dtype]],
                 named_parameters: Optional[Set[str]] = None):
        order = {c: [] for c in CLASSES}
        kinds = set()
        for name, shape, dtype in names_shapes_dtypes:
            is_param = named_parameters is None or name in named_parameters
            cls = classify(dtype, is_param)
            order[cls].append((name, tuple(shape), dtype))
            if cls in (STEP, MEAN_F32):
                kinds.add(dtype)
        if len(kinds) > 1:
            raise NotImplementedError(
                f"a state dict that mixes {sorted(str(k) for k in kinds)} entries is not supported by "
                "the MI355X aggregation kernels: fp32, bfloat16 and float16 tensors each need a state "
                "dict of their own kind (float64 and integer entries may accompany any of them)")
        # the type of the entries of the two fp32 classes: float32, or the 16-bit type of a
        # bfloat16 / float16 model (`half_dtype` names it, None for fp32)
        self.float_dtype: torch.dtype = next(iter(kinds), torch.float32)
        self.half_dtype: Optional[torch.dtype] = self.float_dtype if self.float_dtype in HALF_DTYPES else None
        self.count = {c: sum(_numel(s) for _, s, _ in order[c]) for c in order}
        self.n_f32 = self.count[STEP] + self.count[MEAN_F32]
        self.n_f64 = self.count[STEP_F64] + self.count[MEAN_F64]
        self._place_segments()
        by_name: Dict[str, Entry] = {}
        for c in CLASSES:
            off = 0
            for name, shape, dtype in order[c]:
                n = _numel(shape)
                by_name[name] = Entry(name, shape, dtype, n, c, off,
                                      self.seg_byte[c] + off * self.itemsize(c))
                off += n
        # global-state (dict) order is the order the names were given in
        self.entries: List[Entry] = [by_name[n] for n, _, _ in names_shapes_dtypes]
        self.by_name = by_name
        self.names = [e.name for e in self.entries]

    def itemsize(self, cls: str) -> int:
        """Bytes per element of class `cls` in a row and in its flat vector."""
        if cls in (STEP, MEAN_F32) and self.half_dtype is not None:
            return 2
        return _ITEMSIZE[cls]

    def _place_segments(self) -> None:
        """Byte offsets of the class segments inside a row, from `count` / `n_f32` / `n_f64`."""
        fsize = self.itemsize(STEP)
        self.seg_byte = {STEP: 0, MEAN_F32: fsize * self.count[STEP]}
        self.seg_byte[STEP_F64] = _round_up(fsize * self.n_f32, 16)
        self.seg_byte[MEAN_F64] = self.seg_byte[STEP_F64] + 8 * self.count[STEP_F64]
        self.seg_byte[MEAN_I64] = self.seg_byte[STEP_F64] + 8 * self.n_f64
        self.seg_byte[MEAN_I32] = self.seg_byte[MEAN_I64] + 8 * self.count[MEAN_I64]
        self.row_bytes = _round_up(self.seg_byte[MEAN_I32] + 4 * self.count[MEAN_I32], ROW_ALIGN)

    # -- flat views ---------------------------------------------------------------------------
    def alloc_state(self, device) -> Dict[str, torch.Tensor]:
        """Flat global-state vectors: 'f32' (step segment first, then mean_f32), 'f64' (same),
        'i64', 'i32'."""
        return {
            "f32": torch.zeros(max(self.n_f32, 1), dtype=self.float_dtype, device=device),
            "f64": torch.zeros(max(self.n_f64, 1), dtype=torch.float64, device=device),
            "i64": torch.zeros(max(self.count[MEAN_I64], 1), dtype=torch.int64, device=device),
            "i32": torch.zeros(max(self.count[MEAN_I32], 1), dtype=torch.int32, device=device),
        }

    @staticmethod
    def flat_key(cls: str) -> str:
        return {STEP: "f32", MEAN_F32: "f32", STEP_F64: "f64", MEAN_F64: "f64", MEAN_I64: "i64",
                MEAN_I32: "i32"}[cls]

    def flat_sizes(self) -> Dict[str, int]:
        """Elements of each flat vector that hold entries."""
        return {"f32": self.n_f32, "f64": self.n_f64, "i64": self.count[MEAN_I64],
                "i32": self.count[MEAN_I32]}

    def flat_offset(self, e: Entry) -> int:
        if e.cls == MEAN_F32:
            return e.offset + self.count[STEP]
        if e.cls == MEAN_F64:
            return e.offset + self.count[STEP_F64]
        return e.offset

    def view(self, flats: Dict[str, torch.Tensor], name: str) -> torch.Tensor:
        e = self.by_name[name]
        off = self.flat_offset(e)
        return flats[self.flat_key(e.cls)][off:off + e.numel].view(e.shape)

    def views(self, flats: Dict[str, torch.Tensor], classes: Optional[Iterable[str]] = None
              ) -> Dict[str, torch.Tensor]:
        keep = None if classes is None else set(classes)
        return {e.name: self.view(flats, e.name) for e in self.entries
                if keep is None or e.cls in keep}

    def matches(self, state: Dict[str, torch.Tensor]) -> bool:
        """True when `state` has exactly the layout's names, shapes and dtypes."""
        if len(state) != len(self.entries):
            return False
        for e in self.entries:
            t = state.get(e.name)
            if t is None or tuple(t.shape) != e.shape or t.dtype != e.dtype:
                return False
        return True


class WindowedLayout(Layout):
    """Row layout of a rank that owns only some ranges of the weighted (step) segment.

    `windows` = (global_lo, global_hi, local_offset) triples: elements [global_lo, global_hi) of
    the full layout's step segment live at [local_offset, ...) of this rank's step storage, which
    is `storage` elements long (a multiple of 16 bytes; what lies beyond the windows is padding).  Rows
    and state vectors keep the shape of :class:`Layout` -- ``[step storage | mean fp32 | ints]``
    -- so the staging slab and the kernels work on them unchanged; `entries` holds one PIECE per
    (step entry, window) intersection, so a name may appear more than once, and the plain-mean
    and float64 entries whole (they are kept on every rank).  Name-based views belong to the
    full layout.
    """

    def __init__(self, full: Layout, windows: Sequence[Tuple[int, int, int]], storage: int):
        self.half_dtype, self.float_dtype = full.half_dtype, full.float_dtype
        fsize = self.itemsize(STEP)
        if storage * fsize % 16:
            raise ValueError("step storage must be a multiple of 16 bytes")
        self.full = full
        self.count = dict(full.count)
        self.count[STEP] = storage
        self.live_step = max((loc + hi - lo for lo, hi, loc in windows if hi > lo), default=0)
        if self.live_step > storage:
            raise ValueError("windows exceed the step storage")
        self.n_f32 = storage + full.count[MEAN_F32]
        self.n_f64 = full.n_f64          # float64 entries (rare, small) stay whole on every rank
        self._place_segments()
        self.entries = []
        for e in full.entries:
            if e.cls != STEP:
                self.entries.append(Entry(e.name, e.shape, e.dtype, e.numel, e.cls, e.offset,
                                          self.seg_byte[e.cls] + e.offset * self.itemsize(e.cls)))
                continue
            for lo, hi, loc in windows:
                a, b = max(e.offset, lo), min(e.offset + e.numel, hi)
                if a < b:
                    off = loc + a - lo
                    self.entries.append(Entry(e.name, e.shape, e.dtype, b - a, STEP, off, fsize * off,
                                              src_offset=a - e.offset, full_numel=e.numel))
        self.by_name = {}
        self.names = list(full.names)

    def view(self, flats, name):
        raise TypeError("a windowed layout has no per-name views; use the full layout")

    def views(self, flats, classes=None):
        raise TypeError("a windowed layout has no per-name views; use the full layout")

    def matches(self, state):
        return self.full.matches(state)


def _numel(shape: Tuple[int, ...]) -> int:
    n = 1
    for s in shape:
        n *= int(s)
    return n


def split_names_by_size(sizes: Dict[str, int], max_chunk_size: int) -> List[List[str]]:
    """Chunking rule of misc/memory_utils.py:317-379 on (name -> bytes): names sorted, greedy
    fill up to max_chunk_size, an oversized tensor gets a chunk of its own."""
    chunks: List[List[str]] = []
    cur: List[str] = []
    cur_size = 0
    for key in sorted(sizes):
        sz = sizes[key]
        if sz > max_chunk_size:
            if cur:
                chunks.append(cur)
                cur, cur_size = [], 0
            chunks.append([key])
            continue
        if cur_size + sz > max_chunk_size and cur:
            chunks.append(cur)
            cur, cur_size = [], 0
        cur.append(key)
        cur_size += sz
    if cur:
        chunks.append(cur)
    return chunks
