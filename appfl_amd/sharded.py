"""Parameter-axis sharding of the FedAvg-family step over the GPUs of one node.

Every element of the flat parameter vector is independent, so the vector is dealt to the R ranks
(one process per GPU, `torch.distributed`, backend "nccl" = RCCL over xGMI) in BLOCKS: with
``pieces`` blocks per rank of ``per`` elements each (a multiple of 64 = 256 bytes; of 4 for tiny vectors), block (p, r)
holds the elements ``[(p*R + r)*per, (p*R + r + 1)*per)`` of the vector padded to
``P = pieces*R*per``, and rank r stores its blocks back to back (``L = pieces*per`` elements).
The point of dealing blocks instead of one contiguous slice per rank: the all-gather of block
row p is ONE `all_gather_into_tensor` that lands in place, as the contiguous stretch
``[p*R*per, (p+1)*R*per)`` of the gathered model -- no list outputs, no device-side reshuffle --
so the gathers of the rows can be pipelined behind the kernels that produce them.  With
``pieces = 1`` this is the plain contiguous split.  The global model g, the moments m / v and
FedBuff-style buffers live sharded and stay resident.

Two ways client updates reach the GPUs (no counterpart in the reference, whose server is one CPU
process -- comm/mpi/mpi_server_communicator.py:56-90):

* **sliced** -- rank r holds its blocks of every client update: one local fused kernel, no
  collective in the arithmetic; the all-gather only hands every rank the whole new model.
  Bit-identical to the single-GPU result.
* **whole**  -- whole client updates sit on different GPUs: each rank folds its clients into a
  full-length partial step against its copy of g, RCCL reduce-scatter (sum) leaves every rank
  the step of its blocks, the optimizer epilogue and the apply run there, an all-gather
  refreshes the full g.  The cross-rank sum reorders the fp32 additions, so results agree with
  the single-GPU order to ~1e-7 relative, not bit for bit.

Vectors are fp32, or bfloat16 / float16 for a 16-bit model (sliced rounds only; `ShardMap` then
counts 2-byte elements, so blocks keep their 16- / 256-byte boundaries).

`backend` is the module that provides the flat kernels (default: :mod:`appfl_amd.ops`, the HIP
library).  It exists ONLY so the tests can rehearse the multi-rank logic on CPU with gloo by
passing a stand-in; the product never passes it and has no CPU compute path.
"""

from __future__ import annotations

from typing import List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist

WHOLE_MODE_16BIT = ("whole-update rounds are not supported for bfloat16 / float16 models: the cross-rank "
                    "sum of 16-bit partial steps would be rounded once per rank (up to 2**-8 relative "
                    "per addition for bfloat16) in an order no single-GPU round has; use sliced rounds")
PIECE_BYTES = 128 << 20     # a block row is gathered on its own once a shard has several of these
MAX_PIECES = 8


def default_pieces(n: int, world: int, itemsize: int = 4) -> int:
    """Blocks per rank: one per PIECE_BYTES of shard, 1..MAX_PIECES.  A function of (n, world,
    bytes per element) only, so every rank derives the same value."""
    shard_bytes = itemsize * ((n + world - 1) // world)
    return max(1, min(MAX_PIECES, shard_bytes // PIECE_BYTES))


class ShardMap:
    """Which elements of an n-element vector rank `rank` of `world` owns, and where.
    `itemsize`: bytes per element (4; 2 for the vectors of a bfloat16 / float16 model)."""

    def __init__(self, n: int, world: int, rank: int, pieces: Optional[int] = None, itemsize: int = 4):
        if n < 0 or world < 1 or not 0 <= rank < world:
            raise ValueError("bad shard geometry")
        if itemsize not in (2, 4):
            raise ValueError("elements of 2 or 4 bytes")
        self.n, self.world, self.rank, self.itemsize = n, world, rank, itemsize
        self.pieces = default_pieces(n, world, itemsize) if pieces is None else max(1, int(pieces))
        q = (n + world * self.pieces - 1) // (world * self.pieces)
        # blocks of a real model start on 256-byte boundaries (64 fp32 elements): rows whose
        # blocks straddle the cache lines of a wave's 1-KiB loads read 5-7 % slower; tiny vectors
        # keep the 16-byte granularity so that every rank still gets elements
        big, small = 256 // itemsize, 16 // itemsize
        self.per = (q + big - 1) // big * big if q >= 1024 else max(small, (q + small - 1) // small * small)
        self.L = self.pieces * self.per          # elements of local storage
        self.P = self.world * self.L             # elements of the padded gathered vector
        self.n_local = sum(self.live(p) for p in range(self.pieces))

    def block(self, p: int, rank: Optional[int] = None) -> Tuple[int, int]:
        """[lo, hi) of block (p, rank) inside the un-padded vector (may be short or empty)."""
        r = self.rank if rank is None else rank
        lo = min(self.n, (p * self.world + r) * self.per)
        return lo, min(self.n, lo + self.per)

    def live(self, p: int, rank: Optional[int] = None) -> int:
        lo, hi = self.block(p, rank)
        return hi - lo

    def windows(self, rank: Optional[int] = None) -> List[Tuple[int, int, int]]:
        """(global_lo, global_hi, local_offset) of this rank's non-empty blocks.  Block starts
        grow with p, so the live elements are a prefix [0, n_local) of the local storage."""
        out = []
        for p in range(self.pieces):
            lo, hi = self.block(p, rank)
            if hi > lo:
                out.append((lo, hi, p * self.per))
        return out

    def row(self, p: int) -> Tuple[int, int]:
        """[lo, hi) of block row p (all ranks' blocks p) inside the padded gathered vector."""
        return p * self.world * self.per, (p + 1) * self.world * self.per

    def take(self, full: torch.Tensor, rank: Optional[int] = None) -> torch.Tensor:
        """This rank's local storage (L elements, zero padded) cut out of a full-length vector."""
        out = torch.zeros(self.L, dtype=full.dtype, device=full.device)
        for lo, hi, loc in self.windows(rank):
            out[loc:loc + hi - lo].copy_(full[lo:hi])
        return out


def _shift(x, elements: int, itemsize: int = 4):
    """Client buffer `x` (device tensor, or a raw device address the caller vouches for) advanced
    by `elements` elements of `itemsize` bytes."""
    return x + itemsize * elements if isinstance(x, int) else x[elements:]


def gather_rows(smap: ShardMap, g: torch.Tensor, out: torch.Tensor, group=None,
                rows: Optional[Sequence[int]] = None, async_op: bool = False):
    """All-gather of block rows of the sharded vector `g` (L elements) into the padded full
    vector `out` (P elements): one `all_gather_into_tensor` per row, landing in place."""
    works = []
    for p in (range(smap.pieces) if rows is None else rows):
        a, b = smap.row(p)
        src = g[p * smap.per:(p + 1) * smap.per]
        if smap.world == 1:
            out[a:b].copy_(src)
        else:
            w = dist.all_gather_into_tensor(out[a:b], src, group=group, async_op=async_op)
            if async_op:
                works.append(w)
    return works


def persistent_beside_gather(opt: str) -> bool:
    """Whether the kernel of block row p+1 may be one of the one-block-per-CU persistent kernels
    while the all-gather of row p runs on the same GPU.  Measured on one MI355X with a stand-in
    for RCCL's gather on a second stream (tools/coresidency_probe.py: 8-64 persistent blocks
    holding 12 / 104 / 264 registers per lane that write one peer's block of the row, at full
    speed or throttled to a link-bound 5-9 ms).  Beside a register-heavy, long-running stand-in a
    persistent round runs 1.4-1.6 times as long as alone (its blocks for the CUs the stand-in
    holds start late), the tile-per-block kernel 1.0-1.1 times.  What counts is when BOTH are
    done: plain FedAvg rows of 81 M elements x 32 clients end 2-6 % earlier with the
    tile-per-block kernel (pipeline of four rows 11.4 vs 11.6 ms, 23.5 vs 24.9 ms), but under the
    two-moment optimizers the tile-per-block kernel is 11 % slower to begin with and the persistent
    one still ends first or level (81 M x 32: 12.0 vs 13.7 ms and 25.7 vs 26.3 ms; 25.6 M x 64:
    6.1 vs 6.8 ms and 9.2 vs 10.6 ms).  Hence: keep FedAvg / FedAvgM rounds off the persistent
    kernels beside a gather, leave FedAdam / FedYogi / FedAdagrad on the default routing."""
    return opt in ("adam", "yogi", "adagrad")


def round_gather_pipelined(ops, smap: ShardMap, x: Sequence, w: Sequence[float], g, m, v, opt: str,
                           hp: dict, out: torch.Tensor, group=None, step_ws=None,
                           persistent: Optional[bool] = None) -> torch.Tensor:
    """The sliced round followed by the all-gather of the new model, pipelined over the block
    rows: the fused kernel runs on block p, then its row's all-gather is enqueued (RCCL works
    on its own stream) while the kernel of block p+1 runs.  Every rank enqueues every gather,
    also for rows in which its own block is empty.  On xGMI the gather of a large model takes
    about as long as the local round, so the pipeline hides whichever is shorter.
    `persistent`: whether the kernels may be the one-block-per-CU ones (default: yes when nothing
    runs beside them -- one rank or one block row --, else `persistent_beside_gather(opt)`; a
    per-call routing flag of the C ABI, FA_ROUTE_NO_PERSISTENT, not process state)."""
    if out.numel() < smap.P:
        raise ValueError("gather buffer shorter than the padded vector")
    if persistent is None:
        persistent = smap.world == 1 or smap.pieces == 1 or persistent_beside_gather(opt)
    extra = {} if persistent else {"persistent": False}
    works = []
    for p in range(smap.pieces):
        a, live = p * smap.per, smap.live(p)
        if live > 0:
            ops.reduce_opt([_shift(c, a, g.element_size()) for c in x], list(w), g[a:],
                           None if m is None else m[a:], None if v is None else v[a:], opt,
                           step_ws=None if step_ws is None else step_ws[a:], n=live, **hp, **extra)
        works += gather_rows(smap, g, out, group, rows=[p], async_op=True)
    for wk in works:
        wk.wait()
    return out


def reduce_scatter_row(smap: ShardMap, partial: torch.Tensor, step: torch.Tensor, p: int, group=None):
    """Sum over ranks of block row p of the full-length partial step `partial` (P elements); this
    rank keeps its block in `step` (L elements).  Returns the pending work or None."""
    a, b = smap.row(p)
    dst = step[p * smap.per:(p + 1) * smap.per]
    if smap.world == 1:
        dst.copy_(partial[a:b])
        return None
    if dist.get_backend(group) == "gloo":
        # gloo has no reduce-scatter; CPU rehearsals all-reduce the row and keep their block
        dist.all_reduce(partial[a:b], group=group)
        dst.copy_(partial[a + smap.rank * smap.per:a + (smap.rank + 1) * smap.per])
        return None
    return dist.reduce_scatter_tensor(dst, partial[a:b], op=dist.ReduceOp.SUM, group=group,
                                      async_op=True)


def round_whole_update(ops, smap: ShardMap, my_clients: Sequence, my_weights: Sequence[float],
                       g, m, v, opt: str, hp: dict, g_full: torch.Tensor, partial: torch.Tensor,
                       step: torch.Tensor, group=None, apply: bool = True,
                       step_full: Optional[torch.Tensor] = None) -> None:
    """One round where this rank holds WHOLE updates of some clients (`my_clients[k]`: n elements,
    device tensors or raw device addresses; `my_weights`: their weights in the whole round; a
    rank may hold none).  Per block row: partial weighted step of the own clients against the
    replicated `g_full`, RCCL reduce-scatter (sum over ranks); then the optimizer epilogue and the
    apply on the own block (`g`, `m`, `v`: L elements) and the all-gather of the row into `g_full`
    -- both collectives pipelined behind the kernels of the next row.  `apply=False` stops after
    the epilogue and gathers the STEP into `step_full` instead (compute_steps)."""
    n = smap.n
    scatters = []
    if not len(my_clients):
        partial.zero_()
    for p in range(smap.pieces):
        a, b = smap.row(p)
        b = min(b, n)
        if len(my_clients) and b > a:
            ops.reduce_step([_shift(c, a, g_full.element_size()) for c in my_clients], list(my_weights), g_full[a:],
                            partial[a:], n=b - a)
        scatters.append(reduce_scatter_row(smap, partial, step, p, group))
    gathers = []
    for p in range(smap.pieces):
        if scatters[p] is not None:
            scatters[p].wait()
        a, live = p * smap.per, smap.live(p)
        if live:
            if opt != "none":
                ops.server_opt_step(step[a:], None if m is None else m[a:], v[a:], opt, n=live, **hp)
            if apply:
                ops.apply_step(g[a:], step[a:], n=live)
        gathers += gather_rows(smap, g if apply else step, g_full if apply else step_full, group,
                               rows=[p], async_op=True)
    for wk in gathers:
        wk.wait()


class ShardedFlatAggregator:
    def __init__(self, n: int, opt: str = "none", lr: float = 0.0, beta1: float = 0.0,
                 beta2: float = 0.0, tau: float = 0.0, device=None, group=None, backend=None,
                 pieces: Optional[int] = None, dtype: torch.dtype = torch.float32):
        """`dtype`: torch.float32, or torch.bfloat16 / torch.float16 for a 16-bit model (sliced
        rounds only: g, m, v and the client slices are then 16-bit vectors and every operation
        is rounded to that type, as on one GPU)."""
        if dtype not in (torch.float32, torch.bfloat16, torch.float16):
            raise NotImplementedError(f"{dtype} vectors are not supported (fp32, bf16 and fp16 are)")
        if backend is None:
            from . import ops as backend   # HIP kernels; raises if the library is not built
        if dtype != torch.float32:
            from . import half_ops
            backend = half_ops.for_dtype(dtype, backend)
        self.ops = backend
        self.dtype = dtype
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.n = n
        self.map = ShardMap(n, self.world, self.rank, pieces, torch.empty((), dtype=dtype).element_size())
        self.L = self.map.L
        self.opt = opt
        self.hp = dict(lr=lr, beta1=beta1, beta2=beta2, tau=tau)
        if device is None:
            device = torch.device("cuda", torch.cuda.current_device())
        self.device = torch.device(device)
        z = dict(dtype=dtype, device=self.device)
        self.g = torch.zeros(self.L, **z)
        self.m = torch.zeros(self.L, **z) if opt in ("adam", "yogi", "adagrad") else None
        self.v = None
        if opt != "none":
            self.v = torch.zeros(self.L, **z)
            if opt != "avgm":
                self.ops.fill(self.v, tau ** 2)
        self.g_full: Optional[torch.Tensor] = None       # whole-update mode keeps g replicated
        self._partial: Optional[torch.Tensor] = None
        self._step: Optional[torch.Tensor] = None

    # -- state ------------------------------------------------------------------------------
    def load_global(self, g_full: torch.Tensor) -> None:
        """Install a full-length global vector (every rank passes the same values)."""
        if g_full.numel() != self.n:
            raise ValueError(f"expected {self.n} elements, got {g_full.numel()}")
        self.g.copy_(self.map.take(g_full.to(self.device)))
        if self.g_full is not None:
            self.g_full[: self.n].copy_(g_full)

    def new_gather_buffer(self) -> torch.Tensor:
        return torch.zeros(self.map.P, dtype=self.dtype, device=self.device)

    def all_gather_global(self, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """RCCL all-gather of the shards: the padded full vector (first n elements are g)."""
        if out is None:
            out = self.new_gather_buffer()
        gather_rows(self.map, self.g, out, self.group)
        return out

    # -- sliced mode ------------------------------------------------------------------------
    def round_sliced(self, client_slices: Sequence, weights: Sequence[float],
                     step_ws: Optional[torch.Tensor] = None) -> None:
        """One round on this rank's blocks: `client_slices[k]` holds client k's update in this
        rank's local storage order (`ShardMap.take`; at least n_local elements).  No collective."""
        n_local = self.map.n_local
        if n_local == 0:
            return
        self.ops.reduce_opt(list(client_slices), list(weights), self.g, self.m, self.v, self.opt,
                            step_ws=step_ws, n=n_local, **self.hp)

    def round_sliced_gather(self, client_slices: Sequence, weights: Sequence[float],
                            out: Optional[torch.Tensor] = None,
                            step_ws: Optional[torch.Tensor] = None) -> torch.Tensor:
        """`round_sliced` + the all-gather of the new global model, pipelined over the block
        rows (`round_gather_pipelined`).  Returns the padded full vector."""
        if out is None:
            out = self.new_gather_buffer()
        return round_gather_pipelined(self.ops, self.map, client_slices, weights, self.g, self.m,
                                      self.v, self.opt, self.hp, out, self.group, step_ws)

    # -- whole-update mode ------------------------------------------------------------------
    def round_whole(self, my_clients: Sequence[torch.Tensor], my_weights: Sequence[float]) -> None:
        """One round where this rank holds whole updates of SOME clients (`my_clients[k]` has n
        elements; `my_weights` are their global weights; a rank may hold none): see
        `round_whole_update`."""
        if self.dtype != torch.float32:
            raise NotImplementedError(WHOLE_MODE_16BIT)
        smap = self.map
        if self.g_full is None:
            self.g_full = self.all_gather_global()
        if self._partial is None:
            self._partial = torch.zeros(smap.P, dtype=torch.float32, device=self.device)
            self._step = torch.zeros(self.L, dtype=torch.float32, device=self.device)
        round_whole_update(self.ops, smap, my_clients, my_weights, self.g, self.m, self.v, self.opt,
                           self.hp, self.g_full, self._partial, self._step, self.group)
