"""Parameter-axis sharding of the FedAvg-family step over the GPUs of one node.

Every element of the flat parameter vector is independent, so rank r of R (one process per GPU,
`torch.distributed`, backend "nccl" = RCCL over xGMI) owns the contiguous slice
``[r*L, (r+1)*L)`` of the vector padded to ``R*L`` with ``L`` a multiple of 4 (16 bytes).  The
global model g, the moments m / v and FedBuff-style buffers live sharded and stay resident.

Two ways client updates reach the GPUs (no counterpart in the reference, whose server is one CPU
process -- comm/mpi/mpi_server_communicator.py:56-90):

* **sliced** -- slice r of every client update is on GPU r: one local fused kernel, no
  collective; `all_gather_global()` (RCCL all-gather) only when every rank needs the whole model.
  Bit-identical to the single-GPU result.
* **whole**  -- whole client updates sit on different GPUs: each rank folds its clients into a
  full-length partial step against its copy of g, RCCL reduce-scatter (sum) leaves every rank
  the step of its slice, the optimizer epilogue and the apply run on the slice, an all-gather
  refreshes the full g.  The cross-rank sum reorders the fp32 additions, so results agree with
  the single-GPU order to ~1e-7 relative, not bit for bit.

`backend` is the module that provides the flat kernels (default: :mod:`appfl_amd.ops`, the HIP
library).  Tests rehearse the collective logic on CPU with gloo by passing a stand-in backend.
"""

from __future__ import annotations

from typing import List, Optional, Sequence

import torch
import torch.distributed as dist


def shard_length(n: int, world: int) -> int:
    """Elements per rank: ceil(n / world) rounded up to a multiple of 4 (16 bytes)."""
    per = (n + world - 1) // world
    return (per + 3) // 4 * 4


def shard_bounds(n: int, world: int, rank: int):
    """[lo, hi) of rank's slice inside the un-padded vector (hi - lo may be < shard_length)."""
    L = shard_length(n, world)
    lo = min(n, rank * L)
    hi = min(n, lo + L)
    return lo, hi


class ShardedFlatAggregator:
    def __init__(self, n: int, opt: str = "none", lr: float = 0.0, beta1: float = 0.0,
                 beta2: float = 0.0, tau: float = 0.0, device=None, group=None, backend=None):
        if backend is None:
            from . import ops as backend   # HIP kernels; raises if the library is not built
        self.ops = backend
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.n = n
        self.L = shard_length(n, self.world)
        self.lo, self.hi = shard_bounds(n, self.world, self.rank)
        self.opt = opt
        self.hp = dict(lr=lr, beta1=beta1, beta2=beta2, tau=tau)
        if device is None:
            device = torch.device("cuda", torch.cuda.current_device())
        self.device = torch.device(device)
        z = dict(dtype=torch.float32, device=self.device)
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
        self.g.zero_()
        self.g[: self.hi - self.lo].copy_(g_full[self.lo:self.hi])
        if self.g_full is not None:
            self.g_full[: self.n].copy_(g_full)

    def all_gather_global(self, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """RCCL all-gather of the shards: the padded full vector (first n elements are g)."""
        if out is None:
            out = torch.empty(self.world * self.L, dtype=torch.float32, device=self.device)
        if self.world == 1:
            out.copy_(self.g)
        else:
            dist.all_gather_into_tensor(out, self.g, group=self.group)
        return out

    # -- sliced mode ------------------------------------------------------------------------
    def round_sliced(self, client_slices: Sequence[torch.Tensor], weights: Sequence[float],
                     step_ws: Optional[torch.Tensor] = None) -> None:
        """One round on this rank's slice: `client_slices[k]` holds elements [lo, hi) of client
        k's update (at least hi - lo elements).  No collective."""
        n_local = self.hi - self.lo
        if n_local == 0:
            return
        self.ops.reduce_opt(list(client_slices), list(weights), self.g, self.m, self.v, self.opt,
                            step_ws=step_ws, n=n_local, **self.hp)

    def round_sliced_gather(self, client_slices: Sequence[torch.Tensor], weights: Sequence[float],
                            out: Optional[torch.Tensor] = None, pieces: int = 4,
                            step_ws: Optional[torch.Tensor] = None) -> torch.Tensor:
        """`round_sliced` followed by the all-gather of the new global model, pipelined: the
        slice is aggregated in `pieces` pieces and the all-gather of piece p runs (on RCCL's own
        stream) while piece p+1 is being aggregated.  On xGMI the gather of a large model takes
        several times longer than the local round, so this hides the round, not the gather.
        Returns the padded full vector (rank r's slice at [r*L, (r+1)*L))."""
        if out is None:
            out = torch.empty(self.world * self.L, dtype=torch.float32, device=self.device)
        if out.numel() < self.world * self.L:
            raise ValueError("gather buffer shorter than world * shard_length")
        n_local = self.hi - self.lo
        pieces = max(1, min(int(pieces), self.L // 4 or 1))
        per = ((self.L + pieces - 1) // pieces + 3) // 4 * 4
        works = []
        for a in range(0, self.L, per):
            b = min(self.L, a + per)
            live = min(b, n_local) - a          # elements of this piece that hold parameters
            if live > 0:
                self.ops.reduce_opt([c[a:] for c in client_slices], list(weights), self.g[a:],
                                    None if self.m is None else self.m[a:],
                                    None if self.v is None else self.v[a:], self.opt,
                                    step_ws=None if step_ws is None else step_ws[a:], n=live,
                                    **self.hp)
            if self.world == 1:
                out[a:b].copy_(self.g[a:b])
            else:
                dst = [out[r * self.L + a:r * self.L + b] for r in range(self.world)]
                works.append(dist.all_gather(dst, self.g[a:b], group=self.group, async_op=True))
        for w in works:
            w.wait()
        return out

    # -- whole-update mode ------------------------------------------------------------------
    def _reduce_scatter(self, out: torch.Tensor, inp: torch.Tensor) -> None:
        if self.world == 1:
            out.copy_(inp)
        elif dist.get_backend(self.group) == "gloo":
            # gloo has no reduce-scatter; CPU rehearsals all-reduce and keep the local slice
            dist.all_reduce(inp, group=self.group)
            out.copy_(inp[self.rank * self.L:(self.rank + 1) * self.L])
        else:
            dist.reduce_scatter_tensor(out, inp, op=dist.ReduceOp.SUM, group=self.group)

    def round_whole(self, my_clients: Sequence[torch.Tensor], my_weights: Sequence[float]) -> None:
        """One round where this rank holds whole updates of SOME clients (`my_clients[k]` has n
        elements; `my_weights` are their global weights; a rank may hold none).  Partial step
        against the replicated g, reduce-scatter, optimizer + apply on the slice, all-gather."""
        P = self.world * self.L
        if self.g_full is None:
            self.g_full = self.all_gather_global()
        if self._partial is None:
            self._partial = torch.zeros(P, dtype=torch.float32, device=self.device)
            self._step = torch.zeros(self.L, dtype=torch.float32, device=self.device)
        if len(my_clients):
            self.ops.reduce_step(list(my_clients), list(my_weights), self.g_full, self._partial,
                                 n=self.n)
        else:
            self._partial.zero_()
        self._reduce_scatter(self._step, self._partial)
        n_local = self.hi - self.lo
        if n_local:
            if self.opt != "none":
                self.ops.server_opt_step(self._step, self.m, self.v, self.opt, n=n_local, **self.hp)
            self.ops.apply_step(self.g, self._step, n=n_local)
        self.all_gather_global(self.g_full)
