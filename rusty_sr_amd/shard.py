"""Multi-GPU sharding of the upscale path: one process per GPU, torch.distributed
(backend "nccl" = RCCL over xGMI on MI355X; "gloo" in the CPU tests).

The reference processes one image in one process (no
collectives exist there).  What shards is the output: every output pixel depends
on a 15x15 input window (receptive radius SR_HALO = 7 input px through
conv0 5x5 -> conv1 5x5 -> 3x3 -> 3x3 -> 3x3, reference network.rs:33,60-72), so

  * a large image splits into contiguous ROW BANDS (rows are the slow NHWC axis, a
    band is one contiguous byte range); each rank receives the 7 input rows above
    and below its band from its neighbours (one batched send/recv pair per
    neighbour, <= 160 KB each: latency- not bandwidth-bound on xGMI), recomputes
    the overlap (14 / band_rows extra work) and writes its own 3*band rows.  No
    other communication: bit-identical to the un-sharded result.
  * many images are simply dealt round-robin, weights replicated (522 KB): no
    communication at all.
"""
from typing import Callable, List, Optional, Tuple

import torch
import torch.distributed as dist

SR_HALO = 7


def split_rows(h: int, world: int) -> List[Tuple[int, int]]:
    """Contiguous row bands, sizes differing by at most one row."""
    q, r = divmod(h, world)
    out, y = [], 0
    for k in range(world):
        n = q + (1 if k < r else 0)
        out.append((y, y + n))
        y += n
    return out


def round_robin(n_images: int, rank: int, world: int) -> List[int]:
    """Throughput mode (BASELINE configs[4]): image i -> rank i mod world."""
    return list(range(rank, n_images, world))


class BandExchange:
    """Persistent halo-exchange state for one rank's band of an (H, W, C) image."""

    def __init__(self, band_rows: int, width: int, channels: int, dtype, device, rank: int, world: int,
                 halo: int = SR_HALO, group=None):
        if world > 1 and band_rows < halo:
            raise ValueError(f"band of {band_rows} rows is narrower than the {halo}-row halo")
        self.rank, self.world, self.halo, self.group = rank, world, halo, group
        self.top = halo if rank > 0 else 0
        self.bot = halo if rank < world - 1 else 0
        self.band_rows = band_rows
        # band lives in the middle of one buffer so the received halos land in place
        self.ext = torch.zeros((self.top + band_rows + self.bot, width, channels), dtype=dtype, device=device)
        # gloo has no device-memory point-to-point: stage the 7-row halos through host memory then
        # (CPU tests, and the single-GPU rehearsal of bench.py --gpus N); RCCL sends from HBM directly
        self._staged = (world > 1 and self.ext.is_cuda and dist.is_initialized() and dist.get_backend(group) == "gloo")

    @property
    def band(self) -> torch.Tensor:
        return self.ext[self.top:self.top + self.band_rows]

    def exchange(self) -> torch.Tensor:
        """Send my first/last `halo` rows to the neighbours, receive theirs. Returns ext."""
        if self.world == 1:
            return self.ext
        h, ops = self.halo, []
        band = self.band
        via = (lambda t: t.cpu()) if self._staged else (lambda t: t)
        landing = []  # (host tensor, device destination) pairs of the staged path
        def recv_into(dst):
            if not self._staged:
                return dst
            landing.append((torch.empty(dst.shape, dtype=dst.dtype), dst))
            return landing[-1][0]
        if self.rank > 0:
            ops.append(dist.P2POp(dist.isend, via(band[:h]), self._peer(self.rank - 1), self.group))
            ops.append(dist.P2POp(dist.irecv, recv_into(self.ext[:h]), self._peer(self.rank - 1), self.group))
        if self.rank < self.world - 1:
            ops.append(dist.P2POp(dist.isend, via(band[-h:]), self._peer(self.rank + 1), self.group))
            ops.append(dist.P2POp(dist.irecv, recv_into(self.ext[-h:]), self._peer(self.rank + 1), self.group))
        for req in dist.batch_isend_irecv(ops):
            req.wait()
        for host, dst in landing:
            dst.copy_(host)
        return self.ext

    def _peer(self, group_rank: int) -> int:
        return dist.get_global_rank(self.group, group_rank) if self.group is not None else group_rank


def upscale_sharded(band: torch.Tensor, rank: int, world: int,
                    compute: Callable[[torch.Tensor, int, int], torch.Tensor],
                    xchg: Optional[BandExchange] = None, group=None) -> torch.Tensor:
    """One sharded upscale: `band` = this rank's (rows, W, C) slice of the image;
    `compute(ext, halo_top, halo_bot)` maps (top+rows+bot, W, C) -> (3*rows, 3W, C')
    (Engine.upscale_band_*_dev on a GPU).  Returns this rank's output rows."""
    if xchg is None:
        xchg = BandExchange(band.shape[0], band.shape[1], band.shape[2], band.dtype, band.device, rank, world,
                            group=group)
    xchg.band.copy_(band)
    ext = xchg.exchange()
    return compute(ext, xchg.top, xchg.bot)


def exchange_rows(t: torch.Tensor, k: int, rank: int, world: int, group=None) -> torch.Tensor:
    """Per-layer feature halos (libsrhip: sr_set_experiment "halo" = "layers"): `t` is this rank's own rows of one
    map, (rows, W, C).  Sends its first / last k rows to ranks r-1 / r+1, receives theirs, and returns (k + rows + k, W, C) with the
    received rows in place -- ZEROS where there is no neighbour (a true image edge: the reference's per-layer zero padding)."""
    rows = t.shape[0]
    if world > 1 and rows < k:
        raise ValueError(f"band of {rows} rows is narrower than the {k} rows its neighbour needs")
    ext = torch.zeros((rows + 2 * k,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
    ext[k:k + rows] = t
    if world == 1:
        return ext
    peer = (lambda r: dist.get_global_rank(group, r)) if group is not None else (lambda r: r)
    ops = []
    if rank > 0:
        ops.append(dist.P2POp(dist.isend, t[:k].contiguous(), peer(rank - 1), group))
        ops.append(dist.P2POp(dist.irecv, ext[:k], peer(rank - 1), group))
    if rank < world - 1:
        ops.append(dist.P2POp(dist.isend, t[rows - k:].contiguous(), peer(rank + 1), group))
        ops.append(dist.P2POp(dist.irecv, ext[k + rows:], peer(rank + 1), group))
    for req in dist.batch_isend_irecv(ops):
        req.wait()
    return ext


LAYER_ROWS = (2, 1, 1, 1)  # rows of f, l1, l2, l3 the next stages read beyond a band (5x5 convs read f; 3x3 convs the others)


def upscale_sharded_layers(band: torch.Tensor, rank: int, world: int, stages, group=None) -> torch.Tensor:
    """One sharded upscale with per-layer feature halos instead of the recomputed overlap.  `band`: this rank's (rows, W, 3) slice.
    `stages`: five callables working on ROW-EXTENDED maps and returning the band's OWN rows only (no vertical padding of their own: the
    extension rows are the neighbours' rows, or zeros at a true edge):
        f  = stages[0](x2)                  x2 = input + 2 rows either side
        l1 = stages[1](f2)                  f2 = f + 2 rows
        l2 = stages[2](f2, l1_1)            l1_1 = l1 + 1 row
        l3 = stages[3](f2, l1_1, l2_1)
        out = stages[4](l1_1, l2_1, l3_1, x1, top_edge, bottom_edge)    x1 = input + 1 row where a neighbour exists (the bilinear residual
                                                                        clamps at true edges, it does not zero-pad)
    Exactly the rows libsrhip moves in that mode (csrc/sr_comm.cpp layer_rows); the same values computed once each."""
    x2 = exchange_rows(band, 2, rank, world, group)
    f2 = exchange_rows(stages[0](x2), LAYER_ROWS[0], rank, world, group)
    l1 = exchange_rows(stages[1](f2), LAYER_ROWS[1], rank, world, group)
    l2 = exchange_rows(stages[2](f2, l1), LAYER_ROWS[2], rank, world, group)
    l3 = exchange_rows(stages[3](f2, l1, l2), LAYER_ROWS[3], rank, world, group)
    top_edge, bot_edge = rank == 0, rank == world - 1
    x1 = x2[(2 if top_edge else 1):x2.shape[0] - (2 if bot_edge else 1)]
    return stages[4](l1, l2, l3, x1, top_edge, bot_edge)


def init_band_comm(engine, rank: int, world: int, group=None) -> None:
    """Give `engine` (this rank's Engine) its RCCL band communicator INSIDE libsrhip (sr_comm_init_rank): rank 0
    draws the 128-byte id (ncclGetUniqueId), torch.distributed only carries those bytes to the other ranks.  After
    this, Engine.upscale_sharded_dev exchanges halos without touching torch.distributed at all -- the same calls a
    Rust / C++ host makes."""
    uid, err = [b""], None
    if rank == 0 and world > 1:
        try:
            uid = [engine.comm_unique_id()]
        except Exception as ex:  # noqa: BLE001 -- the other ranks are about to wait in the broadcast: tell them instead of leaving them there
            err = ex
    if world > 1:
        dist.broadcast_object_list(uid, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
        if not uid[0]:
            raise err if err is not None else RuntimeError("rank 0 could not draw an RCCL unique id (librccl not loadable there?)")
    engine.comm_init_rank(uid[0], rank, world)


def upscale_batch_round_robin(images, rank: int, world: int, compute: Callable, group=None, gather: bool = False):
    """Throughput mode (BASELINE configs[4], reference main.rs:164-171 once per file): image i is processed by rank
    i mod world, parameters replicated, NO communication on the data path.  `images` is the whole batch (indexable
    by image: numpy (n,H,W,C) or a list -- a rank only touches its own entries); `compute(stack)` maps this rank's
    (m,H,W,C) stack to its (m,3H,3W,C') outputs (Engine.upscale_rgba8 / upscale_f32 on a GPU).
    Returns (indices, outputs); with gather=True every rank's outputs are collected on rank 0 (all_gather_object:
    test / verification helper, not part of the timed path) and rank 0 gets the batch in image order."""
    import numpy as np
    idx = round_robin(len(images), rank, world)
    mine = np.stack([np.asarray(images[i]) for i in idx]) if idx else None
    outs = compute(mine) if idx else None
    if not gather:
        return idx, outs
    if world == 1:
        return idx, outs
    parts = [None] * world
    dist.all_gather_object(parts, (idx, None if outs is None else np.asarray(outs)), group=group)
    if rank != 0:
        return idx, outs
    n = len(images)
    first = next(o for _, o in parts if o is not None)
    full = np.empty((n,) + first.shape[1:], dtype=first.dtype)
    for ids, o in parts:
        for j, i in enumerate(ids):
            full[i] = o[j]
    return list(range(n)), full
