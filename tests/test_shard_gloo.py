"""The N>1 path on CPU: two processes, gloo backend, the same BandExchange /
upscale_sharded code bench.py and the multi-GPU driver use -- with the CPU oracle
standing in for Engine.upscale_band_f32_dev as the per-band compute."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import oracle
from conftest import ROOT, synth_u8
from rusty_sr_amd.shard import BandExchange, round_robin, split_rows, upscale_batch_round_robin, upscale_sharded, upscale_sharded_layers


def test_split_rows_and_round_robin():
    assert split_rows(2160, 8) == [(270 * k, 270 * (k + 1)) for k in range(8)]
    b = split_rows(1081, 4)
    assert b[0] == (0, 271) and b[-1][1] == 1081 and all(e - s in (270, 271) for s, e in b)
    assert sum(e - s for s, e in split_rows(17, 5)) == 17
    assert round_robin(64, 3, 8) == list(range(3, 64, 8))
    assert sorted(sum((round_robin(10, r, 4) for r in range(4)), [])) == list(range(10))


def test_single_rank_is_identity():
    x = torch.arange(5 * 4 * 3, dtype=torch.float32).reshape(5, 4, 3)
    got = upscale_sharded(x, 0, 1, lambda ext, t, b: (ext.clone(), t, b))
    assert torch.equal(got[0], x) and got[1:] == (0, 0)


def test_band_narrower_than_halo_is_refused():
    with pytest.raises(ValueError):
        BandExchange(5, 8, 3, torch.float32, "cpu", 0, 2)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _oracle_band(params):
    def compute(ext, top, bot):
        y = oracle.forward(params, ext.numpy())[0]
        h = ext.shape[0] - top - bot
        return torch.from_numpy(np.ascontiguousarray(y[3 * top:3 * (top + h)]))
    return compute


def _worker(rank, world, port, h, w, tmp):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        with open(os.path.join(ROOT, "rusty_sr_amd", "res", "imagenet.rsr"), "rb") as f:
            params = oracle.rsr_decode(f.read())
        x = torch.from_numpy(oracle.img_to_data(synth_u8(21, 1, h, w)[0]))
        a, b = split_rows(h, world)[rank]
        xchg = BandExchange(b - a, w, 3, torch.float32, "cpu", rank, world)
        assert (xchg.top, xchg.bot) == (0 if rank == 0 else 7, 0 if rank == world - 1 else 7)
        for _ in range(2):  # the exchange state is reusable step after step
            out = upscale_sharded(x[a:b], rank, world, _oracle_band(params), xchg)
        # halo rows really are the neighbours' rows
        if rank > 0:
            assert torch.equal(xchg.ext[:7], x[a - 7:a])
        if rank < world - 1:
            assert torch.equal(xchg.ext[-7:], x[b:b + 7])
        torch.save(out, os.path.join(tmp, f"out{rank}.pt"))
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,h", [(2, 40), (3, 31)])
def test_sharded_equals_unsharded_gloo(tmp_path, params, world, h):
    w = 24
    mp.spawn(_worker, args=(world, _free_port(), h, w, str(tmp_path)), nprocs=world, join=True)
    got = torch.cat([torch.load(os.path.join(tmp_path, f"out{r}.pt")) for r in range(world)]).numpy()
    want = oracle.forward(params["imagenet"], oracle.img_to_data(synth_u8(21, 1, h, w)))[0]
    np.testing.assert_array_equal(got, want)  # bit-identical, SURVEY.md 8(e)


# ---- config D (BASELINE configs[4]): a batch dealt one image per rank round-robin, no communication ----------------
def _worker_batch(rank, world, port, n, h, w, tmp):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        with open(os.path.join(ROOT, "rusty_sr_amd", "res", "imagenet.rsr"), "rb") as f:
            params = oracle.rsr_decode(f.read())
        batch = oracle.img_to_data(synth_u8(4, n, h, w))  # seed 4 = SURVEY.md 8(d) config D
        calls = []
        def compute(stack):
            calls.append(stack.shape[0])
            return oracle.forward(params, stack)
        idx, outs = upscale_batch_round_robin(batch, rank, world, compute)
        assert idx == list(range(rank, n, world)) and calls == [len(idx)] and outs.shape[0] == len(idx)
        idx2, full = upscale_batch_round_robin(batch, rank, world, compute, gather=True)
        if rank == 0:
            np.save(os.path.join(tmp, "full.npy"), full)
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,n", [(2, 5), (3, 7)])
def test_batch_round_robin_equals_single_rank_gloo(tmp_path, params, world, n):
    h, w = 12, 20
    mp.spawn(_worker_batch, args=(world, _free_port(), n, h, w, str(tmp_path)), nprocs=world, join=True)
    got = np.load(os.path.join(tmp_path, "full.npy"))
    want = oracle.forward(params["imagenet"], oracle.img_to_data(synth_u8(4, n, h, w)))
    np.testing.assert_array_equal(got, want)  # the dealt batch IS the single-rank batch, image for image


def test_batch_round_robin_more_ranks_than_images():
    idx, outs = upscale_batch_round_robin(np.zeros((2, 4, 4, 3), np.float32), 3, 8, lambda s: s)
    assert idx == [] and outs is None


# ---- config C's actual geometry: 3840x2160 over 8 ranks = 270-row bands with 7-row halos (width cut down) ------------
def _worker_c(rank, world, port, h, w, tmp):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        with open(os.path.join(ROOT, "rusty_sr_amd", "res", "imagenet.rsr"), "rb") as f:
            params = oracle.rsr_decode(f.read())
        x = torch.from_numpy(oracle.img_to_data(synth_u8(3, 1, h, w)[0]))  # seed 3 = config C
        a, b = split_rows(h, world)[rank]
        out = upscale_sharded(x[a:b], rank, world, _oracle_band(params))
        torch.save(out, os.path.join(tmp, f"out{rank}.pt"))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_config_c_band_geometry_gloo(tmp_path, params):
    """Ranks 3 and 4 of the 8-way split of 2160 rows (two interior 270-row bands side by side), as a 2-rank job on
    the 554 rows they can see: each band's output equals the same rows of the undivided image."""
    world, w = 2, 16
    bands = split_rows(2160, 8)
    assert bands[3] == (810, 1080) and bands[4] == (1080, 1350)
    h = 540
    mp.spawn(_worker_c, args=(world, _free_port(), h, w, str(tmp_path)), nprocs=world, join=True)
    got = torch.cat([torch.load(os.path.join(tmp_path, f"out{r}.pt")) for r in range(world)]).numpy()
    want = oracle.forward(params["imagenet"], oracle.img_to_data(synth_u8(3, 1, h, w)))[0]
    np.testing.assert_array_equal(got, want)


class _FakeEngine:
    """Stands in for rusty_sr_amd.Engine in init_band_comm: records the calls, fails where told to."""

    def __init__(self, fail_id):
        self.fail_id, self.joined = fail_id, None

    def comm_unique_id(self):
        if self.fail_id:
            raise OSError("librccl.so.1: cannot open shared object file")
        return bytes(range(128))

    def comm_init_rank(self, uid, rank, world):
        self.joined = (uid, rank, world)


def _id_worker(rank, world, port, fail_id, tmp):
    from rusty_sr_amd.shard import init_band_comm
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        eng = _FakeEngine(fail_id and rank == 0)
        try:
            init_band_comm(eng, rank, world)
            outcome = "joined" if eng.joined == (bytes(range(128)), rank, world) else f"wrong {eng.joined}"
        except Exception as ex:  # noqa: BLE001
            outcome = type(ex).__name__
        with open(os.path.join(tmp, f"id_{rank}.txt"), "w") as f:
            f.write(outcome)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("fail_id", [False, True])
def test_band_comm_id_travels_and_a_failure_on_rank_0_reaches_every_rank(tmp_path, fail_id):
    """init_band_comm: rank 0 draws the RCCL id, torch.distributed carries it; if rank 0 cannot draw one, every rank
    raises instead of waiting in the broadcast for ever (the first multi-GPU run must fail fast, not hang)."""
    world = 3
    mp.spawn(_id_worker, args=(world, _free_port(), fail_id, str(tmp_path)), nprocs=world, join=True)
    got = [open(tmp_path / f"id_{r}.txt").read() for r in range(world)]
    assert got == (["OSError", "RuntimeError", "RuntimeError"] if fail_id else ["joined"] * world), got


def _worker_layers(rank, world, port, h, w, tmp):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle.torch_ref import TorchNet
        with open(os.path.join(ROOT, "rusty_sr_amd", "res", "imagenet.rsr"), "rb") as f:
            params = oracle.rsr_decode(f.read())
        x = torch.from_numpy(oracle.img_to_data(synth_u8(22, 1, h, w)[0]))
        a, b = split_rows(h, world)[rank]
        out = upscale_sharded_layers(x[a:b], rank, world, TorchNet(params).band_stages())
        assert out.shape == (3 * (b - a), 3 * w, 3)
        torch.save(out, os.path.join(tmp, f"lay{rank}.pt"))
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,h", [(2, 23), (3, 16)])
def test_per_layer_feature_halos_equal_unsharded_gloo(tmp_path, params, world, h):
    """SURVEY.md 8(e)(ii), the protocol libsrhip runs with sr_set_experiment("halo", "layers"): every stage computes a band's own rows,
    the neighbours' edge rows of its output (f 2, l1 / l2 / l3 one each) travel before the next stage -- here over gloo with the
    torch-CPU restatement as the per-stage compute.  Nothing is recomputed; the ranks' rows together are the unsharded image."""
    w = 20
    mp.spawn(_worker_layers, args=(world, _free_port(), h, w, str(tmp_path)), nprocs=world, join=True)
    got = torch.cat([torch.load(os.path.join(str(tmp_path), f"lay{r}.pt")) for r in range(world)]).numpy()
    x = oracle.img_to_data(synth_u8(22, 1, h, w)[0])
    want = oracle.forward(params["imagenet"], x[None])[0]
    assert got.shape == want.shape and np.abs(got - want).max() < 2e-5  # (oneDNN sums in another order than the C oracle)
