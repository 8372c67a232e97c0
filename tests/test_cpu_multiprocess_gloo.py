"""CPU, world_size 2, gloo: the N > 1 path -- unit sharding and the all-gather of per-tile instance
tables -- gives every rank the same global table, in (unit id, detector order) order, bit-exact."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _make_unit(unit_id, h, w):
    rng = np.random.default_rng(1000 + unit_id)
    n = int(rng.integers(0, 5))
    masks = np.zeros((n, h, w), dtype=bool)
    for i in range(n):
        if rng.random() < 0.15:
            continue  # an empty mask
        y0, x0 = rng.integers(0, h - 8), rng.integers(0, w - 8)
        y1, x1 = rng.integers(y0 + 1, h + 1), rng.integers(x0 + 1, w + 1)
        masks[i, y0:y1, x0:x1] = rng.random((y1 - y0, x1 - x0)) > 0.4
    scores = rng.uniform(0.3, 1.0, n) * 0.6   # float64 (ensemble scores are score * weight products)
    classes = rng.integers(0, 2, n)
    return masks, scores, classes


def _pack(masks):
    m, h, w = masks.shape
    p = np.packbits(masks.reshape(m, h, w // 32, 32), axis=-1, bitorder="little").view(np.uint32).reshape(m, h, w // 32)
    return torch.from_numpy(p.view(np.int32).copy())


def _bbox_area(masks):
    bb = np.full((masks.shape[0], 4), -1, dtype=np.int64)
    for i, m in enumerate(masks):
        ys, xs = np.nonzero(m)
        if len(ys):
            bb[i] = (ys.min(), xs.min(), ys.max(), xs.max())
    return bb, masks.sum((1, 2))


def _worker(rank, world, port, n_units, h, w, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from deepemia_amd import parallel as PL

    mine = PL.shard_indices(n_units, rank, world)
    masks, scores, classes, units = [], [], [], []
    for u in mine:
        m, s, c = _make_unit(u, h, w)
        masks.append(m)
        scores += list(s)
        classes += list(c)
        units += [u] * len(s)
    masks = np.concatenate(masks) if masks else np.zeros((0, h, w), bool)
    bb, area = _bbox_area(masks)
    hdr, pay = PL.encode_instance_table(_pack(masks) if len(masks) else None, scores, classes, units, bb, area)
    gh, gp = PL.all_gather_instance_tables(hdr, pay)
    packed, gs, gc, gu = PL.decode_instance_table(gh, gp, h, w)
    out[rank] = (packed.numpy(), gs, gc, gu)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,n_units", [(2, 7), (2, 1)])
def test_all_gather_instance_tables_world2(world, n_units):
    h, w = 48, 96
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), n_units, h, w, out), nprocs=world, join=True)
    # expected global table: units in order, detector order inside a unit
    em, es, ec, eu = [], [], [], []
    for u in range(n_units):
        m, s, c = _make_unit(u, h, w)
        em.append(m)
        es += [float(v) for v in s]
        ec += [int(v) for v in c]
        eu += [u] * len(s)
    em = np.concatenate(em)
    exp_packed = _pack(em).numpy() if len(em) else np.zeros((0, h, w // 32), np.int32)
    for r in range(world):
        packed, gs, gc, gu = out[r]
        assert gu == eu and gc == ec and gs == es
        np.testing.assert_array_equal(packed, exp_packed)


def test_single_process_roundtrip_and_sharding():
    from deepemia_amd import parallel as PL

    assert PL.shard_indices(16, 3, 8) == [3, 11] and PL.shard_indices(5, 7, 8) == []
    assert sorted(sum((PL.shard_indices(16, r, 8) for r in range(8)), [])) == list(range(16))
    m, s, c = _make_unit(3, 64, 64)
    bb, area = _bbox_area(m)
    hdr, pay = PL.encode_instance_table(_pack(m) if len(m) else None, s, c, [5] * len(s), bb, area)
    gh, gp = PL.all_gather_instance_tables(hdr, pay)          # not initialised -> local table, sorted
    packed, gs, gc, gu = PL.decode_instance_table(gh, gp, 64, 64)
    np.testing.assert_array_equal(packed.numpy(), _pack(m).numpy() if len(m) else packed.numpy())
    assert gs == [float(v) for v in s] and gu == [5] * len(s)
    assert int(pay.numel()) * 4 < max(1, m.size // 8)           # cropped payload is smaller than the full frames


def _worker_capacity(rank, world, port, h, w, out):
    """Four exchanges in one job with tables of very different sizes: the agreed capacity is used when it fits (one
    collective, one host wait), outgrown once (every rank redoes that exchange exactly), and the result is always exact."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from deepemia_amd import parallel as PL

    PL.reset_capacity()
    tables, log = [], []
    for step, n_units in enumerate((6, 5, 40, 3)):          # 40 units: far more rows than 1.5 x the previous maximum
        mine = PL.shard_indices(n_units, rank, world)
        masks, scores, classes, units = [np.zeros((0, h, w), bool)], [], [], []
        for u in mine:
            m, s, c = _make_unit(100 * step + u, h, w)
            masks.append(m)
            scores += list(s)
            classes += list(c)
            units += [u] * len(s)
        masks = np.concatenate(masks)
        bb, area = _bbox_area(masks)
        hdr, pay = PL.encode_instance_table(_pack(masks) if len(masks) else None, scores, classes, units, bb, area)
        before = dict(PL.stats)
        gh, gp = PL.all_gather_instance_tables(hdr, pay)
        log.append((PL.stats["size_exchanges"] - before["size_exchanges"], PL.stats["host_syncs"] - before["host_syncs"]))
        packed, gs, gc, gu = PL.decode_instance_table(gh, gp, h, w)
        tables.append((packed.numpy(), gs, gc, gu))
    out[rank] = (tables, log)
    dist.barrier()
    dist.destroy_process_group()


def test_agreed_capacity_exchange_is_exact_and_needs_one_host_wait_in_steady_state():
    h, w, world = 48, 96, 2
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker_capacity, args=(world, _free_port(), h, w, out), nprocs=world, join=True)
    for step, n_units in enumerate((6, 5, 40, 3)):
        em, es, ec, eu = [np.zeros((0, h, w), bool)], [], [], []
        for u in range(n_units):
            m, s, c = _make_unit(100 * step + u, h, w)
            em.append(m)
            es += [float(v) for v in s]
            ec += [int(v) for v in c]
            eu += [u] * len(s)
        exp = _pack(np.concatenate(em)).numpy()
        for r in range(world):
            packed, gs, gc, gu = out[r][0][step]
            assert gu == eu and gc == ec and gs == es
            np.testing.assert_array_equal(packed, exp)
    for r in range(world):
        log = out[r][1]
        assert log[0] == (1, 2)            # first exchange of a job: sizes first
        assert log[1] == (0, 1)            # fits the agreed capacity: ONE collective, ONE host wait
        assert log[2] == (1, 3)            # outgrown: detected from the gathered sizes, redone exactly by every rank
        assert log[3] == (0, 1)
    assert out[0][1] == out[1][1]
