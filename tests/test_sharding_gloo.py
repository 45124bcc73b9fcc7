"""World-size-2 CPU tests (gloo) of the multi-GPU sharding layer: partition, factor broadcast, gather.
The per-rank solver here is the CPU oracle standing in for a device (test infrastructure); on the GPU box the same
layer is driven by bench.py with the HIP path."""
import os
import socket
import sys

import numpy as np
import pytest

from rsparse_amd import sharding, synth


def test_column_block_covers_everything_once():
    for total in (0, 1, 7, 256, 4096):
        for world in (1, 2, 3, 8):
            seen = []
            for r in range(world):
                s, c = sharding.column_block(total, world, r)
                seen += list(range(s, s + c))
            assert seen == list(range(total))
            sizes = [sharding.column_block(total, world, r)[1] for r in range(world)]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        sharding.column_block(4, 2, 2)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, nrhs, out_dir):
    import torch
    import torch.distributed as dist
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from oracle import pyoracle as orc
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cpu")
    A = synth.laplacian_2d(12, 9)
    n = A[0]
    L = None
    if rank == 0:  # rank 0 "factors"
        st, L = orc.chol(A, orc.schol(A, -1))
        assert st == orc.OK
    L = sharding.broadcast_factor(L, 0, dev, dist)
    B = np.asfortranarray((2.0 * synth.splitmix64(11, n * nrhs) - 1.0).reshape(nrhs, n).T)

    def solve_block(blk):
        for c in range(blk.shape[1]):
            blk[:, c] = orc.trisolve("ltsolve", L, orc.trisolve("lsolve", L, blk[:, c]))
        return blk

    X = sharding.solve_sharded(solve_block, B, 0, dev, dist)
    if rank == 0:
        np.save(os.path.join(out_dir, "X.npy"), X)
        np.save(os.path.join(out_dir, "Lx.npy"), L[4])
    else:
        assert X is None
        np.save(os.path.join(out_dir, "Lx1.npy"), L[4])
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("nrhs", [5, 8])
def test_world_size_2_sharded_solve_matches_single_process(tmp_path, nrhs):
    import torch.multiprocessing as mp
    from oracle import pyoracle as orc
    port = _free_port()
    mp.spawn(_worker, args=(2, port, nrhs, str(tmp_path)), nprocs=2, join=True)
    A = synth.laplacian_2d(12, 9)
    n = A[0]
    st, L = orc.chol(A, orc.schol(A, -1))
    B = (2.0 * synth.splitmix64(11, n * nrhs) - 1.0).reshape(nrhs, n).T
    want = np.stack([orc.trisolve("ltsolve", L, orc.trisolve("lsolve", L, B[:, c])) for c in range(nrhs)], axis=1)
    got = np.load(tmp_path / "X.npy")
    assert np.array_equal(got, want)                       # same arithmetic per column: bit-identical
    assert np.array_equal(np.load(tmp_path / "Lx.npy"), np.load(tmp_path / "Lx1.npy"))  # the broadcast factor


def _batch_worker(rank, world, port, nsys, out_dir):
    import torch
    import torch.distributed as dist
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from oracle import pyoracle as orc
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    mats = [synth.circuit_spd(40, 6, 100 + s) for s in range(nsys)]
    values = np.stack([m[4] for m in mats])
    B = np.stack([synth.splitmix64(1000 + s, 40) for s in range(nsys)])
    pattern = mats[0]

    def solve_systems(vals, rhs):  # the CPU oracle stands in for cholsol_batch on this rank's GPU
        for k in range(rhs.shape[0]):
            A = (pattern[0], pattern[1], pattern[2], pattern[3], vals[k])
            st, rhs[k] = orc.cholsol(A, rhs[k].copy(), 0)
            assert st == orc.OK
        return rhs

    X = sharding.solve_batch_sharded(solve_systems, values, B, 0, torch.device("cpu"), dist)
    if rank == 0:
        np.save(os.path.join(out_dir, "Xb.npy"), X)
    else:
        assert X is None
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("nsys", [1, 5, 8])
def test_world_size_2_sharded_batch_matches_single_process(tmp_path, nsys):
    import torch.multiprocessing as mp
    from oracle import pyoracle as orc
    port = _free_port()
    mp.spawn(_batch_worker, args=(2, port, nsys, str(tmp_path)), nprocs=2, join=True)
    got = np.load(tmp_path / "Xb.npy")
    for s in range(nsys):
        A = synth.circuit_spd(40, 6, 100 + s)
        st, want = orc.cholsol(A, synth.splitmix64(1000 + s, 40), 0)
        assert np.array_equal(got[s], want)
