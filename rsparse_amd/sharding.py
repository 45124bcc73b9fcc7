"""Sharding of the solve path across the GPUs of one node (one process per GPU, torch.distributed).

The path shards by INDEPENDENT right-hand-side columns (or independent systems of a batch): every rank holds the whole
factor and owns a contiguous block of columns, so the solve itself needs no collective.  RCCL (backend "nccl") or gloo
is used for exactly two things: broadcasting the factor arrays once from the rank that factored, and sending the
solution blocks to one rank (point-to-point) when the caller wants them there.  No counterpart in the reference (single-threaded CPU).
"""
import numpy as np


def column_block(total, world, rank):
    """Contiguous block [start, start + count) of `total` columns (or systems) owned by `rank`; sizes differ by <= 1."""
    if world <= 0 or not (0 <= rank < world) or total < 0:
        raise ValueError("bad partition request")
    base, extra = divmod(total, world)
    start = rank * base + min(rank, extra)
    return start, base + (1 if rank < extra else 0)


def broadcast_factor(L, src, device, dist):
    """L = (m, n, p, i, x) on rank `src` (ignored elsewhere) -> the same tuple on every rank.

    int64 / int64 / float64 arrays travel as three broadcasts (p, i, x) after a 3-word header; on the nccl backend the
    staging tensors live on `device` so the transfer goes GPU to GPU over xGMI."""
    import torch
    rank = dist.get_rank()
    hdr = torch.zeros(3, dtype=torch.int64, device=device)
    if rank == src:
        m, n, p, i, x = L
        hdr = torch.tensor([m, n, len(x)], dtype=torch.int64, device=device)
    dist.broadcast(hdr, src)
    m, n, nz = (int(v) for v in hdr.cpu())
    if rank == src:
        tp = torch.as_tensor(np.ascontiguousarray(p, np.int64), device=device)
        ti = torch.as_tensor(np.ascontiguousarray(i, np.int64), device=device)
        tx = torch.as_tensor(np.ascontiguousarray(x, np.float64), device=device)
    else:
        tp = torch.empty(n + 1, dtype=torch.int64, device=device)
        ti = torch.empty(nz, dtype=torch.int64, device=device)
        tx = torch.empty(nz, dtype=torch.float64, device=device)
    for t in (tp, ti, tx):
        dist.broadcast(t, src)
    return (m, n, tp.cpu().numpy(), ti.cpu().numpy(), tx.cpu().numpy())


def gather_columns(local, total, dst, device, dist):
    """local: (n, count) array holding this rank's column block -> (n, total) array on rank `dst`, None elsewhere."""
    import torch
    rank, world = dist.get_rank(), dist.get_world_size()
    n = local.shape[0]
    counts = [column_block(total, world, r)[1] for r in range(world)]
    # column blocks travel as contiguous rows, each straight to `dst` (point-to-point: xGMI links are point-to-point, and
    # only one rank wants the result -- an all_gather would land total x n doubles on every rank)
    mine = torch.as_tensor(np.ascontiguousarray(local.T), device=device)
    bufs = None
    if rank == dst:
        bufs = []
        for r in range(world):
            if r == dst:
                bufs.append(mine)
            else:
                t = torch.empty((counts[r], n), dtype=torch.float64, device=device)
                if counts[r]:
                    dist.recv(t, src=r)
                bufs.append(t)
    elif counts[rank]:
        dist.send(mine, dst=dst)
    if rank != dst:
        return None
    out = np.empty((n, total), dtype=np.float64, order="F")
    for r in range(world):
        s, c = column_block(total, world, r)
        out[:, s:s + c] = bufs[r][:c].cpu().numpy().T
    return out


def solve_sharded(solve_block, B, dst, device, dist):
    """Solve all columns of B (n x nrhs, present on every rank) with `solve_block(block) -> block`, each rank taking
    its own column block; the assembled solution is returned on rank `dst`."""
    rank, world = dist.get_rank(), dist.get_world_size()
    s, c = column_block(B.shape[1], world, rank)
    mine = np.asfortranarray(B[:, s:s + c].copy())
    if c:
        mine = solve_block(mine)
    return gather_columns(mine, B.shape[1], dst, device, dist)


def solve_batch_sharded(solve_systems, values, B, dst, device, dist):
    """Batched independent systems (one pattern, `values[s]` / `B[s]` per system, present on every rank): rank r takes
    the contiguous range column_block(nsys, world, r) and runs `solve_systems(values_block, B_block) -> X_block` on
    it (on a GPU box: rsparse_amd.cholsol_batch); no rank needs anything from another one.  The (nsys, n) solution is
    assembled on rank `dst`, None elsewhere."""
    rank, world = dist.get_rank(), dist.get_world_size()
    nsys = B.shape[0]
    s, c = column_block(nsys, world, rank)
    mine = np.ascontiguousarray(B[s:s + c].copy())
    if c:
        mine = solve_systems(np.ascontiguousarray(values[s:s + c]), mine)
    out = gather_columns(np.asfortranarray(mine.T), nsys, dst, device, dist)
    return None if out is None else np.ascontiguousarray(out.T)
