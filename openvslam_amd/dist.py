"""Multi-GPU layer: one process per GPU (torch.distributed, backend "nccl" = RCCL over xGMI).

Two sharding schemes, both from the structure of the path:
  * offline frame batches: ranks own disjoint contiguous chunks of the sequence and re-extract the
    one halo frame a chunk needs for its first match; no data-path collective at all;
  * keyframe-database brute-force search: the descriptor database is partitioned by rows, every
    rank searches its shard for the (replicated) query block and the per-shard top-2 lists
    (24 bytes per query) are exchanged with ONE all-gather, then merged by the lexicographic
    (distance, global row) order, which reproduces the serial strict-< scan over the whole database.
"""
import numpy as np

from . import _abi as A


def shard_rows(n_rows, world_size, rank):
    """Contiguous row partition: (first_row, n_rows_of_rank)."""
    base, rem = divmod(n_rows, world_size)
    first = rank * base + min(rank, rem)
    return first, base + (1 if rank < rem else 0)


def shard_frames(n_frames, world_size, rank):
    """Contiguous chunk of frames to MATCH on this rank: (first, count).  The rank additionally
    extracts frame first - 1 (the halo) when first > 0."""
    return shard_rows(n_frames, world_size, rank)


def merge_topk_host(lists, n_lists, nq):
    """Host merge of per-shard results (structured array of KNN2_DTYPE, shape [n_lists, nq])."""
    from ._lib import check, lib
    lists = np.ascontiguousarray(lists, dtype=A.KNN2_DTYPE).reshape(n_lists, nq)
    out = np.zeros(nq, dtype=A.KNN2_DTYPE)
    check(lib().ovs_knn2_merge(A.ptr(lists, A.knn2p), n_lists, nq, A.ptr(out, A.knn2p)))
    return out


def sharded_search(local_search, queries, dist=None, device=None):
    """Top-2 over a database sharded across the ranks of `dist` (torch.distributed, already
    initialised).  `local_search(queries) -> KNN2 array` searches this rank's shard and reports
    GLOBAL row numbers.  Returns the merged result, identical on every rank.

    With a CUDA/HIP `device` the exchange is an RCCL all-gather of a uint8 tensor on that device;
    otherwise (CPU tests, gloo) the tensors stay on the host."""
    import torch
    local = np.ascontiguousarray(local_search(queries), dtype=A.KNN2_DTYPE)
    nq = len(local)
    if dist is None or dist.get_world_size() == 1:
        return merge_topk_host(local, 1, nq)
    world = dist.get_world_size()
    send = torch.from_numpy(local.view(np.uint8).reshape(-1).copy())
    if device is not None:
        send = send.to(device)
    recv = torch.empty(world * send.numel(), dtype=torch.uint8, device=send.device)
    dist.all_gather_into_tensor(recv, send)
    gathered = recv.cpu().numpy().reshape(-1).view(A.KNN2_DTYPE).reshape(world, nq)
    return merge_topk_host(gathered, world, nq)
