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


def sharded_search(local_search, queries, dist=None, device=None, stream=None):
    """Top-2 over a database sharded across the ranks of `dist` (torch.distributed, already
    initialised).  `local_search(queries)` searches this rank's shard and reports GLOBAL row numbers.
    Returns the merged result (KNN2 array), identical on every rank.

    With a CUDA/HIP `device` the whole exchange stays on that device: `local_search` is called with
    `stream` as torch's current stream and enqueues its search there (ovs_desc_db_search_dev on
    torch.cuda.current_stream(device)); it may return its
    result as a uint8 tensor on the device (the output buffer of ovs_desc_db_search_dev), the
    per-shard lists travel in ONE RCCL all-gather and are merged by ovs_knn2_merge_dev.  Without a
    device (CPU tests, gloo) `local_search` returns a numpy array and the merge runs on the host."""
    import torch
    world = 1 if dist is None else dist.get_world_size()
    rec = np.dtype(A.KNN2_DTYPE).itemsize
    if device is None:
        local = np.ascontiguousarray(local_search(queries), dtype=A.KNN2_DTYPE)
        nq = len(local)
        if world == 1:
            return merge_topk_host(local, 1, nq)
        send = torch.from_numpy(local.view(np.uint8).reshape(-1).copy())
        recv = torch.empty(world * send.numel(), dtype=torch.uint8)
        dist.all_gather_into_tensor(recv, send)
        return merge_topk_host(recv.numpy().reshape(-1).view(A.KNN2_DTYPE).reshape(world, nq), world, nq)
    from ._lib import check, lib
    with torch.cuda.device(device):
        caller = torch.cuda.current_stream(device)
        if stream is None:
            stream = caller
        else:
            # what the caller queued before this call (queries, database rows) is complete first
            stream.wait_stream(caller)
        # The local search runs with `stream` as the current stream, so that the all-gather and the merge,
        # which are enqueued on `stream`, are ordered behind it.  A `local_search` that enqueues on another
        # stream has to synchronise that stream before it returns.
        with torch.cuda.stream(stream):
            local = local_search(queries)
            if isinstance(local, torch.Tensor):
                send = local.to(device).contiguous().view(torch.uint8).reshape(-1)
            else:
                send = torch.from_numpy(np.ascontiguousarray(local, dtype=A.KNN2_DTYPE).view(np.uint8).reshape(-1).copy()).to(device)
            nq = send.numel() // rec
            if world > 1 or dist is not None:
                recv = torch.empty(world * send.numel(), dtype=torch.uint8, device=device)
                dist.all_gather_into_tensor(recv, send)
            else:
                recv = send
            merged = torch.empty(nq * rec, dtype=torch.uint8, device=device)
            check(lib().ovs_knn2_merge_dev(recv.data_ptr(), world, nq, merged.data_ptr(), stream.cuda_stream))
        stream.synchronize()
    return merged.cpu().numpy().view(A.KNN2_DTYPE).reshape(nq)
