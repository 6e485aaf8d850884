"""Sharding of the parameter-set axis over the GPUs of one node.

The reference has no distributed code: its scalable axis, untouched beyond a
Python loop (reference: rrmpg/models/hbvedu.py:199-209), is the axis of
parameter realisations.  The sets are independent, the forcing is read-only
and replicated, so the sweep shards into contiguous blocks of sets, one block
per process/GPU, with NO collective on the data path.  The only exchange is
one all-gather of the per-set scores (8 B per set; 1 MB per rank at 1M sets
over 8 GPUs) at the end -- RCCL over xGMI when the tensors live on the GPUs
(backend "nccl"), gloo on the CPU.  The [timesteps, sets] discharge stays
sharded where it was written: gathering it (87.7 GB at 1M sets) over xGMI
would cost more than computing it.
"""

import torch
import torch.distributed as dist


def shard_bounds(num_sets, world_size, rank):
    """[start, stop) of rank's contiguous block of the parameter-set axis.

    Blocks differ by at most one set; the first num_sets % world_size ranks
    hold the longer ones.
    """
    if world_size < 1 or not (0 <= rank < world_size):
        raise ValueError("bad world_size / rank")
    base, extra = divmod(int(num_sets), int(world_size))
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


def allgather_scores(local_scores, num_sets=None, group=None,
                     always_collective=False):
    """All-gather the per-set scores of every rank's block, in set order.

    local_scores: 1-D tensor (this rank's block; blocks may differ in length
    by one).  Returns the concatenated [num_sets] tensor on every rank.
    Without an initialised process group (single process) it is the identity;
    so it is for a group of one rank unless always_collective is set (tests:
    the RCCL call itself on a single-GPU box).
    """
    if not (dist.is_available() and dist.is_initialized()):
        return local_scores
    world = dist.get_world_size(group)
    if world == 1 and not always_collective:
        return local_scores
    rank = dist.get_rank(group)
    if num_sets is None:
        n = torch.tensor([local_scores.numel()], device=local_scores.device)
        dist.all_reduce(n, group=group)
        num_sets = int(n.item())
    lens = [shard_bounds(num_sets, world, r) for r in range(world)]
    lens = [b - a for a, b in lens]
    if lens[rank] != local_scores.numel():
        raise ValueError("rank %d holds %d scores, expected %d"
                         % (rank, local_scores.numel(), lens[rank]))
    longest = max(lens)
    if min(lens) == longest:
        out = torch.empty(num_sets, dtype=local_scores.dtype,
                          device=local_scores.device)
        dist.all_gather_into_tensor(out, local_scores.contiguous(),
                                    group=group)
        return out
    # ragged: pad every block to the longest, gather, drop the padding
    padded = torch.zeros(longest, dtype=local_scores.dtype,
                         device=local_scores.device)
    padded[:local_scores.numel()] = local_scores
    out = torch.empty(world * longest, dtype=local_scores.dtype,
                      device=local_scores.device)
    dist.all_gather_into_tensor(out, padded, group=group)
    return torch.cat([out[r * longest:r * longest + lens[r]]
                      for r in range(world)])
