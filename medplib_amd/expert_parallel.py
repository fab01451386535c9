"""Expert parallelism: the two all-to-alls of DeepSpeed's MOELayer (sharded_moe.py `_AllToAll`)
over `torch.distributed` (RCCL on the GPUs, gloo in the CPU tests).

Layout contract (same as DeepSpeed's): a rank routes ITS tokens into a dispatch buffer `[E, capacity, d]` (experts global, in
order); experts are sharded `E_local = E / ep_size` per rank, rank r of the expert-parallel group owning experts
`[r*E_local, (r+1)*E_local)`.  `dispatch` sends chunk r of the buffer to rank r and returns `[ep, E_local, capacity, d]`: for every
source rank the rows routed to this rank's experts (+ the per-(source, expert) row counts, so the expert GEMMs skip the unused tail
of each capacity slab).  `combine` is the inverse exchange of the expert outputs.

Rank groups (DeepSpeed `_create_expert_and_data_parallel`): with world = ep_size * replicas, expert-parallel groups are runs of
ep_size consecutive ranks; the expert-data-parallel group of a rank (the ranks holding the SAME expert shard, over which expert
gradients are reduced) takes every ep_size-th rank."""
from typing import List, Optional, Tuple

import torch
import torch.distributed as dist


def group_ranks(world: int, ep_size: int) -> Tuple[List[List[int]], List[List[int]]]:
    assert world % ep_size == 0, "world size must be a multiple of ep_size"
    ep_groups = [list(range(i, i + ep_size)) for i in range(0, world, ep_size)]
    edp_groups = [list(range(i, world, ep_size)) for i in range(ep_size)]
    return ep_groups, edp_groups


def build_groups(ep_size: int):
    """Create this rank's expert-parallel and expert-data-parallel process groups (every rank must call this)."""
    world, rank = dist.get_world_size(), dist.get_rank()
    ep_groups, edp_groups = group_ranks(world, ep_size)
    ep = edp = None
    for ranks in ep_groups:
        g = dist.new_group(ranks)
        if rank in ranks:
            ep = g
    for ranks in edp_groups:
        g = dist.new_group(ranks)
        if rank in ranks:
            edp = g
    return ep, edp


class ExpertParallel:
    def __init__(self, group, ep_size: int, num_experts: int):
        assert num_experts % ep_size == 0, "num_experts % ep_size must be 0 (deepspeed MoE asserts the same)"
        self.group, self.ep, self.E = group, ep_size, num_experts
        self.E_local = num_experts // ep_size
        self.rank_in_group = dist.get_rank(group) if group is not None else 0

    def local_expert_ids(self):
        return list(range(self.rank_in_group * self.E_local, (self.rank_in_group + 1) * self.E_local))

    def dispatch(self, buf: torch.Tensor, kept: torch.Tensor):
        """buf [E, cap, d], kept [E] int32 -> (recv [ep, E_local, cap, d], recv_counts [ep, E_local] int32)."""
        E, cap, d = buf.shape
        assert E == self.E and buf.is_contiguous()
        recv = torch.empty((self.ep, self.E_local, cap, d), dtype=buf.dtype, device=buf.device)
        dist.all_to_all_single(recv.view(self.ep, -1), buf.view(self.ep, -1), group=self.group)
        counts = torch.empty((self.ep, self.E_local), dtype=kept.dtype, device=kept.device)
        dist.all_to_all_single(counts, kept.view(self.ep, self.E_local).contiguous(), group=self.group)
        return recv, counts

    def combine(self, y: torch.Tensor):
        """y [ep, E_local, cap, d] (expert outputs, slab s = rows that came from source rank s) -> [E, cap, d] on the owner of
        the tokens: slab e holds the outputs of global expert e for THIS rank's tokens."""
        ep, El, cap, d = y.shape
        assert ep == self.ep and El == self.E_local and y.is_contiguous()
        out = torch.empty((self.E, cap, d), dtype=y.dtype, device=y.device)
        dist.all_to_all_single(out.view(self.ep, -1), y.view(self.ep, -1), group=self.group)
        return out
