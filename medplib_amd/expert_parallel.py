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


def build_host_group(ep_size: int):
    """gloo twin of this rank's expert-parallel group (host-side scalars: ExpertParallel.exchange_capacity)."""
    world, rank = dist.get_world_size(), dist.get_rank()
    mine = None
    for ranks in group_ranks(world, ep_size)[0]:
        g = dist.new_group(ranks, backend="gloo")
        if rank in ranks:
            mine = g
    return mine


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
    """One exchange per direction and MoE layer.  The dispatch buffer is `[E, capx + 1, d]`: rows 0..capx-1 of slab e are the capacity
    slots of global expert e, row capx is a HEADER whose first 4 bytes hold the number of routed rows (int32), so the row counts ride
    in the same all-to-all as the rows (DeepSpeed sends capacity-padded slabs too — `dispatched_input` is `[E, C, M]` — but needs no
    counts because it multiplies all C rows; here the expert GEMMs skip the unrouted tail).  `capx` is the exchange capacity: the MAX
    over the expert-parallel group of every rank's own capacity `ceil(T/E * cf)` — ranks of one group see different padded sequence
    lengths T, and an equal-split all-to-all with rank-local slab sizes would hang or corrupt (the routing itself still uses the
    rank's own capacity, like DeepSpeed).  The agreement is one scalar MAX all-reduce per forward pass on `host_group` (a gloo twin of
    the group: host data, so it never orders the host behind the GPU queue); without one the device group is used and read back.
    Why not "routed rows only" with variable split sizes: torch.distributed / RCCL take the split sizes from the HOST, i.e. a
    device -> host read of the counts in every one of the 32 layers, which stops the host from running ahead of the GPU;
    the padding is (cf - 1) / cf = 1/3 of the bytes at the stage-IV capacity factor."""

    def __init__(self, group, ep_size: int, num_experts: int, host_group=None, capi_comm=None, variable_split: bool = False):
        assert num_experts % ep_size == 0, "num_experts % ep_size must be 0 (deepspeed MoE asserts the same)"
        self.group, self.ep, self.E = group, ep_size, num_experts
        self.host_group = host_group
        self.capi_comm = capi_comm                             # medplib_amd.comm.RcclComm over the same ranks: mp_alltoall_tokens
        self.E_local = num_experts // ep_size
        self.rank_in_group = dist.get_rank(group) if group is not None else 0
        self._agreed = {}
        # round 4 (review item 4): ROUTED ROWS ONLY behind a flag.  The padded exchange ships capx + 1 rows per slab whatever was routed
        # ((cf - 1) / cf = 1/3 of the bytes at cf 1.5 are padding: +10-25 % time on a byte-bound link, profiles/r03_ep_exchange_gloo.json);
        # the variable form first trades the E row counts (a tiny equal-split exchange), reads them on the HOST (one device -> host sync per
        # MoE layer: the price, it ends the host's run-ahead for that layer) and then moves exactly the routed rows, one message per
        # (peer, local expert).  dispatch / combine only; `exchange` (the training backward) stays padded.
        self.variable_split = bool(variable_split)
        self._var_counts = None                                # (send counts [E], recv counts [ep, E_local]) of the last variable dispatch, host ints
        # measurement (bench.py --ep): bytes this rank SENT per exchange and, when `timing` is a list, HIP-event pairs around every
        # `sample_every`-th exchange on the stream it was issued on
        self.stats = {"exchanges": 0, "bytes_sent": 0}
        self.timing = None
        self.sample_every = 1

    def local_expert_ids(self):
        return list(range(self.rank_in_group * self.E_local, (self.rank_in_group + 1) * self.E_local))

    def exchange_capacity(self, cap: int, key=None) -> int:
        """MAX of `cap` over the expert-parallel group; cached under `key` (the forward-pass counter: one agreement per pass)."""
        if self.ep == 1:
            return cap
        if key is not None and key in self._agreed:
            return self._agreed[key]
        if self.host_group is not None:
            t = torch.tensor([cap], dtype=torch.int64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.host_group)
        else:
            dev = "cuda" if dist.get_backend(self.group) == "nccl" else "cpu"
            t = torch.tensor([cap], dtype=torch.int64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.group)
        capx = int(t.item())
        if key is not None:
            self._agreed = {key: capx}
        return capx

    @staticmethod
    def write_header(buf: torch.Tensor, kept: torch.Tensor):
        """Row counts into the header row (the last row) of every slab of buf [E, capx + 1, d]."""
        E, rows, d = buf.shape
        hdr = buf[:, rows - 1, :].view(torch.int32) if buf.element_size() * d % 4 == 0 else None
        assert hdr is not None, "row bytes must be a multiple of 4"
        hdr[:, 0] = kept.to(torch.int32)

    def dispatch(self, buf: torch.Tensor, kept: torch.Tensor):
        """buf [E, capx + 1, d] (rows written by the dispatch kernel with slab stride capx + 1), kept [E] int32 ->
        (recv [ep, E_local, capx + 1, d], recv_counts [ep, E_local] int32): ONE all-to-all; recv[s, e, :recv_counts[s, e]] are the
        rows source rank s routed to this rank's local expert e."""
        E, rows, d = buf.shape
        assert E == self.E and buf.is_contiguous()
        if self.variable_split and self.ep > 1:
            return self._dispatch_variable(buf, kept)
        self.write_header(buf, kept)
        recv = torch.empty((self.ep, self.E_local, rows, d), dtype=buf.dtype, device=buf.device)
        self._a2a(recv.view(self.ep, -1), buf.view(self.ep, -1))
        counts = recv[:, :, rows - 1, :].view(torch.int32)[:, :, 0].contiguous()
        return recv, counts

    def exchange(self, t: torch.Tensor):
        """Plain slab exchange without headers: t [E, rows, d] (slab e for global expert e) -> [ep, E_local, rows, d] on the owners —
        the backward of `combine` (output gradients travel to the experts' ranks); `combine` is its inverse."""
        E, rows, d = t.shape
        assert E == self.E and t.is_contiguous()
        recv = torch.empty((self.ep, self.E_local, rows, d), dtype=t.dtype, device=t.device)
        self._a2a(recv.view(self.ep, -1), t.view(self.ep, -1))
        return recv

    def combine(self, y: torch.Tensor):
        """y [ep, E_local, capx, d] (expert outputs, slab s = rows that came from source rank s) -> [E, capx, d] on the owner of
        the tokens: slab e holds the outputs of global expert e for THIS rank's tokens."""
        ep, El, cap, d = y.shape
        assert ep == self.ep and El == self.E_local and y.is_contiguous()
        out = torch.empty((self.E, cap, d), dtype=y.dtype, device=y.device)
        if self.variable_split and self.ep > 1 and self._var_counts is not None:
            # (consumed: a combine that does not directly follow a variable dispatch — the training backward's, behind `exchange` — is padded)
            (sc, rc), self._var_counts = self._var_counts, None
            El_ = self.E_local
            # the reverse road: what arrived from source rank p for local expert e goes back to p; what this rank routed to global
            # expert g comes back from its owner
            send = [(p, y[p, e, :rc[p][e]]) for p in range(ep) for e in range(El_) if rc[p][e]]
            recv = [(g // El_, out[g, :sc[g]]) for g in range(self.E) if sc[g]]
            self._a2av(send, recv)
            return out
        self._a2a(out.view(self.ep, -1), y.view(self.ep, -1))
        return out

    def _dispatch_variable(self, buf: torch.Tensor, kept: torch.Tensor):
        E, rows, d = buf.shape
        El = self.E_local
        mine = kept.to(torch.int32).contiguous()
        theirs = torch.empty((self.ep, El), dtype=torch.int32, device=buf.device)
        self._a2a(theirs.view(self.ep, -1), mine.view(self.ep, -1), count=False)
        sc = [int(v) for v in mine.cpu().tolist()]             # the host read the padded form avoids
        rc = [[int(v) for v in row] for row in theirs.cpu().tolist()]
        self._var_counts = (sc, rc)
        recv = torch.empty((self.ep, El, rows, d), dtype=buf.dtype, device=buf.device)
        send = [(g // El, buf[g, :sc[g]]) for g in range(E) if sc[g]]
        rcv = [(p, recv[p, e, :rc[p][e]]) for p in range(self.ep) for e in range(El) if rc[p][e]]
        self._a2av(send, rcv)
        return recv, theirs

    # ---- transports
    def _mark(self, nbytes):
        """Count an exchange; -> a closing function that records the end event when this exchange is sampled."""
        self.stats["exchanges"] += 1
        self.stats["bytes_sent"] += int(nbytes)
        if self.timing is None or not torch.cuda.is_available() or self.stats["exchanges"] % self.sample_every:
            return None
        s = torch.cuda.Event(enable_timing=True)
        s.record(torch.cuda.current_stream())

        def close():
            e = torch.cuda.Event(enable_timing=True)
            e.record(torch.cuda.current_stream())
            self.timing.append((int(nbytes), s, e))
        return close

    def _a2a(self, recv, send, count=True):
        close = self._mark(send.numel() * send.element_size() * (self.ep - 1) // self.ep) if count else None
        if self.capi_comm is not None:
            self.capi_comm.all_to_all(recv, send)
        elif self.ep == 1 and not dist.is_initialized():
            recv.copy_(send)                                   # a one-rank "group" without a process group (bench.py --ep on one GPU): the exchange with oneself
        else:
            dist.all_to_all_single(recv, send, group=self.group)
        if close is not None:
            close()

    def _a2av(self, send, recv):
        """send / recv: lists of (peer rank IN THE GROUP, contiguous tensor view); segments between one pair of ranks are listed in the
        same order on both sides (ascending expert), which is the order NCCL / gloo match them in.  The rank's own segments are copies."""
        me = self.rank_in_group
        own_s = [t for p, t in send if p == me]
        own_r = [t for p, t in recv if p == me]
        for a, b in zip(own_r, own_s):
            a.copy_(b)
        send = [(p, t) for p, t in send if p != me]
        recv = [(p, t) for p, t in recv if p != me]
        close = self._mark(sum(t.numel() * t.element_size() for _, t in send))
        if self.capi_comm is not None:
            self.capi_comm.all_to_all_v(send, recv)
        elif send or recv:
            to_global = (lambda p: dist.get_global_rank(self.group, p)) if self.group is not None else (lambda p: p)
            ops_ = [dist.P2POp(dist.irecv, t, to_global(p), self.group) for p, t in recv]
            ops_ += [dist.P2POp(dist.isend, t, to_global(p), self.group) for p, t in send]
            for w in dist.batch_isend_irecv(ops_):
                w.wait()
        if close is not None:
            close()
