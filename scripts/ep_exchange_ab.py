"""Expert-parallel exchange: capacity-padded equal-split slabs with the counts in the header rows (what ExpertParallel ships) against
"routed rows only" with variable split sizes (a counts exchange + a host read of them first), at BASELINE config 4's sizes
(T = 5112 tokens, E = 2, ep = 2, d = 4096, capacity factor 1.5) on two gloo ranks of this host.  What it can show on a CPU: the bytes
of each form and the extra host-side step of the variable form; the link rate itself is xGMI's on the GPUs, not loopback's.
    python scripts/ep_exchange_ab.py            (spawns its two ranks)"""
import os, sys, time, json, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

if os.environ.get("EP_AB_RANK") is None:
    port = 34000 + os.getpid() % 2000
    ps = [subprocess.Popen([sys.executable, __file__], env=dict(os.environ, EP_AB_RANK=str(r), EP_AB_PORT=str(port), OMP_NUM_THREADS="4"),
                           stdout=subprocess.PIPE, text=True) for r in range(2)]
    outs = [p.communicate(timeout=900)[0] for p in ps]
    print(outs[0].strip())
    sys.exit(max(p.returncode for p in ps))

import torch, torch.distributed as dist
rank = int(os.environ["EP_AB_RANK"])
dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{os.environ['EP_AB_PORT']}", rank=rank, world_size=2)
T, E, ep, d, cf = 5112, 2, 2, 4096, 1.5
cap = -(-int(T / E * cf) // 1)
g = torch.Generator().manual_seed(rank)
counts = torch.tensor([2500 + 56 * rank, T - 2500 - 56 * rank], dtype=torch.int64)        # routed rows per global expert on this rank
buf = torch.zeros(E, cap + 1, d, dtype=torch.bfloat16)
for e in range(E):
    buf[e, :counts[e]] = torch.randn(int(counts[e]), d, generator=g).to(torch.bfloat16)
    buf[e, cap].view(torch.int32)[0] = int(counts[e])

def padded():
    recv = torch.empty_like(buf)
    dist.all_to_all_single(recv.view(ep, -1), buf.view(ep, -1))
    return recv, recv[:, cap].view(torch.int32)[:, 0]

def variable():
    theirs = torch.empty(ep, dtype=torch.int64)
    dist.all_to_all_single(theirs, counts.clone())                  # (1) the counts travel first ...
    send_rows, recv_rows = counts.tolist(), theirs.tolist()         # (2) ... and must be READ ON THE HOST to size the exchange (on a GPU: a sync per layer)
    send = torch.cat([buf[e, :send_rows[e]] for e in range(E)])
    recv = torch.empty(sum(recv_rows), d, dtype=torch.bfloat16)
    dist.all_to_all_single(recv, send, output_split_sizes=recv_rows, input_split_sizes=send_rows)
    return recv, theirs

res = {}
for name, fn in (("padded_equal_split", padded), ("variable_split", variable)):
    for _ in range(2): fn()
    dist.barrier(); t0 = time.perf_counter()
    for _ in range(5): out = fn()
    dist.barrier(); res[name + "_ms"] = round((time.perf_counter() - t0) / 5 * 1e3, 2)
r_p, c_p = padded(); r_v, c_v = variable()
off = 0
for s in range(ep):                                                  # same rows either way
    n = int(c_v[s]); assert int(c_p[s]) == n and torch.equal(r_p[s, :n], r_v[off:off + n]); off += n
res.update(bytes_padded_per_direction=int(buf.numel() * 2), bytes_variable_per_direction=int(counts.sum()) * d * 2,
           padding_fraction=round(1 - int(counts.sum()) / (E * (cap + 1)), 4), capacity=cap,
           note="gloo over loopback on the host CPU: bytes and the extra counts round trip + host read are what carries over to RCCL / xGMI")
if rank == 0:
    print(json.dumps(res))
dist.destroy_process_group()
