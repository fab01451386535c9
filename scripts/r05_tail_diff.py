"""Debug aid: run the tail program on the device and through the CPU interpreter on the same inputs and list, in allocation order, the
workspace buffers whose contents differ (first differing buffer = the op to look at).  python scripts/r05_tail_diff.py [n] [Dh]"""
import os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from medplib_amd import tail_program as TP
from oracle import tail_program_emu as EMU
from test_gpu_tail_program import _modules, _offsets

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4
Dh = int(sys.argv[2]) if len(sys.argv) > 2 else 256
dev = torch.device("cuda:0")
dec, pe, fc1, fc2 = _modules(33, Dh, dev)
cdec, cpe, cfc1, cfc2 = _modules(33, Dh, torch.device("cpu"))
gp = [fc1.weight, fc1.bias, fc2.weight, fc2.bias] + list(dec.parameters())
cp = [cfc1.weight, cfc1.bias, cfc2.weight, cfc2.bias] + list(cdec.parameters())
goffs, tot = _offsets(gp); coffs, _ = _offsets(cp)
gprog = TP.TailProgram(dec, n, pe.dense_pe_tokens(), pe.no_mask_embed.weight.detach(), fcs=(fc1, fc2), grad_offsets=goffs)
cprog = TP.TailProgram(cdec, n, cpe.dense_pe_tokens(), cpe.no_mask_embed.weight.detach(), fcs=(cfc1, cfc2), grad_offsets=coffs)
torch.manual_seed(7)
x, img = torch.randn(n, Dh), torch.randn(n, 256, 256) * 0.5
d_src, d_hy, d_iou = torch.randn(2, n, 256, 256) * 0.1, torch.randn(n, 32), torch.randn(n)
d_src[1].zero_()
cws, cg, cwb = torch.zeros(cprog.fwd_bytes // 4), torch.zeros(tot), torch.zeros(max(cprog.bwd_bytes, 256) // 4)
EMU.run(cprog.fwd_packed, [0, cws.data_ptr(), 0, x.data_ptr(), 0, img.data_ptr(), 0, 0])
EMU.run(cprog.bwd_packed, [0, cws.data_ptr(), cwb.data_ptr(), 0, cg.data_ptr(), d_src.data_ptr(), d_hy.data_ptr(), d_iou.data_ptr()])
ws, src, hy, iou4 = gprog.run_forward(x.to(dev), img.to(dev))
gg = torch.zeros(tot, device=dev)
gprog._upload()["wb"].zero_()
gprog.run_backward(ws, d_src.to(dev), d_hy.to(dev), d_iou.to(dev), gg.data_ptr())
torch.cuda.synchronize()
for tag, arena, dbuf, cbuf in (("fwd", gprog.L.wf, ws.cpu(), cws), ("bwd", gprog.L.wb, gprog._upload()["wb"].cpu(), cwb)):
    items = sorted(arena.names.items(), key=lambda kv: kv[1])
    for i, (name, off) in enumerate(items):
        end = items[i + 1][1] if i + 1 < len(items) else arena.size
        a, b = dbuf[off // 4: end // 4], cbuf[off // 4: end // 4]
        err = (a - b).abs().max().item(); ref = b.abs().max().item()
        if err > 1e-5 * max(ref, 1e-3) + 1e-7:
            idx = int((a - b).abs().argmax())
            print(f"{tag} {name:28s} off {off:9d} len {a.numel():8d}  max|diff| {err:.3e} (ref max {ref:.3e}) at element {idx}  differing {(((a - b).abs() > 1e-5 * max(ref, 1e-3) + 1e-7)).sum().item()}")
print("done")
