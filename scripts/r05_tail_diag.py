import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo")); sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests"))
import torch
from test_gpu_tail_program import _modules
from test_tail_program import _reference
from oracle import sam as OS
dev = torch.device("cuda:0")
n, Dh = 4, 256
torch.manual_seed(int(sys.argv[1]) if len(sys.argv) > 1 else 7)
x, img = torch.randn(n, Dh, device=dev), torch.randn(n, 256, 256, device=dev) * 0.5
gl, gi = torch.randn(n, 64, 64, device=dev), torch.randn(n, device=dev)
res = {}
for use in (True, False):
    dec, pe, fc1, fc2 = _modules(33, Dh, dev)
    dec.use_program = use
    xi = x.clone().requires_grad_()
    low, iou = dec(img, pe.dense_pe_tokens(), pe.no_mask_embed.weight, None, fcs=(fc1, fc2), hidden_rows=xi)
    ((low * gl).sum() + (iou * gi).sum()).backward()
    res[use] = (low.detach().cpu(), iou.detach().cpu(), xi.grad.cpu(), {k: p.grad.cpu() for k, p in dec.named_parameters() if p.grad is not None})
# CPU truth (fp64) through the oracle
cdec, cpe, cfc1, cfc2 = _modules(33, Dh, torch.device("cpu"))
W = OS.init_weights(seed=33)
Wr = {k: v.double() for k, v in W.items()}
for k, v in cdec.named_parameters(): Wr["mask_decoder." + k] = v.detach().double().requires_grad_()
import torch.nn.functional as F
xr = x.cpu().double().requires_grad_()
text = F.linear(F.relu(F.linear(xr, cfc1.weight.double(), cfc1.bias.double())), cfc2.weight.double(), cfc2.bias.double()).view(n, 1, 256)
sp, de = OS.prompt_encoder_text(text, Wr)
emb = img.cpu().double().view(n, 16, 16, 256).permute(0, 3, 1, 2)
m, io = OS.mask_decoder(emb, OS.dense_pe(Wr).double(), sp, de, Wr)
((m[:, 0] * gl.cpu().double()).sum() + (io[:, 0] * gi.cpu().double()).sum()).backward()
for use in (True, False):
    low, iou, dx, g = res[use]
    print("program" if use else "op-by-op", "low err", (low - m[:, 0].float()).abs().max().item(), "dx per-row err", (dx - xr.grad.float()).abs().max(1).values.tolist(), "dx row max", xr.grad.abs().max(1).values.tolist())
    worst = max(((g[k] - Wr["mask_decoder." + k].grad.float()).abs().max() / (Wr["mask_decoder." + k].grad.abs().max() + 1e-6)).item() for k in g)
    print("   worst rel param grad err", worst)
