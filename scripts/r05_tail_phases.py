"""Per-phase timeline of the mask-tail program at the training step's geometry (n = 8 prompts, hidden 4096): the 100 MHz stamp workgroup 0
writes at every phase end, next to the phase's ops and tile counts; totals by HIP events for several grid sizes, and the op-by-op path's
forward + backward beside them.  Usage (GPU box): python scripts/r05_tail_phases.py [n] [Dh] > gpurun_out/r05_tail_phases.txt"""
import os
import sys
import time

ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402

from medplib_amd import tail_program as TP  # noqa: E402
from test_gpu_tail_program import _modules, _offsets  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    Dh = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
    dev = torch.device("cuda:0")
    dec, pe, fc1, fc2 = _modules(3, Dh, dev)
    gp = [fc1.weight, fc1.bias, fc2.weight, fc2.bias] + list(dec.parameters())
    goffs, tot = _offsets(gp)
    prog = TP.TailProgram(dec, n, pe.dense_pe_tokens(), pe.no_mask_embed.weight.detach(), fcs=(fc1, fc2), grad_offsets=goffs)
    x, img = torch.randn(n, Dh, device=dev), torch.randn(n, 256, 256, device=dev) * 0.5
    d_src, d_hy, d_iou = torch.randn(2, n, 256, 256, device=dev) * 0.1, torch.randn(n, 32, device=dev), torch.randn(n, device=dev)
    g = torch.zeros(tot, device=dev)
    print(f"n={n} hidden={Dh}: forward {len(prog.fwd_packed[0])} ops / {len(prog.fwd_packed[2])} phases, backward {len(prog.bwd_packed[0])} ops / "
          f"{len(prog.bwd_packed[2])} phases; workspaces {prog.fwd_bytes / 1e6:.1f} + {prog.bwd_bytes / 1e6:.1f} MB")
    for grid in (256, 128, 64, 32):
        for _ in range(3):
            ws, *_ = prog.run_forward(x, img, grid=grid)
            prog.run_backward(ws, d_src, d_hy, d_iou, g.data_ptr(), grid=grid)
        torch.cuda.synchronize()
        e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        reps = 20
        tf = tb = 0.0
        for _ in range(reps):
            e[0].record(); ws, *_ = prog.run_forward(x, img, grid=grid); e[1].record()
            prog.run_backward(ws, d_src, d_hy, d_iou, g.data_ptr(), grid=grid); e[2].record()
            torch.cuda.synchronize()
            tf += e[0].elapsed_time(e[1]); tb += e[1].elapsed_time(e[2])
        print(f"grid {grid:4d}: forward {tf / reps * 1e3:8.1f} us, backward {tb / reps * 1e3:8.1f} us   (give-up flag clear: {prog.check_sync()})")
    # per-phase stamps at grid 256
    for tag, packed in (("forward", prog.fwd_packed), ("backward", prog.bwd_packed)):
        nph = len(packed[2])
        st = torch.zeros(nph, dtype=torch.int64, device=dev)
        acc = torch.zeros(nph, dtype=torch.float64)
        reps = 10
        for _ in range(reps):
            if tag == "forward":
                ws, *_ = prog.run_forward(x, img, grid=256, stamps=st)
            else:
                ws, *_ = prog.run_forward(x, img, grid=256)
                prog.run_backward(ws, d_src, d_hy, d_iou, g.data_ptr(), grid=256, stamps=st)
            torch.cuda.synchronize()
            s = st.cpu().double()
            d = torch.cat([torch.zeros(1, dtype=torch.float64), (s[1:] - s[:-1]) / 100.0])       # us per phase (the first has no start stamp)
            acc += d
        acc /= reps
        print(f"\n{tag}: per-phase us (phases 1..) sum {acc.sum():.1f}")
        ops, po, pt, notes = packed
        for ph in range(nph):
            names = ", ".join(f"{notes[i][1]}:{notes[i][2]}[{int(ops[i]['ntiles'])}]" for i in range(int(po[ph]), int(po[ph + 1])))
            print(f"  {ph:3d} {acc[ph]:7.1f} us  tiles {int(pt[ph]):5d}  {names[:230]}")
    # the op-by-op path at the same geometry
    for use in (True, False):
        dec.use_program = use
        xi = x.clone()
        gl, gi = torch.randn(n, 64, 64, device=dev), torch.randn(n, device=dev)
        for p in gp:
            p.grad = None

        def step():
            low, iou = dec(img, pe.dense_pe_tokens(), pe.no_mask_embed.weight, None, fcs=(fc1, fc2), hidden_rows=xi)
            ((low * gl).sum() + (iou * gi).sum()).backward()
        for _ in range(3):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            step()
        torch.cuda.synchronize()
        print(f"\nMaskDecoder forward + backward, {'program' if use else 'op-by-op'} path (fp32 upscaling chain on both): {(time.perf_counter() - t0) * 100:.2f} ms wall per call")


if __name__ == "__main__":
    main()
