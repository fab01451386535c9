"""Worker of tests/test_gpu_properties.py::test_identical_ranks_are_one_rank — one training step of the tiny model on THIS rank's copy of the same
batch, then rank 0 saves the trainable parameters.  Launched by torch.distributed.run with 1 or 2 ranks sharing cuda:0 over gloo
(RCCL refuses two ranks on one device; the reduce path, grad_scale = 1 / world and the optimizer are the real ones)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


def main():
    out_path, mode = sys.argv[1], sys.argv[2]
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:
        dist.init_process_group(backend="gloo")
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    from medplib_amd import engine
    from medplib_amd.model.config import MedPLIBConfig
    from medplib_amd.model.medplib import LISAForCausalLM, MedPLIBForCausalLM
    from oracle import model as OM
    moe = mode in ("moe", "moe_lora")
    cfg = MedPLIBConfig.tiny(moe_enable=moe, sam_depth=2, num_hidden_layers=2)
    W = OM.init_hf_weights(cfg)
    batch = OM.make_batch(cfg, 3, seed=23)
    torch.manual_seed(1234)
    m = (MedPLIBForCausalLM if moe else LISAForCausalLM)(cfg, device=dev)
    m.load_hf_state_dict(W)
    m.train()
    if mode.endswith("lora"):
        lo = m.enable_lora(lora_r=8, lora_alpha=16, lora_dropout=0.0, lora_target_modules="gate_proj,up_proj,down_proj", sft_modules="mask_decoder,text_hidden_fcs")
        g = torch.Generator().manual_seed(33)
        for n, p_ in zip(lo.names, lo.params):
            if "lora_" in n:
                p_.data.copy_((torch.randn(p_.shape, generator=g) * 0.04).to(torch.bfloat16).float().to(dev))
    eng, _, _, _ = engine.initialize(model=m, model_parameters=m.trainable_parameters(),
                                     config={"optimizer": {"params": {"lr": 1e-3, "betas": (0.9, 0.95)}}, "gradient_clipping": 1.0})
    assert eng.world == world
    gb = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in batch.items()}
    gb["masks_list"] = [x.to(dev) for x in batch["masks_list"]]
    for _ in range(2):
        out = eng(**gb)
        eng.backward(out["loss"])
        eng.step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    if int(os.environ.get("RANK", "0")) == 0:
        torch.save([p_.detach().float().cpu() for p_ in eng.optimizer.params], out_path)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
