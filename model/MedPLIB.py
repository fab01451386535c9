"""`model.MedPLIB.MedPLIBForCausalLM` — the MoE class the reference's drivers construct (reference model/MedPLIB.py:192-706) with the
whole call surface they use (medplib_amd/surface.py) over the HIP path (medplib_amd/model/medplib.py).

    from model.MedPLIB import MedPLIBForCausalLM
    model = MedPLIBForCausalLM.from_pretrained(args.version, torch_dtype=torch.bfloat16, low_cpu_mem_usage=True,
                                               ignore_mismatched_sizes=True, **vars(args))
"""
from medplib_amd.model import medplib as _core
from medplib_amd.surface import SurfaceMixin


class MedPLIBForCausalLM(SurfaceMixin, _core.MedPLIBForCausalLM):
    def __init__(self, config=None, device="cuda", **kwargs):
        _core.MedPLIBForCausalLM.__init__(self, config, device=device, **kwargs)
        self._surface_init()
        self.vision_pretrained = kwargs.get("vision_pretrained")
