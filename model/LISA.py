"""`model.LISA.LISAForCausalLM` — the dense class of the reference's drivers (reference model/LISA.py:180-600; `train_ds_medplib.py`
takes this branch when `--moe_enable` is off) with the drivers' call surface (medplib_amd/surface.py) over the HIP path.  Accepts the
collator's `attention_mask` as well as `model_forward`'s own `attention_masks` spelling."""
from medplib_amd.model import medplib as _core
from medplib_amd.surface import SurfaceMixin


class LISAForCausalLM(SurfaceMixin, _core.LISAForCausalLM):
    def __init__(self, config=None, device="cuda", **kwargs):
        _core.LISAForCausalLM.__init__(self, config, device=device, **kwargs)
        self._surface_init()
        self.vision_pretrained = kwargs.get("vision_pretrained")
