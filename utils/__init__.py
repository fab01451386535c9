"""Import face: the reference's drivers do `from utils.utils import ...` (train_ds_medplib.py:24-26).  Logic: medplib_amd/refutils.py."""
