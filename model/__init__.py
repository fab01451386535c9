"""Import points of the reference's drivers (`from model.MedPLIB import MedPLIBForCausalLM`, `from model.LISA import LISAForCausalLM`,
train_ds_medplib.py:19-20, model/eval/vqa_infer.py:24-25) over the MI355X build in `medplib_amd/`."""
