"""`from datasets import ICLLazySupervisedDataset, LazySupervisedDataset, DataCollatorForSupervisedDataset` — the reference
drivers' import line (train_ds_medplib.py:23, model/eval/vqa_infer.py:26), served by this build's sample assembly
(medplib_amd/dataset.py: device-side PIL-exact preprocessing) and collator (medplib_amd/collate.py).  Same constructor
arguments `(data_path, tokenizer, data_args, sam_img_size)` and the same collator call `(samples, inference=False)`.

Like the reference's own `datasets/` directory, this package shadows HuggingFace `datasets` for a process started at the repo root;
nothing on this path imports that library."""
from medplib_amd.collate import collate as DataCollatorForSupervisedDataset      # noqa: F401
from medplib_amd.dataset import ICLLazySupervisedDataset, LazySupervisedDataset  # noqa: F401

__all__ = ["ICLLazySupervisedDataset", "LazySupervisedDataset", "DataCollatorForSupervisedDataset"]
