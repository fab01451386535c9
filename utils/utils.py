"""`from utils.utils import (DEFAULT_IM_END_TOKEN, DEFAULT_IM_START_TOKEN, AverageMeter, ProgressMeter, Summary, dict_to_cuda,
intersectionAndUnionGPU, ADD_OTHERS_TOKENS)` — the reference drivers' import line (train_ds_medplib.py:24-26,
model/eval/vqa_infer.py:27-29), served by this build."""
from medplib_amd.refutils import (ADD_OTHERS_TOKENS, DEFAULT_IM_END_TOKEN, DEFAULT_IM_START_TOKEN, DEFAULT_IMAGE_PATCH_TOKEN,  # noqa: F401
                                  DEFAULT_IMAGE_TOKEN, DEFAULT_REGION_REFER_TOKEN_0, DEFAULT_REGION_REFER_TOKEN_1, IGNORE_INDEX,
                                  IMAGE_TOKEN_INDEX, REGION_TOKEN_INDEX, AverageMeter, ProgressMeter, Summary, dict_to_cuda,
                                  intersectionAndUnionGPU)
