"""Import point: the reference keeps its inference driver at model/eval/vqa_infer.py; this build's surface walk of it lives at the same path."""
