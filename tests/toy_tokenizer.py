"""A deterministic stand-in with the sentencepiece properties the prompt code depends on (shared by oracle/make_golden.py and the
tests so both sides tokenise identically): BOS first; pieces are space-prefixed words, so a trailing space is a piece of its
own and `ROLE: ` + answer has one piece fewer than the two halves tokenised apart; '</s>', '<SEG>', '<region>', '</region>',
'<im_start>', '<im_end>' are single special tokens; '\\n' is a piece.  Ids are CRC32-derived, stable across runs."""
import re
import zlib
from types import SimpleNamespace

import torch

SPECIALS = ["</s>", "<SEG>", "<region>", "</region>", "<im_start>", "<im_end>"]
_SPLIT = re.compile("(" + "|".join(re.escape(s) for s in SPECIALS) + ")")


class ToyTokenizer:
    bos_token_id = 1
    eos_token_id = 2
    pad_token_id = 0
    unk_token_id = 0

    def __init__(self, model_max_length=2048, vocab_size=None, seg_token_idx=None):
        """vocab_size=None: Llama-like id range (the goldens); an int: every id below it (tiny test models), <SEG> = seg_token_idx."""
        self.model_max_length = model_max_length
        self.special_ids = {"</s>": 2, "<SEG>": 32003, "<region>": 32001, "</region>": 32002, "<im_start>": 32004, "<im_end>": 32005}
        self.base, self.span = 1000, 30000
        if vocab_size is not None:
            top = vocab_size - 1
            self.special_ids = {"</s>": 2, "<SEG>": seg_token_idx, "<region>": top, "</region>": top - 1, "<im_start>": top - 2, "<im_end>": top - 3}
            self.base, self.span = 10, min(seg_token_idx, top - 3) - 10

    def _pieces(self, text):
        ids, first = [], True
        for seg in _SPLIT.split(text):
            if seg == "":
                continue
            if seg in self.special_ids:
                ids.append(self.special_ids[seg]); first = False
                continue
            s = (" " + seg) if first else seg          # the dummy prefix of the first text segment
            first = False
            for piece in re.findall(r" ?[^ \n]+|\n| ", s):
                ids.append(self.base + zlib.crc32(piece.encode()) % self.span)
        return ids

    def encode(self, text, add_special_tokens=True):
        return ([self.bos_token_id] if add_special_tokens else []) + self._pieces(text)

    def __call__(self, text, add_special_tokens=True, return_tensors=None, padding=None, max_length=None, truncation=False):
        if isinstance(text, str):
            return SimpleNamespace(input_ids=self.encode(text, add_special_tokens))
        rows = [self.encode(t, add_special_tokens) for t in text]
        if truncation and max_length:
            rows = [r[:max_length] for r in rows]
        if return_tensors == "pt":
            n = max(len(r) for r in rows)
            out = torch.full((len(rows), n), self.pad_token_id, dtype=torch.long)
            for i, r in enumerate(rows):
                out[i, :len(r)] = torch.tensor(r, dtype=torch.long)
            return SimpleNamespace(input_ids=out)
        return SimpleNamespace(input_ids=rows)


class SentencePieceLlamaLike(ToyTokenizer):
    """The same interface over a REAL sentencepiece BPE model (tests/golden/tiny_llama_like_sp.model: byte fallback, identity
    normalisation, dummy prefix, no whitespace stripping -- the Llama settings), with the legacy slow-tokenizer rule that every
    text segment between special tokens is encoded on its own.  Pins the length bookkeeping of the target masking on genuine
    sentencepiece behaviour (merges across word boundaries, the lone trailing-space piece)."""

    def __init__(self, model_file, model_max_length=2048):
        import sentencepiece as spm
        self.sp = spm.SentencePieceProcessor(model_file=model_file)
        self.model_max_length = model_max_length
        n = self.sp.get_piece_size()
        self.special_ids = {"</s>": 2, "<SEG>": n, "<region>": n + 1, "</region>": n + 2, "<im_start>": n + 3, "<im_end>": n + 4}

    def _pieces(self, text):
        ids = []
        for seg in _SPLIT.split(text):
            if seg == "":
                continue
            ids.extend([self.special_ids[seg]] if seg in self.special_ids else self.sp.encode(seg))
        return ids
