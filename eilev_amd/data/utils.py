"""Host-side token layout of interleaved video/text samples (integer work; feeds the hot path).

Mirrors the public functions of ref:eilev/data/utils.py that the two drop-in callers import
(ref:samples/eilev_generate_action_narration.py:10, ref:scripts/general/train_v2.py:22-26):
same names, arguments, return keys and token order — re-implemented around one block builder.
"""
from __future__ import annotations

import re
import string

import torch

_RULES = (
    (re.compile(r"^\#C\s+C", re.IGNORECASE), "The camera wearer"),  # "#C C ..." -> subject
    (re.compile(r"\<\|eos\|\>$", re.IGNORECASE), ""),               # trailing <|eos|>
    (re.compile(r"#unsure\.?$", re.IGNORECASE), ""),                # trailing #unsure
)
_UNSURE_INSIDE = re.compile(r"#unsure", re.IGNORECASE)


def clean_narration_text(narration_text: str) -> str:
    """Ego4D narration clean-up (behaviour pinned by ref:tests/data/test_utils.py:19-54)."""
    out = narration_text.strip()
    for pattern, repl in _RULES:
        out = pattern.sub(repl, out).strip()
    out = _UNSURE_INSIDE.sub("something", out)
    if out and out[-1] not in string.punctuation:
        out += "."
    return out


def generate_chunks(list_to_chunk, chunk_size: int):
    """Consecutive slices of at most ``chunk_size`` items (ref:eilev/data/utils.py:229-231; used by icl_eval's class batching)."""
    for start in range(0, len(list_to_chunk), chunk_size):
        yield list_to_chunk[start:start + chunk_size]


def parse_timestamp(timestamp: str) -> float:
    """``hh:mm:ss.cc`` -> seconds (ref:eilev/data/utils.py:234-241; vectors ref:tests/data/test_utils.py:865-874)."""
    hours, minutes, seconds = timestamp.split(":")
    return float(hours) * 60 * 60 + float(minutes) * 60 + float(seconds)


def _ids(tokenizer, text, **kw):
    return list(tokenizer(text, **kw).input_ids)


def generate_input_ids_and_labels(tokenizer, prompt: str, text: str, decoder_only_lm: bool):
    """Single-clip prompt/label tokenisation (ref:eilev/data/utils.py:95-140)."""
    if not decoder_only_lm:
        enc = tokenizer(prompt, return_attention_mask=False)
        enc["input_ids"] = torch.tensor(enc["input_ids"])
        enc["labels"] = torch.tensor(_ids(tokenizer, text, return_attention_mask=False))
        return enc
    head = _ids(tokenizer, prompt, return_attention_mask=False)
    enc = tokenizer(" " + text, return_attention_mask=False, add_special_tokens=False)
    tail = list(enc["input_ids"]) + [tokenizer.eos_token_id]
    ids = torch.tensor(head + tail)
    labels = ids.clone()
    labels[: len(head)] = -100
    enc["input_ids"], enc["labels"] = ids, labels
    return enc


def generate_input_ids_and_labels_from_interleaved(tokenizer, prompts, text, num_query_tokens: int,
                                                   decoder_only_lm: bool):
    """Token layout of an interleaved sample (ref:eilev/data/utils.py:143-223).

    ``prompts`` is a list of (text, number_of_preceding_videos).  Every video contributes
    ``num_query_tokens`` pad placeholders (video_input_mask = 1) followed by a newline token; text blocks
    other than the last end with a newline.  Decoder-only LMs get a leading BOS and (when ``text`` is given)
    the target `` {text}\\n`` + EOS as labels; encoder-decoder LMs get EOS after the last prompt and the
    tokenised ``text`` as labels.
    """
    newline = _ids(tokenizer, "\n", add_special_tokens=False)[0]  # Flan-T5 maps every whitespace to one id
    video_block = [tokenizer.pad_token_id] * num_query_tokens + [newline]
    video_flags = [1] * num_query_tokens + [0]

    ids, flags = [], []
    last = len(prompts) - 1
    for i, (prompt, num_videos) in enumerate(prompts):
        ids += video_block * num_videos
        flags += video_flags * num_videos
        if i == 0 and decoder_only_lm:
            ids.insert(0, tokenizer.bos_token_id)
            flags.insert(0, 0)
        toks = _ids(tokenizer, prompt if i == last else prompt + "\n", add_special_tokens=False)
        if i == last and not decoder_only_lm:
            toks.append(tokenizer.eos_token_id)
        ids += toks
        flags += [0] * len(toks)

    if decoder_only_lm:
        labels = [-100] * len(ids)
        if text is not None:
            target = _ids(tokenizer, " " + text + "\n", add_special_tokens=False) + [tokenizer.eos_token_id]
            ids += target
            flags += [0] * len(target)
            labels += target
    else:
        labels = _ids(tokenizer, text) if text is not None else []
    return {"input_ids": torch.tensor(ids), "labels": torch.tensor(labels), "video_input_mask": torch.tensor(flags)}


def _seq2seq_base():
    from transformers import DataCollatorForSeq2Seq

    return DataCollatorForSeq2Seq



def _make_collators():
    Base = _seq2seq_base()

    class DataCollatorForVideoSeq2Seq(Base):
        """Stacks ``pixel_values`` when every feature has them (ref:eilev/data/utils.py:19-32)."""

        def __call__(self, features, return_tensors=None):
            frames = None
            if all("pixel_values" in f for f in features):
                frames = torch.stack([f.pop("pixel_values") for f in features])
            batch = super().__call__(features, return_tensors=return_tensors)
            if frames is not None:
                batch["pixel_values"] = frames
            return batch

    class DataCollatorForInterleavedVideoSeq2Seq(Base):
        """Concatenates the clips of all samples along the clip axis and pads ``video_input_mask`` on the
        tokenizer's padding side (ref:eilev/data/utils.py:35-66; pinned by ref:tests/data/test_utils.py:674-862)."""

        def __call__(self, features, return_tensors=None):
            frames = [f.pop("pixel_values") for f in features] if "pixel_values" in features[0] else None
            masks = [f.pop("video_input_mask") for f in features] if "video_input_mask" in features[0] else None
            batch = super().__call__(features, return_tensors=return_tensors)
            if masks is not None:
                width = batch["input_ids"].size(1)
                left = self.tokenizer.padding_side != "right"
                rows = []
                for m in masks:
                    fill = m.new_zeros(width - len(m))
                    rows.append(torch.cat([fill, m] if left else [m, fill]))
                batch["video_input_mask"] = torch.stack(rows)
            if frames is not None:
                batch["pixel_values"] = torch.cat(frames)
            return batch

    return DataCollatorForVideoSeq2Seq, DataCollatorForInterleavedVideoSeq2Seq


def __getattr__(name):  # PEP 562: build the collator classes on first access
    if name in ("DataCollatorForVideoSeq2Seq", "DataCollatorForInterleavedVideoSeq2Seq"):
        a, b = _make_collators()
        globals().update(DataCollatorForVideoSeq2Seq=a, DataCollatorForInterleavedVideoSeq2Seq=b)
        return globals()[name]
    raise AttributeError(name)
