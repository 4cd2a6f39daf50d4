"""A real (tiny) PreTrainedTokenizerFast built in memory — no files, no network — for the code paths that need `tokenizer.pad`
(the collators; ref:tests/data/test_utils.py uses the Salesforce/blip2-opt-2.7b processor, which cannot be downloaded here).
Special ids follow OPT: pad 1, bos/eos 2."""


def tiny_opt_like_tokenizer(padding_side="right"):
    from tokenizers import Tokenizer
    from tokenizers.models import WordLevel
    from tokenizers.pre_tokenizers import Whitespace
    from transformers import PreTrainedTokenizerFast

    vocab = {"<unk>": 0, "<pad>": 1, "</s>": 2, "a": 3, "b": 4, "c": 5}
    tk = Tokenizer(WordLevel(vocab, unk_token="<unk>"))
    tk.pre_tokenizer = Whitespace()
    return PreTrainedTokenizerFast(tokenizer_object=tk, pad_token="<pad>", bos_token="</s>", eos_token="</s>", unk_token="<unk>",
                                   padding_side=padding_side)
