"""Caption -> vocabulary indices (host side).

Own restatement of Foreground_Instance_Colorization/data_processing/text_processing.py:10-51,
pinned against the reference module by tests/golden/text_goldens.json.
"""
import re

UNK_IDENTIFIER = '<unk>'
PAD_IDENTIFIER = '<pad>'
_SPLIT = re.compile(r'(\W+)')


def sentence2vocab_indices(sentence, vocab_dict):
    """Tokenise on non-word runs (kept as tokens), lower-case, drop a trailing '.', a leading 'a' and every
    'the'; ',' reads as 'and'; unknown words map to <unk>."""
    toks = [t.lower() for t in _SPLIT.split(sentence.strip()) if len(t.strip()) > 0]
    if toks[-1] == '.':
        toks = toks[:-1]
    if toks[0] in ['a']:
        toks = toks[1:]
    toks = [t for t in toks if t != 'the']
    toks = ['and' if t in (',', ', ') else t for t in toks]
    unk = vocab_dict[UNK_IDENTIFIER]
    return [vocab_dict.get(t, unk) for t in toks]


def load_vocab_dict_from_file(dict_file):
    with open(dict_file) as f:
        words = [w.strip() for w in f.readlines()]
    return {w: n for n, w in enumerate(words)}


def preprocess_sentence(sentence, vocab_dict, T):
    """Truncate to T tokens, LEFT-pad with <pad> (index 0) -- text_processing.py:40-51."""
    idx = sentence2vocab_indices(sentence, vocab_dict)[:T]
    return [vocab_dict[PAD_IDENTIFIER]] * (T - len(idx)) + idx
