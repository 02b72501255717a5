"""Default caption vocabulary of the foreground module (58 tokens, index = position; 0 = <pad>, 1 = <unk>).

Used when the working directory has no ``data/vocab.txt`` (the reference CLI always reads that file,
main_procedure.py:503-506).  tests/golden/text_goldens.json pins this list against the reference module's output.
"""

FG_TOKENS = (
    '<pad>', '<unk>', 'bench', 'is', 'light', 'gray', 'orange', 'red', 'purple', 'brown', 'dark', 'green',
    'black', 'cyan', 'pink', 'blue', 'yellow', 'bird', 'has', 'body', 'and', 'wing', 'with', 'white', 'bus',
    'windows', 'butterfly', 'edge', 'car', 'cat', 'chair', 'chicken', 'tail', 'head', 'cloud', 'cow', 'dog',
    'duck', 'horse', 'house', 'roof', 'moon', 'person', 'hair', 'in', 'shirt', 'pants', 'skirt', 'pig',
    'rabbit', 'road', 'sheep', 'star', 'sun', 'tree', 'truck', 'carriage', 'grass',
)


def default_vocab_dict():
    return {w: i for i, w in enumerate(FG_TOKENS)}
