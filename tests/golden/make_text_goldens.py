"""Generate tests/golden/text_goldens.json by IMPORTING the reference's pure-Python
Foreground_Instance_Colorization/data_processing/text_processing.py (only possible in the build
container, where /root/reference exists).  The JSON (inputs + expected indices) is the committed fixture."""
import importlib.util
import json
import os

REF = '/root/reference/Foreground_Instance_Colorization'
spec = importlib.util.spec_from_file_location('ref_tp', os.path.join(REF, 'data_processing', 'text_processing.py'))
ref = importlib.util.module_from_spec(spec)
spec.loader.exec_module(ref)
vocab = ref.load_vocab_dict_from_file(os.path.join(REF, 'data', 'vocab.txt'))

SENTENCES = [
    'the car is yellow with blue window',
    'the bus is orange with gray windows',
    'A yellow bus, with blue window.',
    'a the the cat has black head , white body and pink tail .',
    'The Dog, is brown and the tail is zzz',
    'the house has red roof and light gray body',
    'a person has black hair , in red shirt and dark green pants',
    'the tree is dark green with brown edge',
    'sun',
    'the moon is light yellow.',
    'the chicken has white body, red head, and orange tail and wing and something more than fifteen tokens long',
    'THE TRUCK IS CYAN WITH PURPLE CARRIAGE',
    'butterfly has pink wing with black edge',
    'the road is   gray',
    'a sheep has white body and black head',
]
out = {'T': 15, 'vocab': vocab, 'cases': [{'sentence': s, 'indices': ref.preprocess_sentence(s, vocab, 15)} for s in SENTENCES]}
out['cases_T8'] = [{'sentence': s, 'indices': ref.preprocess_sentence(s, vocab, 8)} for s in SENTENCES[:6]]
with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'text_goldens.json'), 'w') as f:
    json.dump(out, f, indent=1)
print('wrote', len(out['cases']), 'cases')
