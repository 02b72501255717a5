"""Defaults of the reference's process-wide settings bag, read from the REFERENCE module itself
(Foreground_Instance_Colorization/obj_lib/config.py imports without TensorFlow).  Build container only; run from the repo root:
python tests/golden/make_config_goldens.py"""
import importlib.util
import json
import os

spec = importlib.util.spec_from_file_location('ref_config', '/root/reference/Foreground_Instance_Colorization/obj_lib/config.py')
ref = importlib.util.module_from_spec(spec)
spec.loader.exec_module(ref)
vals = {k: v for k, v in vars(ref.Config).items() if not k.startswith('__') and not isinstance(v, staticmethod)}
ref.Config.set_from_dict({'batch_size': 7, 'sn': False})
after = {k: v for k, v in vars(ref.Config).items() if not k.startswith('__') and not isinstance(v, staticmethod)}
try:
    ref.Config.set_from_dict([('a', 1)])
    bad = 'accepted'
except AssertionError:
    bad = 'AssertionError'
json.dump({'defaults': vals, 'after_set_from_dict': after, 'non_dict_argument': bad},
          open(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'config_goldens.json'), 'w'), indent=1, sort_keys=True)
print(vals, after, bad)
