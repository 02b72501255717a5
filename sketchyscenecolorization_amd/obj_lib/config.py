"""``Config``: the process-wide settings bag every obj_lib module reads (reference obj_lib/config.py:4-17).

The CLI copies its whole parameter dict onto it (``Config.set_from_dict``), so beyond the defaults below it carries
``batch_size``, ``block_type``, ``ckpt_dir``, ``results_dir`` ... at run time.
"""

_DEFAULTS = {
    'data_format': 'NCHW',                                   # the only layout the API accepts
    'SPECTRAL_NORM_UPDATE_OPS': 'spectral_norm_update_ops',  # name of the u-assign collection (sn.py)
    'sn': True,                        # spectral normalisation in the discriminators
    'proj_d': False,                   # projection discriminator: dead branch in the reference, not built
    'wgan': False,                     # only read when sn is False: dead branch, not built
    'pre_calculated_dist_map': False,  # distance maps are never precomputed
}


class Config(object):
    @staticmethod
    def set_from_dict(mapping):
        if not isinstance(mapping, dict):
            raise AssertionError('Config.set_from_dict expects a dict')
        for name in mapping:
            setattr(Config, name, mapping[name])


Config.set_from_dict(dict(_DEFAULTS))
