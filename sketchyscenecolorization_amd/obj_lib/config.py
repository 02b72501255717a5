"""Global config (reference: obj_lib/config.py:4-17) -- a mutable bag of class attributes."""


class Config(object):
    data_format = 'NCHW'    # DO NOT CHANGE THIS
    SPECTRAL_NORM_UPDATE_OPS = "spectral_norm_update_ops"
    sn = True               # spectral normalisation on the discriminator's dense head
    proj_d = False          # projection discriminator (reference dead branch: not built)
    wgan = False            # only effective if sn is False (reference dead branch: not built)
    pre_calculated_dist_map = False

    @staticmethod
    def set_from_dict(d):
        assert type(d) is dict
        for k, v in d.items():
            setattr(Config, k, v)
