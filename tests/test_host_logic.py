"""Host-side logic that needs no GPU: parameter registry, flat-buffer sections, CLI flags, Config,
split_inputs, checkpoint naming, lr schedule, sketch pre-processing."""
import json
import os

import numpy as np
import pytest
import torch


def test_param_registry_matches_reference_counts():
    from sketchyscenecolorization_amd.params import ParamStore
    st = ParamStore('Pix2Pix', 58, 192, device='cpu', seed=0)
    assert st.parameter_count('generator') == 24101760        # 24.1 M
    assert st.parameter_count("discriminator") == 2781465     # row A4: 2.78 M (+25 for the non-trainable SN u)
    assert st['generator/encoder_1/conv/filter'].shape == (4, 4, 3, 64)
    assert st['generator/decoder_1/deconv/filter'].shape == (4, 4, 3, 128)
    assert st['generator/TextLSTM/RNN/ALSTM/multi_rnn_cell/cell_0/basic_lstm_cell/kernel'].shape == (2048, 2048)
    assert st['discriminator/fully_connected/u'].shape == (1, 25)
    # every view is 16-byte aligned inside the flat buffer (float4 loads in the kernels)
    for sc in (st.generator, st.discriminator):
        for n, (off, k, shape) in sc.offsets.items():
            assert off % 4 == 0
    # reference initialisers
    assert abs(float(st['generator/encoder_2/scale'].mean()) - 1.0) < 0.02
    assert float(st['generator/encoder_2/offset'].abs().max()) == 0.0
    assert float(st['generator/TextLSTM/embedding'].abs().max()) <= 0.08
    with pytest.raises(NotImplementedError):
        ParamStore('NoSuchBlock', 58, 192, device='cpu')


def test_generator_registries_match_oracle_shapes_and_reference_counts():
    """Variable names / shapes / order of the MRU, Residual and BG generators."""
    from oracle import mru as OM
    from oracle import residual as OR
    from sketchyscenecolorization_amd.params import ParamStore, mru_generator_specs, residual_generator_specs
    for specs, shapes, count in ((mru_generator_specs(), OM.generator_shapes(), 30306499),
                                 (residual_generator_specs('fg'), OR.generator_shapes('fg'), 43516550),
                                 (residual_generator_specs('bg'), OR.generator_shapes('bg'), 79839866)):
        assert [n for n, _, _ in specs] == list(shapes.keys())
        assert all(tuple(s) == tuple(shapes[n]) for n, s, _ in specs)
        assert sum(int(np.prod(s)) for _, s, _ in specs) == count
    st = ParamStore('MRU', 58, 64, device='cpu', seed=0)
    assert float(st['generator/mru_conv_unit_t_1_layer_0/update_gate/biases'].min()) == 0.5      # mru.py:360
    assert float(st['generator/mru_conv_unit_t_1_layer_0/Conv_1/scale'].min()) == 1.0
    assert st['generator/mru_deconv_unit_t_0_layer_0/Conv_2/weights'].shape == (3, 3, 579, 384)


def test_state_dict_roundtrip_and_tf_names(tmp_path):
    from sketchyscenecolorization_amd.params import ParamStore
    a = ParamStore('Pix2Pix', 58, 64, device='cpu', seed=1)
    a.generator.adam_t = 7
    a.generator.adam_v.fill_(0.5)
    path = os.path.join(tmp_path, 'model_9.ckpt-9')
    torch.save(a.state_dict(), path)
    b = ParamStore('Pix2Pix', 58, 64, device='cpu', seed=2)
    b.load_state_dict(torch.load(path))
    for n in a.names():
        assert torch.equal(a[n], b[n]), n
    assert b.generator.adam_t == 7 and float(b.generator.adam_v[0]) == 0.5
    assert 'generator/decoder_5/deconv/filter' in a.state_dict()


def test_generator_sections_are_contiguous_and_cover_everything():
    from sketchyscenecolorization_amd.params import ParamStore
    from sketchyscenecolorization_amd.trainer import Pix2PixTrainer
    st = ParamStore('Pix2Pix', 58, 192, device='cpu')
    sec = Pix2PixTrainer._sections(st.generator.offsets)
    assert sec['encoders'][0] == 0 and sec['encoders'][1] == sec['text'][0] and sec['text'][1] == sec['decoders'][0]
    assert sec['decoders'][1] == st.generator.numel
    assert st.generator.offsets['generator/TextLSTM/embedding'][0] == sec['text'][0]
    assert st.generator.offsets['generator/fully_connected/weights'][0] == sec['decoders'][0]


def test_trainer_fails_loudly_without_gpu():
    if torch.cuda.is_available():
        pytest.skip('GPU present')
    from sketchyscenecolorization_amd.trainer import Pix2PixTrainer
    with pytest.raises(RuntimeError):
        Pix2PixTrainer(img=64)


def test_split_inputs_and_num_classes():
    from sketchyscenecolorization_amd.obj_lib.input_pipeline import get_num_classes, split_inputs
    x = np.arange(8 * 3).reshape(8, 3)
    parts = split_inputs(x, 2, [1, 1, 1, 1], 4)
    assert [p.shape[0] for p in parts] == [2, 2, 2, 2] and (parts[2] == x[4:6]).all()
    parts = split_inputs(torch.arange(6), 1, [1, 2, 3], 3)
    assert [int(p.numel()) for p in parts] == [1, 2, 3]
    assert get_num_classes() == 25


def test_config_and_cli_flags_verbatim():
    import obj_colorization_main as cli
    from sketchyscenecolorization_amd.obj_lib.config import Config
    a = cli.build_parser().parse_args([])
    d = {key: getattr(a, name) for name, _s, _t, _d, _c, key, _h in cli.FLAGS}
    assert (a.mode, a.batch_size, a.max_iter, a.optimizer, a.lr_G, a.lr_D) == ('train', 2, 100000, 'Adam', 2e-4, 1e-4)
    assert (a.small_img, a.lstm_hybrid, a.distance_map, a.block_type, a.vocab_size) == (0, 1, 0, 'MRU', 58)
    assert (a.disc_iterations, a.ld, a.num_gpu, a.summary_write_freq, a.save_model_freq) == (1, 10, 1, 100, 10000)
    assert (a.count_left_time_freq, a.count_inception_score_freq) == (100, -1)
    b = cli.build_parser().parse_args(['-md', 'inference', '-rf', '2026-01-02-03-04-05', '-bt', 'Pix2Pix', '-in', 'car.png',
                                       '-ins', 'the car is red', '-bs', '4', '-gpu', '2', '-si', '1', '-lh', '0'])
    assert (b.mode, b.resume_from, b.block_type, b.infer_name, b.batch_size, b.num_gpu) == \
        ('inference', '2026-01-02-03-04-05', 'Pix2Pix', 'car.png', 4, 2)
    assert set(d) >= {'dataset_type', 'max_iter_step', 'LSTM_hybrid', 'lr_G', 'lr_D', 'disc_iterations'}
    Config.set_from_dict({'batch_size': 5})
    assert Config.batch_size == 5 and Config.data_format == 'NCHW' and Config.sn is True
    json.dumps(d)


def test_invalid_resume_folder_is_reported(capsys):
    import obj_colorization_main as cli
    cli.main(['--mode', 'inference', '-rf', 'nope', '--infer_name', 'car.png', '--instruction', 'x'])
    assert 'Invalid resume folder' in capsys.readouterr().out


def test_checkpoint_naming_like_tf_saver(tmp_path):
    from sketchyscenecolorization_amd.obj_lib import main_procedure as mp
    from sketchyscenecolorization_amd.params import ParamStore
    st = ParamStore('Pix2Pix', 58, 64, device='cpu')
    d = str(tmp_path)
    assert mp.latest_checkpoint(d) is None
    p = mp.save_checkpoint(st, d, 'model_%d.ckpt' % 9999, global_step=9999)
    assert os.path.basename(p) == 'model_9999.ckpt-9999'
    assert mp.latest_checkpoint(d) == p
    assert int(os.path.split(p)[1].split('-')[1]) + 1 == 10000      # the CLI's iter_from rule


def test_lr_decay_matches_reference_formula():
    from sketchyscenecolorization_amd.trainer import Pix2PixTrainer
    t = Pix2PixTrainer.__new__(Pix2PixTrainer)
    t.max_iter_step = 100000
    assert t.decay(0) == 1.0 and abs(t.decay(50000) - 0.55) < 1e-6 and t.decay(99999) == pytest.approx(0.2)


def test_sketch_preprocessing():
    from PIL import Image
    from sketchyscenecolorization_amd.obj_lib.input_pipeline import resize_and_padding_mask_image, thicken_drawings
    img = Image.fromarray(np.full((100, 60, 3), 255, np.uint8))
    out = resize_and_padding_mask_image(img, 192, margin_size=10)
    assert out.shape == (192, 192, 3) and out.dtype == np.uint8 and (out == 255).all()
    rng = np.random.RandomState(0)
    sk = np.full((32, 32, 3), 255, np.uint8)
    sk[rng.randint(0, 32, 20), rng.randint(0, 32, 20)] = 0
    th = thicken_drawings(sk)
    inv = 255 - sk[:, :, 0].astype(np.int32)
    ref = np.zeros_like(inv)
    for i in range(32):
        for j in range(32):
            ref[i, j] = inv[i:min(i + 2, 32), j:min(j + 2, 32)].max()      # neighbourhood rows {i,i+1} x cols {j,j+1}
    assert (th[:, :, 0] == 255 - ref).all() and (th[:, :, 1] == th[:, :, 0]).all()


def test_synthetic_batch_shapes():
    from sketchyscenecolorization_amd.synthetic import synthetic_batch
    b = synthetic_batch(3, 7, img=64, device='cpu')
    assert b['sketches'].shape == (3, 3, 64, 64) and b['text'].shape == (3, 15) and b['noise_vec'].shape == (3, 256)
    assert set(np.unique(b['sketches'].numpy())) <= {-1.0, 1.0}
    assert (b['text'][:, :5] == 0).all() and (b['text'][:, -4:] >= 2).all()      # left-padded captions
    assert b['class_id'].dtype == torch.int32 and int(b['class_id'].max()) < 25


def test_config_matches_reference_module_goldens():
    """obj_lib.config.Config against values read from the reference module itself (tests/golden/make_config_goldens.py)."""
    import json
    import os
    g = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'config_goldens.json')))
    import sketchyscenecolorization_amd.obj_lib.config as cfg
    C = cfg.Config

    def fields():
        return {k: v for k, v in vars(C).items() if not k.startswith('__') and not isinstance(v, staticmethod)}

    saved = fields()            # other tests copy CLI parameters onto the class: back to a fresh import's state, restored below
    try:
        for k in saved:
            delattr(C, k)
        C.set_from_dict(dict(cfg._DEFAULTS))
        assert fields() == g['defaults']
        C.set_from_dict({'batch_size': 7, 'sn': False})
        assert fields() == g['after_set_from_dict']
        try:
            C.set_from_dict([('a', 1)])
            bad = 'accepted'
        except AssertionError:
            bad = 'AssertionError'
        assert bad == g['non_dict_argument']
    finally:
        for k in fields():
            delattr(C, k)
        C.set_from_dict(saved)
