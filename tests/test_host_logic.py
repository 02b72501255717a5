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


def test_tower_graph_dequeue_order_with_run_ahead_and_real_ahead():
    """TowerGraph.run keeps the reference's queue order (D batch, G batch, next D batch, ...) while handing the trainer the G
    batch one step early (g_follows -> run_ahead) and the next D batch one step early (d_follows -> real_ahead); a D-step only
    claims the run-ahead real pass for the very batch that was announced."""
    from sketchyscenecolorization_amd.obj_lib.graph_single import Counter, Fetch, TowerGraph

    class StubTrainer(object):
        run_ahead = real_ahead = True
        lr_g = 1e-4

        def __init__(self):
            self.calls = []

        def decay(self, c):
            return 1.0

        def d_step(self, batch, c, ahead=None, use_real=False):
            self.calls.append(('d', batch['id'], None if ahead is None else ahead['id'], use_real))
            return torch.tensor(0.5)

        def g_step(self, batch, c, use_ahead=False, next_d=None):
            self.calls.append(('g', batch['id'], use_ahead, None if next_d is None else next_d['id']))
            return torch.tensor(0.25)

    import sketchyscenecolorization_amd.obj_lib.graph_single as gs
    n = {'v': 0}

    def deq():
        n['v'] += 1
        return {'id': n['v']}

    tr = StubTrainer()
    g = TowerGraph(tr, [None] * 6, Counter(0), 0, 1, 2, [1])
    g._dequeue = deq
    od, og, lg, ld = Fetch(g, 'opt_d'), Fetch(g, 'opt_g'), Fetch(g, 'loss_g'), Fetch(g, 'loss_d')
    real_check = gs.__dict__.get('hip')
    import sketchyscenecolorization_amd.hip as hip_mod
    saved = hip_mod.check_sk
    hip_mod.check_sk = lambda where='': None         # no device in this test
    try:
        for it in range(3):
            g.run([od, ld], g_follows=True)
            g.run([og, lg], d_follows=(it < 2))
    finally:
        hip_mod.check_sk = saved
    assert tr.calls == [('d', 1, 2, False), ('g', 2, True, 3),        # D0 dequeues 1 (+ G0 = 2 early); G0 dequeues D1 = 3 early
                        ('d', 3, 4, True), ('g', 4, True, 5),
                        ('d', 5, 6, True), ('g', 6, True, None)], tr.calls
    assert n['v'] == 6                                                  # the same six dequeues, in the reference's order


def test_independent_bundle_writer_against_the_reader(tmp_path):
    """tests/golden/bundle_writer.py (own varint / protobuf / block / CRC code, CRC of large tensors by a gcc-compiled helper
    that is checked against the bitwise form) -> tf_checkpoint.read_checkpoint: several data blocks, scalars, int64, a tensor
    larger than a block, corrupted data rejected."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden'))
    import bundle_writer as W
    from sketchyscenecolorization_amd import tf_checkpoint as C
    rng = np.random.RandomState(0)
    t = {'generator/v%03d/weights' % i: rng.randn(3, 5).astype(np.float32) for i in range(400)}
    t['generator/big/filter'] = rng.randn(300, 1000).astype(np.float32)
    t['global_step'] = np.array(7, np.int64)
    t['beta2_power'] = np.float32(0.9 ** 7)
    pre = os.path.join(str(tmp_path), 'model_7.ckpt-7')
    index_bytes, data_bytes = W.write_bundle(pre, t, block_bytes=1024)
    assert index_bytes > 10000 and data_bytes == sum(np.asarray(v).nbytes for v in t.values())
    r = C.read_checkpoint(pre)
    assert set(r) == set(t)
    for k in t:
        assert r[k].shape == np.asarray(t[k]).shape and np.array_equal(r[k], np.asarray(t[k])), k
    assert W.crc32c(bytes(range(256)) * 40) == W.crc32c_bitwise(bytes(range(256)) * 40)
    raw = bytearray(open(pre + '.data-00000-of-00001', 'rb').read())
    raw[5000] ^= 0x01
    open(pre + '.data-00000-of-00001', 'wb').write(bytes(raw))
    import pytest
    with pytest.raises(IOError):
        C.read_checkpoint(pre)


def test_bf16_form_of_a_launch_descriptor():
    """Which launches get bf16 planes (hip.bf_form; the library's fwd_is_bf decides again on the device side): 32-multiple
    sources, or ONE source with a multiple of 4 channels above 32 whose chunk count matches the filter's (MRU's materialised
    concats, mru.py:400-411); few columns, few rows, odd column offsets and two ragged sources stay on the exact-fp32 kernels."""
    from sketchyscenecolorization_amd import hip

    def desc(c0, c1, k_real, nstore=128, n_off=0, rows=4096, bmode=0):
        d = hip.ConvDesc()
        d.x.C0, d.x.C1 = c0, c1
        d.k_real, d.Nstore, d.n_off, d.bmode = k_real, nstore, n_off, bmode
        d.wC0, d.wC1 = (k_real, nstore) if bmode == 0 else (nstore, k_real)
        d.NB, d.PH, d.PW, d.nphase = 1, rows, 1, 1
        return d
    assert hip.bf_form(desc(128, 0, 128)) == 'uniform'
    assert hip.bf_form(desc(256, 256, 512, bmode=1)) == 'uniform'
    assert hip.bf_form(desc(132, 0, 131)) == 'partial'          # [state 128 | image 3] in rows of 132
    assert hip.bf_form(desc(68, 0, 67)) == 'partial'
    assert hip.bf_form(desc(388, 0, 387)) == 'partial'
    assert hip.bf_form(desc(132, 0, 96)) is None                # the filter's chunk count differs from the rows'
    assert hip.bf_form(desc(36, 0, 35)) == 'partial' and hip.bf_form(desc(32 + 0, 4, 35)) is None      # two ragged sources
    assert hip.bf_form(desc(8, 0, 8)) is None and hip.bf_form(desc(12, 0, 11)) is None
    assert hip.bf_form(desc(128, 0, 128, nstore=32)) is None and hip.bf_form(desc(128, 0, 128, nstore=3)) is None
    assert hip.bf_form(desc(128, 0, 128, n_off=16)) is None
    assert hip.bf_form(desc(128, 0, 128, rows=32)) is None


def test_cli_hands_an_explicit_argv_to_the_self_launched_ranks(monkeypatch):
    """main(argv) with -gpu N: the ranks must be started with THAT argv under the host script, not with the host script's
    own sys.argv (dist_utils.launch_towers re-executes what it is given)."""
    import sys
    import obj_colorization_main as cli
    from sketchyscenecolorization_amd import dist_utils
    seen = {}

    def fake(num_gpu, argv=None):
        seen['num_gpu'], seen['argv'] = num_gpu, argv
        return 0
    monkeypatch.setattr(dist_utils, 'launch_towers', fake)
    monkeypatch.setattr(sys, 'argv', ['host_script.py', '--unrelated'])
    cli.main(['--mode', 'train', '-bt', 'Pix2Pix', '-gpu', '2', '-bs', '4'])
    assert seen['num_gpu'] == 2 and seen['argv'] == ['host_script.py', '--mode', 'train', '-bt', 'Pix2Pix', '-gpu', '2', '-bs', '4']
    seen.clear()
    monkeypatch.setattr(sys, 'argv', ['obj_colorization_main.py', '--mode', 'train', '-gpu', '2'])
    cli.main()
    assert seen['argv'] is None         # the command line itself: launch_towers re-executes sys.argv


def test_pending_losses_are_settled_in_launch_order(capsys):
    """main_procedure._settle_first: the losses of launched steps are looked at oldest first; the first NaN names its step, drops
    what was launched after it and returns -1 (train() then ends in front of any later snapshot); otherwise 0."""
    import numpy as np
    from sketchyscenecolorization_amd.obj_lib.main_procedure import _settle_first

    class Lazy(object):         # what graph_single.LazyLoss looks like to numpy
        def __init__(self, v):
            self.v, self.read = v, False

        def __array__(self, dtype=None, copy=None):
            self.read = True
            return np.asarray(np.float32(self.v), dtype=dtype)

    a, b, c = Lazy(1.0), Lazy(2.0), Lazy(3.0)
    pending = [('D', a), ('G', b), ('D', c)]
    assert _settle_first(pending, 2) == 0
    assert a.read and b.read and not c.read and pending == [('D', c)]
    x, y, z = Lazy(0.5), Lazy(float('nan')), Lazy(1.0)
    pending = [('D', x), ('G', y), ('D', z)]
    assert _settle_first(pending, 3) == -1
    assert 'NaN occurred during training G' in capsys.readouterr().out
    assert pending == [] and not z.read
