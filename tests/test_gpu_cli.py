"""The module CLI end to end on the GPU: train a few iterations (64x64), checkpoint, resume, inference."""
import glob
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_cli_train_resume_inference(tmp_path, monkeypatch):
    from PIL import Image, ImageDraw
    import obj_colorization_main as cli
    monkeypatch.chdir(tmp_path)
    cli.main(['--mode', 'train', '-bt', 'Pix2Pix', '-si', '1', '-bs', '2', '-mi', '4', '-smf', '2', '-swf', '1', '-clt', '2'])
    runs = sorted(os.listdir('outputs'))
    assert len(runs) == 1 and len(runs[0].split('-')) == 6
    run = os.path.join('outputs', runs[0])
    p0 = json.load(open(os.path.join(run, 'log', 'param_0.json')))
    assert p0['block_type'] == 'Pix2Pix' and p0['batch_size'] == 2 and p0['iter_from'] == 0
    assert os.path.exists(os.path.join(run, 'snapshot', 'model_1.ckpt-1'))
    assert os.path.exists(os.path.join(run, 'snapshot', 'model_3.ckpt-3'))
    scal = [json.loads(l) for l in open(os.path.join(run, 'log', 'scalars.jsonl'))]
    assert len(scal) == 4 and all(np.isfinite(s['total_loss/g']) for s in scal)
    # resume: iter_from = latest step + 1
    cli.main(['--mode', 'train', '-rf', runs[0], '-bt', 'Pix2Pix', '-si', '1', '-bs', '2', '-mi', '6', '-smf', '2'])
    assert os.path.exists(os.path.join(run, 'log', 'param_4.json'))
    assert os.path.exists(os.path.join(run, 'snapshot', 'model_5.ckpt-5'))
    # inference on a wild sketch (256x256 grey strokes on white, like the reference's examples/*.png)
    os.makedirs('examples')
    im = Image.new('L', (256, 256), 255)
    d = ImageDraw.Draw(im)
    d.rectangle([40, 120, 220, 190], outline=0, width=3)
    d.ellipse([60, 180, 100, 220], outline=0, width=3)
    d.ellipse([160, 180, 200, 220], outline=0, width=3)
    im.save('examples/car.png')
    cli.main(['--mode', 'inference', '-rf', runs[0], '-bt', 'Pix2Pix', '-si', '1', '--infer_name', 'car.png',
              '--instruction', 'the car is yellow with blue window'])
    out = os.path.join(run, 'inference_results', 'car_output.png')
    inp = os.path.join(run, 'inference_results', 'car_input.png')
    assert os.path.exists(out) and os.path.exists(inp)
    o = np.array(Image.open(out))
    assert o.shape == (64, 64, 3) and o.dtype == np.uint8
    i = np.array(Image.open(inp))
    assert i.shape == (64, 64, 3) and i.min() < 128 and i.max() == 255
    cli.main(['--mode', 'val', '-rf', runs[0], '-bt', 'Pix2Pix', '-si', '1', '-bs', '2'])
    assert len(glob.glob(os.path.join(run, 'validation_results', 'with_text', '*_output.png'))) == 2


def test_cli_train_losses_read_one_launch_late_give_the_same_run(tmp_path, monkeypatch):
    """main_procedure.train reads an iteration's losses behind the NEXT launch (graph_single.LazyLoss; the device does not wait
    for the host between steps) except where it writes a scalar line or a snapshot: the same run, number for number, as with
    every loss read where the reference reads it (SSC_CLI_LAZY_LOSS=0)."""
    import obj_colorization_main as cli
    from sketchyscenecolorization_amd.obj_lib import graph_single
    monkeypatch.chdir(tmp_path)
    made = []
    real = graph_single.LazyLoss.__init__

    def counting(self, *a, **kw):
        made.append(1)
        real(self, *a, **kw)

    monkeypatch.setattr(graph_single.LazyLoss, '__init__', counting)
    lines, runs = {}, []
    for lazy in ('1', '0'):
        monkeypatch.setenv('SSC_CLI_LAZY_LOSS', lazy)
        torch.manual_seed(11)
        before = set(os.listdir('outputs')) if os.path.isdir('outputs') else set()
        n0 = len(made)
        cli.main(['--mode', 'train', '-bt', 'Pix2Pix', '-si', '1', '-bs', '2', '-mi', '7', '-smf', '4', '-swf', '3', '-clt', '2'])
        run = os.path.join('outputs', (set(os.listdir('outputs')) - before).pop())
        runs.append(run)
        lines[lazy] = [json.loads(l) for l in open(os.path.join(run, 'log', 'scalars.jsonl'))]
        assert os.path.exists(os.path.join(run, 'snapshot', 'model_3.ckpt-3'))
        assert (len(made) - n0 == 14) if lazy == '1' else (len(made) == n0)      # two fetched losses per iteration, all lazy
        import time
        time.sleep(1.1)         # run directories are named by the second
    assert [s['step'] for s in lines['1']] == [0, 3, 6]
    # (a loss is a sum of double atomics over many blocks: its last digits depend on their order from run to run; the weights
    # do not -- the snapshots agree bit for bit)
    for a, b in zip(lines['1'], lines['0']):
        assert a.keys() == b.keys()
        for k in a:
            assert abs(a[k] - b[k]) <= 1e-12 * max(1.0, abs(b[k])), (k, a[k], b[k])
    w = [torch.load(os.path.join(r, 'snapshot', 'model_3.ckpt-3'), map_location='cpu') for r in runs]
    assert w[0].keys() == w[1].keys() and len(w[0]) > 50
    for k in w[0]:
        if torch.is_tensor(w[0][k]):
            assert torch.equal(w[0][k], w[1][k]), k
        else:
            assert w[0][k] == w[1][k], k


def test_obj_lib_api_inference_and_gradients():
    """build_single_graph through the drop-in API: training=False and training=True."""
    import torch
    from sketchyscenecolorization_amd.obj_lib import graph_single, models_collection
    from sketchyscenecolorization_amd.synthetic import synthetic_batch
    models_collection.reset_default_graph()
    b = synthetic_batch(2, 3, img=64)
    gen, images, sketches = graph_single.build_single_graph(b['images'], b['sketches'], None, b['class_id'], None, b['text'],
                                                            batch_size=2, training=False, LSTM_hybrid=True, vocab_size=58,
                                                            data_format='NCHW', distance_map=False, block_type='Pix2Pix',
                                                            noise_vec=b['noise_vec'])
    assert gen.shape == (2, 3, 64, 64) and float(gen.abs().max()) <= 1.0
    img2, noise = models_collection.generator_pix2pix(b['sketches'], b['text'], True, 3, 25, 58, noise_vec=b['noise_vec'])
    assert torch.equal(img2, gen) and noise.shape == (2, 256)
    disc, logits = models_collection.discriminator_pix2pix(b['sketches'], gen, 25)
    assert disc.shape == (2, 1, 6, 6) and logits.shape == (2, 25)
    lg, ld, grad_g, grad_d = graph_single.build_single_graph(b['images'], b['sketches'], b['images_d'], b['class_id'],
                                                             b['class_id_d'], b['text'], batch_size=2, training=True,
                                                             LSTM_hybrid=True, vocab_size=58, distance_map=False,
                                                             block_type='Pix2Pix', noise_vec=b['noise_vec'])
    assert np.isfinite(lg) and np.isfinite(ld)
    names = [n for _, n in grad_g]
    assert 'generator/encoder_1/conv/filter' in names and len(grad_d) == 13
    lg, ld, grad_g, grad_d = graph_single.build_single_graph(b['images'], b['sketches'], b['images_d'], b['class_id'],
                                                             b['class_id_d'], b['text'], batch_size=2, training=True,
                                                             LSTM_hybrid=True, vocab_size=58, noise_vec=b['noise_vec'])
    assert np.isfinite(lg) and np.isfinite(ld)      # default block_type = 'MRU'
    assert 'generator/mru_conv_unit_t_1_layer_0/update_gate/weights' in [n for _, n in grad_g]


@pytest.mark.parametrize('bt', ['MRU', 'Residual'])
def test_cli_inference_default_block_types(tmp_path, monkeypatch, bt):
    """--mode inference with the reference's default block type (MRU, what Pipeline_utils/fg_color_utils.py runs)
    and with Residual, restoring a snapshot keyed by the TF variable names."""
    from PIL import Image, ImageDraw
    import obj_colorization_main as cli
    from sketchyscenecolorization_amd.obj_lib.main_procedure import save_checkpoint
    from sketchyscenecolorization_amd.params import ParamStore
    monkeypatch.chdir(tmp_path)
    ts = '2018-01-02-03-04-05'
    run = os.path.join('outputs', ts)
    save_checkpoint(ParamStore(bt, 58, 64, 'cuda', seed=3), os.path.join(run, 'snapshot'), 'model_9.ckpt', 9)
    os.makedirs('examples')
    im = Image.new('L', (256, 256), 255)
    ImageDraw.Draw(im).rectangle([40, 120, 220, 190], outline=0, width=3)
    im.save('examples/bus.png')
    args = ['--mode', 'inference', '-rf', ts, '-si', '1', '--infer_name', 'bus.png', '--instruction',
            'the bus is orange with gray windows']
    cli.main(args + ([] if bt == 'MRU' else ['-bt', bt]))
    o = np.array(Image.open(os.path.join(run, 'inference_results', 'bus_output.png')))
    assert o.shape == (64, 64, 3) and o.dtype == np.uint8 and o.std() > 0


def test_bg_cli_train_resume_test(tmp_path, monkeypatch):
    """Background_Colorization CLI on synthetic scenes: train, snapshot, resume, test-mode PNGs."""
    import bg_colorization_main as bgcli
    from PIL import Image
    monkeypatch.chdir(tmp_path)
    bgcli.main(['--mode', 'train', '--image_size', '64', '--max_steps', '3', '--save_freq', '2', '--progress_freq', '1',
                '--summary_freq', '1'])
    stamp = sorted(os.listdir('outputs'))[0]
    snap = os.path.join('outputs', stamp, 'snapshot')
    assert os.path.exists(os.path.join(snap, 'snapshot-2')) and os.path.exists(os.path.join(snap, 'snapshot-3'))
    scal = [json.loads(l) for l in open(os.path.join('outputs', stamp, 'log', 'scalars.jsonl'))]
    assert len(scal) == 3 and all(np.isfinite(s['gen_loss']) for s in scal)
    bgcli.main(['--mode', 'train', '--resume_from', stamp, '--image_size', '64', '--max_steps', '4', '--save_freq', '1'])
    assert os.path.exists(os.path.join(snap, 'snapshot-4'))
    bgcli.main(['--mode', 'test', '--resume_from', stamp, '--image_size', '64'])
    res = os.path.join('outputs', stamp, 'results')
    o = np.array(Image.open(os.path.join(res, 'synthetic_0_outputs.png')))
    assert o.shape == (64, 64, 3) and o.dtype == np.uint8
    assert len(glob.glob(os.path.join(res, '*_inputs.png'))) == 8


def test_cli_trains_from_tfrecords(tmp_path, monkeypatch):
    """--mode train reads data/tfrecord/train (the reference's dataset location) through the TensorFlow-free reader."""
    import obj_colorization_main as cli
    from sketchyscenecolorization_amd import tfrecord as tf
    monkeypatch.chdir(tmp_path)
    rng = np.random.RandomState(0)
    os.makedirs('data/tfrecord/train')
    recs = []
    for i in range(6):
        sk = np.full((384, 384, 3), 255, np.uint8)
        sk[60 * i:60 * i + 6, 40:340] = 0
        text = np.zeros(15, np.uint8)
        text[-3:] = [3, 4, 5]
        recs.append(tf.make_example({'ImageName': b'x.png', 'cartoon_data': rng.randint(0, 256, (384, 384, 3)).astype(np.uint8).tobytes(),
                                     'sketch_data': sk.tobytes(), 'Category': b'car', 'Category_id': i % 25,
                                     'Color_text': b'the car is red', 'Text_vocab_indices': text.tobytes()}))
    tf.write_records('data/tfrecord/train/a.tfrecord', recs)
    cli.main(['--mode', 'train', '-bt', 'Pix2Pix', '-si', '1', '-bs', '2', '-mi', '2', '-smf', '1', '-swf', '1'])
    run = os.path.join('outputs', sorted(os.listdir('outputs'))[0])
    scal = [json.loads(l) for l in open(os.path.join(run, 'log', 'scalars.jsonl'))]
    assert len(scal) == 2 and all(np.isfinite(s['total_loss/g']) for s in scal)


def test_paired_queue_device_decode_equals_host_decode(tmp_path):
    """PairedQueue keeps raw records and decodes a batch on the device (default on a GPU box): same examples in the same
    order as the host-decoding queue with the same seed, sketches bit-equal, images equal up to the dequantisation noise."""
    from sketchyscenecolorization_amd import tfrecord as tf
    from sketchyscenecolorization_amd.obj_lib.input_pipeline import PairedQueue
    rng = np.random.RandomState(5)
    d = os.path.join(tmp_path, 'data', 'tfrecord', 'train')
    os.makedirs(d)
    recs = []
    for i in range(7):
        sk = np.full((384, 384, 3), 255, np.uint8)
        sk[50 * i:50 * i + 5, 30:350] = 0
        text = np.zeros(15, np.uint8)
        text[-2:] = [7, 9]
        recs.append(tf.make_example({'ImageName': ('n%d.png' % i).encode(), 'cartoon_data': rng.randint(0, 256, (384, 384, 3)).astype(np.uint8).tobytes(),
                                     'sketch_data': sk.tobytes(), 'Category': b'car', 'Category_id': i,
                                     'Color_text': b'the car is red', 'Text_vocab_indices': text.tobytes()}))
    tf.write_records(os.path.join(d, 'a.tfrecord'), recs)
    base = os.path.join(tmp_path, 'data')
    qd = PairedQueue('train', 3, min_after_dequeue=2, data_base_dir=base, seed=11)
    qh = PairedQueue('train', 3, min_after_dequeue=2, data_base_dir=base, seed=11, device_decode=False)
    assert qd.device_decode and not qh.device_decode
    for _ in range(3):
        di, ds, dc, dt = qd.dequeue()
        hi, hs, hc, ht = qh.dequeue()
        assert torch.is_tensor(di) and di.is_cuda and di.shape == (3, 3, 192, 192)
        assert (dc == hc).all() and (dt == ht).all()
        assert np.array_equal(ds.cpu().numpy(), hs)
        assert np.abs(di.cpu().numpy() - hi).max() <= 2.0 / 256 + 1e-6
    # the prefetching queue of the training procedure (host half of the next batches on a thread of its own, pinned staging
    # buffers reused in turn): the same batches, bit for bit, as the synchronous device-decoding queue -- over more batches
    # than the staging ring has buffers
    qs = PairedQueue('train', 3, min_after_dequeue=2, data_base_dir=base, seed=11)
    qp = PairedQueue('train', 3, min_after_dequeue=2, data_base_dir=base, seed=11, prefetch=True)
    assert qp.prefetch and not qs.prefetch
    try:
        for _ in range(8):
            a, b = qs.dequeue(), qp.dequeue()
            torch.cuda.synchronize()
            assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and (a[2] == b[2]).all() and (a[3] == b[3]).all()
        assert qp._thread is not None and qp._thread.is_alive()
    finally:
        qp.close()


@pytest.mark.parametrize('opt', ['RMSprop', 'AdaDelta', 'AdaGrad'])
def test_cli_other_optimizers(tmp_path, monkeypatch, opt):
    """--optimizer choices of the reference CLI (graph_single.py:584-593) all train."""
    import obj_colorization_main as cli
    monkeypatch.chdir(tmp_path)
    cli.main(['--mode', 'train', '-bt', 'Pix2Pix', '-si', '1', '-bs', '2', '-mi', '2', '-smf', '1', '-swf', '1', '-opt', opt])
    run = os.path.join('outputs', sorted(os.listdir('outputs'))[0])
    scal = [json.loads(l) for l in open(os.path.join(run, 'log', 'scalars.jsonl'))]
    assert len(scal) == 2 and all(np.isfinite(s['total_loss/g']) for s in scal)


def test_cli_train_fails_on_handoff_timeout(tmp_path):
    """A conv launch whose K-slice hand-off times out stores a partial sum: the CLI must die with the launch's name, not go
    on training with a finite loss.  Forced through the test hook (producers withhold their flags, every tile sliced in 2,
    20 ms bound) in a process of its own."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, SSC_SK_TEST_WITHHOLD='1', SSC_SK_TIMEOUT_MS='20', SSC_TS_FORCE='0,2', PYTHONPATH=root)
    r = subprocess.run([sys.executable, os.path.join(root, 'obj_colorization_main.py'), '--mode', 'train', '-bt', 'Pix2Pix',
                        '-si', '1', '-bs', '8', '-mi', '3', '-smf', '2', '-swf', '1'], cwd=str(tmp_path), env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, universal_newlines=True, timeout=900)
    assert r.returncode != 0, r.stdout[-2000:]
    assert 'hand-off timed out' in r.stderr and ('conv_fwd<' in r.stderr or 'conv_bf16x6<' in r.stderr), r.stderr[-3000:]
    assert not glob.glob(os.path.join(str(tmp_path), 'outputs', '*', 'snapshot', 'model_*'))
