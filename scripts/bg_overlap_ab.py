"""Background train step: in-line vs D(real) beside the generator forward -- ms per step and
the weights after three steps compared bit for bit (same seed, same inputs).  usage: python scripts/bg_overlap_ab.py [img]"""
import os, subprocess, sys, json
img = sys.argv[1] if len(sys.argv) > 1 else '768'
if len(sys.argv) > 2:       # child
    import hashlib, time, torch
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from sketchyscenecolorization_amd.bg_colorization import BGTrainer
    n, im = 1, int(img)
    tr = BGTrainer(image_size=im)
    g = torch.Generator(device='cuda').manual_seed(1)
    x = torch.rand(n, im, im, 3, device='cuda', generator=g) * 2 - 1
    y = torch.rand(n, im, im, 3, device='cuda', generator=g) * 2 - 1
    text = torch.randint(1, 18, (n, 8), dtype=torch.int32, generator=torch.Generator().manual_seed(2)).numpy()
    lab = torch.randint(0, 3, (n, im, im), dtype=torch.int32, generator=torch.Generator().manual_seed(3)).cuda()
    for _ in range(3):
        tr.train_step(x, y, text, lab)
    torch.cuda.synchronize()
    h = hashlib.sha256()
    for sc in (tr.store.generator, tr.store.discriminator):
        h.update(sc.flat.cpu().numpy().tobytes())
    for _ in range(5):
        tr.train_step(x, y, text, lab)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(30):
        tr.train_step(x, y, text, lab)
    torch.cuda.synchronize()
    print(json.dumps({'ms': (time.perf_counter() - t0) / 30 * 1e3, 'weights_sha': h.hexdigest()[:16]}))
    sys.exit(0)
for rep in range(2):
    for name, env in (('in line', {'SSC_BG_OVERLAP_REAL': '0'}), ('real beside G forward', {})):
        r = subprocess.run([sys.executable, os.path.abspath(__file__), img, 'child'], env=dict(os.environ, **env), stdout=subprocess.PIPE,
                           stderr=subprocess.PIPE, universal_newlines=True)
        line = [l for l in r.stdout.splitlines() if l.startswith('{')]
        print('%-40s %s' % (name, line[-1] if line else 'FAILED: ' + r.stderr[-300:]), flush=True)
