"""Synthetic sketch + caption batches: the shapes and value ranges the
reference input pipeline produces (input_pipeline.py:131-154, 235-239), generated from a seed
instead of TFRecords so benchmarks never touch disk."""
import numpy as np
import torch

T_STEPS = 15
NUM_CLASSES = 25


def synthetic_batch(n, seed, img=192, vocab_size=58, device='cuda'):
    """Dict of device tensors (NCHW fp32 in [-1,1], int32 labels) + host caption indices."""
    rng = np.random.RandomState(seed)
    sk = np.ones((n, 1, img, img), dtype=np.float32)         # +1 background, -1 strokes (~5% of pixels)
    for i in range(n):
        for _ in range(6):
            y, x = rng.randint(8, img - 8, size=2)
            dy = rng.randint(-2, 3, size=img)
            dx = rng.randint(-2, 3, size=img)
            for s in range(img):
                sk[i, 0, y:y + 2, x:x + 2] = -1.0
                y = int(np.clip(y + dy[s], 0, img - 2))
                x = int(np.clip(x + dx[s], 0, img - 2))
    sketches = np.repeat(sk, 3, axis=1)                        # 3 identical channels
    images = rng.uniform(-1, 1, size=(n, 3, img, img)).astype(np.float32)
    images_d = rng.uniform(-1, 1, size=(n, 3, img, img)).astype(np.float32)   # the second, independent queue
    text = np.zeros((n, T_STEPS), dtype=np.int32)
    for i in range(n):
        ln = rng.randint(4, 11)
        text[i, T_STEPS - ln:] = rng.randint(2, vocab_size, size=ln)          # left-padded with <pad>=0
    dev = lambda a: torch.from_numpy(a).to(device)
    return {'images': dev(images), 'sketches': dev(sketches), 'images_d': dev(images_d),
            'class_id': dev(rng.randint(0, NUM_CLASSES, size=n).astype(np.int32)),
            'class_id_d': dev(rng.randint(0, NUM_CLASSES, size=n).astype(np.int32)),
            'text': text, 'noise_vec': dev(rng.randn(n, 256).astype(np.float32))}
