#!/usr/bin/env python3
"""Row f2 end to end: MPII-format JSON + JPEG files -> host decode (thread / process pool) -> padded uint8 frames on the
device -> device augmentation -> training step.  Writes N synthetic 1280x720 JPEGs and an annotation file to a temp
directory, then reports (a) decoded images/sec of the feeder alone for several worker counts and (b) images/sec of
stack_hg.train over that feed beside the engine's rate on resident frames.

    python tools/bench_feed.py [N=240]"""
import json
import os
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402


def make_dataset(root, n):
    from PIL import Image
    from tests import inputs
    base = inputs.warp_frame('smooth').astype(np.int16)
    g = np.random.default_rng(0)
    anno = []
    for i in range(n):
        img = np.clip(base + g.integers(-12, 13, base.shape, dtype=np.int16) + (i % 7) * 5, 0, 255).astype(np.uint8)
        Image.fromarray(img).save(os.path.join(root, 'im%04d.jpg' % i), quality=90)
        c = [640 + float(g.uniform(-100, 100)), 360 + float(g.uniform(-100, 100))]
        sp = float(g.uniform(1.5, 3.5))
        joints = [[float(np.clip(c[0] + g.normal() * 60 * sp, 1, 1279)), float(np.clip(c[1] + g.normal() * 60 * sp, 1, 719)), 1.0] for _ in range(16)]
        anno.append(dict(dataset='MPII', isValidation=0.0, img_paths='im%04d.jpg' % i, joint_self=joints, objpos=c, scale_provided=sp,
                         normalizer=float(g.uniform(40, 120))))
    path = os.path.join(root, 'mpii-hr-lsp-normalizer.json')
    json.dump(anno, open(path, 'w'))
    return path


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 960
    from pose_adv_aug_amd.mpii_for_mpii import MPII
    from pose_adv_aug_amd.models.asn_stacked_hg import create_hg
    from pose_adv_aug_amd.utils.optim import RMSprop
    from pose_adv_aug_amd.data import Augmenter
    from pose_adv_aug_amd import stack_hg
    import types
    root = tempfile.mkdtemp(prefix='mpii_feed_')
    path = make_dataset(root, n)
    ds = MPII(path, root, is_train=True, log=lambda m: None)
    print('host: %d cores (%s)' % (os.cpu_count(), open('/proc/cpuinfo').read().split('model name')[1].split('\n')[0].strip(': \t')))
    for decoder, workers in (('thread', 8), ('thread', 32), ('process', 32), ('process', 64), ('process', 96)):
        feed = ds.batches(24, shuffle=False, drop_last=True, workers=workers, decoder=decoder)
        if decoder == 'process':
            for b in feed:                                   # first pass forks the pool and page-locks the slots
                pass
        t0 = time.perf_counter(); cnt = 0
        for b in feed:
            cnt += b.B
        torch.cuda.synchronize()
        print('feeder alone   decoder=%-7s workers=%2d : %7.1f img/s (decode + pad + H2D of %d images)' % (decoder, workers, cnt / (time.perf_counter() - t0), cnt))
    net = create_hg(2, 1, 16, 256, default_batch=24); net.reset_parameters(seed=0)
    opt_ = RMSprop(net, lr=2.5e-4)
    aug = Augmenter(seed=1)
    opt = types.SimpleNamespace(print_freq=10 ** 9)
    feed = ds.batches(24, shuffle=True, drop_last=True, workers=64, decoder='process')
    print('frame slots page-locked:', ds._frame_slots(24, 64, 6).pinned)
    stack_hg.train(feed, net, opt_, aug, 0, opt, log=lambda m: None)                   # warm-up pass (pools, kernels)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    stack_hg.train(feed, net, opt_, aug, 1, opt, log=lambda m: None)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print('stack_hg.train over MPII.batches (64 forked decode workers, 6 batches in flight): %.1f img/s fed (%d images, %.2f s)' % (feed.num_samples / dt, feed.num_samples, dt))
    resident = list(feed)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    stack_hg.train(resident, net, opt_, aug, 2, opt, log=lambda m: None)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print('the same batches resident in HBM (engine rate on real-size padded frames): %.1f img/s' % (feed.num_samples / dt))


if __name__ == '__main__':
    main()
