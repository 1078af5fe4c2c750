"""data/mpii_for_mpii.py of the reference as a feeder of `DeviceBatch`es.

The reference's `MPII(data.Dataset)` loads ONE annotated person per item: JSON record -> JPEG -> host augmentation ->
crop (data/mpii_for_mpii.py:83-163) in DataLoader worker processes.  Here the host only parses the JSON and decodes the
images (PIL, a thread pool); the frames of a batch are copied, top-left aligned, into one padded uint8 buffer on the
device together with their own (width, height), and everything from the MPII pre-normalisation on -- law, flip, colour,
crop, joint transform, heat maps -- happens in the device kernels (data.Augmenter).

JSON record fields used (data/mpii_for_mpii.py:29-47, 86-104):
  dataset ('MPII'), isValidation, img_paths, joint_self [16][3], objpos [2], scale_provided, normalizer."""
import json
import os
from concurrent.futures import ProcessPoolExecutor, ThreadPoolExecutor

import numpy as np
import torch

from .data import DeviceBatch


class MPII(object):
    def __init__(self, jsonfile, img_folder, inp_res=256, out_res=64, is_train=True, sigma=1,
                 scale_factor=0.25, rot_factor=30, std_size=200, log=print):
        self.img_folder, self.is_train = img_folder, is_train
        self.inp_res, self.out_res, self.sigma = inp_res, out_res, sigma
        self.scale_factor, self.rot_factor, self.std_size = scale_factor, rot_factor, std_size
        with open(jsonfile, 'r') as anno_file:
            self.anno = json.load(anno_file)
        log('loading json file is done...')
        self.train, self.valid = [], []
        for idx, val in enumerate(self.anno):                                  # data/mpii_for_mpii.py:33-42
            if val['dataset'] == 'MPII':
                if val['objpos'][0] <= 0 or val['objpos'][1] <= 0:
                    log('invalid center: %s\nimage name: %s\ndataset: %s' % (val['objpos'], val['img_paths'], val['dataset']))
                if val['isValidation'] == True:                                 # noqa: E712 (the JSON stores 0/1 floats too)
                    self.valid.append(idx)
                else:
                    self.train.append(idx)
        log(('total training images: %d' % len(self.train)) if is_train else ('total validation images: %d' % len(self.valid)))

    def __len__(self):
        return len(self.train) if self.is_train else len(self.valid)

    def record(self, index):
        """The per-person numbers of __getitem__ (:86-104) BEFORE any augmentation: (img_path, pts [16][2], c [2], s, normalizer)."""
        a = self.anno[(self.train if self.is_train else self.valid)[index]]
        pts = np.asarray(a['joint_self'], dtype=np.float32)[:, 0:2]
        c = np.asarray(a['objpos'], dtype=np.float32).copy()
        s = np.float32(a['scale_provided'])
        if a['dataset'] == 'MPII':
            c[1] = c[1] + 15 * s                                                # :101
            s = s * np.float32(1.25)                                            # :102
            normalizer = np.float32(a['normalizer']) * np.float32(0.6)          # :103
        else:
            raise ValueError('no such dataset {}'.format(a['dataset']))
        return os.path.join(self.img_folder, a['img_paths']), pts, c, s, normalizer

    @staticmethod
    def load_image(img_path):
        """utils/imutils.py:38-41 (RGB); kept as uint8 H x W x 3 -- the /255 happens in the warp kernel."""
        from PIL import Image
        with Image.open(img_path) as im:
            return np.array(im.convert('RGB'), dtype=np.uint8)          # a writable copy

    def load_batch(self, indices, pool=None):
        """Decode the images of `indices` and build the DeviceBatch (frames padded to the largest image of the batch)."""
        recs = [self.record(i) for i in indices]
        imgs = list(pool.map(self.load_image, [r[0] for r in recs])) if pool is not None else [self.load_image(r[0]) for r in recs]
        B = len(recs)
        Hs, Ws = max(i.shape[0] for i in imgs), max(i.shape[1] for i in imgs)
        frames = torch.zeros((B, Hs, Ws, 3), dtype=torch.uint8).pin_memory() if torch.cuda.is_available() else torch.zeros((B, Hs, Ws, 3), dtype=torch.uint8)
        sizes = np.zeros((B, 2), dtype=np.int32)
        for b, im in enumerate(imgs):
            h, w = im.shape[0], im.shape[1]
            frames[b, :h, :w] = torch.from_numpy(im)
            sizes[b] = (w, h)
        return DeviceBatch(frames, np.stack([r[2] for r in recs]), np.asarray([r[3] for r in recs]),
                           np.stack([r[1] for r in recs]), np.asarray([r[4] for r in recs]), sizes=sizes, index=list(indices))

    def batches(self, batch_size, shuffle=None, seed=0, drop_last=False, workers=8, rank=0, world=1, decoder='thread'):
        """A sized BatchFeed over the split (shuffle defaults to is_train, stack-hg.py:73-83): len() = number of batches,
        every iter() is one pass in a fresh order (seed + pass number); the next batch is decoded by a thread pool
        (PIL releases the GIL while decoding) while the caller trains on the current one.  Data parallel: all ranks draw
        the SAME order and rank r takes batches r, r + world, ... (equal counts on every rank).
        decoder: 'thread' (PIL drops the GIL inside the JPEG decoder) or 'process' (the reference's DataLoader workers,
        stack-hg.py:75 num_workers: decoded frames come back through a pipe; no GIL at all)."""
        from .data import BatchFeed
        n = len(self)
        nb_all = n // batch_size if drop_last else (n + batch_size - 1) // batch_size
        nb = nb_all // world if world > 1 else nb_all
        passes = [0]

        def one_pass():
            order = np.arange(n)
            if self.is_train if shuffle is None else shuffle:
                np.random.default_rng(seed + passes[0]).shuffle(order)
            passes[0] += 1
            chunks = [order[i:i + batch_size].tolist() for i in range(0, n, batch_size)][:nb_all]
            chunks = chunks[rank::world][:nb] if world > 1 else chunks
            import multiprocessing as mp
            make_pool = (lambda: ProcessPoolExecutor(max_workers=max(1, workers), mp_context=mp.get_context('spawn'))) if decoder == 'process' \
                else (lambda: ThreadPoolExecutor(max_workers=max(1, workers)))
            with make_pool() as pool, ThreadPoolExecutor(max_workers=1) as ahead:
                fut = ahead.submit(self.load_batch, chunks[0], pool) if chunks else None
                for k in range(len(chunks)):
                    batch = fut.result()
                    fut = ahead.submit(self.load_batch, chunks[k + 1], pool) if k + 1 < len(chunks) else None
                    yield batch

        total = sum(min(batch_size, n - k * batch_size) for k in (range(rank, nb_all, world) if world > 1 else range(nb_all)))
        return BatchFeed(nb, total if world == 1 else min(total, nb * batch_size), one_pass)
