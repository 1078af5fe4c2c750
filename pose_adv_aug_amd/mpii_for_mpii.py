"""data/mpii_for_mpii.py of the reference as a feeder of `DeviceBatch`es.

The reference's `MPII(data.Dataset)` loads ONE annotated person per item: JSON record -> JPEG -> host augmentation ->
crop (data/mpii_for_mpii.py:83-163) in DataLoader worker processes.  Here the host only parses the JSON and decodes the
images (PIL, a thread pool); the frames of a batch are copied, top-left aligned, into one padded uint8 buffer on the
device together with their own (width, height), and everything from the MPII pre-normalisation on -- law, flip, colour,
crop, joint transform, heat maps -- happens in the device kernels (data.Augmenter).

JSON record fields used (data/mpii_for_mpii.py:29-47, 86-104):
  dataset ('MPII'), isValidation, img_paths, joint_self [16][3], objpos [2], scale_provided, normalizer."""
import json
import mmap
import os
import queue
import threading
from concurrent.futures import ProcessPoolExecutor, ThreadPoolExecutor

import numpy as np
import torch

from .data import DeviceBatch


# ---- decode workers (forked BEFORE they are needed, like the reference's DataLoader workers, stack-hg.py:75): a worker decodes
# one JPEG straight into a frame slot that it shares with the parent (anonymous shared mapping inherited over fork), so no
# pixel ever crosses a pipe; it returns only (width, height).
_SLOTS = {}            # slot id -> numpy uint8 view of the shared mapping (filled before the fork)


def _decode_into(args):
    path, slot, B, Hs, Ws, b = args
    from PIL import Image
    arr = _SLOTS[slot][:B * Hs * Ws * 3].reshape(B, Hs, Ws, 3)
    with Image.open(path) as im:
        a = np.asarray(im.convert('RGB'))
    h, w = a.shape[0], a.shape[1]
    arr[b, :h, :w] = a
    return w, h


def _nap(_):
    import time
    time.sleep(0.05)
    return os.getpid()


def _image_size(path):
    from PIL import Image
    with Image.open(path) as im:          # header only: PIL decodes lazily
        return im.size


class FrameSlots(object):
    """`depth` host buffers of `slot_bytes` shared with the forked decode pool, page-locked when the runtime allows it
    (hipHostRegister through torch's cudart binding), handed out through a free list."""

    def __init__(self, depth, slot_bytes, workers):
        import multiprocessing as mp
        self.slot_bytes = slot_bytes
        self.maps = []
        base = len(_SLOTS)
        for k in range(depth):
            m = mmap.mmap(-1, slot_bytes)                 # MAP_SHARED | MAP_ANONYMOUS: children of a fork see the same pages
            self.maps.append(m)
            _SLOTS[base + k] = np.frombuffer(m, dtype=np.uint8)
        self.ids = list(range(base, base + depth))
        self.pool = ProcessPoolExecutor(max_workers=max(1, workers), mp_context=mp.get_context('fork'))
        self.n_workers = len(set(self.pool.map(_nap, range(4 * max(1, workers)))))   # fork every worker NOW, before any helper thread exists (they inherit _SLOTS)
        self.pinned = False
        if torch.cuda.is_available():
            try:
                rt = torch.cuda.cudart()
                self.pinned = all(int(rt.cudaHostRegister(_SLOTS[i].ctypes.data, slot_bytes, 0)) == 0 for i in self.ids)
            except Exception:
                self.pinned = False
        self.free = queue.Queue()
        for i in self.ids:
            self.free.put(i)

    def close(self):
        self.pool.shutdown(wait=False, cancel_futures=True)


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
        batch = DeviceBatch(frames, np.stack([r[2] for r in recs]), np.asarray([r[3] for r in recs]),
                            np.stack([r[1] for r in recs]), np.asarray([r[4] for r in recs]), sizes=sizes, index=list(indices))
        if torch.cuda.is_available():            # the copies from the page-locked staging tensor are asynchronous: consumers on
            batch.ready = torch.cuda.Event()     # other streams (data.AugmentAhead) wait for this event
            batch.ready.record()
        return batch

    def batches(self, batch_size, shuffle=None, seed=0, drop_last=False, workers=8, rank=0, world=1, decoder='thread', prefetch=6):
        """A sized BatchFeed over the split (shuffle defaults to is_train, stack-hg.py:73-83): len() = number of batches,
        every iter() is one pass in a fresh order (seed + pass number).  Data parallel: all ranks draw the SAME order and
        rank r takes batches r, r + world, ... (equal counts on every rank).
        decoder='thread': the next batch is decoded by a thread pool (PIL drops the GIL inside the JPEG decoder) while the
        caller trains on the current one -- a few hundred images/s.
        decoder='process': the reference's DataLoader workers (stack-hg.py:75 num_workers) at the engine's rate: `workers`
        forked processes decode straight into page-locked frame slots shared with this process, `prefetch` batches are
        assembled concurrently and copied to the device on their own stream."""
        from .data import BatchFeed
        n = len(self)
        nb_all = n // batch_size if drop_last else (n + batch_size - 1) // batch_size
        nb = nb_all // world if world > 1 else nb_all
        passes = [0]

        def chunks_of_pass():
            order = np.arange(n)
            if self.is_train if shuffle is None else shuffle:
                np.random.default_rng(seed + passes[0]).shuffle(order)
            passes[0] += 1
            chunks = [order[i:i + batch_size].tolist() for i in range(0, n, batch_size)][:nb_all]
            return chunks[rank::world][:nb] if world > 1 else chunks

        def one_pass_threads():
            chunks = chunks_of_pass()
            with ThreadPoolExecutor(max_workers=max(1, workers)) as pool, ThreadPoolExecutor(max_workers=1) as ahead:
                fut = ahead.submit(self.load_batch, chunks[0], pool) if chunks else None
                for k in range(len(chunks)):
                    batch = fut.result()
                    fut = ahead.submit(self.load_batch, chunks[k + 1], pool) if k + 1 < len(chunks) else None
                    yield batch

        def one_pass_processes():
            chunks = chunks_of_pass()
            slots = self._frame_slots(batch_size, workers, prefetch)
            copy_stream = torch.cuda.Stream()
            with ThreadPoolExecutor(max_workers=max(1, prefetch)) as assemblers:
                futs = [assemblers.submit(self._assemble, c, slots, copy_stream) for c in chunks[:prefetch]]
                released = 0
                try:
                    for k in range(len(chunks)):
                        batch, ev, slot = futs[k].result()
                        torch.cuda.current_stream().wait_event(ev)
                        batch.record_stream(torch.cuda.current_stream())     # allocated on the copy stream, consumed on the caller's
                        if k + prefetch < len(chunks):
                            futs.append(assemblers.submit(self._assemble, chunks[k + prefetch], slots, copy_stream))
                        yield batch
                        ev.synchronize()                      # (long done: the copy finished before the batch was trained on)
                        slots.free.put(slot)
                        released = k + 1
                finally:                                      # a pass abandoned early (next(iter(feed))): hand every slot back
                    for f in futs[released:]:
                        try:
                            _, ev, slot = f.result()
                            ev.synchronize()
                            slots.free.put(slot)
                        except Exception:
                            pass

        total = sum(min(batch_size, n - k * batch_size) for k in (range(rank, nb_all, world) if world > 1 else range(nb_all)))
        return BatchFeed(nb, total if world == 1 else min(total, nb * batch_size), one_pass_processes if decoder == 'process' else one_pass_threads,
                         dataset_size=n)

    # ---- process-pool path
    MAX_FRAME = (1088, 1920)                  # slot size: MPII images are at most 1920 x 1080

    def _frame_slots(self, batch_size, workers, prefetch):
        key = (batch_size, workers, prefetch)
        if getattr(self, '_slots_key', None) != key:
            if getattr(self, '_slots', None) is not None:
                self._slots.close()
            self._slots = FrameSlots(prefetch + 2, batch_size * self.MAX_FRAME[0] * self.MAX_FRAME[1] * 3, workers)
            self._slots_key = key
        return self._slots

    def _assemble(self, indices, slots, copy_stream):
        """one batch: image headers -> decode tasks into a free slot -> asynchronous copy to the device on `copy_stream`"""
        recs = [self.record(i) for i in indices]
        sizes = np.asarray([_image_size(r[0]) for r in recs], dtype=np.int32)             # (w, h) per person
        B, Hs, Ws = len(recs), int(sizes[:, 1].max()), int(sizes[:, 0].max())
        if B * Hs * Ws * 3 > slots.slot_bytes:
            raise ValueError('images larger than the frame slots (%d x %d)' % (Ws, Hs))
        slot = slots.free.get()
        list(slots.pool.map(_decode_into, [(r[0], slot, B, Hs, Ws, b) for b, r in enumerate(recs)]))
        host = torch.from_numpy(_SLOTS[slot][:B * Hs * Ws * 3].reshape(B, Hs, Ws, 3))
        with torch.cuda.stream(copy_stream):
            frames = host.to(torch.device('cuda', torch.cuda.current_device()), non_blocking=slots.pinned)
            batch = DeviceBatch(frames, np.stack([r[2] for r in recs]), np.asarray([r[3] for r in recs]),
                                np.stack([r[1] for r in recs]), np.asarray([r[4] for r in recs]), sizes=sizes, index=list(indices))
            ev = torch.cuda.Event()
            ev.record(copy_stream)
        batch.ready = ev
        return batch, ev, slot
