"""On-device augmentation front end: replaces the reference's DataLoader workers
(data/mpii_for_mpii.py, data/joint_train_pose.py, data/joint_train_s_r_agent.py) for frames that are
already resident in HBM.  Everything below runs as HIP kernels; the host only sequences them.

A `DeviceBatch` holds MPII-shape annotations for B people:
  frames  uint8  [B][Hs][Ws][3]   source images
  meta    fp32   [B][4]           objpos_x, objpos_y, scale (already MPII-normalised: c.y += 15 s; s *= 1.25,
                                  data/mpii_for_mpii.py:101-104), frame width
  joints  fp32   [B][16][2]       joint_self in image pixels (<= 0 means not annotated)
  normalizer fp32 [B]             0.6 * head size
"""
import torch

from . import _lib
from ._lib import lib, check, ptr, stream, require_gpu
from .pylib import HumanAug


class DeviceBatch(object):
    def __init__(self, frames, objpos, scale, joints, normalizer, sizes=None, index=None):
        """sizes: optional [B][2] (width, height) of each person's own image when the frames are padded to a common size
        (real MPII images); index: optional dataset indices of the samples (validation predictions are stored by index)."""
        require_gpu()
        dev = _lib.device()
        self.frames = frames.to(dev).contiguous()
        B, Hs, Ws, _ = self.frames.shape
        self.B, self.Hs, self.Ws = B, Hs, Ws
        meta = torch.zeros(B, 4, dtype=torch.float32)
        meta[:, 0:2] = torch.as_tensor(objpos, dtype=torch.float32)
        meta[:, 2] = torch.as_tensor(scale, dtype=torch.float32).reshape(-1)
        meta[:, 3] = float(Ws)
        self.sizes = None
        if sizes is not None:
            sz = torch.as_tensor(sizes, dtype=torch.int32).reshape(B, 2)
            meta[:, 3] = sz[:, 0].float()          # the flip of the centre mirrors about the sample's own width
            self.sizes = sz.to(dev).contiguous()
        self.index = None if index is None else [int(i) for i in index]
        self.meta = meta.to(dev)
        self.joints = torch.as_tensor(joints, dtype=torch.float32).to(dev).contiguous()
        self.normalizer = torch.as_tensor(normalizer, dtype=torch.float32).to(dev).contiguous()
        self.params = torch.zeros(B, 8, dtype=torch.float64, device=dev)
        self.ready = None          # event after which the tensors above are valid on ANY stream (set by asynchronous feeders)

    def record_stream(self, stream):
        """the batch was built on another stream (an asynchronous feeder): tell the allocator who consumes it"""
        for t in (self.frames, self.meta, self.joints, self.normalizer, self.params, self.sizes):
            if t is not None and t.is_cuda:
                t.record_stream(stream)

    @staticmethod
    def synthetic(B, seed=0, Hs=720, Ws=1280):
        """MPII-shape synthetic people."""
        g = torch.Generator().manual_seed(seed)
        frames = torch.randint(0, 256, (B, Hs, Ws, 3), generator=g, dtype=torch.uint8)
        objpos = torch.stack([Ws / 2 + (torch.rand(B, generator=g) * 200 - 100), Hs / 2 + (torch.rand(B, generator=g) * 200 - 100)], 1)
        sp = 1.5 + 2.0 * torch.rand(B, generator=g)
        objpos[:, 1] += 15 * sp
        scale = sp * 1.25
        joints = objpos[:, None, :] + torch.randn(B, 16, 2, generator=g) * (60 * sp)[:, None, None]
        joints[..., 0].clamp_(1, Ws - 1); joints[..., 1].clamp_(1, Hs - 1)
        joints[torch.rand(B, 16, generator=g) < 0.1] = 0
        normalizer = (40 + 80 * torch.rand(B, generator=g)) * 0.6
        return DeviceBatch(frames, objpos, scale, joints, normalizer)


class BatchFeed(object):
    """A SIZED, re-iterable source of DeviceBatches (what `len(train_loader)` / `enumerate(train_loader)` are to the
    reference's loops, stack-hg.py:133,183): len() = number of batches, .num_samples = number of people, iter() starts a
    fresh pass (`make_iter` is called once per pass)."""

    def __init__(self, num_batches, num_samples, make_iter, dataset_size=None):
        self._n, self.num_samples, self._make_iter = int(num_batches), int(num_samples), make_iter
        # people in the SPLIT the feed draws from (>= num_samples: a drop_last / sharded feed visits fewer per pass)
        self.dataset_size = int(num_samples if dataset_size is None else dataset_size)

    def __len__(self):
        return self._n

    def __iter__(self):
        return iter(self._make_iter())

    @staticmethod
    def of(batches):
        """a list of resident batches (the synthetic benchmark input)"""
        batches = list(batches)
        return BatchFeed(len(batches), sum(b.B for b in batches), lambda: batches)


def num_samples(batches):
    """people in a feed WITHOUT consuming it"""
    return batches.num_samples if hasattr(batches, 'num_samples') else sum(b.B for b in batches)


def dataset_size(batches):
    """people in the split behind a feed (what a dataset-order, keep-everything pass over it visits): a shuffled training
    feed drops its last partial batch, so its num_samples can be smaller"""
    return batches.dataset_size if hasattr(batches, 'dataset_size') else num_samples(batches)


class Augmenter(object):
    """Draw augmentation parameters by the reference's laws and produce the network input + targets."""

    def __init__(self, seed=0, inp_res=256, out_res=64):
        self.seed, self.inp_res, self.out_res = int(seed), inp_res, out_res
        self.step = 0

    def _finish(self, batch, want_nchw=False):
        t_out, _ = HumanAug.affine_params(batch.params, self.inp_res, self.out_res)
        img4, imgf, _ = HumanAug.crop_batch(batch.frames, batch.params, res=self.inp_res, want_nchw=want_nchw, sizes=batch.sizes)
        pts_heat, pts_img = HumanAug.transform_pts_batch(batch.joints, batch.params, t_out, batch.Ws, sizes=batch.sizes)
        # fp32 c, s, r for the metric calls from ONE launch of the library (three framework elementwise kernels per step before round 6)
        B = batch.B
        csr = torch.empty(4 * B, dtype=torch.float32, device=batch.params.device)
        check(lib().pa_params_csr(ptr(batch.params), B, ptr(csr), stream()), 'pa_params_csr')
        return {'img4': img4, 'img': imgf, 'pts': pts_heat, 'grnd_pts': pts_img,
                'c': csr[:2 * B].view(B, 2), 's': csr[2 * B:3 * B], 'r': csr[3 * B:], 'normalizer': batch.normalizer}

    def regular(self, batch, want_nchw=False):
        """data/mpii_for_mpii.py:119-135."""
        check(lib().pa_sample_aug(ptr(batch.meta), None, None, 0, self.seed, self.step, batch.B, ptr(batch.params), stream()))
        self.step += 1
        return self._finish(batch, want_nchw)

    def agent(self, batch, scale_idx, rot_idx, mode=1, want_nchw=False):
        """data/joint_train_s_r_agent.py:134-177: mode 1 = both bins (+flip, colour), 2 = scale only, 3 = rotation only."""
        check(lib().pa_sample_aug(ptr(batch.meta), ptr(scale_idx), ptr(rot_idx), mode, self.seed, self.step, batch.B,
                                  ptr(batch.params), stream()))
        self.step += 1
        return self._finish(batch, want_nchw)

    def fixed(self, batch, scale_exp=0.0, rot=0.0, want_nchw=False):
        """deterministic crop at scale * 2**scale_exp and rotation `rot` degrees, no flip / colour jitter: the 7 + 7 crops
        per person of the distribution collection (data/collect_scale_distri.py:149-157, data/collect_rotation_distri.py)."""
        p = batch.params
        p.zero_()
        p[:, 0:3] = batch.meta[:, 0:3].double()
        p[:, 2] *= 2.0 ** float(scale_exp)
        p[:, 3] = float(rot)
        p[:, 5:8] = 1.0
        return self._finish(batch, want_nchw)

    def standard(self, batch, want_nchw=False):
        """un-augmented crop (inp_std, data/joint_train_s_r_agent.py:160): scale as annotated, no rotation / flip / jitter."""
        p = batch.params
        p.zero_()
        p[:, 0:3] = batch.meta[:, 0:3].double()
        p[:, 5:8] = 1.0
        return self._finish(batch, want_nchw)


class AugmentAhead(object):
    """The reference's DataLoader workers prepare batch i + 1 while batch i trains (stack-hg.py:73-77, num_workers); here the
    device augmentation of the NEXT batch (law, crop, joint transform: ~10 small launches, ~0.2 ms) is enqueued on its own
    stream and overlaps the current step's forward pass instead of sitting in front of it.

        ahead = AugmentAhead(augmenter); ahead.start(first_batch)
        for batch, nxt in pairs:  data = ahead.take(); ahead.start(nxt); train_step(..., data=data)
    """

    def __init__(self, augmenter, kind='regular'):
        self.augmenter, self.kind = augmenter, kind
        self.stream = torch.cuda.Stream() if _lib.device().type == 'cuda' else None     # (None: the ABI-stub tests of the host flow)
        self.pending = None

    def start(self, batch):
        if batch is None:
            self.pending = None
            return
        if self.stream is None:
            self.pending = getattr(self.augmenter, self.kind)(batch)
            return
        ready = getattr(batch, 'ready', None)          # an asynchronous feeder's copy event (mpii_for_mpii.MPII.batches)
        if ready is not None:
            self.stream.wait_event(ready)
            batch.record_stream(self.stream)
        with torch.cuda.stream(self.stream):
            self.pending = getattr(self.augmenter, self.kind)(batch)

    def take(self):
        data = self.pending
        self.pending = None
        if self.stream is None:
            return data
        main = torch.cuda.current_stream()
        main.wait_stream(self.stream)
        for v in data.values():
            if isinstance(v, torch.Tensor):
                v.record_stream(main)              # allocated on the augmentation stream, consumed on the caller's
        return data


def with_next(batches):
    """(batch, next batch or None) pairs of a feed"""
    it = iter(batches)
    try:
        cur = next(it)
    except StopIteration:
        return
    for nxt in it:
        yield cur, nxt
        cur = nxt
    yield cur, None
