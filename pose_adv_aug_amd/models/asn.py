"""ASN agent of the reference (models/asn_stacked_hg.py:349-439) on the HIP engine: the scale/rotation head (is_aug) or the
occlusion-mask head (is_dropout)."""
import torch

from .._lib import lib, check, ptr
from .asn_stacked_hg import _HipModule


class ASN(_HipModule):
    def __init__(self, chan, scale_num, rotation_num, res=256, default_batch=24, is_dropout=False):
        super().__init__()
        self.chan, self.scale_num, self.rotation_num, self.res = chan, scale_num, rotation_num, res
        self.is_dropout = bool(is_dropout)
        self.default_batch = default_batch
        self._last_B = None
        self._last_pose = None

    def _create(self, B):
        if self.is_dropout:
            return lib().pa_asn_create_dropout(self.chan, B, self.res)
        return lib().pa_asn_create(self.chan, self.scale_num, self.rotation_num, B, self.res)

    # ---- occlusion agent (create_asn(is_dropout=True))
    def _forward_masks_from_pose(self, pose, x=None, img4=None, update_running=True):
        """hg(img, asn, is_half_hg=True, is_dropout=True) (models/asn_stacked_hg.py:313-316): half hourglass of the pose net
        + ASN.forward(is_dropout=True) -> mask logits [B][1][4][4].  update_running=False: the pose net's running BatchNorm
        estimates are left alone (the whole-hourglass call updates them once, in its full forward)."""
        B = x.shape[0] if x is not None else img4.shape[0]
        hp = pose._net(B)
        pose._last_B = B
        xin = x.contiguous().float() if x is not None else None
        mode = (1 if update_running else 2) if pose.training else 0
        check(lib().pa_hg_forward_half(hp, ptr(xin), ptr(img4), mode), 'pa_hg_forward_half')
        if pose.training and update_running:
            pose._nbt += 1
        return self.forward_masks(pose)

    def forward_masks(self, pose, update_running=True):
        B = pose._last_B
        hp, ha = pose._net(B), self._net(B)
        out = torch.empty((B, 1, 4, 4), dtype=torch.float32, device=self.flat_params.device)
        mode = (1 if update_running else 2) if self.training else 0
        check(lib().pa_asn_forward_masks(ha, hp, mode, ptr(out)), 'pa_asn_forward_masks')
        if self.training and update_running:
            self._nbt += 1
        self._last_B, self._last_pose = B, pose
        return out

    def backward_masks(self, grad_pred_mask):
        """pred_mask.backward(grad) of the reference's autograd: d(loss)/d(mask logits) [B][1][4][4] -> flat_grads (the
        reference ships no loss for the occlusion agent; the caller supplies the gradient of its own)."""
        B, pose = self._last_B, self._last_pose
        g = grad_pred_mask.to(self.flat_params.device, torch.float32).reshape(B, 16).contiguous()
        check(lib().pa_asn_backward_masks(self._net(B), pose._net(B), ptr(g)), 'pa_asn_backward_masks')

    def _forward_from_pose(self, pose, x=None, img4=None, is_half_hg=True, update_running=True):
        """hg(img, asn, is_half_hg=True, is_aug=True) of the reference (models/asn_stacked_hg.py:300-304): the
        pose net runs its stem + first hourglass down path in ITS current mode, the agent (in its own mode) maps the
        detached features to (scale_logits [B][S], rotation_logits [B][R])."""
        if not is_half_hg:
            raise NotImplementedError('the joint loop only uses the half-hourglass agent forward')
        B = x.shape[0] if x is not None else img4.shape[0]
        hp, ha = pose._net(B), self._net(B)
        pose._last_B = B
        xin = x.contiguous().float() if x is not None else None
        check(lib().pa_hg_forward_half(hp, ptr(xin), ptr(img4), 1 if pose.training else 0), 'pa_hg_forward_half')
        if pose.training:
            pose._nbt += 1
        return self.forward_features(pose, update_running)

    def forward_features(self, pose, update_running=True):
        """agent forward on the features of the pose net's last (half or full) forward"""
        B = pose._last_B
        hp, ha = pose._net(B), self._net(B)
        dev = self.flat_params.device
        ls = torch.empty((B, self.scale_num), dtype=torch.float32, device=dev)
        lr = torch.empty((B, self.rotation_num), dtype=torch.float32, device=dev)
        mode = (1 if update_running else 2) if self.training else 0
        check(lib().pa_asn_forward(ha, hp, mode, ptr(ls), ptr(lr)), 'pa_asn_forward')
        if self.training and update_running:
            self._nbt += 1
        self._last_B, self._last_pose = B, pose
        return ls, lr

    @staticmethod
    def kl_loss(scale_logits, rot_logits, grnd_scale_distri, grnd_rotation_distri, log_eps=1e-7):
        """the same loss value without a backward pass (validation, pretrain-s-r-agent.py:228-238): [B][7] device tensors."""
        total = 0.0
        for logits, t in ((scale_logits, grnd_scale_distri), (rot_logits, grnd_rotation_distri)):
            t = t.to(logits.device, torch.float32)
            logp = torch.log(torch.softmax(logits, 1) + log_eps) if log_eps > 0 else torch.log_softmax(logits, 1)
            kl = torch.where(t > 0, t * (torch.log(t.clamp(min=1e-30)) - logp), torch.zeros_like(t))
            total = total + kl.mean() * t.shape[1]
        return total

    def loss_and_backward(self, grnd_scale_distri, grnd_rotation_distri, log_eps=1e-7):
        """KL(log(softmax + log_eps) || target) * K for both heads and its gradient w.r.t. the agent's parameters
        (flat_grads).  log_eps = 1e-7: joint-train-pose-s-r-agent.py:399-407; log_eps = 0: the LogSoftmax form of the
        agent pre-training (pretrain-s-r-agent.py:177-190).  Returns the loss as a 0-d GPU tensor."""
        B, pose = self._last_B, self._last_pose
        dev = self.flat_params.device
        check(lib().pa_asn_set_log_eps(self._net(B), float(log_eps)), 'pa_asn_set_log_eps')
        ts = grnd_scale_distri.to(dev, torch.float32).contiguous()
        tr = grnd_rotation_distri.to(dev, torch.float32).contiguous()
        loss = torch.zeros(1, dtype=torch.float32, device=dev)
        check(lib().pa_asn_backward(self._net(B), pose._net(B), ptr(ts), ptr(tr), ptr(loss)), 'pa_asn_backward')
        return loss[0]
