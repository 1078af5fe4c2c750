"""utils/checkpoint.py of the reference: same file names, same dict layout, same lenient by-name load."""
import os
import shutil

import scipy.io
import torch


class Checkpoint(object):
    def __init__(self):
        self.save_prefix = ''
        self.load_prefix = ''

    @staticmethod
    def _lr_prefix(train_history):
        return ('lr-%.15f' % train_history.lr[-1]['lr']).rstrip('0').rstrip('.')       # utils/checkpoint.py:15

    def save_checkpoint(self, net, optimizer, train_history, preds=None, is_asn=False):
        """utils/checkpoint.py:14-36.  Keys carry the 'module.' prefix of the reference's DataParallel wrapper
        so that files interchange with the reference."""
        lr_prefix = self._lr_prefix(train_history)
        epoch = train_history.epoch[-1]['epoch']
        save_path = self.save_prefix + lr_prefix + ('-%d.pth.tar' % epoch)
        sd = {k: v.detach().cpu().clone() for k, v in net.state_dict(prefix='module.').items()}
        torch.save({'train_history': train_history.state_dict(), 'state_dict': sd, 'optimizer': optimizer.state_dict()}, save_path)
        print("=> saving checkpoint '{}'".format(save_path))
        if not is_asn:
            save_pred_path = self.save_prefix + lr_prefix + ('-%d-preds.mat' % epoch)
            scipy.io.savemat(save_pred_path, mdict={'preds': torch.as_tensor(preds).cpu().numpy()})
            print("=> saving predictions '{}'".format(save_pred_path))
            if train_history.is_best:
                shutil.copyfile(save_path, self.save_prefix + lr_prefix + ('-%d-model-best.pth.tar' % epoch))
                shutil.copyfile(save_pred_path, self.save_prefix + lr_prefix + ('-%d-preds-best.mat' % epoch))
        return save_path

    def save_preds(self, preds):
        """utils/checkpoint.py:38-43."""
        scipy.io.savemat(self.save_prefix + 'preds.mat', mdict={'preds': torch.as_tensor(preds).cpu().numpy()})

    def load_checkpoint(self, net, optimizer=None, train_history=None):
        """utils/checkpoint.py:45-71: weights are copied by name, unknown keys are skipped."""
        save_path = self.load_prefix + '.pth.tar'
        if not os.path.isfile(save_path):
            print("=> no checkpoint found at '{}'".format(save_path))
            return False
        print("=> loading checkpoint '{}'".format(save_path))
        ck = torch.load(save_path, map_location='cpu', weights_only=False)
        if train_history is not None:
            train_history.load_state_dict(ck['train_history'])
        if optimizer is not None:
            optimizer.load_state_dict(ck['optimizer'])
        _, unexpected = net.load_state_dict(ck['state_dict'], strict=False)
        for name in unexpected:
            print("=> not load weights '{}'".format(name))
        return True
