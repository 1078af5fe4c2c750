"""options/base_options.py of the reference (same flags, defaults and quirks)."""
import argparse
import os

from ..utils import util


def _bool(v):
    # the reference declares these flags with type=bool: any non-empty string is True (Appendix A.10)
    return bool(v)


class BaseOptions(object):
    def __init__(self):
        self.parser = argparse.ArgumentParser()
        self.initialized = False

    def initialize(self):
        p = self.parser
        p.add_argument('--data_dir', type=str, default='./dataset', help='training data or listfile path')
        p.add_argument('--exp_dir', type=str, default='./exp', help='root experimental directory')
        p.add_argument('--exp_id', type=str, default='', help='experimental name')
        p.add_argument('--gpu_id', type=str, default='0', help='gpu ids: e.g. 0  0,1,2, 0,2')
        p.add_argument('--nThreads', type=int, default=4, help='number of data loading threads')
        p.add_argument('--is_train', type=_bool, default=False, help='training mode')
        p.add_argument('--use_visdom', type=_bool, default=True, help='use visdom to display')
        p.add_argument('--vis_env', type=str, default='main', help='environment name for visdom')
        p.add_argument('--use_html', type=_bool, default=False, help='use html to store images')
        p.add_argument('--display_winsize', type=int, default=256, help='display window size')
        p.add_argument('--dataset', type=str, default='mpii', help='dataset type')
        self.initialized = True

    @staticmethod
    def _normalise_prefix(v):
        """What options/base_options.py:62-85 intends (it reads the stale names resume_prefix_* and crashes
        as shipped): strip '.pth.tar' and append '-', because the scripts do opt.load_prefix_pose[0:-1]."""
        if v != '':
            v = v[0:v.index('pth') - 1] + '-'
        return v

    def parse(self, args=None):
        if not self.initialized:
            self.initialize()
        self.opt = self.parser.parse_args(args)
        kv = vars(self.opt)
        print('------------ Options -------------')
        for k, v in sorted(kv.items()):
            print('%s: %s' % (str(k), str(v)))
        print('-------------- End ----------------')
        if self.opt.exp_id == '':
            print('Please set the experimental ID with option --exp_id')
            raise SystemExit(1)
        exp_dir = os.path.join(self.opt.exp_dir, self.opt.exp_id)
        util.mkdirs(exp_dir)
        for name in ('load_prefix_pose', 'load_prefix_sr', 'load_prefix_occ', 'load_prefix_aug'):
            if hasattr(self.opt, name):
                setattr(self.opt, name, self._normalise_prefix(getattr(self.opt, name)))
        with open(os.path.join(exp_dir, 'opt.txt'), 'wt') as f:
            f.write('------------ Options -------------\n')
            for k, v in sorted(kv.items()):
                f.write('%s: %s\n' % (str(k), str(v)))
            f.write('-------------- End ----------------\n')
        return self.opt
