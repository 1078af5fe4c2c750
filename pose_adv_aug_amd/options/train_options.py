"""options/train_options.py of the reference (same flags and defaults)."""
from .base_options import BaseOptions, _bool


class TrainOptions(BaseOptions):
    def initialize(self):
        BaseOptions.initialize(self)
        p = self.parser
        p.add_argument('--lr', type=float, default=2.5e-4, help='initial learning rate')
        p.add_argument('--agent_lr', type=float, default=5.0e-5, help='initial agent learning rate')
        p.add_argument('--bs', type=int, default=12, help='mini-batch size')
        p.add_argument('--load_checkpoint', type=_bool, default=False, help='resume from checkpoint model')
        p.add_argument('--load_checkpoint_pose', type=_bool, default=False, help='use checkpoint model')
        p.add_argument('--load_checkpoint_asn', type=_bool, default=False, help='use checkpoint model')
        p.add_argument('--load_prefix_pose', type=str, default='', help='checkpoint name for resuming')
        p.add_argument('--load_prefix_sr', type=str, default='', help='checkpoint name for loading sr agent')
        p.add_argument('--load_prefix_occ', type=str, default='', help='checkpoint name for loading occ agent')
        p.add_argument('--load_prefix_aug', type=str, default='', help='checkpoint name')
        p.add_argument('--occ_dir', type=str, default='occ-dir', help='occlusion dir')
        p.add_argument('--sr_dir', type=str, default='sr-dir', help='sr dir')
        p.add_argument('--joint_dir', type=str, default='joint', help='model dir for joint training')
        p.add_argument('--nEpochs', type=int, default=100, help='number of total training epochs to run')
        p.add_argument('--best_pckh', type=float, default=0., help='best result until now')
        p.add_argument('--train_list', type=str, default='train_list.txt', help='train image list')
        p.add_argument('--val_list', type=str, default='val_list.txt', help='validation image list')
        p.add_argument('--print_freq', type=int, default=10, help='print log every n iterations')
        p.add_argument('--display_freq', type=int, default=10, help='display figures every n iterations')
        p.add_argument('--pose_gpu_id', type=int, default=0, help='gpu ids')
        p.add_argument('--asn_gpu_id', type=int, default=1, help='gpu ids')
