"""pose_adv_aug_amd -- MI355X-native hot path of zhiqiangdon/pose-adv-aug: stacked-hourglass pose
training with adversarial scale/rotation augmentation as hand-written HIP (gfx950) kernels behind the
reference's Python surface (models.create_hg / create_asn, pylib.*, utils.Checkpoint, options).
There is no CPU compute path: importing works anywhere, running needs the HIP library and a GPU."""
from ._lib import PoseAdvError, build, lib, LIB_PATH, EXPORTS, set_dtype, act_dtype, grad_scale  # noqa: F401

__all__ = ['PoseAdvError', 'build', 'lib', 'LIB_PATH', 'EXPORTS', 'set_dtype', 'act_dtype', 'grad_scale']
