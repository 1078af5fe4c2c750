"""ctypes binding of libposeadv_hip.so (C ABI: include/poseadv.h).

The HIP library IS the product path: if it is missing or fails to load this module raises --
there is no CPU fallback anywhere in the package."""
import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# One storage type per process: bfloat16 (default) or IEEE half (POSEADV_DTYPE=fp16 in the environment BEFORE the first call,
# or set_dtype('fp16')): two builds of the same sources (csrc/build.sh), see csrc/common.h.
DTYPE = os.environ.get('POSEADV_DTYPE', 'bf16').lower()
if DTYPE not in ('bf16', 'fp16'):
    raise ValueError('POSEADV_DTYPE must be bf16 or fp16')
LIB_PATH = os.path.join(_HERE, 'libposeadv_hip.so' if DTYPE == 'bf16' else 'libposeadv_hip_fp16.so')


def set_dtype(name):
    """Select the library build ('bf16' / 'fp16') -- only before the library is loaded."""
    global DTYPE, LIB_PATH
    name = name.lower()
    if name not in ('bf16', 'fp16'):
        raise ValueError('dtype must be bf16 or fp16')
    if _lib is not None and name != DTYPE:
        raise PoseAdvError('the %s library is already loaded in this process' % DTYPE)
    DTYPE = name
    LIB_PATH = os.path.join(_HERE, 'libposeadv_hip.so' if name == 'bf16' else 'libposeadv_hip_fp16.so')


def act_dtype():
    """torch dtype of the library's 16-bit activations (network input NHWC4, operator-level tensors)"""
    return torch.bfloat16 if DTYPE == 'bf16' else torch.float16


def grad_scale():
    """Factor carried by every gradient the loaded library returns (1 for bf16; the fp16 build runs its backward pass on
    scaled gradients); the optimizer divides it out."""
    return float(lib().pa_grad_scale())


class PoseAdvError(RuntimeError):
    pass


_lib = None


def build(verbose=False):
    """Compile the HIP sources for gfx950 in-tree (hipcc cross-compiles without a GPU)."""
    import subprocess
    r = subprocess.run(['bash', os.path.join(_HERE, 'csrc', 'build.sh')], capture_output=True, text=True)
    if verbose or r.returncode != 0:
        print(r.stdout[-4000:])
        print(r.stderr[-4000:])
    if r.returncode != 0:
        raise PoseAdvError('building libposeadv_hip.so failed')
    return LIB_PATH



_vp, _i, _f, _sz, _u64 = C.c_void_p, C.c_int, C.c_float, C.c_size_t, C.c_uint64

_PROTOS = {
    'pa_last_error': (C.c_char_p, []),
    'pa_version': (_i, []),
    'pa_grad_scale': (_f, []),
    'pa_dtype': (_i, []),
    'pa_gaussian_heatmap': (_i, [_vp, _vp, _i, _i, _i, _i, _vp]),
    'pa_weighted_l2': (_i, [_vp, _vp, _vp, _sz, _vp, _vp]),
    'pa_get_preds': (_i, [_vp, _i, _i, _i, _i, _vp, _vp, _vp]),
    'pa_final_preds': (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp]),
    'pa_pck': (_i, [_vp, _vp, _vp, _f, _vp, _i, _f, _vp, _i, _i, _vp, _vp, _vp, _vp]),
    'pa_affine_params': (_i, [_vp, _i, _i, _i, _vp, _vp, _vp]),
    'pa_transform_pts': (_i, [_vp, _vp, _vp, _i, _i, _f, _vp, _vp, _vp]),
    'pa_affine_warp_bilinear': (_i, [_vp, _i, _i, _vp, _vp, _i, _i, _vp, _vp, _vp]),
    'pa_affine_warp_bilinear_sized': (_i, [_vp, _i, _i, _vp, _vp, _vp, _i, _i, _vp, _vp, _vp]),
    'pa_transform_pts_sized': (_i, [_vp, _vp, _vp, _i, _i, _vp, _vp, _vp, _vp]),
    'pa_crop_workspace_bytes': (_sz, [_i, _i, _i, _i]),
    'pa_crop_design_bytes': (_i, [_i, _i, _i, _i, C.POINTER(C.c_double)]),
    'pa_crop': (_i, [_vp, _i, _i, _vp, _vp, _i, _i, _vp, _vp, _vp, _vp, _vp]),
    'pa_flip_lr_nhwc4': (_i, [_vp, _vp, _i, _i, _i, _vp]),
    'pa_flip_tta_merge': (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    'pa_sample_aug': (_i, [_vp, _vp, _vp, _i, _u64, _u64, _i, _vp, _vp]),
    'pa_sample_aug_given': (_i, [_vp, _vp, _vp, _i, _vp, _i, _vp, _vp]),
    'pa_sample_categorical': (_i, [_vp, _i, _i, _u64, _u64, C.c_uint, _vp, _vp, _vp]),
    'pa_rmsprop_step': (_i, [_vp, _vp, _vp, _sz, _f, _f, _f, _f, _vp]),
    'pa_rmsprop_skipped_steps': (_i, [_vp, _vp]),
    'pa_rmsprop_step_state': (_i, [_vp, _vp, _vp, _sz, _f, _f, _f, _f, _vp, _vp]),
    'pa_rmsprop_skipped_steps_state': (_i, [_vp, _vp, _vp]),
    'pa_residual_workspace_bytes': (_sz, [_i, _i, _i, _i]),
    'pa_residual_fwd_bwd': (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp]),
    'pa_conv2d_workspace_bytes': (_sz, [_i, _i, _i, _i, _i, _i]),
    'pa_conv2d': (_i, [_i, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _vp]),
    'pa_nchw_to_nhwc': (_i, [_vp, _vp, _i, _i, _i, _i, _vp]),
    'pa_nhwc_to_nchw': (_i, [_vp, _vp, _i, _i, _i, _i, _vp]),
    'pa_hg_create': (_vp, [_i, _i, _i, _i, _i]),
    'pa_asn_create': (_vp, [_i, _i, _i, _i, _i]),
    'pa_asn_create_dropout': (_vp, [_i, _i, _i]),
    'pa_asn_forward_masks': (_i, [_vp, _vp, _i, _vp]),
    'pa_asn_backward_masks': (_i, [_vp, _vp, _vp]),
    'pa_hg_set_dropout_masks': (_i, [_vp, _vp]),
    'pa_sample_dropout_masks': (_i, [_vp, _i, _i, _i, _u64, _u64, _vp, _vp, _vp, _vp, _vp]),
    'pa_cell_mask': (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    'pa_net_destroy': (None, [_vp]),
    'pa_net_num_tensors': (_i, [_vp]),
    'pa_net_tensor_info': (_i, [_vp, _i, C.c_char_p, _i, C.POINTER(_i), C.POINTER(_i), C.POINTER(_sz), C.POINTER(_sz),
                                C.POINTER(_i)]),
    'pa_net_param_floats': (_sz, [_vp]),
    'pa_net_buffer_floats': (_sz, [_vp]),
    'pa_net_workspace_bytes': (_sz, [_vp]),
    'pa_net_bind': (_i, [_vp, _vp, _vp, _vp, _vp, _vp]),
    'pa_net_prepare_weights': (_i, [_vp]),
    'pa_hg_forward': (_i, [_vp, _vp, _vp, _vp, _i, _vp]),
    'pa_hg_heatmap_nhwc': (_vp, [_vp, _i]),
    'pa_hg_heatmap_nchw': (_i, [_vp, _i, _vp]),
    'pa_hg_backward': (_i, [_vp]),
    'pa_hg_backward_phase': (_i, [_vp, _i]),
    'pa_hg_bucket_range': (_i, [_vp, _i, C.POINTER(_sz), C.POINTER(_sz)]),
    'pa_hg_bucket_wait': (_i, [_vp, _i, _vp]),
    'pa_hg_train_step': (_i, [_vp, _vp, _vp, _i, _i, _vp]),
    'pa_hg_set_loss_total': (_i, [_vp, _vp]),
    'pa_hg_accuracy': (_i, [_vp, _i, _vp, _i, _vp, _vp]),
    'pa_hg_forward_half': (_i, [_vp, _vp, _vp, _i]),
    'pa_asn_forward': (_i, [_vp, _vp, _i, _vp, _vp]),
    'pa_asn_probs': (_vp, [_vp]),
    'pa_asn_backward': (_i, [_vp, _vp, _vp, _vp, _vp]),
    'pa_asn_set_log_eps': (_i, [_vp, _f]),
    'pa_hg_pckh': (_i, [_vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp]),
    'pa_net_meters_async': (_i, [_vp, _i]),
    'pa_net_profile_begin': (_i, [_vp]),
    'pa_net_profile_report': (_i, [_vp, C.POINTER(C.c_double), _i, C.POINTER(_i)]),
    'pa_net_design_bytes': (_i, [_vp, C.POINTER(C.c_double)]),
    'pa_copy_probe': (_i, [_vp, _vp, _sz, _vp]),
    'pa_copy_probe_form': (_i, [_vp, _vp, _sz, _i, _vp]),
    'pa_params_csr': (_i, [_vp, _i, _vp, _vp]),
    'pa_net_set_fin_prologue': (_i, [_vp, _i]),
    'pa_wgrad_group_workspace_bytes': (_sz, [_vp, _i]),
    'pa_wgrad_group': (_i, [_vp, _i, _i, _vp, _vp]),
    'pa_conv2d_time': (_i, [_i] * 9 + [_vp, C.POINTER(C.c_float), _vp]),
    'pa_net_profile_classes': (_i, [_vp, C.POINTER(C.c_int32), _i]),
    'pa_net_set_multi_stream': (_i, [_vp, _i]),
    'pa_hg_debug_tensor': (_i, [_vp, C.c_char_p, _i, _vp, C.POINTER(_i)]),
    'pa_asn_debug_tensor': (_i, [_vp, C.c_char_p, _i, _vp, C.POINTER(_i)]),
}

EXPORTS = sorted(_PROTOS)


def lib():
    """The loaded library; raises PoseAdvError (never falls back) when it is absent."""
    global _lib
    if _lib is None:
        if not os.path.isfile(LIB_PATH):
            raise PoseAdvError('%s not found: run `python -c "import __graft_entry__ as g; g.build()"` '
                               '(there is no CPU fallback)' % LIB_PATH)
        try:
            l = C.CDLL(LIB_PATH)
        except OSError as e:
            raise PoseAdvError('cannot load %s: %s' % (LIB_PATH, e))
        for name, (res, args) in _PROTOS.items():
            fn = getattr(l, name)
            fn.restype = res
            fn.argtypes = args
        if l.pa_dtype() != (0 if DTYPE == 'bf16' else 1):
            raise PoseAdvError('%s is not the %s build' % (LIB_PATH, DTYPE))
        _lib = l
    return _lib


def check(rc, what=''):
    if rc != 0:
        msg = lib().pa_last_error()
        raise PoseAdvError('%s failed (code %d): %s' % (what or 'libposeadv_hip call', rc, msg.decode() if msg else ''))


def ptr(t):
    """Device pointer of a torch tensor (None -> NULL).  Tensors must be contiguous CUDA(HIP) tensors."""
    if t is None:
        return None
    if not t.is_cuda:
        raise PoseAdvError('expected a GPU tensor: the pose-adv-aug hot path runs only on the HIP device')
    if not t.is_contiguous():
        raise PoseAdvError('expected a contiguous tensor')
    return C.c_void_p(t.data_ptr())


def stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def device():
    """The HIP device all tensors of the package live on (one per process: LOCAL_RANK's GPU)."""
    require_gpu()
    return torch.device('cuda', torch.cuda.current_device())


def require_gpu():
    if not torch.cuda.is_available():
        raise PoseAdvError('no HIP device visible: pose_adv_aug_amd has no CPU path')
    lib()
