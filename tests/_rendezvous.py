"""Rendezvous for the multi-process tests without TCP ports: a torch FileStore in a fresh temporary directory
(`file://` init method).  Ports derived from the pid collided between back-to-back spawns (round-2 flake)."""
import os
import tempfile


def file_init_method():
    d = tempfile.mkdtemp(prefix='poseadv_rdzv_')
    return 'file://' + os.path.join(d, 'store')


def set_env(rank, world, init_method, local_rank=None, **extra):
    """the environment stack_hg.init_distributed reads"""
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank if local_rank is None else local_rank),
                      POSEADV_DIST_INIT=init_method, **extra)


def engine_rank_env(rank):
    """How the two-rank ENGINE tests place their ranks: both on the one GPU of the test box with gloo carrying the exchange (default), or
    -- POSEADV_TEST_DIST_BACKEND=nccl on a node with >= 2 GPUs (tools/run_scale.sh) -- one rank per GPU over RCCL."""
    backend = os.environ.get('POSEADV_TEST_DIST_BACKEND', 'gloo')
    return dict(local_rank=rank if backend == 'nccl' else 0, POSEADV_DIST_BACKEND=backend)
