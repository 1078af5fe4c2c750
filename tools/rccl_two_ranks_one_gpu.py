"""Can two RCCL ranks share the one GPU of a test box?  (run the overlapped gradient exchange through RCCL's stream
semantics once before an 8-GPU node appears.)  Spawns two processes on cuda:0 with backend nccl (= RCCL), tries a communicator and one
all-reduce, prints what happened -- the refusal text is the record (profiles/round6_rccl_two_ranks_one_gpu.txt).
    timeout 120 python tools/rccl_two_ranks_one_gpu.py"""
import os
import sys
import tempfile
import traceback

import torch
import torch.multiprocessing as mp


def worker(rank, world, path):
    import torch.distributed as dist
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK='0', NCCL_DEBUG='WARN')
    torch.cuda.set_device(0)
    try:
        dist.init_process_group('nccl', init_method='file://' + path, rank=rank, world_size=world)
        t = torch.full((1 << 20,), float(rank + 1), device='cuda')
        dist.all_reduce(t)
        torch.cuda.synchronize()
        print('rank %d: all_reduce over RCCL with both ranks on cuda:0 -> %s (expected 3.0)' % (rank, float(t[0])), flush=True)
        dist.destroy_process_group()
    except Exception as e:        # the refusal is the result
        print('rank %d: RCCL refused: %s: %s' % (rank, type(e).__name__, str(e).strip().splitlines()[-1][:400]), flush=True)
        traceback.print_exc(limit=2)


if __name__ == '__main__':
    print('torch %s, devices visible: %d' % (torch.__version__, torch.cuda.device_count()), flush=True)
    d = tempfile.mkdtemp()
    mp.spawn(worker, args=(2, os.path.join(d, 'rdv')), nprocs=2, join=True)
