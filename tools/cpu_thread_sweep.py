#!/usr/bin/env python3
"""Thread-count sweep of bench.py's `cpu_baseline` leg (the fp32 PyTorch-CPU oracle's training step, BASELINE configs[1] shape) on this
host: justifies the thread count bench.py uses (SURVEY.md section 8d asks for the host's cores; PyTorch's CPU convolutions stop scaling
well before the core count of a GPU host).  Writes profiles-style JSON to stdout.

    python tools/cpu_thread_sweep.py [--threads 4 8 12 16 24 32 64] [--bs 24] > gpurun_out/cpu_thread_sweep.json"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--threads', type=int, nargs='+', default=[4, 8, 12, 16, 24, 32, 64])
    ap.add_argument('--bs', type=int, default=24, help='images per step of the sample (the network is never shrunk)')
    ap.add_argument('--steps', type=int, default=2)
    args = ap.parse_args()
    from oracle import model as om, step as ostep
    from tests import inputs
    import bench
    net = om.create_hg(2, 1, 16, 256)
    om.deterministic_fill_(net, seed=0)
    opt = ostep.make_optimizer(net)
    img = torch.from_numpy(inputs.images(1, args.bs, 256))
    heat = torch.from_numpy(inputs.heatmaps_from_pts(inputs.heat_pts(2, args.bs, res=64), res=64))
    rows = []
    for t in args.threads:
        if t > (os.cpu_count() or 1):
            continue
        torch.set_num_threads(t)
        ostep.pose_train_step(net, opt, img, heat)
        t0 = time.time()
        for _ in range(args.steps):
            ostep.pose_train_step(net, opt, img, heat)
        dt = (time.time() - t0) / args.steps
        rows.append({'threads': t, 'images_per_sec': round(args.bs / dt, 3), 's_per_step': round(dt, 3)})
        print('# %d threads: %.2f img/s' % (t, args.bs / dt), file=sys.stderr)
    best = max(rows, key=lambda r: r['images_per_sec'])
    json.dump({'host_cores': os.cpu_count(), 'cpu_model': bench.cpu_model(), 'workload': '2-stack chan 256, B=%d, 256x256, oracle/step.py pose_train_step, %d timed steps per point' % (args.bs, args.steps),
               'sweep': rows, 'best_threads': best['threads']}, sys.stdout, indent=1)
    print()


if __name__ == '__main__':
    main()
