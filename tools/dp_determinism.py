"""Repeatability probe for tests/test_hip_dp.py: the single-process reference against itself and the 2-rank run against it, several times."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch
import torch.multiprocessing as mp
import test_hip_dp as T

def diff(a, b):
    worst, where = 0.0, None
    for k in a:
        d = float((a[k].float().cpu() - b[k].float().cpu()).abs().max())
        if d > worst: worst, where = d, k
    return worst, where

if __name__ == '__main__':
    dev = torch.device('cuda', 0)
    ref = None
    for i in range(int(sys.argv[1]) if len(sys.argv) > 1 else 4):
        G, Ge, D, _ = T._single_process(dev)
        cur = {'G.' + k: v.detach().clone() for k, v in G.state_dict().items()}
        cur.update({'D.' + k: v.detach().clone() for k, v in D.state_dict().items()})
        if ref is None: ref = cur
        print('single vs single', i, diff(cur, ref), flush=True)
    for i in range(int(sys.argv[2]) if len(sys.argv) > 2 else 4):
        out = f'/tmp/dpdet{i}'
        mp.start_processes(T._worker, args=(2, T._free_port(), out), nprocs=2, join=True, start_method='spawn')
        st = torch.load(out + '.0')
        cur = {'G.' + k: v for k, v in st['G'].items()}
        cur.update({'D.' + k: v for k, v in st['D'].items()})
        print('2-rank vs single', i, diff(cur, ref), flush=True)
