"""Deterministic mode under data parallelism on the 256 x 256 networks: two ranks on GPU 0 against the accumulated single-process run, bit for bit
(the body of tests/test_hip_dp.py::test_deterministic_mode_two_ranks_at_256_...).    python tools/dp_determinism_256.py [out prefix]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault('AGF_DP_TEST_FULL', '1'); os.environ.setdefault('AGF_DP_TEST_DETERMINISTIC', '1')
os.environ['PYTHONPATH'] = os.pathsep.join([ROOT, os.path.join(ROOT, 'tests'), os.environ.get('PYTHONPATH', '')])
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch
import torch.multiprocessing as mp
import test_hip_dp as T
from animeface_amd import _lib

if __name__ == '__main__':
    out = sys.argv[1] if len(sys.argv) > 1 else '/tmp/dpdet256'
    assert T.CFG['image_size'] == 256 and T.DETERMINISTIC
    prev = None
    for rep in range(int(os.environ.get('REPEAT', '1'))):
        mp.start_processes(T._worker, args=(2, T._free_port(), out, torch.bfloat16), nprocs=2, join=True, start_method='spawn')
        st = [torch.load(f'{out}.{r}') for r in range(2)]
        if prev is not None:
            nd = sum(int(not torch.equal(st[0][n][k], prev[0][n][k])) for n in ('G', 'D', 'G_ema') for k in st[0][n])
            print(f'2-rank run {rep} vs run {rep - 1}: {nd} tensors differ; losses {st[0]["losses"][1]} / {prev[0]["losses"][1]}', flush=True)
            for r in range(1):
                for ta, tb in zip(st[r].get('trace', []), prev[r].get('trace', [])):
                    if ta[0] == 'params':
                        bad = [x[0] for x, y in zip(ta[2], tb[2]) if x != y]
                        if bad:
                            print(f'   rank {r}: parameters after iteration {ta[1]}: {len(bad)} of {len(ta[2])} differ, first {bad[:5]}', flush=True)
                            break
                        print(f'   rank {r}: parameters after iteration {ta[1]}: identical', flush=True)
                        continue
                    if ta[2:4] != tb[2:4]:
                        which = [i for i, (x, y) in enumerate(zip(ta[4], tb[4])) if x != y]
                        nb = len(st[r]['names'])
                        names = st[r]['names'][ta[0] % nb]
                        print(f'   rank {r}: first bucket exchange whose LOCAL gradients differ: #{ta[0]} = iteration {ta[0] // nb}, bucket {ta[0] % nb} of {nb} ({ta[1]} floats): '
                              f'{[names[i] if i < len(names) else i for i in which[:6]]} of {names}', flush=True)
                        break
        prev = st
    _lib.set_deterministic(True)
    G, G_ema, D, losses = T._single_process(torch.device('cuda', 0), torch.bfloat16)
    bad = []
    for name, mod in (('G', G), ('D', D), ('G_ema', G_ema)):
        for k, v in mod.state_dict().items():
            if not torch.equal(st[0][name][k], st[1][name][k]):
                bad.append(('replicas', name, k))
            if not torch.equal(st[0][name][k], v.detach().cpu()):
                bad.append(('vs one process', name, k, float((st[0][name][k].float() - v.detach().cpu().float()).abs().max())))
    print('losses 2-rank :', st[0]['losses'], st[1]['losses'], flush=True)
    print('losses single :', losses[0], losses[1], flush=True)
    same_losses = [tuple(x) for x in st[0]['losses']] == [tuple(x) for x in losses[0]] and [tuple(x) for x in st[1]['losses']] == [tuple(x) for x in losses[1]]
    print('differing tensors:', len(bad), bad[:8], flush=True)
    assert same_losses, 'losses differ'
    assert not bad
    print('BIT-IDENTICAL: %d tensors' % sum(len(m.state_dict()) for m in (G, D, G_ema)))
