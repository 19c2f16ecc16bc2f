import sys, time, numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, './tests')
from fastsk_amd import _native
z = np.load('./tests/golden/tokens_EP300.npz')
tok, off, ntr, nte = z['tokens'].astype(np.int32), z['offsets'].astype(np.int64), int(z['n_train']), int(z['n_test'])
for skip in (False, True, False, True):
    e = _native.Engine(10, 6, skip_test_block=skip)
    best = 1e9
    for _ in range(6):
        t0 = time.perf_counter(); e.compute(tok, off, ntr, nte); best = min(best, time.perf_counter() - t0)
    print('skip', skip, 'ms', best * 1e3); e.close()
