"""times prl_lbr_checkdown_equity before the flop (C(50, 5) boards per range): the call the host LBR worker makes per pre-flop decision and the batched
engine per cache miss.   python scripts/lbr_preflop_equity_timing.py [n_calls] [n_q]"""
import ctypes
import sys
import time

import numpy as np

sys.path.insert(0, __file__.rsplit("/scripts/", 1)[0])
from pokerrl_amd import _native  # noqa: E402
from pokerrl_amd.game.games import DiscretizedNLHoldem  # noqa: E402

n_calls, n_q = int(sys.argv[1]) if len(sys.argv) > 1 else 20, int(sys.argv[2]) if len(sys.argv) > 2 else 3
L = _native.lib()
_native.require_device()
rules = DiscretizedNLHoldem.native_rules()
rng = np.random.RandomState(0)
rg = rng.random_sample((n_q, 1326)).astype(np.float32)
rg /= rg.sum(axis=1, keepdims=True)
out = np.zeros(n_q, np.float32)
hand = np.array([3, 40], np.int8)
board = np.zeros(5, np.int8)
for i in range(n_calls + 1):
    if i == 1:
        t0 = time.perf_counter()
    _native.check(L.prl_lbr_checkdown_equity(ctypes.byref(rules), board.ctypes.data_as(ctypes.c_void_p), 0, hand.ctypes.data_as(ctypes.c_void_p),
                                            rg.ctypes.data_as(ctypes.c_void_p), n_q, out.ctypes.data_as(ctypes.c_void_p)), L)
print("%d calls, %d ranges each: %.1f ms per call; wp %s" % (n_calls, n_q, (time.perf_counter() - t0) * 1e3 / n_calls, out))
