"""Round 6 experiment: per-workgroup phase clocks of prl_k_ebf_random_step (variant build with -DPRL_EB_TIMELINE).
   POKERRL_AMD_LIB=.../libpokerrl_hip_ebtl.so python scripts/r6_env_timeline.py"""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pokerrl_amd import _native
from pokerrl_amd.game import bet_sets, games as G
cls = G.DiscretizedNLHoldem
ea = cls.ARGS_CLS(n_seats=2, starting_stack_sizes_list=[20000, 20000], bet_sizes_list_as_frac_of_pot=bet_sets.B_5)
b = _native.NativeEnvBatch.with_cards(cls.native_game(ea), cls.native_rules(), 1 << 20, deck_seed=11)
b.random_steps_full(20, 1)
b.random_steps_full(1, 2)
L = _native.lib()
buf = np.zeros(4096 * 4, np.uint64)
L.prl_debug_eb_timeline.argtypes = [ctypes.c_void_p]
assert L.prl_debug_eb_timeline(buf.ctypes.data) == 0
t = buf.reshape(4096, 4).astype(np.int64)
t = t[t[:, 0] > 0]  # the persistent grid has fewer than 4096 workgroups; a workgroup's LAST chunk is what it leaves here
t0 = t[:, 0].min()
start, step_end, emit_end, hw = t[:, 0] - t0, t[:, 1] - t0, t[:, 2] - t0, t[:, 3]
print("clock ticks (s_memrealtime / readcyclecounter units); kernel span", emit_end.max())
print("step duration: mean %.0f  p10 %.0f p50 %.0f p90 %.0f" % ((step_end - start).mean(), *np.percentile(step_end - start, [10, 50, 90])))
print("emit duration: mean %.0f  p10 %.0f p50 %.0f p90 %.0f" % ((emit_end - step_end).mean(), *np.percentile(emit_end - step_end, [10, 50, 90])))
print("start times percentiles", np.percentile(start, [0, 10, 25, 50, 75, 90, 100]).astype(int))
# one CU's workgroups in start order
cu = (hw >> 8) & 0xF; se = (hw >> 13) & 0x7; sh = (hw >> 12) & 1; xcc = np.arange(len(t)) % 8
key = xcc * 10000 + se * 1000 + sh * 100 + cu
k0 = key[0]
idx = np.where(key == k0)[0]
idx = idx[np.argsort(start[idx])]
print("workgroups on the CU of workgroup 0 (n=%d): bid, wave slot, start, step_end, emit_end" % len(idx))
for i in idx:
    print("  %5d slot %2d  %8d %8d %8d" % (i, hw[i] & 0xF, start[i], step_end[i], emit_end[i]))
# (the clocks of different XCDs do not share an origin: only one CU's workgroups are put side by side)
