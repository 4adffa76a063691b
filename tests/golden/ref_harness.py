"""
Golden-vector harness: imports the READ-ONLY reference (/root/reference) in THIS container with test-only shims
 so its outputs can be captured as fixtures under tests/golden/. Never imported by product code
and never run on the GPU box (the reference does not travel).
"""
import os
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.environ.get("POKERRL_REFERENCE", "/root/reference")


def setup():
    sys.dont_write_bytecode = True  # never write into the read-only reference tree
    os.environ.setdefault("OMP_NUM_THREADS", "1")
    os.environ["HOME"] = tempfile.mkdtemp(prefix="prl_home_")
    for p in (os.path.join(HERE, "_stubs"), REF):
        if p not in sys.path:
            sys.path.insert(0, p)
    import numpy as np

    # NumPy-2 shim (LocalLBRWorker.py:394-396, PokerRange.py:94-99): np.delete with float-typed empty/integral index
    _orig_delete = np.delete

    def _delete(arr, obj, axis=None):
        if isinstance(obj, np.ndarray) and obj.dtype.kind == "f":
            obj = obj.astype(np.intp)
        return _orig_delete(arr, obj, axis=axis)

    np.delete = _delete

    # BigLeduc only: int8 + 10000 overflows under NEP 50 (game_rules.py:133-140); same values as NumPy 1
    from PokerRL.game._.rl_env import game_rules

    def _big_rank(self, hand_2d, board_2d):
        if board_2d[0, 0] == hand_2d[0, 0]:
            return 10000 + int(hand_2d[0, 0])
        return int(hand_2d[0, 0])

    game_rules.BigLeducRules.get_hand_rank = _big_rank
    return np
