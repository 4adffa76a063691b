"""
Drop-in proof at the reference's OWN boundary: PokerRL's ctypes wrappers
CppHandeval / CppLibHoldemLuts are instantiated UNMODIFIED except for the library path (the one-line change of INTEGRATION.md,
here applied by substituting `path_to_dll` in CppWrapper.__init__), so every call below goes reference Python -> ctypes ->
libpokerrl_hip.so. Then
  * the reference's own tests of that boundary (test/game/test_look_up_table.py, test_CppLibPoker.py) are run against it,
  * LutHolderHoldem's tables and 2000 scalar hand ranks are compared with what the reference's binary lib_*.so return.
Runs only where /root/reference exists (this container); the `-m gpu` suite checks the batched legacy symbol with the same
row-pointer marshalling (test_gpu_parity.py::test_gpu_legacy_batched_symbol_row_pointers).
"""
import os
import sys
import unittest

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.environ.get("POKERRL_REFERENCE", "/root/reference")
pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "PokerRL")), reason="the reference does not travel to the GPU box")


@pytest.fixture(scope="module")
def ref():
    sys.path.insert(0, os.path.join(HERE, "golden"))
    import ref_harness
    ref_harness.setup()
    from PokerRL._ import CppWrapper as cw
    from pokerrl_amd import _native
    state = {"redirect": False, "loaded": []}
    orig_init = cw.CppWrapper.__init__

    def init(self, path_to_dll):  # the path is the only thing a maintainer changes
        if state["redirect"]:
            path_to_dll = _native.LIB_PATH
        state["loaded"].append(path_to_dll)
        orig_init(self, path_to_dll)

    cw.CppWrapper.__init__ = init
    yield state
    cw.CppWrapper.__init__ = orig_init


def test_reference_wrappers_load_libpokerrl_hip_and_agree_with_the_reference_binaries(ref):
    from PokerRL.game._.cpp_wrappers.CppHandeval import CppHandeval
    from PokerRL.game._.cpp_wrappers.CppLUT import CppLibHoldemLuts
    from pokerrl_amd import _native
    n_boards, n_out = {0: 1, 1: 22100, 2: 270725, 3: 2598960}, {0: 0, 1: 3, 2: 4, 3: 5}  # look_up_table.py:58-75
    ref["redirect"] = False
    theirs_h, theirs_l = CppHandeval(), CppLibHoldemLuts(n_boards_lut=n_boards, n_cards_out_lut=n_out)
    ref["redirect"] = True
    ref["loaded"].clear()
    ours_h, ours_l = CppHandeval(), CppLibHoldemLuts(n_boards_lut=n_boards, n_cards_out_lut=n_out)  # binds all 7 symbols (CppLUT.py:21-34)
    assert ref["loaded"] == [_native.LIB_PATH, _native.LIB_PATH]
    # LUTs through the reference's own marshalling (CppLUT.py:38-47)
    assert np.array_equal(ours_l.get_idx_2_hole_card_lut(), theirs_l.get_idx_2_hole_card_lut())
    assert np.array_equal(ours_l.get_hole_card_2_idx_lut(), theirs_l.get_hole_card_2_idx_lut())
    for c in range(52):
        c2 = ours_l.get_2d_card(c)
        assert np.array_equal(c2, theirs_l.get_2d_card(c))
        assert ours_l.get_1d_card(card_2d=np.asarray(c2, dtype=np.int8)) == c == theirs_l.get_1d_card(card_2d=np.asarray(c2, dtype=np.int8))
    # the board tables the reference binds but never calls (its own binary crashes in them): ascending lexicographic boards
    import itertools
    flop = ours_l.get_idx_2_flop_lut()
    assert flop.shape == (22100, 3) and np.array_equal(flop, np.array(list(itertools.combinations(range(52), 3)), np.int8))
    turn = ours_l.get_idx_2_turn_lut()
    assert turn.shape == (270725, 4) and np.array_equal(turn[[0, 1, -1]], np.array([[0, 1, 2, 3], [0, 1, 2, 4], [48, 49, 50, 51]], np.int8))
    assert np.all(turn[:, :-1] < turn[:, 1:]) and len({tuple(r) for r in turn[::997].tolist()}) == len(turn[::997])
    # scalar evaluator (CppHandeval.py:34-43) on 2000 seeded 7-card deals
    rng = np.random.RandomState(11)
    for _ in range(2000):
        cards = rng.choice(52, 7, replace=False)
        c2d = np.stack([cards // 4, cards % 4], axis=1).astype(np.int8)
        h, b = np.ascontiguousarray(c2d[:2]), np.ascontiguousarray(c2d[2:])
        assert ours_h.get_hand_rank_52_holdem(hand_2d=h, board_2d=b) == theirs_h.get_hand_rank_52_holdem(hand_2d=h, board_2d=b)


@pytest.mark.parametrize("module", ["test_look_up_table", "test_CppLibPoker"])
def test_reference_unit_tests_pass_against_libpokerrl_hip(ref, module):
    """the reference's own unittest files for this boundary, with its wrappers bound to our library"""
    ref["redirect"] = True
    ref["loaded"].clear()
    sys.path.insert(0, os.path.join(REF, "test", "game"))
    for m in [k for k in sys.modules if k.startswith("PokerRL.game._.look_up_table") or k.startswith("PokerRL.game._.cpp_wrappers")]:
        del sys.modules[m]  # LUT holders built earlier were bound to the reference's binaries: rebuild them through ours
    sys.modules.pop(module, None)
    suite = unittest.defaultTestLoader.loadTestsFromName(module)
    assert suite.countTestCases() > 0
    res = unittest.TextTestRunner(verbosity=0).run(suite)
    assert res.wasSuccessful(), (res.failures, res.errors)
    from pokerrl_amd import _native
    assert ref["loaded"] and all(p == _native.LIB_PATH for p in ref["loaded"]), ref["loaded"]
