"""The reference's OWN unit tests run against this package (container only: they are read from /root/reference, the kernels run on
the SIMT emulator build). Every `PokerRL.<module>` import of a test file resolves to `pokerrl_amd.<module>` (tests/ref_suite_runner.py):
  test/cfr/test_cfr.py            VanillaCFR / CFRPlus / LinearCFR on DiscretizedNLLeduc through CFRBase.iteration()
  test/game/test_tree.py          PublicTree: build, node states against env observations, uniform fill + compute_ev values
  test/game/test_rangeManager.py  PokerRange: range sizes, normalisation, blockers, card removal, save / load
  test/game/test_pokerEnv.py      NoLimitHoldem env with its table-size loops set to 2 seats: chip consistency over random episodes (equal
                                  and random stacks), pot-fraction <-> chips, rewards, get / set state -- 8 of its 12 tests; the other four
                                  build 3-seat tables outright (this package is heads-up only, as the reference's CFR / BR / LBR are)
Not run: test_Deck.py (a class of
the reference's env internals), test_look_up_table.py / test_CppLibPoker.py (run through the reference's own ctypes wrappers in
tests/test_reference_binding.py)."""
import os
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.environ.get("POKERRL_REFERENCE", "/root/reference")
pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "test")), reason="the reference does not travel to the GPU box")

FILES = {"test/cfr/test_cfr.py": 3, "test/game/test_tree.py": None, "test/game/test_rangeManager.py": 15, "test/game/test_pokerEnv.py": 8}


def test_reference_unit_tests_pass_against_this_package():
    sys.path.insert(0, os.path.join(HERE, "emu"))
    import build_emu
    env = dict(os.environ, POKERRL_AMD_LIB=build_emu.build())
    r = subprocess.run([sys.executable, os.path.join(HERE, "ref_suite_runner.py")] + [os.path.join(REF, f) for f in FILES], env=env,
                       cwd=os.path.dirname(HERE), capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    lines = [x for x in r.stdout.splitlines() if " run=" in x]
    assert len(lines) == len(FILES), r.stdout
    for line, (f, n) in zip(lines, FILES.items()):
        assert line.startswith(os.path.basename(f)) and "failures=0 errors=0" in line, line
        ran = int(line.split("run=")[1].split()[0])
        assert ran >= 1 and (n is None or ran == n), line


def test_observation_entry_names_match_the_reference():
    """PokerEnv.obs_idx_dict / obs_parts_idxs_dict (what print_obs and the reference's neural modules index observations by)"""
    sys.path.insert(0, os.path.join(HERE, "golden"))
    import ref_harness
    ref_harness.setup()
    from PokerRL.game import bet_sets as rb
    from PokerRL.game import games as RG
    from pokerrl_amd.game import bet_sets as mb
    from pokerrl_amd.game import games as MG
    for name, bets in (("StandardLeduc", None), ("DiscretizedNLHoldem", "B_5"), ("LimitHoldem", None), ("Flop5Holdem", None)):
        rcls, mcls = getattr(RG, name), getattr(MG, name)
        ra = rcls.ARGS_CLS(n_seats=2, bet_sizes_list_as_frac_of_pot=getattr(rb, bets)) if bets else rcls.ARGS_CLS(n_seats=2)
        ma = mcls.ARGS_CLS(n_seats=2, bet_sizes_list_as_frac_of_pot=getattr(mb, bets)) if bets else mcls.ARGS_CLS(n_seats=2)
        theirs = rcls(env_args=ra, is_evaluating=True, lut_holder=rcls.get_lut_holder())
        ours = mcls(env_args=ma, is_evaluating=True, lut_holder=mcls.get_lut_holder())
        assert dict(ours.obs_idx_dict) == dict(theirs.obs_idx_dict), name
        assert list(ours.obs_idx_dict) == list(theirs.obs_idx_dict), name
        assert ours.obs_parts_idxs_dict == theirs.obs_parts_idxs_dict, name


def test_no_limit_holdem_random_episodes_match_the_reference():
    """Heads-up NoLimitHoldem side by side with the reference for 200 seeds: the stacks a freshly constructed env shows, the stacks after
    reset(), the (type, chips) actions get_random_action draws, the rewards and the final stacks of whole random episodes -- with equal
    and with randomised starting stacks -- and get_fraction_of_pot_raise / get_frac_from_chip_amt along the way, also after the hand is over."""
    sys.path.insert(0, os.path.join(HERE, "golden"))
    import ref_harness
    np = ref_harness.setup()
    from PokerRL.game.games import NoLimitHoldem as RN
    from pokerrl_amd.game.games import NoLimitHoldem as MN

    def mk(cls, randomise):
        args = cls.ARGS_CLS(n_seats=2, stack_randomization_range=((100 - 1000) if randomise else 0, 0), starting_stack_sizes_list=[1000, 1000])
        return cls(env_args=args, is_evaluating=True, lut_holder=cls.get_lut_holder())

    for trial in range(200):
        outs = []
        for cls in (RN, MN):
            np.random.seed(trial)
            env = mk(cls, trial % 2 == 1)
            rec = [[p.stack for p in env.seats]]
            env.reset()
            rec.append([p.stack for p in env.seats])
            done = False
            while not done:
                a = env.get_random_action()
                nxt = env.current_player.seat_id
                rec.append((a, env.get_fraction_of_pot_raise(fraction=0.7, player_that_bets=env.seats[nxt]),
                            round(env.get_frac_from_chip_amt(amt=233, player_that_bets=env.seats[nxt]), 12)))
                _o, r, done, _i = env.step(action=a)
            rec.append(([float(x) for x in r], [p.stack for p in env.seats], env.get_fraction_of_pot_raise(fraction=1.4, player_that_bets=env.seats[0])))
            outs.append(rec)
        assert outs[0] == outs[1], trial
