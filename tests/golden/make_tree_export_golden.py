"""Golden data for PublicTree.get_tree_as_dict (the PokerViz export, PublicTree.py:143-149,313-420): the REFERENCE's tree on
StandardLeduc (stacks 13/13, as test/game/test_tree.py) and DiscretizedNLLeduc + POT_ONLY, uniform strategies + compute_ev.
Stored: SHA-256 of json.dumps(tree dict), node count, and the text blocks of a few nodes. -> tests/golden/tree_export.npz"""
import hashlib
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_harness  # noqa: E402

np = ref_harness.setup()

from PokerRL.game import bet_sets  # noqa: E402
from PokerRL.game._.tree.PublicTree import PublicTree  # noqa: E402
from PokerRL.game.games import DiscretizedNLLeduc, StandardLeduc  # noqa: E402
from PokerRL.game.wrappers import HistoryEnvBuilder  # noqa: E402


def flatten(d, out):
    out.append(d["text"])
    for c in d["children"]:
        flatten(c, out)
    return out


def one(game_cls, args, stack):
    tree = PublicTree(env_bldr=HistoryEnvBuilder(env_cls=game_cls, env_args=args), stack_size=stack, stop_at_street=None)
    tree.build_tree()
    tree.fill_uniform_random()
    tree.compute_ev()
    d = tree.get_tree_as_dict()
    texts = flatten(d, [])
    pick = [0, 1, 2, len(texts) // 3, len(texts) // 2, len(texts) - 1]
    return hashlib.sha256(json.dumps(d).encode()).hexdigest(), len(texts), json.dumps([texts[i] for i in pick]), pick


out = {}
for tag, game_cls, args, stack in [
    ("StandardLeduc", StandardLeduc, StandardLeduc.ARGS_CLS(n_seats=2, starting_stack_sizes_list=[13, 13]), [13, 13]),
    ("DiscretizedNLLeduc", DiscretizedNLLeduc, DiscretizedNLLeduc.ARGS_CLS(n_seats=2, starting_stack_sizes_list=[20000, 20000],
                                                                          bet_sizes_list_as_frac_of_pot=bet_sets.POT_ONLY), [20000, 20000]),
]:
    h, n, sample, pick = one(game_cls, args, stack)
    out[tag + "_sha256"] = np.array(h)
    out[tag + "_n"] = np.array(n)
    out[tag + "_sample"] = np.array(sample)
    out[tag + "_pick"] = np.array(pick)
    print(tag, n, h)
np.savez_compressed(os.path.join(HERE, "tree_export.npz"), **out)
