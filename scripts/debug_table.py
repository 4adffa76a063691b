import os, sys, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_lbr as T
from pokerrl_amd.rl.tabular_agent import make_table_agent_cls
from pokerrl_amd.eval.head_to_head import BatchedHead2Head, H2HArgs, LocalHead2HeadMaster
from pokerrl_amd.game.games import StandardLeduc
table, cfr = T.solved_table(StandardLeduc, None, 10)
try:
    T.check_table_on_device(table); print("probe ok")
except AssertionError as e:
    print("probe FAILED", str(e)[:500])
tmp = tempfile.mkdtemp()
n = 300
t_prof = T.TrainingProfileBase(name="h2h_tab", log_verbose=False, log_export_freq=1, checkpoint_freq=10 ** 9, eval_agent_export_freq=10 ** 9, game_cls=StandardLeduc,
    env_bldr_cls=T.HistoryEnvBuilder, start_chips=None, eval_modes_of_algo=("TABLE", "HASH2"), eval_stack_sizes=None,
    module_args={"env": StandardLeduc.ARGS_CLS(n_seats=2), "h2h": H2HArgs(n_hands=n)}, path_data=tmp)
record = []
m = LocalHead2HeadMaster(t_prof=t_prof, chief_handle=T._Chief(), eval_agent_cls=make_table_agent_cls(T.EvalAgentBase, table, seed=11, record=record))
m.set_modes(["TABLE", "HASH2"])
np.random.seed(99)
want = m.play(stack_size=t_prof.eval_stack_sizes[0])
decks = T.decks_from_record(record[::2], StandardLeduc.get_lut_holder(), 1)
for kinds, tabs in ((("table", "hash"), (table, None)), (("uniform", "hash"), (None, None)), (("hash", "hash"), (None, None))):
    b = BatchedHead2Head(t_prof, kinds=kinds, seeds=(11, 12), tables=tabs)
    got = b.play(n_hands=n, decks=decks)
    print(kinds, "differ from host table agent:", int(np.sum(got != want)), "of", 2 * n, "first", np.flatnonzero(got != want)[:6], got[:10], want[:10])
# a table whose every row is one-hot on CHECK/CALL: the device must then never fold or raise
import copy
t2 = copy.copy(table); t2._dev = None
p = np.zeros_like(table.probs); p[:, 1, :] = 1.0; t2.probs = p
b = BatchedHead2Head(t_prof, kinds=("table", "table"), seeds=(11, 12), tables=(t2, t2))
got = b.play(n_hands=n, decks=decks)
print("always-call table vs itself: winnings set", np.unique(got)[:10], "steps", b.last_stats)
