"""
Records the oracle tapes of the GPU suite's multi-street parity checks (tests/oracle_tape.py): the CHECK FUNCTIONS THEMSELVES, called exactly as the GPU
tests call them (their pytest parametrisations are read off the test functions), with PRL_ORACLE_TAPE=record -- the oracle runs, no solver is built, no
GPU is needed; every value the check asks of the oracle goes onto tests/golden/oracle_tapes/<kind>_<key>.npz (big arrays as SHA-256 digests).

    python tests/golden/make_oracle_tapes.py [name filter]

On the GPU box a check whose tape exists replays it instead of running the oracle (the GPU suite's oracle time: ~5 minutes of its 15); a check whose
problem or sequence of questions changed no longer finds / matches its tape and says so.
"""
import itertools
import os
import sys
import time

os.environ["PRL_ORACLE_TAPE"] = "record"
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, ".."))
sys.path.insert(0, os.path.join(HERE, "..", ".."))

TAPED = ["test_gpu_multistreet_limit_holdem_full_betting_vs_oracle", "test_gpu_streets_engine_limit_holdem_full_betting_vs_oracle",
         "test_gpu_streets_engine_other_street_shapes_vs_oracle", "test_gpu_streets_engine_many_outcomes_per_deal_vs_oracle",
         "test_gpu_streets_engine_cfr_plus_with_averaging_delay_vs_oracle", "test_gpu_streets_engine_float32_running_average_opt_in",
         "test_gpu_streets_engine_best_response_of_an_explicit_strategy_vs_oracle", "test_gpu_multistreet_short_stack_run_outs_vs_oracle",
         "test_gpu_all_in_before_the_deal_run_out_vs_oracle", "test_gpu_streets_engine_discretized_nl_holdem_vs_oracle",
         "test_gpu_streets_engine_all_in_run_outs_vs_oracle", "test_gpu_streets_engine_mixed_best_response_and_f32_average_vs_oracle"]


def cases(fn):
    """the cartesian product of a test function's parametrize marks -> keyword dicts"""
    axes = []
    for m in getattr(fn, "pytestmark", []):
        if m.name != "parametrize":
            continue
        names = [n.strip() for n in m.args[0].split(",")] if isinstance(m.args[0], str) else list(m.args[0])
        vals = [v if isinstance(v, (tuple, list)) and len(names) > 1 else (v,) for v in m.args[1]]
        axes.append([dict(zip(names, v)) for v in vals])
    for combo in itertools.product(*axes):
        kw = {}
        for d in combo:
            kw.update(d)
        yield kw


def main(flt=None):
    import test_gpu_parity as T
    from pokerrl_amd import _native
    L = _native.lib()  # host entry points only (tree builder): nothing here touches a device
    for name in TAPED:
        if flt and flt not in name:
            continue
        fn = getattr(T, name)
        for kw in cases(fn):
            t0 = time.time()
            try:
                fn(L, **kw)
            except (AttributeError, TypeError) as e:  # the tail of a test that looks at the solver it did not get
                if "NoneType" not in str(e):
                    raise
            print("%-80s %-60s %.0f s" % (name, kw, time.time() - t0), flush=True)


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else None)
