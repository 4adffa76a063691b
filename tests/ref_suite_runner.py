"""Child process of tests/test_reference_suite.py: runs unit-test FILES OF THE REFERENCE (read where they lie under /root/reference,
never copied) with every `PokerRL...` import resolved to the `pokerrl_amd` module of the same name -- the drop-in claim of INTEGRATION.md
section 3 checked by the reference's own tests. Prints one line per file: `<file> run=<n> failures=<n> errors=<n>`.
argv: test files. The library is whatever POKERRL_AMD_LIB / the default resolves to (the CPU suite points it at the emulator build)."""
import importlib
import importlib.util
import os
import sys
import unittest

sys.dont_write_bytecode = True  # nothing is written into the read-only reference tree
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

# reference-internal module paths that have a differently placed twin here
SPECIAL = {"PokerRL.game._.tree.PublicTree": "pokerrl_amd.game.PublicTree"}


class Alias:
    """meta-path finder + loader: `PokerRL.x.y` IS the module object `pokerrl_amd.x.y`"""

    def find_spec(self, name, path=None, target=None):
        if name != "PokerRL" and not name.startswith("PokerRL."):
            return None
        real = SPECIAL.get(name, "pokerrl_amd" + name[len("PokerRL"):])
        try:
            self._module = importlib.import_module(real)
        except ImportError:
            if name in SPECIAL or not any(k.startswith(name + ".") for k in SPECIAL):
                return None
            import types
            self._module = types.ModuleType(name)  # a package on the way to a SPECIAL module
            self._module.__path__ = []
        return importlib.util.spec_from_loader(name, loader=self, origin="alias of " + real, is_package=hasattr(self._module, "__path__"))

    def create_module(self, spec):
        return self._module

    def exec_module(self, module):
        pass


sys.meta_path.insert(0, Alias())
rc = 0
for f in sys.argv[1:]:
    spec = importlib.util.spec_from_file_location("ref_" + os.path.basename(f)[:-3], f)
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    suite = unittest.defaultTestLoader.loadTestsFromModule(m)
    if hasattr(m, "TestPokerEnv"):
        # its tests loop over table sizes MIN_P .. MAX_P (2 .. 6): this package is heads-up only, so the loops run for 2 seats and the
        # four tests that build 3-seat tables outright are left out
        m.TestPokerEnv.MIN_P = m.TestPokerEnv.MAX_P = 2
        three_seats = {"test_action_space_sample", "test_get_current_obs", "test_get_filtered_action_but_change_nothing", "test_sync_deck"}
        keep = unittest.TestSuite()
        for group in suite:
            for t in group:
                if t._testMethodName not in three_seats:
                    keep.addTest(t)
        suite = keep
    res = unittest.TextTestRunner(verbosity=0, stream=open(os.devnull, "w")).run(suite)
    print("%s run=%d failures=%d errors=%d" % (os.path.basename(f), res.testsRun, len(res.failures), len(res.errors)), flush=True)
    for t, tb in res.failures + res.errors:
        print("   ", t, "|", tb.strip().splitlines()[-1][:300], flush=True)
        rc = 1
sys.exit(rc)
