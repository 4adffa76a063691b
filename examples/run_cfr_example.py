"""150 iterations of vanilla CFR on DiscretizedNLLeduc (reference: examples/run_cfr_example.py)."""
from _common import run

from pokerrl_amd.cfr.VanillaCFR import VanillaCFR

if __name__ == "__main__":
    run(VanillaCFR, "CFR_EXAMPLE")
