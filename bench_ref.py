"""The reference's own Python path as timed by scripts/time_reference.py (profiles/reference_cpu.json; the reference does not travel to the GPU
box, so it is timed in the build container, one process, one thread, core count in the file). Every bench*.py quotes its `reference_python_*`
figures from that file through this module: no bench holds a reference number of its own."""
import json
import os

PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "reference_cpu.json")
SOURCE = "profiles/reference_cpu.json (scripts/time_reference.py: /root/reference itself, one process, one thread, the build container's host cores)"


def load():
    try:
        with open(PATH) as f:
            return json.load(f)
    except (OSError, ValueError):
        return None


def figure(section, key, field):
    """e.g. figure("env_step_random_play", "DiscretizedNLHoldem_B_5", "steps_per_s") -> float or None (file or entry missing)"""
    d = load()
    try:
        return float(d[section][key][field])
    except (TypeError, KeyError, ValueError):
        return None


def host():
    d = load()
    return d.get("host") if d else None
