"""CPU suite: the host side of the plugin surface (PublicTree facade, CFR classes, LocalBRMaster, tree export, batched agent
queries) exercised without a GPU -- tests/test_gpu_plugin_surface.py re-run in a child process whose library is the SIMT
emulator build of the same kernel sources (tests/emu, test infrastructure; the package only loads it because POKERRL_AMD_LIB
points at it here). The GPU suite runs the same file against the product library."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))


def test_plugin_surface_on_the_emulator():
    sys.path.insert(0, os.path.join(HERE, "emu"))
    import build_emu
    env = dict(os.environ, POKERRL_AMD_LIB=build_emu.build())
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(HERE, "test_gpu_plugin_surface.py"), "-q", "-x", "-m", "gpu",
                        "-p", "no:cacheprovider"], env=env, cwd=os.path.dirname(HERE), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert " passed" in r.stdout and "failed" not in r.stdout, r.stdout[-1000:]
