"""Test-side access to the package's host mirror of plonk/evaluation.rs (halo2-pse_amd/evaluation.py): the package
directory is not an importable identifier, so it is loaded through conftest.load_pkg()."""
import importlib

from conftest import load_pkg

load_pkg()
_ev = importlib.import_module("halo2_pse_amd.evaluation")
globals().update({k: v for k, v in vars(_ev).items() if not k.startswith("__")})
