"""CPU: the design study for the next sampling kernel stays true — incremental inversion of a masked autoregressive
layer equals the reference's sweep loop and costs one forward pass of multiply-adds (scripts/design/)."""

import importlib.util
import os


def test_incremental_inverse_study():
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scripts", "design", "incremental_inverse.py")
    spec = importlib.util.spec_from_file_location("incremental_inverse", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    mod.main()  # asserts equality with the sweep loop and the multiply-add count
    diag, bulk = mod.tile_level()  # zuko's own masks, classes aligned to MFMA k-steps: equality asserted inside
    assert diag < 300 and bulk < 1600  # (a dense forward over the same layout is 3128 MFMAs)
