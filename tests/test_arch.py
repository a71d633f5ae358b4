"""The architecture table reproduces the reference checkpoint format exactly."""
import ast
import hashlib
import os

import numpy as np

from calipsync_amd import arch, recipe

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


def _ref_manifest():
    rows = []
    for line in open(os.path.join(GOLDEN, "state_dict_manifest.txt")):
        key, rest = line.split(" ", 1)
        shape = ast.literal_eval(rest[:rest.rindex(")") + 1])
        rows.append((key, tuple(shape), rest[rest.rindex(")") + 1:].strip()))
    return rows


def test_manifest_matches_reference_state_dict():
    ref = _ref_manifest()          # written from the reference's own state_dict()
    mine = [(k, s, d) for k, s, d, _ in arch.manifest()]
    assert len(ref) == 582
    assert mine == ref             # names, shapes, dtypes and registration order


def test_parameter_count():
    n = sum(int(np.prod(s)) for k, s, d, r in arch.manifest()
            if r in ("conv_weight", "linear_weight", "conv_bias", "bn_weight", "bn_bias", "gamma"))
    assert n == 19_793_937         # README.md:22 "19.79 M"; SURVEY §6 exact count


def test_work_figures_match_survey():
    w = arch.work_per_frame()
    assert w["conv_macs"] == 3_927_822_336
    assert w["attention_macs"] == 23_040_000
    assert w["canonical_elems"] == 69_173_248


def test_recipe_is_bit_reproducible(golden, recipe_sd):
    h = hashlib.sha256()
    for k, _s, _d, _r in arch.manifest():
        h.update(np.ascontiguousarray(recipe_sd[k]).tobytes())
    assert bytes(golden["weights_sha256"]) == h.digest()
    x, a = recipe.make_inputs(int(golden["batch"][0]))
    assert bytes(golden["inputs_sha256"]) == hashlib.sha256(x.tobytes() + a.tobytes()).digest()


def test_inputs_are_shard_invariant():
    x, a = recipe.make_inputs(3)
    x2, a2 = recipe.make_inputs_range(1, 2)
    assert np.array_equal(x[1:], x2) and np.array_equal(a[1:], a2)


def test_stagewise_bound_matches_survey():
    """fp32 stage-wise bound 60.7 us/frame = 16.5 k frames/s per GPU, HBM governs inc, down1,
    up3, up4, outc; bf16 57.6-57.8 k frames/s with HBM governing (almost) everywhere."""
    f32 = arch.stagewise_bound(157.3e12, 8e12, 4)
    assert abs(f32["seconds_per_frame"] * 1e6 - 60.7) < 0.05
    assert f32["hbm_governed"] == ["down1", "inc", "outc", "up3", "up4"]
    bf16 = arch.stagewise_bound(2.5e15, 8e12, 2)
    assert 57.5e3 < bf16["frames_per_s"] < 57.9e3 and "mlp" not in bf16["hbm_governed"]
