"""CPU: the PRODUCT's guidance weight schedule (diffusionhandles_amd.guided_stable_diffuser.build_weight_schedule /
StepGuidanceWeightSchedule) against g6 -- the table the reference's own StepGuidanceWeightSchedule produced for the
three schedule types (reference guided_stable_diffuser.py:336-373, 622-665; tools/make_golden.py G6)."""
import numpy as np
import pytest

from diffusionhandles_amd.guided_stable_diffuser import StepGuidanceWeightSchedule, build_weight_schedule


@pytest.mark.parametrize("kind", ["constant", "linear", "quadratic"])
def test_product_schedule_matches_reference_table(golden, kind):
    tab = golden("g6_schedule.npz")[kind]            # [50 timesteps][4 iterations][fg / bg][3 layers], float64
    sched = build_weight_schedule(1.5, 1.25, 38, kind)
    for t in range(50):
        for it in range(4):
            fg, bg = sched(t, it)
            assert fg == tab[t, it, 0].tolist() and bg == tab[t, it, 1].tolist(), (kind, t, it)
    # act0 never carries weight; nothing is weighted from guidance_max_step on
    assert not tab[:, :, :, 0].any() and not tab[38:].any()


def test_schedule_errors_match_reference():
    with pytest.raises(ValueError):
        build_weight_schedule(1.5, 1.25, 38, "cubic")
    with pytest.raises(ValueError):       # fg / bg length mismatch (reference :631-636)
        StepGuidanceWeightSchedule([(0, [1.0] * 3, [1.0] * 2)], [(0, [1.0] * 3, [1.0] * 3)])
    with pytest.raises(ValueError):       # denoising vs optimisation layer count (reference :659-660)
        StepGuidanceWeightSchedule([(0, [1.0] * 3, [1.0] * 3)], [(0, [1.0] * 2, [1.0] * 2)])
    s = StepGuidanceWeightSchedule([(5, [1.0] * 3, [1.0] * 3)], [(0, [1.0] * 3, [1.0] * 3)])
    with pytest.raises(ValueError):       # no entry at or before the query
        s(4, 0)
