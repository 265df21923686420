"""Closed-form known answers (tests/kats.py) for every kernel body on the path: against the oracle here (CPU suite), and
against the HIP kernels through the C ABI under -m gpu. The expected values never come from oracle/hns_oracle.c."""
import numpy as np
import pytest

import kats
from hnanosolver_amd import fields  # noqa: F401

BOXES = {"box24": dict(B=24, origin=(0, 0, 0)), "box16_negative": dict(B=16, origin=(-4104, 4088, -8))}  # second one straddles NanoVDB node borders


def make_box(name):
    return kats.Box(**BOXES[name])


@pytest.mark.parametrize("box_name", list(BOXES))
@pytest.mark.parametrize("case", list(kats.CASES))
def test_oracle_known_answers(case, box_name):
    from oracle_lib import OracleGrid

    box = make_box(box_name)
    kats.CASES[case](OracleGrid(box.leaves), box)


from hip_kernels import HipKernels  # noqa: E402


@pytest.mark.gpu
@pytest.mark.parametrize("box_name", list(BOXES))
@pytest.mark.parametrize("case", list(kats.CASES))
def test_hip_known_answers(case, box_name):
    box = make_box(box_name)
    kats.CASES[case](HipKernels(box.leaves), box)


class FusedSorKernels(HipKernels):
    """rbgs through the production path: one fused (red, black) launch, split back into colours by running it on copies --
    the red result is what a fused iteration leaves on the red voxels of ... no: the fused kernel has no single-colour mode,
    so this engine answers `rbgs(color=0)` with the input and `rbgs(color=1)` with the full iteration applied to the input
    it saw at color 0. Only meaningful for cases that call red then black on the same data."""

    def rbgs(self, div, p, dx, color, omega):
        if color == 0:
            self._p0 = np.array(p, dtype=np.float32)
            return p
        a, b = self._d(self._p0), self.t.zeros(len(p), device="cuda")
        return self._h(self.D.rbgs_iterate(self.grid, self._d(div), a, b, dx, omega, 1))


@pytest.mark.gpu
@pytest.mark.parametrize("form", [("rbgs", "auto"), ("sor_block_lb", "1"), ("sor_block_lb", "2"), ("rbgs", "color")])
def test_fused_sor_forms_known_sweep(form):
    """the closed-form answer through hns_dev_rbgs_iterate, one iteration per call: the one-iteration launch of the 16^3 kernel, the two-launch form behind one-leaf blocks"""
    import hnanosolver_amd as H

    box = make_box("box24")
    H.set_option(*form)
    try:
        kats.kat_rbgs_known_sweep(FusedSorKernels(box.leaves), box)
    finally:
        H.set_option(form[0], None)
