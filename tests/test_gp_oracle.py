"""Pins oracle/gp_oracle.py to vectors produced by the reference's GPRegression (scikit-learn underneath). CPU only."""
import os

import numpy as np
import pytest
from conftest import GOLDEN

from oracle import gp_oracle as gpo

NAMES = ["ship", "syn130", "syn300"]


def _data(g, name):
    x = np.insert(np.cumsum(g[f"{name}_dts"]), 0, 0)
    y = np.column_stack([g[f"{name}_lon"], g[f"{name}_lat"]])
    return x, y


@pytest.mark.parametrize("name", NAMES)
def test_lml_gradient_and_predict(name):
    g = np.load(os.path.join(GOLDEN, "gp.npz"))
    x, y = _data(g, name)
    for i, th in enumerate(g["thetas"]):
        lml, grad, _, _ = gpo.lml_and_grad(th, x, y)
        assert np.isclose(lml, g[f"{name}_lml"][i], rtol=1e-10, atol=1e-8)
        np.testing.assert_allclose(grad, g[f"{name}_grad"][i], rtol=1e-7, atol=1e-6)
    for i, th in enumerate(g["thetas"][:3]):
        m, sd = gpo.predict(th, x, y, g[f"{name}_tq"])
        np.testing.assert_allclose(m, g[f"{name}_pred"][i], rtol=1e-9, atol=1e-8)
        np.testing.assert_allclose(sd, g[f"{name}_std"][i], rtol=1e-6, atol=1e-7)
