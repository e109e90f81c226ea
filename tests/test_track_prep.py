"""
Observation preparation on the device (ste_track_prep_f64).

Sphere model (haversine_formula + heading): pinned to arrays produced by RUNNING the reference's ShipTrack
(tests/golden/track_prep.npz, made by tests/golden/make_golden.py ``prep_cases``: 40 ragged random tracks, duplicate
timestamps, the 0/360 seam).  WGS84 model: geographiclib is not installed anywhere this runs, so beyond the one noise-free
number the reference's CLI fixture holds (row 0) the WGS84 results are **parity unpinned**; those tests compare the
device with this package's own host restatement of Karney's algorithm (track_estimators/geodesic.py) and say so in their names.

Floating-point tolerance: the device evaluates the same formulas with its own libm and contracted FMAs; distances and
headings agree to ~1e-13 relative, and the rates -- differences of neighbouring values divided by the gap -- to 1e-9
absolute at the magnitudes of these tracks.  Non-finite values (duplicate timestamps, gap = 0) must match exactly.
"""
import math
import os

import numpy as np
import pytest

from track_estimators import batch
from track_estimators.ship_track import ShipTrack
from track_estimators.utils import generate_dts, haversine_formula, heading

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(__file__), "golden")
FUNCS = {"sphere": dict(calc_distance_func=haversine_formula, calc_heading_func=heading), "wgs84": {}}


def host_track(lon, lat, dts, model):
    st = ShipTrack(**FUNCS[model])
    st.lon, st.lat, st.dts = np.asarray(lon, float), np.asarray(lat, float), np.asarray(dts, float)
    with np.errstate(all="ignore"):
        st.calculate_cog()
        st.calculate_sog()
        st.calculate_sog_rate()
        st.calculate_cog_rate()
        st.get_measurements(include_sog=True, include_cog=True)
    return st


def random_tracks(rng, B, tmin, tmax):
    out = []
    for _ in range(B):
        T = int(rng.integers(tmin, tmax + 1))
        lon = rng.uniform(-170, 170) + np.cumsum(rng.normal(0, 0.3, T))
        lat = rng.uniform(-60, 60) + np.cumsum(rng.normal(0, 0.2, T))
        dts = rng.choice([0.5, 1.0, 6.0, 23.0, 24.0, 25.0], T - 1)
        out.append((lon, lat, dts))
    return out


def compare(res, st, rate_atol=1e-9):
    np.testing.assert_allclose(res["sog"], st.sog, rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(res["cog"], st.cog, rtol=1e-11, atol=1e-10)
    np.testing.assert_allclose(res["sog_rate"], st.sog_rate, rtol=1e-8, atol=rate_atol)
    np.testing.assert_allclose(res["cog_rate"], st.cog_rate, rtol=1e-8, atol=rate_atol)
    np.testing.assert_allclose(res["z"], st.z, rtol=1e-10, atol=1e-10)
    assert res["z"].shape == st.z.shape


def _ref_case(g, b):
    T = int(g["nobs"][b])
    return (g["lon"][b, :T], g["lat"][b, :T], g["dts"][b, : T - 1]), {k: g[k][b, :T] for k in ("sog", "cog", "sog_rate", "cog_rate")}, g["z"][b, :, :T]


def test_prep_sphere_vs_reference_shiptrack():
    """ste_track_prep_f64 (sphere) against sog / cog / rates / z computed by the reference's ShipTrack on 42 tracks in one
    ragged launch; inf / NaN from zero gaps must sit in the same places."""
    g = np.load(os.path.join(GOLDEN, "track_prep.npz"))
    B = len(g["nobs"])
    cases = [_ref_case(g, b) for b in range(B)]
    res = batch.prepare_observations([c[0][0] for c in cases], [c[0][1] for c in cases], [c[0][2] for c in cases], model="sphere")
    nonfinite = 0
    for r, (_, ref, z) in zip(res, cases):
        for k in ("sog", "cog", "sog_rate", "cog_rate"):
            a, b_ = r[k], ref[k]
            np.testing.assert_array_equal(np.isnan(a), np.isnan(b_), err_msg=k)
            np.testing.assert_array_equal(np.isposinf(a), np.isposinf(b_), err_msg=k)
            np.testing.assert_array_equal(np.isneginf(a), np.isneginf(b_), err_msg=k)
            ok = np.isfinite(b_)
            nonfinite += int((~ok).sum())
            tol = dict(rtol=1e-10, atol=1e-10) if k in ("sog", "cog") else dict(rtol=1e-8, atol=1e-9)
            np.testing.assert_allclose(a[ok], b_[ok], err_msg=k, **tol)
        assert r["z"].shape == z.shape
        ok = np.isfinite(z)
        np.testing.assert_array_equal(np.isfinite(r["z"]), ok)
        np.testing.assert_allclose(r["z"][ok], z[ok], rtol=1e-10, atol=1e-10)
    assert nonfinite >= 4


@pytest.mark.parametrize("model", ["sphere", "wgs84"], ids=["sphere", "wgs84-unpinned-beyond-cli-row0"])
def test_prep_ragged_random_device_vs_host_class(model):
    tracks = random_tracks(np.random.default_rng(11), 37, 2, 90)
    res = batch.prepare_observations([t[0] for t in tracks], [t[1] for t in tracks], [t[2] for t in tracks], model=model)
    assert len(res) == len(tracks)
    for r, (lon, lat, dts) in zip(res, tracks):
        compare(r, host_track(lon, lat, dts, model))


def test_prep_wgs84_reference_cli_fixture_row0():
    # the reference's CLI example output pins the WGS84 path: first row of output_01203823_predictions.txt
    st = ShipTrack()
    st.read_csv(os.path.join(GOLDEN, "ship_01203823.csv"), ship_id="01203823", id_col="primary.id", lat_col="lat", lon_col="lon")
    batch.prepare_ship_tracks([st])
    np.testing.assert_allclose(st.z[:, 0], [-30.5, -0.5, 14.578418614021368, 198.52495095065817], rtol=1e-11)
    host = host_track(st.lon, st.lat, st.dts, "wgs84")
    compare({"sog": st.sog, "cog": st.cog, "sog_rate": st.sog_rate, "cog_rate": st.cog_rate, "z": st.z}, host)


@pytest.mark.parametrize("model", ["sphere", "wgs84"], ids=["sphere", "wgs84-unpinned-beyond-cli-row0"])
def test_prep_duplicate_timestamps_and_coincident_points_device_vs_host_class(model):
    # gap = 0 -> distance / 0 (ship_track.py:217): inf for a moved ship, NaN for a stationary one, and the rates
    # built on them; data/modern_ships has thousands of these
    lon = np.array([10.0, 10.2, 10.2, 10.5, 10.5, 10.9, 11.0])
    lat = np.array([50.0, 50.1, 50.1, 50.3, 50.3, 50.2, 50.2])
    dts = np.array([1.0, 0.0, 2.0, 1.0, 0.0, 1.0])
    lon[4] = 10.6  # moved within a zero gap -> inf
    r = batch.prepare_observations([lon], [lat], [dts], model=model)[0]
    st = host_track(lon, lat, dts, model)
    for k in ("sog", "cog", "sog_rate", "cog_rate"):
        a, b = r[k], getattr(st, k)
        np.testing.assert_array_equal(np.isnan(a), np.isnan(b), err_msg=k)
        np.testing.assert_array_equal(np.isposinf(a), np.isposinf(b), err_msg=k)
        np.testing.assert_array_equal(np.isneginf(a), np.isneginf(b), err_msg=k)
        ok = np.isfinite(b)
        np.testing.assert_allclose(a[ok], b[ok], rtol=1e-9, atol=1e-9, err_msg=k)
    assert np.isnan(st.sog).any() and np.isinf(st.sog).any()


def test_prep_rejects_bad_input():
    with pytest.raises(IndexError):
        batch.prepare_observations([[1.0]], [[2.0]], [[]])
    with pytest.raises(ValueError):
        batch.prepare_observations([[1.0, 2.0]], [[2.0, 3.0]], [[1.0]], model="flat-earth")
    st = ShipTrack(calc_distance_func=lambda *a: 0.0)
    st.lon, st.lat, st.dts = np.zeros(3), np.zeros(3), np.ones(2)
    with pytest.raises(ValueError):
        batch.prepare_ship_tracks([st])


def test_prep_wgs84_unpinned_feeds_filter_end_to_end():
    # raw positions -> device preparation -> batched UKF + URTSS, against the same run on host-prepared tracks
    H, Q, R = np.diag([1.0, 1, 0, 0]), np.diag([1e-4, 1e-4, 1e-6, 1e-6]), np.diag([0.25, 0.25, 0, 0])
    P0 = np.eye(4)
    rng = np.random.default_rng(5)
    tracks = []
    for _ in range(12):  # steady ships: ~20 km/h on a slowly turning course, positions jittered by 0.01 deg
        T = int(rng.integers(20, 60))
        dts = rng.choice([1.0, 2.0, 6.0], T - 1)
        course = np.radians(rng.uniform(0, 360) + np.cumsum(rng.normal(0, 2.0, T - 1)))
        lat = rng.uniform(-50, 50) + np.concatenate([[0], np.cumsum(0.18 * dts * np.cos(course))])
        lon = rng.uniform(-150, 150) + np.concatenate([[0], np.cumsum(0.18 * dts * np.sin(course))])
        tracks.append((lon + rng.normal(0, 0.01, T), lat + rng.normal(0, 0.01, T), dts))
    host = [host_track(*t, "wgs84") for t in tracks]
    dev = []
    for lon, lat, dts in tracks:
        st = ShipTrack()
        st.lon, st.lat, st.dts = lon, lat, dts
        dev.append(st)
    batch.prepare_ship_tracks(dev)
    outs = []
    for sts in (host, dev):
        hb = batch.pack_tracks(sts, [generate_dts(st.dts, 2) for st in sts], [st.z[:, 0] for st in sts], H, Q, R, P0)
        outs.append(batch.run_batch(hb))
    assert not outs[0]["status"].any() and not outs[1]["status"].any()
    for b, st in enumerate(host):
        n = outs[0]["nsteps"][b] + 1
        a, c = outs[0]["means_smoothed"][b, :n], outs[1]["means_smoothed"][b, :n]
        assert np.max(np.abs(a - c) / np.maximum(np.abs(a), 1e-12)) < 1e-6


def test_prep_wgs84_solves_nearly_antipodal_legs():
    """Nearly antipodal legs -- which the Vincenty iteration of rounds 1-3 flagged instead of solving -- and meridional,
    equatorial and polar ones: the device's Karney solver agrees with the host's (track_estimators/geodesic.py, checked on
    its own in tests/test_geodesic_karney.py) to 1e-9 km and, where the azimuth is well conditioned, 1e-9 deg; no status
    bit, no warning."""
    import warnings

    from track_estimators import geodesic

    rng = np.random.default_rng(3)
    lons, lats, gaps = [], [], []
    for _ in range(64):
        la, lo = rng.uniform(-80, 80), rng.uniform(-180, 180)
        lons.append(np.array([lo, lo + 180 + rng.uniform(-0.6, 0.6), lo + rng.uniform(-1, 1)]))
        lats.append(np.array([la, -la + rng.uniform(-0.6, 0.6), la + rng.uniform(-1, 1)]))
        gaps.append(np.ones(2))
    lons += [np.array([0.0, 179.7, 179.9]), np.array([5.0, 5.0, 5.0]), np.array([-30.0, 100.0, -100.0]), np.array([0.0, 180.0, 0.0])]
    lats += [np.array([0.0, 0.2, 0.4]), np.array([-40.0, 60.0, 89.9]), np.array([0.0, 0.0, 0.0]), np.array([89.9, 89.9, -89.9])]
    gaps += [np.ones(2)] * 4
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        res = batch.prepare_observations(lons, lats, gaps, model="wgs84")
    assert [r["status"] for r in res] == [0] * len(res)
    for r, lon, lat in zip(res, lons, lats):
        for j in range(2):
            s12, azi1, _, _ = geodesic.inverse(lat[j], lon[j], lat[j + 1], lon[j + 1])
            assert abs(r["sog"][j] - s12 * 1e-3) < 1e-9, (lon, lat, j)
            cond = abs(math.sin(min(s12, 2.0e7 - s12) / 6.4e6))  # azimuths at the antipode move by degrees per micrometre
            assert abs((r["cog"][j] - azi1 + 180.0) % 360.0 - 180.0) * cond < 1e-9, (lon, lat, j)
