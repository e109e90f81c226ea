"""CPU-only tests of the host side: utils, ShipTrack, the update schedule and the SoA packer."""
import os
import types

import numpy as np
import pytest
from conftest import GOLDEN, load_cases


def test_generate_dts_exact():
    """Restates reference tests/test_utils.py:103-149 and the golden values."""
    from track_estimators.utils import generate_dts

    k = np.load(os.path.join(GOLDEN, "kats.npz"))
    assert np.array_equal(generate_dts([10, 5, 3, 2, 1, 10, 10], 2), k["gdts_2"])
    assert np.array_equal(generate_dts([10, 5, 3, 2, 1, 10, 10], 4), k["gdts_4"])
    assert np.array_equal(generate_dts([23.0, 24.0, 25.0], 3), k["gdts_3"])
    assert np.array_equal(generate_dts([10, 5], 2), np.array([5, 5, 2.5, 2.5]))


def test_distances_and_headings():
    """Known answers of reference tests/test_utils.py:11-100 (same tolerances), for both implementations."""
    from track_estimators.utils import geographiclib_distance, geographiclib_heading, haversine_formula, heading

    for dist in (haversine_formula, geographiclib_distance):
        assert np.isclose(dist(-74.0060, 40.7128, -118.2437, 34.0522), 3933.96, rtol=1e-2)
        assert np.isclose(dist(-9.13333, 38.7167, -8.6291, 41.1579), 273.59, rtol=1e-2)
        assert np.isclose(dist(12.5, -33.0, 12.5, -33.0), 0.0)
    for hd in (heading, geographiclib_heading):
        assert np.isclose(hd(-94.581213, 39.099912, -90.200203, 38.627089), 96.51, rtol=1e-3)
        assert np.isclose(hd(3.0, 4.0, 3.0, 4.0), 0.0)


def test_wgs84_inverse_against_cli_fixture_row0():
    """Row 0 of the reference's committed CLI output (examples/cli_example/output_01203823_predictions.txt) is
    noise-free: sog = s12/24 h and cog = azi1 of the WGS84 inverse problem (-30.5,-0.5) -> (-31.5,-3.5)."""
    from track_estimators.utils import geographiclib_distance, geographiclib_heading

    assert geographiclib_distance(-30.5, -0.5, -31.5, -3.5) / 24.0 == 14.578418614021368  # to the last bit
    assert geographiclib_heading(-30.5, -0.5, -31.5, -3.5) == 198.52495095065817


def test_ship_track_from_csv_matches_reference_arrays():
    """ShipTrack.read_csv + calculate_* with the haversine/heading injection vs arrays produced by the reference."""
    from track_estimators.ship_track import ShipTrack
    from track_estimators.utils import haversine_formula, heading

    c = load_cases("ukf_ship_01203823.npz")[0]
    st = ShipTrack(calc_distance_func=haversine_formula, calc_heading_func=heading)
    lat, lon, dts = st.read_csv(os.path.join(GOLDEN, "ship_01203823.csv"), ship_id="01203823", id_col="primary.id")
    z = st.get_measurements(include_sog=True, include_cog=True)
    st.calculate_cog_rate()
    st.calculate_sog_rate()
    assert np.array_equal(z, c["z"]) and np.array_equal(dts, c["dts"])
    assert np.array_equal(st.sog_rate, c["sog_rate"]) and np.array_equal(st.cog_rate, c["cog_rate"])
    assert np.array_equal(lon, c["lon"]) and np.array_equal(lat, c["lat"])
    with pytest.raises(ValueError):
        ShipTrack().read_csv(os.path.join(GOLDEN, "ship_01203823.csv"), ship_id="nope", id_col="primary.id")
    rev = ShipTrack(calc_distance_func=haversine_formula, calc_heading_func=heading)
    rlat, rlon, rdts = rev.read_csv(os.path.join(GOLDEN, "ship_01203823.csv"), ship_id=1203823 if False else "01203823",
                                    id_col="primary.id", reverse=True)
    assert np.array_equal(rlat, lat[::-1]) and np.array_equal(rdts, dts[::-1])


def test_update_schedule_matches_reference_trigger():
    from track_estimators import batch

    for c in load_cases("ukf_synthetic.npz"):
        upd, ridx, t_end = batch.update_schedule(c["dt"], c["dts"])
        assert np.array_equal(upd >= 0, c["fires"])
        assert np.array_equal(upd[upd >= 0], np.arange(1, c["fires"].sum() + 1))
        assert np.array_equal(ridx, np.cumsum(c["fires"]) - c["fires"])
        t = 0
        for d in c["dt"]:
            t += d
        assert t_end == t


def test_pack_tracks_layout_and_ragged_padding():
    from track_estimators import batch

    cs = load_cases("ukf_synthetic.npz")[:4] + load_cases("ukf_synthetic.npz")[7:8]
    tr = [types.SimpleNamespace(z=c["z"], dts=c["dts"], sog_rate=c["sog_rate"], cog_rate=c["cog_rate"]) for c in cs]
    hb = batch.pack_tracks(tr, [c["dt"] for c in cs], [c["x0"] for c in cs], cs[0]["H"], cs[0]["Q"], cs[0]["R"], cs[0]["P0"],
                           bucket_by_length=False)
    assert hb.B == 5 and hb.Nmax == 500 and hb.Tmax == 501 and hb.shared_p0
    assert hb.nsteps.tolist() == [500, 500, 500, 500, 200]
    for b, c in enumerate(cs):
        N, T = len(c["dt"]), c["z"].shape[1]
        assert np.array_equal(hb.dt[:N, b], c["dt"]) and not hb.dt[N:, b].any()
        assert np.array_equal(hb.z[:T, :, b], c["z"].T)
        assert (hb.upd_idx[N:, b] == -1).all()
        fires = hb.upd_idx[:N, b] >= 0
        assert np.array_equal(fires, c["fires"])
        ridx = np.cumsum(fires) - fires
        assert np.array_equal(hb.sog_rate[:N, b], c["sog_rate"][ridx])
    # the non-dyadic track needs its own smoother rates (forward index stalls at 1 update, smoother uses k // 10)
    assert hb.sog_rate_rts is not None
    c = cs[4]
    assert np.array_equal(hb.sog_rate_rts[:200, 4], np.repeat(c["sog_rate"], 10)[:200])
    with pytest.raises(ValueError):
        batch.pack_tracks(tr, [c["dt"] for c in cs], [c["x0"] for c in cs], np.eye(2), cs[0]["Q"], cs[0]["R"], cs[0]["P0"])


def test_length_bucketing_permutation():
    """Ragged batches are laid out longest-first; ``order`` maps batch slots back to the caller's tracks."""
    from track_estimators import batch

    cs = load_cases("ukf_synthetic.npz")
    pick = [7, 0, 9, 1]  # N = 200, 500, 117, 500
    tr = [types.SimpleNamespace(z=cs[i]["z"], dts=cs[i]["dts"], sog_rate=cs[i]["sog_rate"], cog_rate=cs[i]["cog_rate"]) for i in pick]
    hb = batch.pack_tracks(tr, [cs[i]["dt"] for i in pick], [cs[i]["x0"] for i in pick], cs[0]["H"], cs[0]["Q"], cs[0]["R"], cs[0]["P0"])
    assert hb.order.tolist() == [1, 3, 0, 2] and hb.nsteps.tolist() == [500, 500, 200, 117]
    for slot, src in enumerate(hb.order):
        c = cs[pick[src]]
        assert np.array_equal(hb.dt[: len(c["dt"]), slot], c["dt"]) and np.array_equal(hb.x0[:, slot], c["x0"])
    same = batch.pack_tracks(tr[1:2] * 3, [cs[0]["dt"]] * 3, [cs[0]["x0"]] * 3, cs[0]["H"], cs[0]["Q"], cs[0]["R"], cs[0]["P0"])
    assert same.order is None  # nothing to reorder when all lengths are equal


def test_pack_uniform_equals_pack_tracks():
    from track_estimators import batch, synthetic
    from track_estimators.utils import generate_dts

    H, Q, R, P0 = synthetic.example_matrices()
    sb = synthetic.make_batch(5, nobs=13, gap_h=1.0, seed0=3)
    hu = batch.pack_uniform(sb, 4, H, Q, R, P0)
    tr = [types.SimpleNamespace(z=sb.z[i], dts=sb.dts[i], sog_rate=sb.sog_rate[i], cog_rate=sb.cog_rate[i]) for i in range(5)]
    ht = batch.pack_tracks(tr, [generate_dts(sb.dts[i], 4) for i in range(5)], [sb.z[i][:, 0] for i in range(5)], H, Q, R, P0)
    for f in ("dt", "sog_rate", "cog_rate", "upd_idx", "z", "x0", "P0", "nsteps"):
        assert np.array_equal(getattr(hu, f), getattr(ht, f)), f
    assert hu.sog_rate_rts is None and ht.sog_rate_rts is None


def test_synthetic_tracks_are_batch_independent():
    from track_estimators import synthetic

    a = synthetic.make_batch(6, nobs=9, seed0=100)
    b = synthetic.make_batch(3, nobs=9, seed0=103)
    assert np.array_equal(a.z[3:], b.z) and np.array_equal(a.sog_rate[3:], b.sog_rate)
    assert np.abs(a.lat).max() < 80


def test_host_shiptrack_sphere_vs_reference_shiptrack():
    """The package's host ShipTrack against sog / cog / rates / z produced by RUNNING the reference's ShipTrack with its
    sphere pair on 42 ragged tracks incl. duplicate timestamps and the 0/360 seam (tests/golden/track_prep.npz): exact,
    NaN / inf in the same places."""
    import os

    from conftest import GOLDEN
    from track_estimators.ship_track import ShipTrack
    from track_estimators.utils import haversine_formula, heading

    g = np.load(os.path.join(GOLDEN, "track_prep.npz"))
    for b in range(len(g["nobs"])):
        T = int(g["nobs"][b])
        st = ShipTrack(calc_distance_func=haversine_formula, calc_heading_func=heading)
        st.lon, st.lat, st.dts = g["lon"][b, :T].copy(), g["lat"][b, :T].copy(), g["dts"][b, : T - 1].copy()
        with np.errstate(all="ignore"):
            st.calculate_cog()
            st.calculate_sog()
            st.calculate_sog_rate()
            st.calculate_cog_rate()
            z = st.get_measurements(include_sog=True, include_cog=True)
        for k in ("sog", "cog", "sog_rate", "cog_rate"):
            np.testing.assert_array_equal(getattr(st, k), g[k][b, :T], err_msg=k)
        np.testing.assert_array_equal(z, g["z"][b, :, :T])


def test_cli_presmoothing_vs_reference():
    """SOG / COG pre-smoothing (reference cli/main_cli.py:99-109): calculate, moving-average with utils.smooth, then
    get_measurements and the rates -- against arrays produced by the reference on its CLI example's ship."""
    import os

    from conftest import GOLDEN
    from track_estimators.ship_track import ShipTrack
    from track_estimators.utils import haversine_formula, heading, smooth

    g = np.load(os.path.join(GOLDEN, "cli_smooth.npz"))
    st = ShipTrack(calc_distance_func=haversine_formula, calc_heading_func=heading)
    st.read_csv(os.path.join(GOLDEN, "ship_01203823.csv"), ship_id="01203823", id_col="primary.id", lat_col="lat", lon_col="lon")
    st.calculate_cog()
    st.calculate_sog()
    assert np.array_equal(st.sog, g["raw_sog"]) and np.array_equal(st.cog, g["raw_cog"])
    st.sog = smooth(st.sog, int(g["box"]))
    st.cog = smooth(st.cog, int(g["box"]))
    z = st.get_measurements(include_sog=True, include_cog=True)
    st.calculate_cog_rate()
    st.calculate_sog_rate()
    for k in ("sog", "cog", "sog_rate", "cog_rate"):
        np.testing.assert_array_equal(getattr(st, k), g[k], err_msg=k)
    np.testing.assert_array_equal(z, g["z"])
    np.testing.assert_array_equal(smooth(np.arange(7.0) ** 2, 4), g["smooth_even"])


def test_non_symmetric_covariances_are_refused():
    """Q, R, P0 (and every covariance handed to the drop-in class) must be symmetric: the kernels use a symmetric square
    root and an eigenvalue pseudo-inverse, the reference calls sqrtm / pinv on whatever it gets, and on a non-symmetric
    matrix those are different computations -- so the host refuses instead of silently symmetrising."""
    import types

    from track_estimators import batch, synthetic

    H, Q, R, P0 = synthetic.example_matrices()
    sb = synthetic.make_batch(3, nobs=6, gap_h=1.0, seed0=1)
    assert batch.pack_uniform(sb, 2, H, Q, R, P0).B == 3  # the regular matrices pass
    skew = np.zeros((4, 4))
    skew[0, 1], skew[1, 0] = 1e-3, -1e-3
    for name in ("Q", "R", "P0"):
        kw = dict(H=H, Q=Q, R=R, P0=P0)
        kw[name] = kw[name] + skew
        with pytest.raises(ValueError, match=f"{name} must be symmetric"):
            batch.pack_uniform(sb, 2, **kw)
    tr = [types.SimpleNamespace(z=sb.z[b], dts=sb.dts[b], sog_rate=sb.sog_rate[b], cog_rate=sb.cog_rate[b]) for b in range(3)]
    dts = [np.repeat(sb.dts[b] / 2, 2) for b in range(3)]
    x0s = [sb.z[b][:, 0] for b in range(3)]
    with pytest.raises(ValueError, match="P0 must be symmetric"):
        batch.pack_tracks(tr, dts, x0s, H, Q, R, np.stack([P0, P0 + skew, P0]))
    # asymmetry at rounding level is what arithmetic leaves behind: accepted
    tiny = np.zeros((4, 4))
    tiny[0, 1] = 1e-17
    assert batch.pack_uniform(sb, 2, H, Q + tiny, R, P0).B == 3
    # non-finite entries are not a symmetry question (config 4's data produces them downstream): no ValueError here
    batch.require_symmetric(np.full((4, 4), np.nan), "P")


def test_wgs84_legs_near_the_antipode_are_solved_silently():
    """Rounds 1-3 fell back to Vincenty's iteration, which does not converge for nearly antipodal points and warned; the
    product now restates Karney's algorithm (track_estimators/geodesic.py), which the reference's geographiclib implements:
    every leg is an ordinary leg (details: tests/test_geodesic_karney.py)."""
    import warnings

    from track_estimators import utils

    with warnings.catch_warnings():
        warnings.simplefilter("error")
        assert 0 < utils.geographiclib_distance(-30.5, -0.5, 20.0, 40.0) < 2.1e4
        d = utils.geographiclib_distance(0.0, 0.0, 179.7, 0.2)
    assert 19_900 < d < 20_004
