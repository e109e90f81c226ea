"""Round 5: the forward passes of many batches (or windows of a fleet) as ONE scheduled launch of resident waves over
(64-track tile, time slice) items -- include/ste.h 0.3.2 ``ste_ukf_forward_sched_f64``, ``batch.forward_schedule``,
``SmootherPipeline.submit_sequence``.  The reference's batch dimension is its per-ship loop
(/root/reference/examples/example_ukf_rts_smoother_batch.py:19-90); like windows and time slices, a schedule only moves
launch boundaries: every result is the per-batch launch's, bit for bit."""
import ctypes as C

import numpy as np
import pytest

from track_estimators import batch, synthetic
from track_estimators._hip import binding


def _check_schedule(items, ntiles, nslices):
    """Every tile of every window runs exactly its slices, at most one per round; returns the completion round per window."""
    seen = {}
    done = np.zeros(len(ntiles), dtype=int)
    for r in range(items.shape[0]):
        this = set()
        for w, t in items[r]:
            if w < 0:
                continue
            assert 0 <= w < len(ntiles) and 0 <= t < ntiles[w]
            assert (w, t) not in this  # one slice of a tile per round
            this.add((w, t))
            seen[(w, t)] = seen.get((w, t), 0) + 1
            done[w] = r + 1
    assert len(seen) == sum(ntiles)
    for (w, t), c in seen.items():
        assert c == nslices[w], (w, t, c)
    return done


@pytest.mark.parametrize("ntiles,nslices,nwaves", [([157] * 20, [8] * 20, 1024), ([224] * 7, [8] * 7, 1024), ([3, 1, 7], [2, 9, 1], 4),
                                                    ([5], [4], 64), ([40, 40], [3, 5], 16)])
def test_forward_schedule_runs_every_slice_once_and_reaches_the_bound(ntiles, nslices, nwaves):
    items = batch.forward_schedule(ntiles, nslices, nwaves)
    done = _check_schedule(items, ntiles, nslices)
    total = sum(a * b for a, b in zip(ntiles, nslices))
    # McNaughton's bound for chains of unit jobs: total work over the waves, and no shorter than the longest chain
    assert items.shape[0] == max(-(-total // nwaves), max(nslices))
    if len(set(nslices)) == 1:
        assert (np.diff(done) >= 0).all()  # equal windows finish in order
    # a tile that keeps running stays on its wave
    moves = 0
    for r in range(1, items.shape[0]):
        prev = {tuple(v): i for i, v in enumerate(items[r - 1]) if v[0] >= 0}
        for i, v in enumerate(items[r]):
            if v[0] >= 0 and tuple(v) in prev and prev[tuple(v)] != i:
                moves += 1
    assert moves == 0


def test_forward_schedule_with_staggered_deadlines_is_still_complete():
    ntiles, nslices = [157] * 20, [8] * 20
    items = batch.forward_schedule(ntiles, nslices, 1024, stagger=1.0)
    done = _check_schedule(items, ntiles, nslices)
    assert len(set(done.tolist())) > 6  # the windows finish spread out, not in generations


def _minimal_windows(n, B=128, Nmax=128, keep=None):
    arr = (binding.SteUkfBatchF64 * n)()
    m = np.eye(4)
    r = np.diag([0.25, 0.25, 0.0, 0.0])
    h = np.diag([1.0, 1.0, 0.0, 0.0])
    keep.extend([m, r, h])
    for s in arr:
        s.B, s.Nmax, s.Tmax, s.n = B, Nmax, 4, 4
        s.w0, s.wi, s.fan_scale = -1.0 / 3.0, 1.0 / 6.0, 3.0
        s.H, s.Q, s.R = h.ctypes.data, m.ctypes.data, r.ctypes.data
        for name in ("x0", "P0", "dt", "sog_rate", "cog_rate", "upd_idx", "z", "fwd_mean", "fwd_cov", "status"):
            setattr(s, name, 0x1000)
    return arr


def test_schedule_validation_refuses_tables_that_could_not_progress():
    """include/ste.h 0.3.2: the item table is checked on the host before anything is launched -- a tile short of a slice,
    scheduled twice in a round, or outside its window is an argument error (no GPU needed: the device is looked at last)."""
    lib, keep = binding.load(), []
    wins = _minimal_windows(2, keep=keep)  # 2 windows x 2 tiles x 2 slices
    ws = np.zeros(1 << 16, dtype=np.uint8)

    def call(items, nwaves, **kw):
        items = np.ascontiguousarray(items, dtype=np.int32)
        sc = binding.SteFwdSchedF64()
        sc.nwindows, sc.windows, sc.slice_steps = 2, C.addressof(wins), kw.get("slice_steps", 64)
        sc.nwaves, sc.nrounds, sc.items = nwaves, items.shape[0], items.ctypes.data
        sc.host_ws, sc.dev_ws, sc.ws_bytes = ws.ctypes.data, 0x1000, kw.get("ws_bytes", ws.nbytes)
        sc.window_done, sc.error = 0x1000, 0x1000
        rc = lib.ste_ukf_forward_sched_f64(C.byref(sc), None)
        return rc, lib.ste_last_error().decode()

    good = batch.forward_schedule([2, 2], [2, 2], 4)
    assert good.shape == (2, 4, 2)
    rc, msg = call(good, 4)
    assert (rc, msg) == (-3, "no HIP device") or rc == 0 or "nwaves exceeds" in msg  # valid table: only the device can object
    bad = good.copy()
    bad[1, 0] = (-1, 0)
    assert "missing slices" in call(bad, 4)[1]
    bad = good.copy()
    bad[0, 0] = bad[0, 1]
    rc, msg = call(bad, 4)
    assert rc == -1 and ("same round" in msg)
    bad = good.copy()
    bad[0, 0] = (1, 2)
    assert "outside its window" in call(bad, 4)[1]
    bad = good.copy()
    bad[0, 0] = (2, 0)
    assert "does not exist" in call(bad, 4)[1]
    three = np.concatenate([good, good[:1]])
    assert "more slices than it has" in call(three, 4)[1]
    assert "workspace too small" in call(good, 4, ws_bytes=64)[1]
    assert "multiple of STE_SLICE_ALIGN" in call(good, 4, slice_steps=96)[1]
    wins[1].rts_work = 0x1000  # one window leaves smoother rows, the other does not: two different kernels
    assert "must agree" in call(good, 4)[1]
    assert lib.ste_ukf_forward_sched_workspace(2, 2, 4, 2, 4) >= 4 * 64 + 8 * 16 + 16


def test_scheduled_smoother_validation():
    """include/ste.h (STE_VERSION 321): ste_urtss_backward_sched_f64 checks windows and tile list before it launches anything
    (no GPU needed): every window must take the one-kernel smoother, every tile appears once."""
    lib, keep = binding.load(), []
    wins = _minimal_windows(2, B=4224, keep=keep)  # 2 windows x 66 tiles, above the two-kernel smoother's bound
    for w in wins:
        w.rts_work = w.sm_mean = w.sm_cov = 0x1000
    ws = np.zeros(1 << 16, dtype=np.uint8)
    tiles = np.array([(w, t) for w in range(2) for t in range(66)], dtype=np.int32)

    def call(items, **kw):
        items = np.ascontiguousarray(items, dtype=np.int32)
        sc = binding.SteBwdSchedF64()
        sc.nwindows, sc.windows, sc.slice_steps = 2, C.addressof(wins), kw.get("slice_steps", 64)
        sc.nitems, sc.items = items.shape[0], items.ctypes.data
        sc.host_ws, sc.dev_ws, sc.ws_bytes = ws.ctypes.data, 0x1000, kw.get("ws_bytes", ws.nbytes)
        sc.progress, sc.error = 0x1000, 0x1000
        rc = lib.ste_urtss_backward_sched_f64(C.byref(sc), None)
        return rc, lib.ste_last_error().decode()

    assert lib.ste_urtss_backward_sched_workspace(2, 132) <= ws.nbytes
    assert "every tile of every window once" in call(tiles[:-1])[1]
    bad = tiles.copy()
    bad[5] = bad[4]
    assert "appears twice" in call(bad)[1]
    bad = tiles.copy()
    bad[5] = (1, 66)
    assert "does not exist" in call(bad)[1]
    assert "workspace too small" in call(tiles, ws_bytes=64)[1]
    assert "multiple of STE_SLICE_ALIGN" in call(tiles, slice_steps=96)[1]
    wins[1].sog_rate_rts = 0x1000  # one window smooths with rates of its own, the other does not: two kernels
    assert "must agree on smoother rates" in call(tiles)[1]
    wins[1].sog_rate_rts = 0
    wins[1].B = 4096  # the two-kernel smoother's territory: other bits
    small = np.array([(0, t) for t in range(66)] + [(1, t) for t in range(64)], dtype=np.int32)
    assert "one-kernel smoother" in call(small)[1]
    # where the forward launch keeps its per-tile counters: behind its parameter blocks and its item table
    off = lib.ste_ukf_forward_sched_progress_offset(2, 2, 132, 3, 128)
    assert 0 < off < lib.ste_ukf_forward_sched_workspace(2, 2, 132, 3, 128) and off % 16 == 0


def _uniform(B, seed0, nobs=33, substeps=4):
    H, Q, R, P0 = synthetic.example_matrices()
    sb = synthetic.make_batch(B, nobs=nobs, gap_h=1.0, seed0=seed0)
    return sb, batch.pack_uniform(sb, substeps, H, Q, R, P0)


_OUT = ("fwd_mean", "fwd_cov", "sm_mean", "sm_cov", "status", "rts_work")


def _clear(db):
    """Histories and work rows are torch.empty: rows past a short track's end, and the parts of a work row a step does not
    need, are never written -- zero them so that two buffer sets can be compared whole."""
    for n in _OUT:
        getattr(db, n).zero_()
    return db


def _same(a, b):
    import torch

    for n in _OUT:
        x, y = getattr(a, n), getattr(b, n)
        if not torch.equal(torch.nan_to_num(x.double(), nan=1.25e300), torch.nan_to_num(y.double(), nan=1.25e300)):  # NaN is NaN
            return False
    return True


@pytest.mark.gpu
def test_sequence_of_batches_is_the_per_batch_launches_bit_for_bit():
    """Five batches of different sizes and lengths (tiles that do not fill a wave, 1 to 5 time slices) as one scheduled
    launch: histories, smoother work rows and status are those of DeviceBatch.run() on each."""
    import torch

    cases = [(700, 40, 4), (64, 9, 2), (1030, 81, 4), (130, 17, 1), (257, 33, 2)]  # (tracks, observations, sub-steps)
    refs, seqs = [], []
    for i, (B, nobs, sub) in enumerate(cases):
        _, hb = _uniform(B, 777_000 + 5000 * i, nobs=nobs, substeps=sub)
        hb.lanes = 1
        r = _clear(batch.DeviceBatch(hb))
        r.run()
        refs.append(r)
        seqs.append(_clear(batch.DeviceBatch(hb)))
    torch.cuda.synchronize()
    with batch.SmootherPipeline(ntracks=1030) as pipe:
        for stagger in (0.0, 1.0):
            for d in seqs:
                _clear(d)
            pipe.submit_sequence(seqs, stagger=stagger)
            pipe.synchronize()
            for r, d in zip(refs, seqs):
                assert _same(r, d), stagger
        # forward only, and a second use of the same buffer sets behind the first (events, not a device-wide wait)
        for d in seqs:
            _clear(d)
        pipe.submit_sequence(seqs, smooth=False)
        pipe.submit_sequence(seqs)
        pipe.synchronize()
        for r, d in zip(refs, seqs):
            assert _same(r, d)
        with pytest.raises(ValueError, match="histories of its own"):
            pipe.submit_sequence([seqs[0], seqs[0]])
        # workspaces of retired launches are kept and reused: the launches above were never more than two at a time
        assert 1 <= len(pipe._sched_free) <= 2 and not pipe._sched_live
        kept = {h.data_ptr() for _cap, h, _d, _c in pipe._sched_free}
        for d in seqs:
            _clear(d)
        pipe.submit_sequence(seqs)
        assert pipe._sched_live[-1][0].data_ptr() in kept
        pipe.synchronize()
        for r, d in zip(refs, seqs):
            assert _same(r, d)
        # a pageable host workspace (what a caller of the C ABI without page-locked memory hands over) takes the staged copy
        # instead of the upload kernel: same table, same bits
        pipe._sched_free, pipe._sched_pinned = [], False
        for d in seqs:
            _clear(d)
        pipe.submit_sequence(seqs)
        assert not pipe._sched_live[-1][0].is_pinned()
        pipe.synchronize()
        for r, d in zip(refs, seqs):
            assert _same(r, d)


@pytest.mark.gpu
def test_scheduled_fleet_with_clamped_priors_and_recorded_noise_is_bit_identical():
    """A resident ragged fleet whose priors are indefinite (every square root clamped: the first-bad-step word, which a later
    slice reads back and which carries the slice's absolute offset) and that replays recorded noise, in windows through one
    scheduled launch: the bits of the one-launch run, and of run_fleet without a schedule."""
    import torch

    H, Q, R, P0 = synthetic.example_matrices()
    rng = np.random.default_rng(5)
    sb = synthetic.make_batch(900, nobs=60, gap_h=1.0, seed0=31_000)
    import types

    tr, dts = [], []
    for i in range(900):
        T = int(rng.integers(3, 61))
        tr.append(types.SimpleNamespace(z=sb.z[i][:, :T], dts=sb.dts[i][: T - 1], sog_rate=sb.sog_rate[i][:T], cog_rate=sb.cog_rate[i][:T]))
        dts.append(np.repeat(sb.dts[i][: T - 1] / 4, 4))
    Pbad = np.diag([1.0, 1.0, -0.5, 1.0])
    hb = batch.pack_tracks(tr, dts, [t.z[:, 0] for t in tr], H, Q, R, Pbad)
    N = hb.Nmax
    hb.noise_pred = rng.normal(0, 1e-3, (N, 4, hb.B))
    hb.noise_upd = rng.normal(0, 1e-3, (N + 1, 4, hb.B))
    hb.noise_rts = rng.normal(0, 1e-3, (N, 4, hb.B))
    hb.lanes = 1
    ref = _clear(batch.DeviceBatch(hb))
    ref.run()
    torch.cuda.synchronize()
    assert (ref.status_host() & binding.STE_STATUS_CLAMPED).all()
    db = _clear(batch.DeviceBatch(hb))
    res = batch.run_fleet(db, chunk=256, scheduled=True)
    assert len(batch.fleet_windows(hb.B, 256)) == 4 and _same(ref, db)
    db2 = _clear(batch.DeviceBatch(hb))
    batch.run_fleet(db2, chunk=256, scheduled=False)  # one forward launch per window
    assert _same(db, db2) and np.array_equal(res["status"], ref.status_host())


@pytest.mark.gpu
def test_slices_that_change_waves_every_round_hand_over_bit_for_bit():
    """The list scheduler keeps a running tile on its wave, so the in-kernel hand-over between waves (release of the history
    row, work rows, status and first-bad word at agent scope; acquire by the next slice -- possibly on another XCD) is
    exercised only by pre-empted tiles.  Here the table is rotated by a different amount every round: EVERY slice of every
    tile runs on another wave than the one before it, twenty times over, and the results must stay those of plain launches."""
    import torch

    _, hb = _uniform(6000, 4_100_000, nobs=81, substeps=4)  # 94 tiles x 5 slices per batch
    hb.lanes = 1
    ref = _clear(batch.DeviceBatch(hb))
    ref.run()
    torch.cuda.synchronize()
    seqs = [_clear(batch.DeviceBatch(hb)) for _ in range(3)]
    with batch.SmootherPipeline(ntracks=6000) as pipe:
        ntiles, nslices = [94] * 3, [5] * 3
        nwaves = 4 * (pipe.forward_cus - pipe.reserve_cus)
        items = batch.forward_schedule(ntiles, nslices, nwaves).copy()
        for r in range(items.shape[0]):
            items[r] = np.roll(items[r], 131 * (r + 1), axis=0)  # 131 waves = 32 compute units and a bit: another XCD, usually
        moved = 0
        for r in range(1, items.shape[0]):
            prev = {tuple(v): i for i, v in enumerate(items[r - 1]) if v[0] >= 0}
            moved += sum(1 for i, v in enumerate(items[r]) if v[0] >= 0 and prev.get(tuple(v), i) != i)
        assert moved >= 3 * 94 * 4  # every hand-over crosses waves
        pipe._schedules[(tuple(ntiles), tuple(nslices), nwaves, 0.0)] = np.ascontiguousarray(items)
        for rep in range(20):
            for d in seqs:
                _clear(d)
            # batches of 6 000 tracks, nothing in flight: the smoothers are ONE launch, a wave per tile waiting for its own
            # tile's last slice (every other repetition: a gate and a smoother launch per batch) -- same bits
            pipe.submit_sequence(seqs, tile_smoothers=None if rep % 2 == 0 else False)
            assert (len(pipe._sched_live[-1]) > 9) == (rep % 2 == 0)
            pipe.synchronize()
            for d in seqs:
                assert _same(ref, d), rep


@pytest.mark.gpu
def test_scheduled_launches_of_two_pipelines_become_resident_one_at_a_time():
    """Every wave of a scheduled launch may wait for any other, so two launches dispatched together could each hold part of
    the chip.  The launch counts its waves as they begin (``started``, include/ste.h) and the next scheduled launch of the
    process -- here: of another pipeline, with nothing else ordering the two -- waits on its stream for all of them."""
    import torch

    cases = [(9000, 33, 4, 31), (7000, 33, 4, 57)]
    refs, seqs = [], []
    for B, nobs, sub, seed in cases:
        _, hb = _uniform(B, 9_000_000 + seed, nobs=nobs, substeps=sub)
        hb.lanes = 1
        r = _clear(batch.DeviceBatch(hb))
        r.run()
        refs.append(r)
        seqs.append([_clear(batch.DeviceBatch(hb)) for _ in range(8)])  # 8 x 141 / 8 x 110 tiles: more than a chip-full each
    torch.cuda.synchronize()
    # (sequence_only: one forward stream each -- two full pipelines would hold 28 of the device's ~24 hardware queues)
    with batch.SmootherPipeline(ntracks=9000, sequence_only=True) as pa, batch.SmootherPipeline(ntracks=7000, sequence_only=True) as pb:
        nwaves = 4 * (pa.forward_cus - pa.reserve_cus)
        for rep in range(3):
            for group in seqs:
                for d in group:
                    _clear(d)
            pa.submit_sequence(seqs[0])
            first = pa._sched_live[-1]
            pb.submit_sequence(seqs[1])
            # the second launch holds the first one's counters -- its gate reads them -- unless the first had already finished
            # when the second was submitted (a host that was held up for milliseconds between the two calls)
            assert pb._sched_live[-1][8] is first[7] or first[5][0].query()
            pb.submit_sequence(seqs[1][:3] + seqs[0][:0], smooth=False)  # (and a third behind the second, same pipeline)
            pa.synchronize()
            pb.synchronize()
            assert int(first[7][len(seqs[0]) + 1].item()) == nwaves  # every wave of the launch counted itself in
            for r, group in zip(refs, seqs):
                for d in group:
                    assert _same(r, d), rep


@pytest.mark.gpu
def test_pipeline_shrinks_to_the_streams_scheduled_launches_need():
    """SmootherPipeline.shrink hands hardware queues back without building new ones (bench.py --sequence auto goes from the
    7 + 6 + 1 streams of per-step launches to 2 + 6 + 1 and 1 + 1 + 1 that way): same results on what is left."""
    import torch

    _, hb = _uniform(5000, 6_600_000, nobs=33, substeps=4)  # above the two-kernel smoother's bound: tile smoothers apply
    hb.lanes = 1
    ref = _clear(batch.DeviceBatch(hb))
    ref.run()
    torch.cuda.synchronize()
    seqs = [_clear(batch.DeviceBatch(hb)) for _ in range(4)]
    with batch.SmootherPipeline(ntracks=5000) as pipe:
        nf, nb = len(pipe.fwd_streams), len(pipe.bwd_streams)
        assert nf > 2 and nb > 1
        for i, d in enumerate(seqs):
            pipe.submit(d, final=(i == len(seqs) - 1))
        pipe.synchronize()
        assert all(_same(ref, d) for d in seqs)
        pipe.shrink(2, nb)
        assert (len(pipe.fwd_streams), len(pipe.bwd_streams), pipe.buffers_needed) == (2, nb, nb + 3) and len(pipe._raw) == nb + 2
        for d in seqs:
            _clear(d)
        pipe.submit_sequence(seqs[:2], final=False)
        pipe.submit_sequence(seqs[2:])
        pipe.synchronize()
        assert all(_same(ref, d) for d in seqs)
        pipe.shrink(1, 1)
        assert (len(pipe.fwd_streams), len(pipe.bwd_streams), pipe.buffers_needed) == (1, 1, 3) and len(pipe._raw) == 2
        for d in seqs:
            _clear(d)
        pipe.submit_sequence(seqs)
        assert len(pipe._sched_live[-1]) > 9  # one smoother launch, a wave per tile
        pipe.synchronize()
        assert all(_same(ref, d) for d in seqs)
        with pytest.raises(ValueError):
            pipe.shrink(0, 1)
