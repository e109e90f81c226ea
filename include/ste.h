/*
 * ste.h — C ABI of the MI355X-native batched UKF + URTSS path ("ship-track-estimators", ste).
 *
 * The reference (NOC-OI/ship-track-estimators) is pure Python and has no FFI of its own; its boundary for this path
 * is the Python API.  Each entry point below is the batched, device-resident counterpart of one
 * reference method and is what a reference-side binding (ctypes) would load:
 *
 *   ste_ukf_forward_f64      KalmanFilterBase.run            src/track_estimators/kalman_filters/kalman_filter.py:36-117
 *                            (+ UnscentedKalmanFilter.predict / .update, kalman_filters/unscented.py:144-265)
 *   ste_urtss_backward_f64   KalmanFilterBase.run_rts_smoother kalman_filter.py:119-137
 *                            (+ UnscentedKalmanFilter.rts_step, unscented.py:267-351)
 *   ste_ukf_urtss_f64        both, back to back on one stream (examples/example_ukf_rts_smoother_batch.py:75-87)
 *   ste_ukf_predict_f64      UnscentedKalmanFilter.predict (one step, many filters)   unscented.py:144-207
 *   ste_ukf_update_f64       UnscentedKalmanFilter.update  (one step, many filters)   unscented.py:209-265
 *   ste_geodetic_dynamics_f64  geodetic_dynamics              kalman_filters/non_linear_process.py:6-85
 *   ste_sigma_points_f64     UnscentedKalmanFilter.compute_sigma_points  unscented.py:76-107
 *   ste_track_prep_f64       ShipTrack.calculate_sog / _cog / _sog_rate / _cog_rate / get_measurements
 *                            src/track_estimators/ship_track.py:197-338 (distance / heading: utils.py:9-147)
 *
 * Conventions
 *   - plain C, no torch / HIP types in signatures; `stream` is a hipStream_t passed as void* (NULL = default stream).
 *   - every array pointer is a DEVICE pointer unless marked HOST; the caller owns every buffer, the library never
 *     allocates outputs or frees inputs.
 *   - calls are asynchronous on `stream`; no global state except a thread-local error string (launch shape and lane
 *     mapping are per call: flags / tuning of the batch struct).
 *   - return 0 on success, a negative STE_E* code on argument / launch errors (message via ste_last_error()).
 *   - per-track numerical trouble never aborts a batch: it is reported in status[] (mirrors the batch example's
 *     try/except/continue, example_ukf_rts_smoother_batch.py:73-90).
 *   - state order [lon deg, lat deg, speed km/h, heading deg], time in hours (non_linear_process.py:54-57).
 *
 * Data layout in HBM: structure-of-arrays with the TRACK index fastest, so that consecutive lanes (= consecutive
 * tracks) read and write consecutive 8-byte words:
 *      per-step scalar   a[k][t]            -> a[k*B + t]
 *      per-step vector   v[k][c][t]         -> v[(k*4 + c)*B + t]
 *      per-step matrix   M[k][r][c][t]      -> M[(k*16 + r*4 + c)*B + t]
 */
#ifndef STE_H
#define STE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define STE_VERSION 321 /* 0.3.2: the forward passes of many windows as one scheduled launch (ste_ukf_forward_sched_f64,
                           ste_stream_wait_counter); 321: their smoothers as one launch too (ste_urtss_backward_sched_f64).  0.3.1: track_stride (windows of a resident fleet), sm_pos, forward pass in
                           time slices (step_begin / step_end).  0.3.0: rts_work rows of 30 doubles (+ B at the end); sigma
                           weights sum to one */

/* error codes */
#define STE_OK 0
#define STE_EINVAL (-1)   /* bad argument (NULL pointer, B <= 0, n != 4 ...) */
#define STE_ELAUNCH (-2)  /* HIP launch / runtime error */
#define STE_ENOGPU (-3)   /* no HIP device visible */

/* ste_ukf_batch_f64.flags */
#define STE_FLAG_SHARED_P0 0x1u        /* P0 is one 4x4 matrix [16] shared by all tracks (else [16][B]) */
#define STE_FLAG_NO_INITIAL_UPDATE 0x2u /* skip the update with z[:,0] that run() performs before the first predict */
#define STE_FLAG_LANES_1 0x10u /* forward pass with one lane per track for this call (default: chosen by batch size) */
#define STE_FLAG_LANES_4 0x20u /* forward pass with one DPP quad (4 lanes) per track for this call */
#define STE_FLAG_PACKED_COV 0x40u /* fwd_cov and sm_cov hold upper triangles, [Nmax+1][10][B] (row-major: 00 01 02 03 11 12 13
                                     22 23 33), instead of full matrices [Nmax+1][16][B]: the covariances are symmetric by
                                     construction, and 96 of the 828 bytes the two passes move per track-step go */
#define STE_FLAG_ROBUST 0x4u /* opt-in Mahalanobis robustification of every update (check_robustness, unscented.py:353-387;
                                the reference ships with its call site commented out, :228) */

#define STE_RTS_WORK_ROWS 30 /* doubles per (step, track) of ste_ukf_batch_f64.rts_work (+ one row at its end) */
#define STE_SLICE_ALIGN 64   /* time slices of the forward pass start and end on multiples of this many steps */

/* status[] bits (per track) */
#define STE_STATUS_NAN 0x1        /* a non-finite value reached the state or covariance */
#define STE_STATUS_CLAMPED 0x2    /* sigma fan: negative eigenvalue clamped to 0 (sqrtm went complex in the reference) */
#define STE_STATUS_NOCONV 0x4     /* Jacobi eigen-solve hit its sweep cap */
#define STE_STATUS_ROBUST_CAP 0x8 /* robust update: criterion still above chi_alpha after robust_max_iter rescalings */
#define STE_STATUS_HOST_INDEX 0x10 /* set by host-side packers, never by the library: the reference would raise IndexError
                                      for this track (update index past its last observation, kalman_filter.py:105) */
#define STE_STATUS_BAD_INDEX 0x20 /* upd_idx[k] >= Tmax at some step: that update was skipped (precondition below) */

/*
 * One batch of B independent tracks, padded to Nmax filter steps and Tmax observations.
 * n must be 4 (the reference hard-codes index 3 as the heading, unscented.py:250).
 */
typedef struct ste_ukf_batch_f64 {
    int32_t B;     /* number of tracks */
    int32_t Nmax;  /* padded number of filter steps (len(dt) in run()) */
    int32_t Tmax;  /* padded number of observations (columns of ShipTrack.z) */
    int32_t n;     /* state dimension, must be 4 */
    uint32_t flags;
    int32_t tuning; /* 0 = defaults.  Bit 8 (0x100): every smoother gain by the eigenvalue route (default: only where P_b is close
                       to singular).  Bits 9 / 10 (0x200 / 0x400): ste_urtss_backward_f64 in its two-kernel / one-kernel form
                       whatever the batch size (default: two kernels -- all gains at once, then a lean recurrence -- up to 4 096
                       tracks, where the smoother is a few waves running a latency chain; one kernel above; same bits either way).
                       Bit 11 (0x800): the lean recurrence of the two-kernel form with a lane per track instead of a DPP quad per
                       track (same bits; measurements).  Must not change between the backward calls made on one forward result.
                       Other bits are ignored */

    /* sigma-fan constants, computed by the host exactly as unscented.py:95,125,132 does (HOST values) */
    double fan_scale; /* n / (1 - W0) */
    double w0;        /* 1 - n/3 */
    double wi;        /* (1 - W0) / (2n) */

    /* shared 4x4 matrices, row-major, HOST pointers read at call time (unscented.py:55-61) */
    const double* H;
    const double* Q;
    const double* R;

    /* per-track inputs (device) */
    const int32_t* nsteps; /* [B] real step count of each track (<= Nmax); NULL = every track has Nmax steps */
    const double* x0;      /* [4][B]  prior mean */
    const double* P0;      /* [16][B] prior covariance, or [16] with STE_FLAG_SHARED_P0 */

    /* per-step inputs (device) */
    const double* dt;           /* [Nmax][B]  kalman_filter.py:88 */
    const double* sog_rate;     /* [Nmax][B]  rate used by predict at step k  (kalman_filter.py:93) */
    const double* cog_rate;     /* [Nmax][B]                                   (kalman_filter.py:94) */
    const double* sog_rate_rts; /* [Nmax][B]  rate used by the smoother at step k (unscented.py:287-292,310); NULL = sog_rate */
    const double* cog_rate_rts; /* [Nmax][B]  NULL = cog_rate */
    const int32_t* upd_idx;     /* [Nmax][B]  observation column consumed after step k, -1 = no update (kalman_filter.py:101-108);
                                   must be < Tmax: a larger value skips that update and sets STE_STATUS_BAD_INDEX */
    const double* z;            /* [Tmax][4][B] measurement matrix columns (ship_track.py:324-336) */

    /* recorded noise draws, already scaled; NULL = zero noise (unscented.py:198,232,320) */
    const double* noise_pred; /* [Nmax][4][B]   added to the predicted mean of step k */
    const double* noise_upd;  /* [Nmax+1][4][B] row 0: initial update; row k+1: update after step k */
    const double* noise_rts;  /* [Nmax][4][B]   added to the back-predicted mean of step k */

    /* outputs (device) */
    double* fwd_mean; /* [Nmax+1][4][B]  row 0 = prior (kalman_filter.py:76) */
    double* fwd_cov;  /* [Nmax+1][16][B]; [Nmax+1][10][B] with STE_FLAG_PACKED_COV */
    double* sm_mean;  /* [Nmax+1][4][B]  smoothed; row nsteps = filtered row nsteps */
    double* sm_cov;   /* [Nmax+1][16][B]; [Nmax+1][10][B] with STE_FLAG_PACKED_COV */
    int32_t* status;  /* [B] OR-ed STE_STATUS_* bits; the forward pass overwrites, the backward pass ORs */

    /*
     * Optional workspace of (Nmax * STE_RTS_WORK_ROWS + 1) * B doubles (device), caller-owned.  When it is non-NULL
     * ste_ukf_forward_f64 also evaluates what the smoother's step k needs of the sigma
     * fan of the filtered state of step k (unscented.py:297-330: back-prediction x_b, P_b, cross-covariance D) while it has
     * that fan in registers, and stores it here: row k = columns 0-1 of D (8) | x_b (4) | P_b upper triangle (10) |
     * columns 2-3 of D (8), each [B].  x_b and P_b are written only for the steps where they do not follow from rows k and
     * k + 1 of the filtered history (steps followed by an update, row 0 of a run that starts with one, runs with recorded
     * noise); columns 2-3 of D only at and after a track's first clamped / unconverged square root (elsewhere they are
     * 2 wi fan_scale times columns 2-3 of the filtered covariance); the last B words hold that step index per track.
     * ste_urtss_backward_f64 on the same batch then forms the gains K = D pinv(P_b) (:333) and runs the recurrence
     * (:337-349); it may be called again on the same forward result (the one-kernel form only reads the workspace; the
     * two-kernel form of small batches rewrites rows into gains once and marks the workspace as such).  Results are those
     * of the stand-alone smoother to rounding.  Smoother rates of its own (sog_rate_rts / cog_rate_rts) are no obstacle:
     * speed and heading pass through the process model as x + rate * dt, so they move x_b[2:4] -- and through it P_b,
     * which is taken about the filtered mean -- by a known amount and leave D alone.
     * NULL = the smoother recomputes everything from fwd_mean / fwd_cov (required when the forward history was not
     * produced by ste_ukf_forward_f64 on this batch).
     */
    double* rts_work;

    /* robust update (STE_FLAG_ROBUST): threshold chi_alpha (the reference hard-codes 50) and iteration cap (0 = 50) */
    double chi_alpha;
    int32_t robust_max_iter;
    int32_t reserved2;

    /* ---- 0.3.1 ---------------------------------------------------------------------------------------------------
     * A WINDOW of a larger resident batch ("fleet"): every per-track array above and below is addressed
     *      a[row * track_stride + t],   0 <= t < B,
     * i.e. the pointers name track 0 of the window inside arrays whose rows hold track_stride tracks.  0 = B (a batch of
     * its own).  The reference's batch dimension is its per-ship loop (examples/example_ukf_rts_smoother_batch.py:19-90);
     * windows are how a fleet of any size goes through the GPU in chip-sized pieces without copying or re-packing
     * (track_estimators.batch.run_fleet).  nsteps and status are [B] and simply point at the window's first entry; the
     * last row of rts_work (first bad step per track) sits after Nmax * STE_RTS_WORK_ROWS rows of track_stride. */
    int64_t track_stride;

    /* Optional output [Nmax+1][2][track_stride]: smoothed longitude / latitude, written by the smoother beside sm_mean
     * (rows 0 .. nsteps of every track; rows past a short track's end are left alone).  This is the tensor
     * BASELINE configs[2] exchanges between GPUs: with it the all-gather sends the smoother's own output. */
    double* sm_pos;

    /* The forward pass in time slices: ste_ukf_forward_f64 runs steps [step_begin, step_end) of every track (step_end = 0
     * means Nmax).  Both must be multiples of STE_SLICE_ALIGN (or 0 / Nmax); a call with step_begin = 0 starts from the
     * prior, a later one from history row step_begin, which the previous call left -- slices must therefore be issued in
     * order (one stream, or the caller's own ordering).  Histories, work rows and status are bit-identical to those of one
     * call over [0, Nmax): at multiples of STE_SLICE_ALIGN the warm start of the fan's eigen-solve restarts in a whole
     * pass too, so the history row is the complete filter state and only the launch boundary moves.  With the quad
     * mapping (STE_FLAG_LANES_4 or a small batch) slices need full covariance histories (no STE_FLAG_PACKED_COV).  The
     * smoother is not sliced.  What this is for: a forward launch is one indivisible wave per 64 tracks for the whole
     * pass; in slices, the waves of several batches re-balance over the chip at every boundary. */
    int32_t step_begin;
    int32_t step_end;
} ste_ukf_batch_f64;

int ste_version(void);

/* Thread-local message of the last failing call on this thread; valid until the next call on this thread. */
const char* ste_last_error(void);

/* Number of HIP devices visible (0 if none / runtime unavailable). Does not select a device. */
int ste_device_count(void);

/* Forward UKF over all steps of every track: writes fwd_mean, fwd_cov, status. */
int ste_ukf_forward_f64(const ste_ukf_batch_f64* b, void* stream);

/* ---- 0.3.2 -------------------------------------------------------------------------------------------------------
 * The forward passes of MANY windows (or batches) as ONE launch.  The reference's batch dimension is its per-ship loop
 * (examples/example_ukf_rts_smoother_batch.py:19-90); a fleet goes through the GPU as windows (track_stride above), and
 * with one launch per window a forward wave -- 64 tracks for the whole pass -- is indivisible: W windows of T tiles on S
 * SIMDs cost ceil(W T / S) pass times for W T / S pass times of work.  Here the unit is a (64-track tile, time slice) item,
 * the launch is `nwaves` resident waves (one per SIMD the stream may use), and `items` says which item every wave runs in
 * every round.  A tile's slices may run on different waves: each slice starts from the history row the one before it
 * left (the slices of ste_ukf_forward_f64: same bits as a whole pass); where a tile changes waves the finished slice
 * publishes the tile's slice count once its rows are in memory (release, agent scope) and the next slice waits for that
 * count (acquire) -- always an item of an EARLIER round; slices that follow one another on one wave are run as ONE item over
 * the whole step range, with no hand-over at all (the library finds those runs in the table).  The library checks the table
 * (every tile of every window runs all its slices, in order, at most one per round) before anything is launched, and
 * every in-kernel wait is bounded: a launch that cannot progress sets *error and ends.
 *
 * window_done[w] counts the finished tiles of window w: ste_stream_wait_counter(window_done + w, tiles of w, ...) on another
 * stream holds that stream until the window's forward pass is complete (its smoother goes behind it).
 *
 * All windows must take the same kernel (same H / R structure, robust flag, rts_work present or not); lane-per-track
 * mapping only.  host_ws should be page-locked (hipHostMalloc / hipHostRegister): a kernel on `stream` then reads the table
 * from it in place, and nothing but kernels sits between two launches; pageable memory goes through a staged copy.  Either
 * way it must stay untouched until the launch has started; dev_ws is filled by the call's own stream operations.  window_done,
 * error and started must be ZERO when the launch starts: the caller clears them on `stream` before the call (and orders
 * every stream that will wait on a counter behind that clearing), the call does not.
 *
 * Residency.  A wave may wait for any other wave of its launch, so all `nwaves` must be on the chip together.  One such
 * launch at a time gets there by itself; two that are dispatched together can each take part of the SIMDs and wait for
 * the rest until their bounds expire.  `started` (optional) counts the waves of this launch that have begun: put
 * ste_stream_wait_counter(started of the launch before, its nwaves, ...) on the stream in front of the next call and the
 * next launch is not dispatched before the one before it is resident (SmootherPipeline does, across pipelines of a process;
 * launches of different processes on one device are the caller's to keep apart). */
typedef struct ste_fwd_sched_f64 {
    int32_t nwindows;
    const ste_ukf_batch_f64* windows; /* HOST [nwindows]; step_begin = step_end = 0 */
    int32_t slice_steps;              /* steps per time slice, a multiple of STE_SLICE_ALIGN; 0 = STE_SLICE_ALIGN */
    int32_t nwaves;                   /* waves of the launch; all must be resident at once: <= SIMDs the stream may use */
    int32_t nrounds;
    const int32_t* items;             /* HOST [nrounds][nwaves][2]: (window, tile of that window), window < 0 = idle */
    void* host_ws;                    /* HOST scratch (page-locked if possible), ws_bytes (ste_ukf_forward_sched_workspace) */
    void* dev_ws;                     /* DEVICE scratch, ws_bytes */
    size_t ws_bytes;
    int32_t* window_done;             /* DEVICE [nwindows], zero at launch */
    int32_t* error;                   /* DEVICE [1], zero at launch: 1 = a forward wait timed out, 2 = a gate (wait_counter) did */
    double timeout_s;                 /* bound of one in-kernel wait; 0 = 2 s */
    int32_t* started;                 /* DEVICE [1] or NULL, zero at launch: waves of this launch that have begun */
} ste_fwd_sched_f64;

/* bytes of host_ws / dev_ws for a schedule of that shape (max_slices = largest ceil(Nmax / slice_steps) over the windows) */
size_t ste_ukf_forward_sched_workspace(int32_t nwindows, int32_t max_slices, int64_t ntiles_total, int32_t nrounds,
                                       int32_t nwaves);
int ste_ukf_forward_sched_f64(const ste_fwd_sched_f64* sc, void* stream);

/* Holds `stream` (a one-wave kernel) until *counter >= need; after timeout_s (0 = 2 s) it gives up and sets *error = 2. */
int ste_stream_wait_counter(const int32_t* counter, int32_t need, int32_t* error, double timeout_s, void* stream);

/* The smoothers of every window of a scheduled forward launch as ONE launch: a wave per (window, 64-track tile) that waits
 * (bounded; *error = 2) until the forward launch has finished THAT tile's last slice and then smooths it -- the reference's
 * run_rts_smoother (kalman_filter.py:119-137) per track, started as soon as the track's forward pass is complete instead of
 * when its whole window's is.  `items` lists every tile of every window once, in the order the forward schedule finishes
 * them (waves are dispatched in that order).  `progress` is the forward launch's per-tile slice counter array: dev_ws of that
 * launch + ste_ukf_forward_sched_progress_offset(same arguments as ste_ukf_forward_sched_workspace).  Every window must take
 * the one-kernel smoother (rts_work; more than 4096 tracks, or tuning bit 10) and agree on sog_rate_rts / cog_rate_rts.
 * Waiting waves hold a wave slot each: put ste_stream_wait_counter(started of the forward launch, its nwaves, ...) on `stream`
 * in front of this call, so that every forward wave has its SIMD before a waiting smoother wave could be in its way.  Results
 * are those of ste_urtss_backward_f64 on every window, bit for bit. */
typedef struct ste_bwd_sched_f64 {
    int32_t nwindows;
    const ste_ukf_batch_f64* windows; /* HOST [nwindows], the forward launch's windows */
    int32_t slice_steps;              /* as in the forward launch */
    int32_t nitems;                   /* tiles of all windows */
    const int32_t* items;             /* HOST [nitems][2]: (window, tile of that window) */
    void* host_ws;                    /* HOST scratch (page-locked if possible), ws_bytes (ste_urtss_backward_sched_workspace) */
    void* dev_ws;                     /* DEVICE scratch, ws_bytes */
    size_t ws_bytes;
    const int32_t* progress;          /* DEVICE: the forward launch's per-tile slice counters */
    int32_t* error;                   /* DEVICE [1]: the forward launch's error word */
    double timeout_s;                 /* bound of a wave's wait; 0 = 2 s */
} ste_bwd_sched_f64;

size_t ste_ukf_forward_sched_progress_offset(int32_t nwindows, int32_t max_slices, int64_t ntiles_total, int32_t nrounds,
                                             int32_t nwaves);
size_t ste_urtss_backward_sched_workspace(int32_t nwindows, int64_t ntiles_total);
int ste_urtss_backward_sched_f64(const ste_bwd_sched_f64* sc, void* stream);

/* Unscented RTS smoother: reads fwd_mean/fwd_cov, writes sm_mean/sm_cov, ORs status. */
int ste_urtss_backward_f64(const ste_ukf_batch_f64* b, void* stream);

/* Forward then backward on the same stream. */
int ste_ukf_urtss_f64(const ste_ukf_batch_f64* b, void* stream);

/*
 * Great-circle process model on `count` independent states.
 * x, out: [4][count]; dt, sog_rate, cog_rate: [count].
 */
int ste_geodetic_dynamics_f64(int64_t count, const double* x, const double* dt, const double* sog_rate,
                              const double* cog_rate, double* out, void* stream);

/*
 * One UKF predict on `count` independent (x, P) pairs.  x, x_out: [4][count]; P, P_out: [16][count];
 * dt, sog_rate, cog_rate: [count]; noise: [4][count] or NULL; Q: HOST 4x4; status: [count] or NULL.
 */
int ste_ukf_predict_f64(int64_t count, const double* x, const double* P, const double* dt, const double* sog_rate,
                        const double* cog_rate, const double* noise, const double* Q, double fan_scale, double w0,
                        double wi, double* x_out, double* P_out, int32_t* status, void* stream);

/*
 * One linear-KF update (pinv gain, heading wrap, Joseph form) on `count` independent (x, P) pairs with observations
 * z: [4][count]; noise: [4][count] or NULL (added to z); H, R: HOST 4x4.
 */
int ste_ukf_update_f64(int64_t count, const double* x, const double* P, const double* z, const double* noise,
                       const double* H, const double* R, double* x_out, double* P_out, int32_t* status, void* stream);

/*
 * Terms of the robustification helpers for `count` independent (x, P, z) triples, with y = z - x as the reference
 * writes it: gamma = |y^T (H P H^T + R)^+ y| (criterion_index, unscented.py:389-428) and
 * denom = y^T S^+ R S^+ y (update_lambda_factor, :468-478).  H, R: HOST 4x4.
 */
int ste_ukf_robust_terms_f64(int64_t count, const double* x, const double* P, const double* z, const double* H,
                             const double* R, double* gamma, double* denom, void* stream);

/*
 * Sigma fans of `count` (x, P) pairs: out[j][c][i] for sigma point j in 0..8, component c, pair i.
 * x: [4][count]; P: [16][count]; out: [9][4][count].  scale = n/(1-W0) (n when weights were never computed).
 */
int ste_sigma_points_f64(int64_t count, const double* x, const double* P, double scale, double* out, void* stream);

/*
 * Same for a general state dimension 1 <= n <= 16 (the reference's constructor and compute_sigma_points accept any n;
 * its unit tests use n = 2).  x: [n][count]; P: [n*n][count]; out: [2n+1][n][count].  Not a hot path.
 */
int ste_sigma_points_generic_f64(int32_t n, int64_t count, const double* x, const double* P, double scale, double* out,
                                 void* stream);

/* ---------------------------------------------------------------------------------------------------------------
 * Second kernel set: Gaussian-process regression (reference: src/track_estimators/gaussian_processes/
 * gaussian_process.py:28-89, a wrapper over scikit-learn's GaussianProcessRegressor).  One batch = B independent tracks;
 * track b has n[b] observations at 1-D inputs x (cumulative time, gaussian_process.py:53-58) with nout outputs
 * (lon, lat: :66) and its own kernel hyper-parameters theta = log(constant, length_scale, noise) of
 * ConstantKernel * RBF + WhiteKernel (examples/example_gaussian_process_batch.py:41).
 * Matrices are row-major [B][ld][ld] with ld = 64 * ceil(nmax / 64) + STE_GP_LD_PAD (the pad keeps consecutive rows off
 * the same HBM channel when 64 * ceil(nmax / 64) is a power of two); all buffers are caller-owned device memory.
 * ------------------------------------------------------------------------------------------------------------- */
#define STE_GP_LD_PAD 16
#define STE_GP_INVERSE_AUTO 0 /* column order (one workgroup per matrix) when B >= 128, else row order (nb workgroups per matrix) */
#define STE_GP_INVERSE_ROWS 1
#define STE_GP_INVERSE_COLS 2

typedef struct ste_gp_batch_f64 {
    int32_t B;     /* number of tracks */
    int32_t nmax;  /* padded number of observations */
    int32_t nout;  /* output columns of y (2: lon, lat) */
    int32_t inverse_order; /* which kernel forms U = L^-T: STE_GP_INVERSE_AUTO (0: by B, see below), _ROWS (1), _COLS (2).  The two
                              sum in different orders, so objectives that must agree bit for bit (a track's first start and
                              its restarts in a replicated batch) have to name the same one */
    double jitter; /* added to the diagonal of K (GaussianProcessRegressor alpha, 1e-10) */
    const int32_t* n;    /* [B] observations per track */
    const double* x;     /* [B][nmax] */
    const double* y;     /* [B][nout][nmax] */
    const double* theta; /* [B][3] log(constant), log(length_scale), log(noise) */
    double* K;      /* [B][ld][ld]  K(X,X) + (noise + jitter) I (ste_gp_rbf_kmatrix_f64), then its Cholesky factor L (lower triangle,
                       diagonal included; what is above the diagonal is not defined).  ste_gp_lml_f64 evaluates the kernel
                       function inside the factorisation and leaves L here without K ever being stored */
    double* U;      /* [B][ld][ld]  workspace: L^-T (upper triangle) */
    double* Dinv;   /* [B][(ld-16)/64][64][64] workspace: inverses of the diagonal blocks of L */
    double* Kinv;   /* [B][ld][ld] K^-1 (both triangles) when non-NULL; needed by ste_gp_predict_f64 */
    double* alpha;  /* [B][nout][nmax] out: K^-1 y (after ste_gp_potrf_f64 alone: the forward substitution L^-1 y, which rides along
                       with the factorisation) */
    double* lml;    /* [B] out: log marginal likelihood summed over outputs */
    double* grad;   /* [B][3] out: d lml / d theta, or NULL to skip the gradient */
    double* tr;     /* [B][3][nt] workspace, nt = nb(nb + 1)/2 with nb = ceil(nmax/64): per-tile partial traces (summed in tile order) */
    int32_t* status; /* [B] out: 0 ok, 1 = K not positive definite */
} ste_gp_batch_f64;

const char* ste_gp_last_error(void);

/* K(X,X) build only (lower 64x64 tiles + identity padding) -- sklearn kernel __call__ (RBF: exp(-pdist^2/2)). */
int ste_gp_rbf_kmatrix_f64(const ste_gp_batch_f64* b, void* stream);

/* In-place blocked Cholesky of the K that ste_gp_rbf_kmatrix_f64 left in the buffer (scipy.linalg.cholesky(K, lower=True) in
 * GaussianProcessRegressor); also writes L^-1 y to alpha and the inverted diagonal blocks to Dinv. */
int ste_gp_potrf_f64(const ste_gp_batch_f64* b, void* stream);

/*
 * One objective evaluation per track (GaussianProcessRegressor.log_marginal_likelihood(theta, eval_gradient=True)):
 * Cholesky of K (the kernel function evaluated in place of a stored K; same values as ste_gp_rbf_kmatrix_f64), L^-T, alpha,
 * lml, and -- when grad != NULL -- the gradient via K^-1 = L^-T L^-1 reduced against dK/dtheta on the fly.
 */
int ste_gp_lml_f64(const ste_gp_batch_f64* b, void* stream);

/* The same evaluation for `count` of the batch's matrices only: active[count] (DEVICE int32, distinct indices in
 * [0, B)) names them.  Outputs of the other matrices are left untouched.  This is what a lock-step batched optimiser
 * uses once some of its tracks have converged (scipy's L-BFGS-B stops per track, gaussian_process.py:63-66 via
 * scikit-learn's _constrained_optimization); per-matrix results do not depend on which other matrices are listed. */
int ste_gp_lml_subset_f64(const ste_gp_batch_f64* b, int32_t count, const int32_t* active, void* stream);

/*
 * Posterior mean [B][nout][mmax] and variance [B][mmax] at m[b] new inputs xs [B][mmax]
 * (GaussianProcessRegressor.predict(return_std=True); std = sqrt(max(var, 0)) is left to the caller).
 * Needs alpha and Kinv from a preceding ste_gp_lml_f64 on the same batch.  Kstar: workspace [B][64*ceil(mmax/64)][ld].
 */
int ste_gp_predict_f64(const ste_gp_batch_f64* b, int32_t mmax, const int32_t* m, const double* xs, double* Kstar,
                       double* mean, double* var, void* stream);

/* ---------------------------------------------------------------------------------------------------------------
 * Observation preparation: speed / course over ground and their rates for a batch of tracks, the
 * device counterpart of ShipTrack.calculate_sog (ship_track.py:197-224), calculate_sog_rate (:226-250),
 * calculate_cog (:252-278), calculate_cog_rate (:280-304) and get_measurements(include_sog=True, include_cog=True)
 * (:306-338).  Arrays are [Tmax][B], track index fastest; observation i >= nobs[t] of a short track is written as 0.
 *   sog[i]      = distance(i, i+1) / gap[i]   (km/h), last value repeated;  gap = 0 gives inf / NaN as in the reference
 *   cog[i]      = heading(i, i+1)             (degrees in [0, 360)), last value repeated
 *   *_rate[i]   = (v[i] - v[i-1]) / gap[i-1], 0 for i = 0
 *   z[i][0..3]  = lon, lat, sog, cog          (optional; the layout ste_ukf_batch_f64.z consumes)
 * A track with fewer than 2 observations has no leg: all outputs 0 (the reference raises IndexError there).
 * ------------------------------------------------------------------------------------------------------------- */
#define STE_PREP_SPHERE 0 /* haversine_formula + heading on the 6378.137 km sphere (utils.py:75-147) */
#define STE_PREP_STATUS_NOCONV 0x1 /* reserved: up to 0.3.0 the WGS84 model used Vincenty's iteration, which does not converge
                                      for nearly antipodal points, and flagged such legs here; Karney's solver (0.3.1) solves
                                      every leg and never sets it */
#define STE_PREP_WGS84 1  /* geographiclib_distance + geographiclib_heading semantics (utils.py:9-72): WGS84 inverse
                             geodesic by Karney's algorithm (J. Geodesy 87, 2013), what geographiclib implements */

typedef struct ste_prep_batch_f64 {
    int32_t B;            /* tracks */
    int32_t Tmax;         /* observations per track (padded) */
    int32_t model;        /* STE_PREP_* */
    int32_t reserved;
    const int32_t* nobs;  /* [B] observations per track, NULL = Tmax for all */
    const double* lon;    /* [Tmax][B] degrees */
    const double* lat;    /* [Tmax][B] degrees */
    const double* gap;    /* [Tmax-1][B] hours between observation i and i+1 (ShipTrack.dts) */
    double* sog;          /* [Tmax][B] out */
    double* cog;          /* [Tmax][B] out */
    double* sog_rate;     /* [Tmax][B] out */
    double* cog_rate;     /* [Tmax][B] out */
    double* z;            /* [Tmax][4][B] out, may be NULL */
    int32_t* status;      /* [B] out, may be NULL: STE_PREP_STATUS_* bits per track (reset by the call) */
} ste_prep_batch_f64;

int ste_track_prep_f64(const ste_prep_batch_f64* b, void* stream);

/* ---------------------------------------------------------------------------------------------------------------
 * Stream partitioning for pipelined batches.  At the batch sizes of BASELINE configs[1..2] the forward kernel is a few
 * hundred long-running waves (10 000 tracks = 625 waves on 1 024 SIMDs) and the smoother is latency-bound, so the
 * smoother of batch i can run beside the forward pass of batch i+1 -- provided they do not share SIMDs, where each
 * would take the other's issue slots.  These two calls create / destroy a HIP stream restricted to the compute units
 * [first_cu, first_cu + num_cus) of the current device (hipExtStreamCreateWithCUMask; on MI355X mask bit n is CU n/8
 * of XCD n%8, so a contiguous range is spread evenly over the eight XCDs).  Every entry point above accepts such a
 * stream.  Host plumbing only: nothing of the reference corresponds to it (the reference runs ships one at a time,
 * examples/example_ukf_rts_smoother_batch.py:19-90).
 * ------------------------------------------------------------------------------------------------------------- */
int ste_stream_create_cu_range(int32_t first_cu, int32_t num_cus, void** stream);
int ste_stream_destroy(void* stream);

#ifdef __cplusplus
}
#endif
#endif /* STE_H */
