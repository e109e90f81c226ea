// ste_prep.hip — observation preparation for a batch of tracks: speed / course over ground and their rates.
//
// Device counterpart of ShipTrack.calculate_sog / calculate_cog / calculate_sog_rate / calculate_cog_rate /
// get_measurements (reference src/track_estimators/ship_track.py:197-338) for many tracks at once: one thread per
// (observation, track), track index fastest, every read and write coalesced.  The kernel is pure streaming work
// (24 B in, 64 B out per observation) with ~10 transcendentals per observation on the sphere and a short fixed-point
// iteration on the ellipsoid, so it is HBM/latency trivial next to the filter; it exists so that raw lon/lat/time can
// go to smoothed tracks without a per-ship Python loop.
//
//   model 0  sphere of radius 6378.137 km: haversine_formula + heading        (reference utils.py:75-147)
//   model 1  WGS84 inverse geodesic: geographiclib_distance + _heading         (reference utils.py:9-72)
//            geographiclib itself is a third-party dependency that is not part of the reference tree; the inverse
//            problem is solved with Vincenty's iteration on the same ellipsoid, exactly as track_estimators/utils.py
//            does on the host (pinned by the reference's CLI fixture, tests/test_host_logic.py).
#include <hip/hip_runtime.h>

#include <cstdint>

#include "../../include/ste.h"
#include "ste_err.h"
#include "ste_math.h"

// The reference's formulas give exact zeros for coincident points (distance 0, heading atan2(0, 0) = 0) and
// 0 / 0 = NaN on a duplicate timestamp; a fused multiply-add across `lat2 * k - lat1 * k` would turn those zeros into
// rounding residue.  This file is therefore compiled without contraction (the library's build uses -ffp-contract=fast).
#pragma clang fp contract(off)

namespace ste {
namespace {

constexpr double kDeg2Rad = 0.017453292519943295;  // np.radians multiplies by this double
constexpr double kRad2Deg = 57.29577951308232;     // np.degrees
constexpr double kPi = 3.141592653589793;
constexpr double kEarthKm = 6378.137;
constexpr double kWgsA = 6378137.0;
constexpr double kWgsF = 1.0 / 298.257223563;

struct Leg {
    double dist_km;
    double head_deg;
    bool converged;  // false: Vincenty's iteration hit its cap (nearly antipodal points); the sphere model always converges
};

// utils.py:75-147 — haversine (atan2 form) and the initial great-circle bearing in [0, 360)
__device__ Leg sphere_leg(double lon1, double lat1, double lon2, double lat2) {
    lon1 *= kDeg2Rad;
    lat1 *= kDeg2Rad;
    lon2 *= kDeg2Rad;
    lat2 *= kDeg2Rad;
    const double dlat = lat2 - lat1, dlon = lon2 - lon1;
    const double sh = sin(dlat / 2.0), sl = sin(dlon / 2.0);
    const double c1 = cos(lat1), c2 = cos(lat2);
    const double a = sh * sh + c1 * c2 * sl * sl;
    Leg r;
    r.converged = true;
    r.dist_km = 2.0 * atan2(sqrt(a), sqrt(1.0 - a)) * kEarthKm;
    const double east = sin(dlon) * c2;
    const double north = c1 * sin(lat2) - sin(lat1) * c2 * cos(dlon);
    r.head_deg = floored_mod360(atan2(east, north) * kRad2Deg + 360.0);
    return r;
}

// utils.py:9-72 with Geodesic.WGS84.Inverse replaced by Vincenty's inverse iteration (see the header of this file)
__device__ Leg wgs84_leg(double lon1, double lat1, double lon2, double lat2) {
    Leg r{0.0, 0.0, true};
    if (fabs(lat1 - lat2) < 1e-8 && fabs(lon1 - lon2) < 1e-8) return r;  // utils.py:32-33, :64-65
    const double b = kWgsA * (1.0 - kWgsF);
    const double phi1 = lat1 * kDeg2Rad, phi2 = lat2 * kDeg2Rad;
    double L = (lon2 - lon1) * kDeg2Rad;
    {
        double m = fmod(L + kPi, 2.0 * kPi);
        if (m < 0.0) m += 2.0 * kPi;
        L = m - kPi;
    }
    const double U1 = atan((1.0 - kWgsF) * tan(phi1)), U2 = atan((1.0 - kWgsF) * tan(phi2));
    const double sU1 = sin(U1), cU1 = cos(U1), sU2 = sin(U2), cU2 = cos(U2);
    double lam = L, sl = 0.0, cl = 1.0, sin_sigma = 0.0, cos_sigma = 1.0, sigma = 0.0, cos2_alpha = 1.0, cos_2sm = 0.0;
    for (int it = 0; it < 200; ++it) {
        sl = sin(lam);
        cl = cos(lam);
        sin_sigma = hypot(cU2 * sl, cU1 * sU2 - sU1 * cU2 * cl);
        if (sin_sigma == 0.0) return r;  // coincident points
        cos_sigma = sU1 * sU2 + cU1 * cU2 * cl;
        sigma = atan2(sin_sigma, cos_sigma);
        const double sin_alpha = cU1 * cU2 * sl / sin_sigma;
        cos2_alpha = 1.0 - sin_alpha * sin_alpha;
        cos_2sm = cos2_alpha != 0.0 ? cos_sigma - 2.0 * sU1 * sU2 / cos2_alpha : 0.0;
        const double Cc = kWgsF / 16.0 * cos2_alpha * (4.0 + kWgsF * (4.0 - 3.0 * cos2_alpha));
        const double lam_new =
            L + (1.0 - Cc) * kWgsF * sin_alpha *
                    (sigma + Cc * sin_sigma * (cos_2sm + Cc * cos_sigma * (-1.0 + 2.0 * cos_2sm * cos_2sm)));
        const bool done = fabs(lam_new - lam) < 1e-15;
        lam = lam_new;
        r.converged = done;
        if (done) break;
    }
    // Vincenty's fixed point does not contract for nearly antipodal points (geographiclib's Karney solver, which the
    // reference calls, handles them); the values below are then the last iterate's, and the track is flagged.
    sl = sin(lam);
    cl = cos(lam);
    const double u2 = cos2_alpha * (kWgsA * kWgsA - b * b) / (b * b);
    const double A = 1.0 + u2 / 16384.0 * (4096.0 + u2 * (-768.0 + u2 * (320.0 - 175.0 * u2)));
    const double Bc = u2 / 1024.0 * (256.0 + u2 * (-128.0 + u2 * (74.0 - 47.0 * u2)));
    const double dsig =
        Bc * sin_sigma *
        (cos_2sm + Bc / 4.0 * (cos_sigma * (-1.0 + 2.0 * cos_2sm * cos_2sm) -
                               Bc / 6.0 * cos_2sm * (-3.0 + 4.0 * sin_sigma * sin_sigma) * (-3.0 + 4.0 * cos_2sm * cos_2sm)));
    r.dist_km = b * A * (sigma - dsig) * 1e-3;
    const double azi = atan2(cU2 * sl, cU1 * sU2 - sU1 * cU2 * cl) * kRad2Deg;
    r.head_deg = floored_mod360(azi + 360.0);
    return r;
}

struct PrepParams {
    int B, T, model;
    const int32_t* nobs;
    const double *lon, *lat, *gap;
    double *sog, *cog, *sog_rate, *cog_rate, *z;
    int32_t* status;
};

template <int kModel>
__device__ __forceinline__ Leg leg_of(const PrepParams& p, int j, int t) {
    const size_t a = (size_t)j * p.B + t, b = a + p.B;
    const double lon1 = p.lon[a], lat1 = p.lat[a], lon2 = p.lon[b], lat2 = p.lat[b];
    return kModel == 0 ? sphere_leg(lon1, lat1, lon2, lat2) : wgs84_leg(lon1, lat1, lon2, lat2);
}

// Observation i of a track with n observations uses leg j = min(i, n-2) (the last value is repeated,
// ship_track.py:220, :275); its rate is the backward difference against observation i-1 over gap[i-1], 0 for i = 0
// (ship_track.py:242-246, :296-300).  The previous observation's leg is recomputed here rather than read back, so the
// kernel is one pass with no ordering between threads; both evaluations run the same code and agree bit for bit.
template <int kModel>
__global__ void __launch_bounds__(256) track_prep(PrepParams p) {
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (size_t)p.T * p.B) return;
    const int t = (int)(idx % p.B), i = (int)(idx / p.B);
    const int n = p.nobs ? min(p.nobs[t], p.T) : p.T;
    double sog = 0.0, cog = 0.0, sr = 0.0, cr = 0.0;
    if (i < n && n >= 2) {
        const int j = min(i, n - 2);
        const Leg cur = leg_of<kModel>(p, j, t);
        if (kModel == 1 && !cur.converged && p.status) atomicOr(&p.status[t], STE_PREP_STATUS_NOCONV);
        sog = cur.dist_km / p.gap[(size_t)j * p.B + t];
        cog = cur.head_deg;
        if (i >= 1) {
            const int jp = min(i - 1, n - 2);
            double sog_p = sog, cog_p = cog;
            if (jp != j) {
                const Leg prev = leg_of<kModel>(p, jp, t);
                sog_p = prev.dist_km / p.gap[(size_t)jp * p.B + t];
                cog_p = prev.head_deg;
            }
            const double g = p.gap[(size_t)(i - 1) * p.B + t];
            sr = (sog - sog_p) / g;
            cr = (cog - cog_p) / g;
        }
    }
    p.sog[idx] = sog;
    p.cog[idx] = cog;
    p.sog_rate[idx] = sr;
    p.cog_rate[idx] = cr;
    if (p.z) {
        const size_t zb = ((size_t)i * 4) * p.B + t;
        const bool live = i < n;
        p.z[zb] = live ? p.lon[idx] : 0.0;
        p.z[zb + p.B] = live ? p.lat[idx] : 0.0;
        p.z[zb + 2 * (size_t)p.B] = sog;
        p.z[zb + 3 * (size_t)p.B] = cog;
    }
}

}  // namespace
}  // namespace ste

extern "C" int ste_track_prep_f64(const ste_prep_batch_f64* b, void* stream) {
    using namespace ste;
    if (!b) return abi_fail(STE_EINVAL, "prep batch pointer is NULL");
    if (b->B <= 0 || b->Tmax < 1) return abi_fail(STE_EINVAL, "B must be > 0 and Tmax >= 1");
    if (b->model != STE_PREP_SPHERE && b->model != STE_PREP_WGS84)
        return abi_fail(STE_EINVAL, "model must be STE_PREP_SPHERE or STE_PREP_WGS84");
    if (!b->lon || !b->lat || !b->sog || !b->cog || !b->sog_rate || !b->cog_rate)
        return abi_fail(STE_EINVAL, "lon, lat, sog, cog, sog_rate and cog_rate are required");
    if (b->Tmax > 1 && !b->gap) return abi_fail(STE_EINVAL, "gap is required when Tmax > 1");
    PrepParams p{b->B, b->Tmax, b->model, b->nobs, b->lon, b->lat, b->gap, b->sog, b->cog, b->sog_rate, b->cog_rate, b->z,
                 b->status};
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (b->status) {
        int rc = abi_check_hip(hipMemsetAsync(b->status, 0, sizeof(int32_t) * (size_t)b->B, s), "track_prep status reset");
        if (rc) return rc;
    }
    const size_t total = (size_t)b->B * b->Tmax;
    const unsigned grid = (unsigned)((total + 255) / 256);
    if (b->model == STE_PREP_SPHERE)
        hipLaunchKernelGGL(track_prep<0>, dim3(grid), dim3(256), 0, s, p);
    else
        hipLaunchKernelGGL(track_prep<1>, dim3(grid), dim3(256), 0, s, p);
    return abi_check_hip(hipGetLastError(), "track_prep launch");
}
