"""
Process models.  Mirrors reference ``track_estimators.kalman_filters.non_linear_process``
(/root/reference/src/track_estimators/kalman_filters/non_linear_process.py:6-85).

``geodetic_dynamics`` keeps the reference signature.  It is the one process model the HIP kernels implement
(``geodetic_step`` in csrc/ste_math.h); the batched filter recognises it by identity.  Called directly it evaluates
the model on the GPU through ``ste_geodetic_dynamics_f64``; ``x`` may be one state ``(n,)`` or a batch ``(count, 4)``.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from ..constants import EARTH_RADIUS  # noqa: F401  (re-exported like the reference module namespace)


def geodetic_dynamics(x, c, dt, sog_rate=0.0, cog_rate=0.0):
    """
    Great-circle dead reckoning of [longitude, latitude, speed, heading] (degrees, km/h, hours).

    Parameters follow the reference (non_linear_process.py:6-12).  A non-empty control vector ``c`` is concatenated
    to the state exactly as the reference does (:47-51) before the first four entries are read.
    Returns the transformed state truncated to ``x.shape[0]`` entries (:85).
    """
    import torch

    from .._hip import binding

    x = np.asarray(x, dtype=np.float64)
    single = x.ndim == 1
    xs = x[None, :] if single else x
    if c is not None and np.size(c):
        cc = np.asarray(c, dtype=np.float64)
        xs = np.concatenate([xs, np.broadcast_to(cc, (xs.shape[0], cc.shape[-1]))], axis=1)
    if xs.shape[1] < 4:
        raise IndexError(f"geodetic_dynamics needs at least 4 state entries (lon, lat, speed, heading); got {xs.shape[1]}")
    n_out = x.shape[-1]
    count = xs.shape[0]
    lib = binding.require_gpu()
    dev = torch.device("cuda", torch.cuda.current_device())
    up = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64)).to(dev)  # noqa: E731
    xin = up(xs[:, :4].T)
    bc = lambda v: up(np.array(np.broadcast_to(np.asarray(v, dtype=np.float64), (count,))))  # noqa: E731
    d, sr, cr = bc(dt), bc(sog_rate), bc(cog_rate)
    out = torch.empty_like(xin)
    stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    binding.check(lib.ste_geodetic_dynamics_f64(count, xin.data_ptr(), d.data_ptr(), sr.data_ptr(), cr.data_ptr(),
                                                out.data_ptr(), stream), "ste_geodetic_dynamics_f64")
    res = out.cpu().numpy().T[:, : min(4, n_out)]
    return res[0] if single else res
