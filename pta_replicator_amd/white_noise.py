"""EFAC/EQUAD and ECORR injection on the MI355X: ``add_measurement_noise`` / ``add_jitter``.

Mirrors ``pta_replicator/white_noise.py`` (quantize_fast :7-44, add_measurement_noise :47-125, add_jitter
:128-198), same signatures, same ValueErrors, same ``added_signals`` keys.  Flag -> per-TOA vector expansion and
the (realisation-independent) epoch bucketing stay on the host - the latter in native code
(pta_quantize_epochs) - while the per-TOA arithmetic runs in HIP kernels.  ECORR is a gather through the epoch
map instead of the reference's dense N x E indicator matrix (17 s of its 18.6 s per realisation at 68 x 5000).
``add_efac`` / ``add_equad`` / ``add_ecorr`` are libstempo-style aliases.

Latency of a call: the flag column of a pulsar is indexed ONCE (``flag_codes``: unique labels + integer codes,
cached on the TOA object) instead of being rebuilt as a string array per call; the operands of a call travel in one pinned staging
copy and the result in one (device.upload_packed / download).  Both functions also accept a LIST of pulsars (with per-pulsar
parameter lists and a list of seeds): one upload, one launch over the concatenated TOAs and one download per signal, every
pulsar's draws exactly the stream ``np.random.seed(seed_i)`` would give.  What remains of a call is host work that the parity
contract pins: the NumPy legacy draws themselves (14 ms per 68 x 5000 realisation) and the per-pulsar longdouble TOA bookkeeping.
"""
import ctypes

import numpy as np

from . import _lib, device as dv
from ._compat import TimeDelta, u


def epoch_map(times, dt=1.0):
    """Greedy bucketing of white_noise.py:21-31 -> (epoch_of int32[N], first_index int32[E]).

    ``epoch_of[i]`` is the column of the reference's U holding TOA i; ``first_index[e]`` is the earliest TOA of
    epoch e (its flag labels the epoch, :35).  Ties are ordered by np.argsort(times) exactly as the reference."""
    times = np.ascontiguousarray(times, dtype=np.float64)
    n = len(times)
    order = np.ascontiguousarray(np.argsort(times), dtype=np.int64)
    epoch_of = np.empty(n, dtype=np.int32)
    first = np.empty(n, dtype=np.int32)
    ne = ctypes.c_int(0)
    _lib.call("pta_quantize_epochs", dv.hptr(times), n, ctypes.c_double(dt), dv.hptr(order), dv.hptr(epoch_of), dv.hptr(first),
              ctypes.byref(ne))
    return epoch_of, first[:ne.value].copy()


def quantize_fast(times, flags=None, dt=1.0):
    """Same returns as the reference's quantize_fast (white_noise.py:7-44): (avetoas, [aveflags,] U).

    The dense U is materialised only for API compatibility; add_jitter itself never builds it."""
    times = np.asarray(times)
    epoch_of, first = epoch_map(times, dt)
    ne = len(first)
    avetoas = np.array([np.mean(times[epoch_of == e]) for e in range(ne)], "d")
    U = np.zeros((len(times), ne), "d")
    U[np.arange(len(times)), epoch_of] = 1
    if flags is not None:
        return avetoas, np.asarray(flags)[first], U
    return avetoas, U


def flag_codes(toas, flagid):
    """(labels, codes): ``labels[codes[i]]`` is the value of flag ``flagid`` of TOA i - what the reference rebuilds on every call
    as ``np.array([f[flagid] for f in toas.table['flags'].data])`` (white_noise.py:98-99,161; half of a call's time at 5000 TOAs).
    Built once per (TOA object, flagid) and cached on the object; flags are injection-invariant (adjust_TOAs does not touch them).
    The cache is keyed by the TOA count and spot-checked against the live flags at five positions (ends, middle, two fixed pseudo-random
    ones).  INVARIANT the caller owns: flags edited in place between add_* calls are not seen - the reference re-reads them on every
    call - so after such an edit call ``clear_caches(psr)``; the errors column has no such caveat (content-checked)."""
    data = toas.table["flags"].data
    n = len(data)
    cache = getattr(toas, "_pta_flag_codes", None)
    hit = cache.get(flagid) if isinstance(cache, dict) else None
    if hit is not None and hit[0] == n and n > 0:
        probe = (0, n // 2, n - 1, 2654435761 % n, 40503 * 7919 % n)
        if all(str(hit[1][hit[2][i]]) == str(data[i][flagid]) for i in probe):
            return hit[1], hit[2]
    col = np.array([f[flagid] for f in data])
    labels, codes = np.unique(col, return_inverse=True)
    codes = codes.astype(np.int32)
    try:
        if not isinstance(cache, dict):
            cache = {}
            toas._pta_flag_codes = cache
        cache[flagid] = (n, labels, codes)
    except AttributeError:      # an object that refuses new attributes: no cache, same result
        pass
    return labels, codes


def clear_caches(obj):
    """forget the per-TOA-object caches (flag index, errors in seconds) of a pulsar or TOA container - after editing flags in place."""
    toas = getattr(obj, "toas", obj)
    for name in ("_pta_flag_codes", "_pta_errors_s"):
        try:
            toas.__dict__.pop(name, None)
        except AttributeError:
            pass


def _per_flag_vector(labels, codes, flags, values):
    """vec[i] = values[ct] where TOA i's flag == flags[ct], else 0 - the loop of white_noise.py:100-103 through the label table
    (a later entry of `flags` overrides an earlier one, as in the reference's in-order assignment)."""
    lut = np.zeros(len(labels))
    for ct, flag in enumerate(flags):
        lut[flag == labels] = values[ct]
    return lut[codes]


def errors_seconds(toas):
    """TOA uncertainties in seconds as float64 (white_noise.py:105: ``get_errors().to('s')``), converted once per TOA object - the
    errors are injection-invariant - and cached on it.  The cache is validated by CONTENT, never by object identity alone: array-backed
    TOAs are checked against their source column (``errors_us``), any other container (PINT-style: ``get_errors()`` returns a fresh
    Quantity on every call) against the raw values of ``get_errors()`` itself - a comparison of N doubles instead of a unit conversion
    through astropy (the round-4 revision cached only the array-backed kind)."""
    src = getattr(toas, "errors_us", None)      # array-backed TOAs: the source column itself (no unit conversion needed to validate)
    hit = getattr(toas, "_pta_errors_s", None)
    if src is not None:
        if hit is not None and hit[0] is src and len(hit[2]) == len(src) and np.array_equal(hit[1], src):
            return hit[2]                        # same column object, same content (a rescaled error column must not be served stale)
        sig = np.asarray(toas.get_errors().to("s").value, dtype=np.float64)
        key = (src, np.array(src, copy=True), sig)
    else:
        q = toas.get_errors()
        raw = np.asarray(getattr(q, "value", q), dtype=np.float64)
        unit = str(getattr(q, "unit", ""))
        if hit is not None and hit[0] == unit and hit[1].shape == raw.shape and np.array_equal(hit[1], raw):
            return hit[2]
        sig = np.asarray(q.to("s").value, dtype=np.float64)
        key = (unit, np.array(raw, copy=True), sig)
    try:
        toas._pta_errors_s = key
    except AttributeError:
        pass
    return sig


def _wn_vectors(psr, efac, equad, flagid, flags):
    """efacvec, equadvec of one pulsar with the reference's checks and messages (white_noise.py:83-103)."""
    ntoas = psr.toas.ntoas
    efacvec = np.zeros(ntoas)
    equadvec = np.zeros(ntoas)
    if flags is None:
        if not np.isscalar(efac) or not np.isscalar(equad):
            raise ValueError("ERROR: If flags is None, efac and equad must be a scalar")
        efacvec = np.ones(ntoas) * efac
        equadvec = np.ones(ntoas) * equad
    if (flags is not None and not np.isscalar(efac)) or (flags is not None and not np.isscalar(equad)):
        if len(efac) == len(flags) and len(equad) == len(flags):
            labels, codes = flag_codes(psr.toas, flagid)
            efacvec = _per_flag_vector(labels, codes, flags, efac)
            equadvec = _per_flag_vector(labels, codes, flags, equad)
        else:
            raise ValueError("ERROR: flags must be same length as efac and log10_equad")
    return efacvec, equadvec


def _legacy_normals(seeds, counts):
    """One list of NumPy legacy-stream normal arrays per pulsar: pulsar i draws ``randn(c)`` for c in counts[i], in order, from the
    stream ``np.random.seed(seeds[i])`` starts (white_noise.py:79-80,105-109,154-155,182) - the same MT19937 + legacy polar Gaussian,
    value for value - and the GLOBAL stream is left where the sequential calls would leave it: in the state after the last pulsar's
    draws.  Integer seeds go through the native restatement of that generator, the pulsars' independent streams on host threads
    (``pta_legacy_randn``: 0.9 ms for 68 streams of 10 000 deviates against 12.7 ms through 68 ``RandomState(seed)`` objects - each
    costs ~100 us to construct, and the legacy generator holds the GIL, so Python threads cannot draw side by side: a thread pool
    measured 2.3x SLOWER than the serial loop); anything else NumPy accepts as a seed goes through ``RandomState(seed)``.  ``seeds is None``: the global stream continues
    through the pulsars in order, like a loop of reference calls with seed=None - one serial stream."""
    if seeds is None:
        return [[np.random.randn(int(c)) for c in cs] for cs in counts]
    if len(seeds) and all(isinstance(sd, (int, np.integer)) and not isinstance(sd, bool) and 0 <= int(sd) < 2 ** 32 for sd in seeds):
        tot = np.array([sum(int(c) for c in cs) for cs in counts], dtype=np.int64)
        off = np.concatenate([[0], np.cumsum(tot)]).astype(np.int64)
        flat = np.empty(int(off[-1]), dtype=np.float64)
        sd = np.array([int(x) for x in seeds], dtype=np.uint32)
        key, ph, g = np.empty(624, dtype=np.uint32), np.zeros(2, dtype=np.int32), np.zeros(1, dtype=np.float64)
        _lib.call("pta_legacy_randn", sd.ctypes.data, tot.ctypes.data, off.ctypes.data, len(sd), flat.ctypes.data, key.ctypes.data,
                  ph.ctypes.data, g.ctypes.data, 0)
        np.random.set_state(("MT19937", key, int(ph[0]), int(ph[1]), float(g[0])))
        out = []
        for i, cs in enumerate(counts):
            edges = off[i] + np.concatenate([[0], np.cumsum([int(c) for c in cs])])
            out.append([flat[edges[k]:edges[k + 1]] for k in range(len(cs))])
        return out
    out, rs = [], None
    for seed, cs in zip(seeds, counts):
        rs = np.random.RandomState(seed)
        out.append([rs.randn(int(c)) for c in cs])
    if rs is not None:
        np.random.set_state(rs.get_state())
    return out


def _broadcast(val, P, what):
    if isinstance(val, (list, tuple)) and len(val) == P:
        return list(val)
    if isinstance(val, np.ndarray) and val.ndim >= 1 and len(val) == P:
        return list(val)
    if val is None or np.isscalar(val) or (isinstance(val, np.ndarray) and val.ndim == 0):
        return [val] * P
    raise ValueError(f"{what}: expected a scalar or one entry per pulsar ({P})")


def _seed_list(seed, P):
    """None, or one seed per pulsar (the list forms re-seed per pulsar exactly as the loop of single calls would)."""
    if seed is None:
        return None
    if np.ndim(seed) == 0:
        raise ValueError("seed must be None or one seed per pulsar (a sequence), not a single value: the loop of single calls re-seeds per pulsar")
    seeds = list(seed)
    if len(seeds) != P:
        raise ValueError("seed must be None or one seed per pulsar")
    return seeds


def _pow10(l10):
    """10 ** log10_value exactly as the single-pulsar path computes it: a Python / NumPy SCALAR goes through libm's pow (``10 ** float``),
    only a genuine vector through NumPy's ufunc loop - whose SIMD build may differ from libm by an ulp, which would break the
    bit-identity of a list call with the loop of single calls."""
    if np.ndim(l10) == 0:
        return 10 ** (l10.item() if isinstance(l10, np.generic) or isinstance(l10, np.ndarray) else l10)
    return 10 ** np.asarray(l10, dtype=float)


def _record_wn(psr, efac, log10_equad, flags, equad_str, dt):
    if flags is None:
        psr.update_added_signals("{}_measurement_noise".format(psr.name),
                                 {"efac": efac, "log10_" + equad_str: log10_equad}, dt)
    else:
        psr.update_added_signals("{}_measurement_noise".format(psr.name), {}, dt)
        for i, flag in enumerate(flags):
            psr.update_added_signals("{}_{}_measurement_noise".format(psr.name, flag),
                                     {"efac": efac[i], "log10_" + equad_str: log10_equad[i]})
    psr.toas.adjust_TOAs(TimeDelta(dt.to("day")))
    psr.update_residuals()


def add_measurement_noise(psr, efac=1.0, log10_equad=None, flagid="f", flags=None, seed=None, tnequad=False):
    """Add EFAC/EQUAD white noise (white_noise.py:47-125): EFAC*(sigma z1 + EQUAD z2) by default,
    EFAC*sigma z1 + EQUAD z2 with ``tnequad``.  z2 is drawn even when EQUAD is zero, like the reference.

    ``psr`` may be a LIST of pulsars; ``efac`` / ``log10_equad`` / ``flags`` / ``seed`` are then one entry per pulsar (or one value
    for all; ``flags`` a list of flag lists): equivalent to the loop of single calls, in one launch."""
    if isinstance(psr, (list, tuple)):
        return _add_measurement_noise_list(list(psr), efac, log10_equad, flagid, flags, seed, tnequad)
    equad_str = "tnequad" if tnequad else "t2equad"
    if log10_equad is not None:
        equad = 10 ** log10_equad
    else:
        equad = 0.0
    if seed is not None:
        np.random.seed(seed)
    ntoas = psr.toas.ntoas
    efacvec, equadvec = _wn_vectors(psr, efac, equad, flagid, flags)
    sigma = errors_seconds(psr.toas)
    z1 = np.random.randn(ntoas)
    z2 = np.random.randn(ntoas)
    sig_d, ef_d, eq_d, z1_d, z2_d = dv.upload_packed([sigma, efacvec, equadvec, z1, z2])
    out = dv.empty((1, ntoas))
    _lib.call("pta_wn", dv.ptr(sig_d), dv.ptr(ef_d), dv.ptr(eq_d), ntoas, 1 if tnequad else 0, dv.ptr(z1_d), dv.ptr(z2_d),
              ntoas, 1, dv.ptr(out), ntoas, 0, dv.stream_ptr())
    dt = dv.download(out[0]) * u.s
    _record_wn(psr, efac, log10_equad, flags, equad_str, dt)


def _add_measurement_noise_list(psrs, efac, log10_equad, flagid, flags, seed, tnequad):
    P = len(psrs)
    equad_str = "tnequad" if tnequad else "t2equad"
    efacs, l10s = _broadcast(efac, P, "efac"), _broadcast(log10_equad, P, "log10_equad")
    if flags is not None and (len(flags) != P or not all(f is None or isinstance(f, (list, tuple, np.ndarray)) for f in flags)):
        raise ValueError("flags must be a per-pulsar list of flag lists")
    flagl = flags if flags is not None else [None] * P
    seeds = _seed_list(seed, P)
    counts = [p.toas.ntoas for p in psrs]
    vecs = []
    for a, p in enumerate(psrs):
        equad = _pow10(l10s[a]) if l10s[a] is not None else 0.0
        vecs.append(_wn_vectors(p, efacs[a], equad, flagid, flagl[a]))
    z = _legacy_normals(seeds, [[n, n] for n in counts])
    sigma = np.concatenate([errors_seconds(p.toas) for p in psrs])
    ntot = int(np.sum(counts))
    sig_d, ef_d, eq_d, z1_d, z2_d = dv.upload_packed([sigma, np.concatenate([v[0] for v in vecs]), np.concatenate([v[1] for v in vecs]),
                                                      np.concatenate([x[0] for x in z]), np.concatenate([x[1] for x in z])])
    out = dv.empty((1, ntot))
    _lib.call("pta_wn", dv.ptr(sig_d), dv.ptr(ef_d), dv.ptr(eq_d), ntot, 1 if tnequad else 0, dv.ptr(z1_d), dv.ptr(z2_d),
              ntot, 1, dv.ptr(out), ntot, 0, dv.stream_ptr())
    res = np.split(dv.download(out[0]), np.cumsum(counts)[:-1])
    for a, p in enumerate(psrs):
        _record_wn(p, efacs[a], l10s[a], flagl[a], equad_str, res[a] * u.s)


def _ecorr_vector(psr, ecorr, flagid, flags, first):
    """per-epoch ECORR of one pulsar with the reference's checks (white_noise.py:166-180)."""
    ne = len(first)
    ecorrvec = np.zeros(ne)
    if flags is None:
        if not np.isscalar(ecorr):
            raise ValueError("ERROR: If flags is None, jitter must be a scalar")
        ecorrvec = np.ones(ne) * ecorr
    if flags is not None and not np.isscalar(ecorr):
        if len(ecorr) == len(flags):
            labels, codes = flag_codes(psr.toas, flagid)
            ecorrvec = _per_flag_vector(labels, codes[first], flags, ecorr)   # first TOA labels the epoch (:35)
        else:
            raise ValueError("ERROR: flags must be same length as jitter")
    return ecorrvec


def _record_jitter(psr, log10_ecorr, flags, dt):
    if flags is None:
        psr.update_added_signals("{}_jitter".format(psr.name), {"log10_ecorr": log10_ecorr}, dt)
    else:
        psr.update_added_signals("{}_jitter".format(psr.name), {}, dt)
        for i, flag in enumerate(flags):
            psr.update_added_signals("{}_{}_jitter".format(psr.name, flag), {"log10_ecorr": log10_ecorr[i]})
    psr.toas.adjust_TOAs(TimeDelta(dt.to("day")))
    psr.update_residuals()


def add_jitter(psr, log10_ecorr, flagid="f", flags=None, coarsegrain=0.1, seed=None):
    """Add correlated (ECORR) noise of rms ecorr [s] per epoch, epochs = greedy buckets of width ``coarsegrain``
    days (white_noise.py:128-198).  ``psr`` may be a LIST of pulsars (per-pulsar ``log10_ecorr`` / ``flags`` / ``seed`` lists)."""
    if isinstance(psr, (list, tuple)):
        return _add_jitter_list(list(psr), log10_ecorr, flagid, flags, coarsegrain, seed)
    ecorr = 10 ** log10_ecorr
    if seed is not None:
        np.random.seed(seed)
    times = np.asarray(psr.toas.get_mjds().value, dtype=np.float64)
    epoch_of, first = epoch_map(times, coarsegrain)
    ne = len(first)
    ecorrvec = _ecorr_vector(psr, ecorr, flagid, flags, first)
    z = np.random.randn(ne)
    n = len(times)
    ep_d, ec_d, z_d = dv.upload_packed([epoch_of, ecorrvec, z])
    out = dv.empty((1, n))
    _lib.call("pta_ecorr", dv.ptr(ep_d), dv.ptr(ec_d), n, ne, dv.ptr(z_d), ne, 1, dv.ptr(out), n, 0, dv.stream_ptr())
    dt = u.s * dv.download(out[0])
    _record_jitter(psr, log10_ecorr, flags, dt)


def _add_jitter_list(psrs, log10_ecorr, flagid, flags, coarsegrain, seed):
    P = len(psrs)
    l10s = _broadcast(log10_ecorr, P, "log10_ecorr")
    if flags is not None and len(flags) != P:
        raise ValueError("flags must be a per-pulsar list of flag lists")
    flagl = flags if flags is not None else [None] * P
    seeds = _seed_list(seed, P)
    eps, vecs, nes = [], [], []
    for a, p in enumerate(psrs):
        times = np.asarray(p.toas.get_mjds().value, dtype=np.float64)
        epoch_of, first = epoch_map(times, coarsegrain)
        ecorr = _pow10(l10s[a])
        vecs.append(_ecorr_vector(p, ecorr, flagid, flagl[a], first))
        eps.append(epoch_of)
        nes.append(len(first))
    z = _legacy_normals(seeds, [[ne] for ne in nes])
    eoff = np.concatenate([[0], np.cumsum(nes)]).astype(np.int64)
    counts = [len(e) for e in eps]
    ntot, etot = int(np.sum(counts)), int(eoff[-1])
    ep_all = np.concatenate([e.astype(np.int64) + eoff[a] for a, e in enumerate(eps)]).astype(np.int32)
    ep_d, ec_d, z_d = dv.upload_packed([ep_all, np.concatenate(vecs), np.concatenate([x[0] for x in z])])
    out = dv.empty((1, ntot))
    _lib.call("pta_ecorr", dv.ptr(ep_d), dv.ptr(ec_d), ntot, etot, dv.ptr(z_d), etot, 1, dv.ptr(out), ntot, 0, dv.stream_ptr())
    res = np.split(dv.download(out[0]), np.cumsum(counts)[:-1])
    for a, p in enumerate(psrs):
        _record_jitter(p, l10s[a], flagl[a], u.s * res[a])


def add_efac(psr, efac=1.0, flagid="f", flags=None, seed=None):
    """libstempo-style alias: EFAC only (BASELINE.json names add_efac/add_ecorr)."""
    return add_measurement_noise(psr, efac=efac, log10_equad=None if flags is None else np.full(len(flags), -np.inf),
                                 flagid=flagid, flags=flags, seed=seed)


def add_equad(psr, log10_equad, flagid="f", flags=None, seed=None):
    """libstempo-style alias: EQUAD only (tnequad convention, EFAC = 0 on the sigma term)."""
    efac = 0.0 if flags is None else np.zeros(len(flags))
    return add_measurement_noise(psr, efac=efac, log10_equad=log10_equad, flagid=flagid, flags=flags, seed=seed, tnequad=True)


add_ecorr = add_jitter
