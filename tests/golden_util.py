"""Loading and checking tests/golden/golden.npz and golden_half.npz (vectors produced by the
reference's classes)."""

import hashlib
import json
import os

import numpy as np

_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
GOLDEN = os.path.join(_DIR, "golden.npz")
GOLDEN_FILES = [GOLDEN, os.path.join(_DIR, "golden_half.npz")]
_cache = {}


class _Merged:
    """The arrays of several .npz files behind one `files` / `[]` (scenario ids are unique)."""

    def __init__(self, zs):
        self._where = {}
        for z in zs:
            for f in z.files:
                self._where[f] = z
        self.files = set(self._where)

    def __getitem__(self, key):
        return self._where[key][key]


def load():
    if "z" not in _cache:
        zs = [np.load(f, allow_pickle=False) for f in GOLDEN_FILES]
        index = {}
        for z in zs:
            index.update(json.loads(bytes(z["__index__"]).decode()))
        _cache["z"], _cache["index"] = _Merged(zs), index
    return _cache["z"], _cache["index"]


def record_hash(d) -> str:
    h = hashlib.sha256()
    for k in sorted(d):
        a = np.ascontiguousarray(d[k])
        h.update(k.encode())
        h.update(str(a.dtype).encode())
        h.update(str(a.shape).encode())
        h.update(a.tobytes())
    return h.hexdigest()


def _bits(a):
    a = np.ascontiguousarray(a)
    if a.dtype == np.float32:
        return a.view(np.uint32)
    return a.view(np.uint64) if a.dtype == np.float64 else a


def canonical_nan(a):
    """Same array with every NaN replaced by one quiet NaN: x86 and gfx950 produce NaNs of
    different sign / payload for the same invalid operation (inf - inf), so bit comparisons
    between host and device treat all NaNs as equal."""
    a = np.ascontiguousarray(a)
    if a.dtype.kind == "f" and np.isnan(a).any():
        a = a.copy()
        a[np.isnan(a)] = a.dtype.type("nan")
    return a


def same_bits(a, b) -> bool:
    a, b = canonical_nan(a), canonical_nan(b)
    return a.dtype == b.dtype and a.shape == b.shape and np.array_equal(_bits(a), _bits(b))


# Element-wise companion of the tensor-scale tolerance: every element within ELEM_RTOL of ITS OWN
# magnitude, plus one unit in the last place of the tensor's largest magnitude (elements that
# nearly cancel carry the absolute error of their neighbours; measured worst case over all sqrt
# scenarios: 8.2e-5 relative / 1024 ulp on such an element, 0.11 of this bound).
ELEM_RTOL = 1e-5
worst_seen = {"scale_rel": 0.0, "elem_rel": 0.0, "ulp": 0, "bound_frac": 0.0}


def ulp_distance(a, b):
    """Distance in units in the last place between two float32 arrays (finite values)."""
    ai = np.ascontiguousarray(a, np.float32).view(np.int32).astype(np.int64)
    bi = np.ascontiguousarray(b, np.float32).view(np.int32).astype(np.int64)
    ai = np.where(ai < 0, -(ai & 0x7FFFFFFF), ai)
    bi = np.where(bi < 0, -(bi & 0x7FFFFFFF), bi)
    return np.abs(ai - bi)


def _close(got, want, rtol):
    """|got - want| <= rtol * max|finite want| for every element AND
    |got - want| <= ELEM_RTOL * |want| + 2**-23 * max|finite want| element by element;
    inf / nan patterns must coincide.  Returns (ok, error relative to the tensor's scale)."""
    got, want = np.asarray(got), np.asarray(want)
    if want.dtype.kind != "f":
        return np.array_equal(got, want), 0.0
    fin = np.isfinite(want)
    if not np.array_equal(fin, np.isfinite(got)):
        return False, float("inf")
    if not np.array_equal(got[~fin], want[~fin], equal_nan=True):
        return False, float("inf")
    if not fin.any():
        return True, 0.0
    g, w = got[fin].astype(np.float64), want[fin].astype(np.float64)
    scale = float(np.max(np.abs(w))) or 1.0
    diff = np.abs(g - w)
    err = float(np.max(diff)) / scale
    frac = float(np.max(diff / (ELEM_RTOL * np.abs(w) + 2.0 ** -23 * scale)))
    nz = w != 0
    worst_seen["scale_rel"] = max(worst_seen["scale_rel"], err)
    worst_seen["bound_frac"] = max(worst_seen["bound_frac"], frac)
    if nz.any():
        worst_seen["elem_rel"] = max(worst_seen["elem_rel"], float(np.max(diff[nz] / np.abs(w[nz]))))
    worst_seen["ulp"] = max(worst_seen["ulp"], int(ulp_distance(got[fin], want[fin]).max()))
    return err <= rtol and frac <= 1.0, err


RESIDUAL_LABEL, RESIDUAL_RTOL = ".residuals", 1e-5


def check_records(sc_id, rec, rtol=None):
    """Comparison of a scenario's records with the golden file: bit-exact when rtol is None,
    else every stored tensor within rtol of the tensor's largest magnitude.  Records whose label
    ends in RESIDUAL_LABEL (norms the reference sums in torch's own order) take RESIDUAL_RTOL."""
    sc_rtol = rtol
    z, index = load()
    want = index[sc_id]
    assert [r["label"] for r in want] == [lbl for lbl, _ in rec], \
        f"{sc_id}: record labels differ: {[lbl for lbl, _ in rec]} vs {[r['label'] for r in want]}"
    worst = 0.0
    for i, ((label, d), w) in enumerate(zip(rec, want)):
        assert sorted(d) == w["names"], f"{sc_id}/{label}: keys {sorted(d)} vs {w['names']}"
        if record_hash(d) == w["sha256"]:
            continue
        rtol = RESIDUAL_RTOL if label.endswith(RESIDUAL_LABEL) else sc_rtol
        for k in sorted(d):
            key, sub = f"{sc_id}|{i}|{k}", False
            if key not in z.files and key + "|s97" in z.files:
                key, sub = key + "|s97", True
            if key not in z.files:
                continue
            g = z[key]
            mine = d[k].ravel()[::97] if sub else d[k]
            assert mine.dtype == g.dtype and mine.shape == g.shape, \
                f"{sc_id}/{label}/{k}: {mine.dtype}{mine.shape} vs {g.dtype}{g.shape}"
            if rtol is None:
                bad = np.nonzero(_bits(mine).ravel() != _bits(g).ravel())[0]
                assert bad.size == 0, (
                    f"{sc_id}/{label}/{k}: {bad.size} of {g.size} elements differ; first {bad[:4]}: "
                    f"got {mine.ravel()[bad[:4]]} want {g.ravel()[bad[:4]]}")
            else:
                ok, err = _close(mine, g, rtol)
                worst = max(worst, err)
                assert ok, (f"{sc_id}/{label}/{k}: error {err:.3g} of the tensor's scale (limit {rtol}) "
                            f"or an element beyond {ELEM_RTOL} of its own magnitude + 1 ulp of the scale")
        assert rtol is not None, f"{sc_id}/{label}: hash differs (in a tensor stored by hash only)"
    return worst
