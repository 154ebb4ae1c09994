"""Helpers shared by tests/golden/gen_goldens.py and the tests (TEST INFRASTRUCTURE ONLY).

Large tensors (parameter gradients, updated weights at ngf=64) are kept in the
fixtures in a compact form so that the committed files stay small: a strided
sample of the flattened tensor plus its sum and L2 norm."""
import numpy as np

FULL_LIMIT = 4096      # tensors up to this many elements are stored whole
N_SAMPLE = 1024


def compact(a):
    """dict of arrays describing `a`: whole tensor if small, else sample + moments."""
    a = np.asarray(a)
    flat = a.reshape(-1).astype(np.float64)
    if flat.size <= FULL_LIMIT:
        return {"full": a}
    stride = flat.size // N_SAMPLE
    return {"sample": flat[::stride][:N_SAMPLE].astype(np.float32),
            "sum": np.array(flat.sum()), "l2": np.array(np.sqrt((flat * flat).sum()))}


def put(out, key, a, small):
    if small:
        out[key] = np.asarray(a)
    else:
        for k, v in compact(a).items():
            out[key + "#" + k] = v


def get(npz, key):
    """Return ('full', array) or ('compact', dict(sample, sum, l2))."""
    if key in npz:
        return "full", npz[key]
    if key + "#full" in npz:
        return "full", npz[key + "#full"]
    return "compact", {k: npz[key + "#" + k] for k in ("sample", "sum", "l2")}


def compare(npz, key, got, rtol, atol, atol_of_max=0.0):
    """Max error of `got` against fixture entry `key`; returns (ok, message).
    Allowed error per element: atol + atol_of_max * max|ref| + rtol * |ref|."""
    kind, ref = get(npz, key)
    got = np.asarray(got, dtype=np.float64)
    if kind == "full":
        ref = ref.astype(np.float64)
        atol = atol + atol_of_max * (float(np.abs(ref).max()) if ref.size else 0.0)
        if ref.shape != got.shape:
            return False, "%s: shape %s vs %s" % (key, got.shape, ref.shape)
        err = np.abs(got - ref)
        tol = atol + rtol * np.abs(ref)
        bad = err > tol
        return (not bad.any()), "%s: max abs err %.3e (ref max %.3e)" % (
            key, float(err.max()) if err.size else 0.0, float(np.abs(ref).max()) if ref.size else 0.0)
    c = compact(got)
    atol = atol + atol_of_max * float(np.abs(ref["sample"]).max())
    err = np.abs(c["sample"].astype(np.float64) - ref["sample"].astype(np.float64))
    tol = atol + rtol * np.abs(ref["sample"])
    ok = not (err > tol).any()
    l2r, l2g = float(ref["l2"]), float(c["l2"])
    ok = ok and abs(l2r - l2g) <= atol + 10 * rtol * abs(l2r)
    return ok, "%s: sample max abs err %.3e, l2 %.6e vs %.6e" % (key, float(err.max()), l2g, l2r)


def compare_update(npz, key, got, init, lr, steps, frac_ok=0.97, grad_key=None, noise_rel=1e-5):
    """Compare an AdamW-updated tensor against the fixture.  At the first steps
    AdamW moves every element by about +-lr whatever the gradient's size, so an
    element whose true gradient is zero up to rounding (a conv bias in front of a
    BatchNorm) moves in a direction that rounding noise decides.  The check is
    therefore: at least frac_ok of the elements moved as the reference's did
    (within 5 % of lr), and none is further than a full sign flip away.
    grad_key: fixture entry of the reference's gradient of the FIRST step.  Elements whose
    reference gradient is below noise_rel of the tensor's largest (f32 sums of O(max) terms carry a
    rounding error of that order, so their sign is not a property of the computation) are left out of
    the agreeing fraction; the sign-flip bound still holds for them."""
    kind, ref = get(npz, key)
    got = np.asarray(got, dtype=np.float64)
    init = np.asarray(init, dtype=np.float64)
    if kind == "full":
        r = ref.astype(np.float64).reshape(-1)
        g = got.reshape(-1)
    else:
        r = ref["sample"].astype(np.float64)
        g = compact(got)["sample"].astype(np.float64)
    err = np.abs(g - r)
    judged = np.ones(err.shape, dtype=bool)
    if grad_key is not None:
        gk, gref = get(npz, grad_key)
        gr = (gref.astype(np.float64).reshape(-1) if gk == "full" else gref["sample"].astype(np.float64))
        if gr.shape == err.shape and gr.size:
            judged = np.abs(gr) > noise_rel * float(np.abs(gr).max())
    worst = float(err.max()) if err.size else 0.0
    close = float((err[judged] <= 0.05 * lr * steps + 1e-6 * np.abs(r[judged])).mean()) if judged.any() else 1.0
    ok = close >= frac_ok and worst <= 2.2 * lr * steps + 1e-5 * float(np.abs(r).max() if r.size else 0)
    return ok, "%s: %.2f%% of the %d judged elements agree, worst %.3e (lr %.1e)" % (key, 100 * close, int(judged.sum()), worst, lr)
