"""Field-by-field comparison of two ColumnStores with the per-variable tolerances of DESIGN.md.

Used by the parity tests (HIP vs oracle, oracle vs reference) and by ad-hoc debugging.
"""
import numpy as np

from noahmp_amd.abi import FIELD_INFO

# (rtol, atol) per field group -- see DESIGN.md "Parity contract" (derived from SURVEY 8c:
# the float32 reference is only reproducible to ~7e-6 in states and ~2e-4 / 0.02 W m-2 in
# fluxes against itself at a different optimisation level).
FLUX = ("hfx", "lh", "grdflx", "qfx", "firaxy", "fsaxy", "savxy", "sagxy", "shgxy", "shcxy", "shbxy",
        "evgxy", "evbxy", "ghvxy", "ghbxy", "irgxy", "ircxy", "irbxy", "trxy", "evcxy", "aparxy",
        "psnxy", "neexy", "gppxy", "nppxy")
RATE = ("runsfxy", "runsbxy", "ecanxy", "edirxy", "etranxy", "qsnowxy")
EXCH = ("chvxy", "chbxy", "chleafxy", "chucxy", "chv2xy", "chb2xy", "cmxy", "chxy", "rssunxy",
        "rsshaxy")


def tolerance(name, steps=1):
    if name in FLUX:
        return (1e-3, 0.05) if steps > 1 else (2e-4, 5e-3)
    if name in RATE:
        return (1e-3, 2e-8) if steps > 1 else (2e-4, 1e-9)
    if name in EXCH:
        return (2e-3, 1e-7) if steps > 1 else (2e-4, 1e-8)
    return (2e-5, 2e-5) if steps > 1 else (1e-5, 1e-6)


def compare(a, b, fields=None, steps=1, mask=None, verbose=False, skip=()):
    """Return list of (field, max_abs, max_rel, n_bad) for fields that violate tolerance."""
    bad = []
    names = fields or [n for n in a.a if FIELD_INFO[n][2] != "in"]
    for n in names:
        if n in skip or n == "dzs":
            continue
        x, y = np.asarray(a.a[n]), np.asarray(b.a[n])
        if mask is not None:
            m = mask if x.ndim == 2 else np.broadcast_to(mask[:, None, :], x.shape)
            x, y = x[m], y[m]
        if x.dtype.kind == "i":
            nb = int((x != y).sum())
            if nb:
                bad.append((n, float(np.abs(x - y).max()), 0.0, nb))
            continue
        x = x.astype(np.float64)
        y = y.astype(np.float64)
        rt, at = tolerance(n, steps)
        both_nan = np.isnan(x) & np.isnan(y)
        d = np.abs(x - y)
        d[both_nan] = 0.0
        lim = at + rt * np.maximum(np.abs(x), np.abs(y))
        viol = ~(d <= lim)
        if verbose or viol.any():
            rel = d / np.maximum(np.maximum(np.abs(x), np.abs(y)), 1e-30)
            if viol.any():
                bad.append((n, float(np.nanmax(d)), float(np.nanmax(rel[viol])), int(viol.sum())))
            elif verbose:
                print("  ok %-12s maxabs %.3e" % (n, float(np.nanmax(d)) if d.size else 0.0))
    return bad


def report(bad):
    return "\n".join("%-12s max|d|=%.4e maxrel=%.3e n_bad=%d" % b for b in bad)
