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
CANOPY = ("canwat", "canliqxy", "canicexy", "fwetxy")
EXCH = ("chvxy", "chbxy", "chleafxy", "chucxy", "chv2xy", "chb2xy", "cmxy", "chxy", "rssunxy",
        "rsshaxy")


def tolerance(name, steps=1):
    if name in FLUX:
        return (1e-3, 0.05) if steps > 1 else (2e-4, 5e-3)
    if name in RATE:
        return (1e-3, 2e-8) if steps > 1 else (2e-4, 1e-9)
    if name in EXCH:
        return (2e-3, 1e-7) if steps > 1 else (2e-4, 1e-8)
    if name == "fwetxy":    # (CANLIQ/MAXLIQ)**0.667: amplifies the absolute noise of a near-empty canopy store
        return (5e-3, 2e-3) if steps > 1 else (1e-3, 5e-4)
    if name in CANOPY:      # small residues of interception minus an evaporation FLUX (x dt / HVAP)
        return (5e-3, 1e-4) if steps > 1 else (1e-3, 1e-5)
    if name == "eahxy":     # canopy-air vapour pressure [Pa], a Newton-loop by-product like the fluxes
        return (1e-3, 0.1) if steps > 1 else (1e-4, 1e-2)
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
        with np.errstate(invalid="ignore"):
            d = np.abs(x - y)
        d[both_nan | (x == y)] = 0.0          # NaN==NaN and inf==inf count as agreement
        lim = at + rt * np.nan_to_num(np.maximum(np.abs(x), np.abs(y)), nan=0.0, posinf=0.0)
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


# ---------------------------------------------------------------------------------------------
# Parity envelope for float32 engines whose libm differs from the reference's (GPU vs CPU).
#
# Measured floor (tests/golden): the reference compiled -O2 against ITSELF compiled -O0,
# one step from an identical state on the 512-column mixed tile, already differs in a few columns by
#   TSK/TV/TAH 8e-3 K, HFX 0.06 W m-2 (rel 8e-3), CM/CH/CHV 2.2e-2 rel, CHV2 1.5e-2 rel, EAH 0.08 Pa,
# because the canopy Newton loop stops on |DTV| <= 0.01 after >= 5 iterations (lsm:3487) and an
# ulp-level difference flips the iteration count.  The contract therefore has two parts:
#   (1) TIGHT  : |d| <= atol + rtol*|x| with the per-group tolerances of `tolerance()` must hold for
#                at least (1 - frac) of the entries of every field, and
#   (2) ENVELOPE: every entry must stay inside a loose physical envelope (no blow-ups, no wrong branch
#                with a visible effect).
TEMP = ("tsk", "tvxy", "tgxy", "tahxy", "tslb", "tsnoxy", "t2mvxy", "t2mbxy", "tradxy", "tgvxy", "tgbxy")


def envelope(name, steps=1):
    """Hard cap for every entry.  Sized ~3x the largest single-step deviation seen over 196 608
    column-steps for BOTH pairs (reference -O2 vs -O0: TSK 0.03 K, HFX 0.65, SHG 1.3 W m-2, CH 3e-2 rel;
    HIP vs oracle: TSK 0.07 K, HFX 4.0, SHC 4.1 W m-2, EAH 15 Pa) -- profiles/r01_parity_stats.md."""
    k = 1.0 if steps == 1 else 3.0
    if name in TEMP:
        return (0.0, 0.5 * k)
    if name in FLUX:
        return (5e-2, 15.0 * k)
    if name in EXCH:
        return (0.5, 1e-6)
    if name in RATE:
        return (0.2, 5e-6 * k)
    if name == "eahxy":
        return (0.0, 50.0 * k)
    if name in ("snow", "sneqvoxy", "snicexy", "snliqxy", "acsnom", "canwat", "canicexy", "canliqxy", "acsnow"):
        return (2e-2 * k, 2e-2 * k)
    if name in ("fwetxy",):
        return (0.0, 0.1 * k)
    return (5e-3 * k, 5e-3 * k)


def parity_check(ref, test, steps=1, frac=0.04, frac_medium=0.02, fields=None, mask=None, skip=()):
    """-> (ok, lines).  Three nested criteria per field:
         TIGHT    tolerance()      may be exceeded by at most `frac` of the entries (min 3 entries),
         MEDIUM   10 x TIGHT       by at most `frac_medium` (min 3 entries),
         ENVELOPE envelope()       by none."""
    names = fields or [n for n in ref.a if FIELD_INFO[n][2] != "in"]
    ok, lines = True, []
    for n in names:
        if n in skip or n == "dzs":
            continue
        x, y = np.asarray(ref.a[n]), np.asarray(test.a[n])
        if mask is not None:
            m = mask if x.ndim == 2 else np.broadcast_to(mask[:, None, :], x.shape)
            x, y = x[m], y[m]
        if x.dtype.kind == "i":
            nb = int((x != y).sum())
            if nb > max(frac_medium * x.size, 1):
                ok = False
                lines.append("%-12s integer mismatches %d/%d" % (n, nb, x.size))
            continue
        x = x.astype(np.float64)
        y = y.astype(np.float64)
        with np.errstate(invalid="ignore"):
            d = np.abs(x - y)
        d[(np.isnan(x) & np.isnan(y)) | (x == y)] = 0.0     # NaN==NaN and inf==inf count as agreement
        mag = np.nan_to_num(np.maximum(np.abs(x), np.abs(y)), nan=0.0, posinf=0.0)
        rt, at = tolerance(n, steps)
        nt = int((~(d <= at + rt * mag)).sum())
        nm = int((~(d <= 10 * (at + rt * mag))).sum())
        re_, ae = envelope(n, steps)
        ne = int((~(d <= ae + re_ * mag)).sum())
        if nt > max(frac * d.size, 3) or nm > max(frac_medium * d.size, 5) or ne:
            ok = False
            lines.append("%-12s tight-viol %d  medium-viol %d  envelope-viol %d  of %d; max|d|=%.3e  (nan ref/test %d/%d)"
                         % (n, nt, nm, ne, d.size, np.nanmax(d), int(np.isnan(x).sum()), int(np.isnan(y).sum())))
    return ok, lines


def exact_check(ref, test, fields=None, skip=(), allow_cols=0):
    """Bit-for-bit comparison (NaN == NaN).  Returns (ok, lines).  `allow_cols` is 0 everywhere in the suite: the one host-dependent
    operation, libm's expf (two builds that differ at two arguments), is pinned on both sides (oracle/nmp_pin_expf.c)."""
    import numpy as np
    from noahmp_amd.abi import FIELD_INFO
    names = fields or [k for k in ref.a if k in FIELD_INFO and FIELD_INFO[k][2] != "in"]
    names = [n for n in names if n in ref.a and n in test.a]
    lines, cols = [], None
    for n in names:
        if n in skip:
            continue
        x, y = np.asarray(ref.a[n]), np.asarray(test.a[n])
        if x.dtype.kind == "f":
            ne = ~((x.view(np.uint32) == y.view(np.uint32)) | (np.isnan(x) & np.isnan(y)))
        else:
            ne = x != y
        if ne.any():
            c = ne.any(axis=1) if ne.ndim == 3 else ne
            cols = c if cols is None else (cols | c)
            idx = tuple(np.argwhere(ne)[0])
            lines.append("%s: %d entries differ, first at %s: %r vs %r" % (n, int(ne.sum()), idx, x[idx], y[idx]))
    nbad = int(cols.sum()) if cols is not None else 0
    if nbad:
        lines.insert(0, "%d column(s) not bit-identical (allowed %d)" % (nbad, allow_cols))
    return nbad <= allow_cols, lines
