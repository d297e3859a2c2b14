"""Extract the dummy-argument name lists of the reference's entry points into tests/golden/signatures.json:
   noahmplsm           phys/module_sf_noahmpdrv.F90:11-46   (the WRF_HYDRO block of #ifdef lines kept apart)
   WTABLE_mmf_noahmp   phys/module_sf_noahmp_groundwater.F90:14-22
A fixture is data: names only, parsed from /root/reference by this script (run in the dev container).  The signature tests
compare the generated Fortran shim and the C-ABI argument blocks with THIS list, not with noahmp_amd/abi_spec.py."""
import json
import os
import re

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "signatures.json")


def dummy_list(path, name):
    """-> (names outside any #ifdef, {macro: names inside #ifdef macro}) of SUBROUTINE `name`'s dummy list."""
    lines = open(path, errors="replace").read().split("\n")
    start = next(i for i, ln in enumerate(lines) if re.match(r"\s*SUBROUTINE\s+%s\b" % name, ln, re.I))
    plain, cond, macro, text_end = [], {}, None, None
    depth = 0
    i = start
    first = True
    while True:
        ln = lines[i]
        if ln.lstrip().startswith("#"):
            m = re.match(r"\s*#\s*ifdef\s+(\w+)", ln)
            if m:
                macro = m.group(1)
            elif re.match(r"\s*#\s*endif", ln):
                macro = None
            i += 1
            continue
        code = ln.split("!")[0]
        if first:
            code = code[code.index("(") + 1:]
            depth = 1
            first = False
        for tok in re.findall(r"[A-Za-z_]\w*|\(|\)", code):
            if tok == "(":
                depth += 1
            elif tok == ")":
                depth -= 1
                if depth == 0:
                    text_end = i
                    break
            else:
                (cond.setdefault(macro, []) if macro else plain).append(tok.lower())
        if text_end is not None:
            break
        i += 1
    return plain, cond, (start + 1, text_end + 1)


def main():
    out = {}
    for key, rel, sub in (("noahmplsm", "phys/module_sf_noahmpdrv.F90", "noahmplsm"),
                          ("wtable_mmf_noahmp", "phys/module_sf_noahmp_groundwater.F90", "WTABLE_mmf_noahmp")):
        plain, cond, span = dummy_list(os.path.join(REF, rel), sub)
        out[key] = {"file": rel, "lines": list(span), "dummies": plain, "conditional": cond}
    json.dump(out, open(OUT, "w"), indent=1)
    for k, v in out.items():
        print(k, v["file"], v["lines"], len(v["dummies"]), "dummies", {m: len(n) for m, n in v["conditional"].items()})


if __name__ == "__main__":
    main()
