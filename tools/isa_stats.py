"""Static instruction mix of one kernel in a `hipcc -S --cuda-device-only` listing.

usage: isa_stats.py listing.s [kernel-name-substring]      (default: the class-range kernel noahmp_ranges_kernel)
Prints resource usage (.vgpr_count, spills, LDS, scratch) and counts by class: VALU total, float64, packed float32, moves,
selects, IEEE-division parts, lane moves of spilled SGPRs, transcendentals, conversions; s_nop, s_waitcnt, LDS and memory
instructions; calls (s_swappc: none may exist in a column kernel, profiles/r03_experiments.md section 3d)."""
import collections
import re
import sys


def kernel_body(lines, key):
    name, start = None, None
    for i, ln in enumerate(lines):
        m = re.match(r"^(_Z\w+):", ln)
        if m and key in m.group(1) and ("noahmp_column_kernel" in m.group(1) or "noahmp_ranges_kernel" in m.group(1)):
            name, start = m.group(1), i
            break
    if start is None:
        raise SystemExit("kernel not found: " + key)
    body = []
    for ln in lines[start + 1:]:
        if ln.startswith(".Lfunc_end"):
            break
        body.append(ln)
    return name, body


def main():
    path = sys.argv[1]
    key = sys.argv[2] if len(sys.argv) > 2 else "noahmp_ranges_kernel"        # the one-launch class-range kernel (land + land-ice + skipped bodies)
    lines = open(path).read().splitlines()
    name, body = kernel_body(lines, key)
    ops = collections.Counter()
    for ln in body:
        t = ln.strip()
        if not t or t.startswith((";", ".", "//")) or t.endswith(":"):
            continue
        ops[t.split()[0]] += 1
    valu = {k: v for k, v in ops.items() if k.startswith("v_")}
    tot = sum(valu.values())

    def cnt(pred):
        return sum(v for k, v in valu.items() if pred(k))
    rows = [("VALU total", tot),
            ("  float64 (v_*_f64, cvt to/from f64)", cnt(lambda k: "f64" in k)),
            ("  packed float32 (v_pk_*)", cnt(lambda k: k.startswith("v_pk_"))),
            ("  v_mov / v_accvgpr", cnt(lambda k: k.startswith(("v_mov", "v_accvgpr")))),
            ("  v_cndmask", cnt(lambda k: k.startswith("v_cndmask"))),
            ("  v_cmp*", cnt(lambda k: k.startswith("v_cmp"))),
            ("  v_div_scale / fmas / fixup (f32)", cnt(lambda k: k.startswith("v_div_") and "f32" in k)),
            ("  v_rcp / v_sqrt / v_rsq / v_exp / v_log (transcendental unit)", cnt(lambda k: re.match(r"v_(rcp|sqrt|rsq|exp|log)_", k) is not None)),
            ("  v_readlane / v_writelane (spilled SGPRs)", cnt(lambda k: k.startswith(("v_readlane", "v_writelane", "v_readfirstlane")))),
            ("  v_cvt_*", cnt(lambda k: k.startswith("v_cvt"))),
            ("  v_fma_f32 / v_mul_f32 / v_add_f32 / v_sub_f32 (+ mac, fmac)", cnt(lambda k: re.match(r"v_(fma|mul|add|sub|subrev|mac|fmac)_f32", k) is not None))]
    print(name)
    for ln in lines:
        if name in ln and ".name:" in ln:
            i = lines.index(ln)
            for q in lines[i - 12:i + 12]:
                if re.search(r"\.(sgpr_count|sgpr_spill_count|vgpr_count|vgpr_spill_count|group_segment_fixed_size|private_segment_fixed_size):", q):
                    print("   ", q.strip())
            break
    for k, v in rows:
        print("%-66s %6d  %5.1f %%" % (k, v, 100.0 * v / tot))
    for k in ("s_nop", "s_waitcnt", "s_swappc_b64", "s_cbranch_execz", "s_cbranch_execnz", "s_cbranch_vccz", "s_cbranch_vccnz", "s_cbranch_scc0", "s_cbranch_scc1"):
        print("%-66s %6d" % (k, ops.get(k, 0)))
    print("%-66s %6d" % ("SALU + SMEM (s_*)", sum(v for k, v in ops.items() if k.startswith("s_"))))
    print("%-66s %6d" % ("ds_read* / ds_write*", sum(v for k, v in ops.items() if k.startswith("ds_"))))
    print("%-66s %6d" % ("global / flat / buffer / scratch", sum(v for k, v in ops.items() if k.startswith(("global_", "flat_", "buffer_", "scratch_")))))
    if len(sys.argv) > 3:
        for k, v in sorted(valu.items(), key=lambda kv: -kv[1])[:int(sys.argv[3])]:
            print("   %-40s %6d" % (k, v))


if __name__ == "__main__":
    main()
