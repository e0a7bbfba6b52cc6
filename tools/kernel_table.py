"""Per-kernel table for profiles/: time from a rocprofv3 kernel trace, HBM traffic and MFMA-busy from separate --pmc
passes of the same command.
usage: kernel_table.py <trace.db> <fetch.db> <write.db> <mfma.db> <warm-up cycles to skip> > table.md

  HBM GB/s   = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 / avg duration      (KB counters; FETCH_SIZE counts 64 B per
               128-B request of a wide stream on gfx950, MI355X_MICROARCH.md "HBM": doubled)
  MFMA busy  = SQ_VALU_MFMA_BUSY_CYCLES / (avg duration x 2.4 GHz x 1024 SIMDs)   (the counter adds 32 cycles per
               v_mfma_f32_32x32x16_bf16 on the issuing SIMD: 100 % = every SIMD issuing MFMAs back to back = 2.5 PF dense)
Counter passes serialise kernels; durations come from the un-instrumented trace only.
"""
import sqlite3
import sys


def trace(db, skip_cycles=0, adam_per_cycle=3):
    """-> ({kernel: (count, total us, avg us)}, number of cycles counted).  Steady state only: the first `skip_cycles`
    training cycles (cache creation, pack-entry set-up) are dropped; a cycle ends with its `adam_per_cycle`-th Adam launch
    (three optimizer groups per G+D cycle)."""
    cur = sqlite3.connect(db).cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    name = "name" if "name" in cols else [c for c in cols if "name" in c][0]
    rows = cur.execute(f"select {name}, start, end from kernels order by start").fetchall()
    out, adam, cycles = {}, 0, 0
    for n, st, en in rows:
        cyc = adam // adam_per_cycle
        if "adam_dev_kernel" in n or "adam_dev4_kernel" in n or n.startswith("adam_kernel"):      # (not adam_prep_kernel: one per optimizer step too)
            adam += 1
        if cyc < skip_cycles:
            continue
        c, s = out.get(n, (0, 0.0))
        out[n] = (c + 1, s + (en - st) / 1e3)
    total_cycles = adam // adam_per_cycle
    return {n: (c, s, s / c) for n, (c, s) in out.items()}, max(1, total_cycles - skip_cycles)


def counters(db, names):
    cur = sqlite3.connect(db).cursor()
    out = {}
    for k, c, n, v in cur.execute("select kernel_name, counter_name, count(*), avg(value) from counters_collection "
                                  "group by kernel_name, counter_name"):
        if c in names:
            out.setdefault(k, {})[c] = v
    return out


def main():
    skip = int(sys.argv[5]) if len(sys.argv) > 5 else 0          # warm-up cycles to drop from the trace
    tr, steps = trace(sys.argv[1], skip_cycles=skip)
    f = counters(sys.argv[2], ("FETCH_SIZE",))
    w = counters(sys.argv[3], ("WRITE_SIZE",))
    m = counters(sys.argv[4], ("SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CYCLES"))
    steps = float(steps)
    total = sum(v[1] for v in tr.values())
    print("| kernel | launches/step | avg us | ms/step | % of step | HBM MB/launch (2xFETCH+WRITE) | HBM GB/s | MFMA busy % |")
    print("|---|---|---|---|---|---|---|---|")
    for n, (c, tot_us, avg_us) in sorted(tr.items(), key=lambda kv: -kv[1][1]):
        if tot_us / total < 0.002:
            continue
        fs, wsz = f.get(n, {}).get("FETCH_SIZE"), w.get(n, {}).get("WRITE_SIZE")
        mb = (2 * fs + wsz) * 1024 / 1e6 if fs is not None and wsz is not None else None
        gbps = mb * 1e6 / (avg_us * 1e-6) / 1e9 if mb is not None and avg_us > 0 else None
        busy = m.get(n, {}).get("SQ_VALU_MFMA_BUSY_CYCLES")
        pct = 100.0 * busy / (avg_us * 1e-6 * 2.4e9 * 1024) if busy is not None and avg_us > 0 else None
        short = n if len(n) < 84 else n[:81] + "..."
        print(f"| `{short}` | {c / steps:.1f} | {avg_us:.1f} | {tot_us / steps / 1e3:.3f} | {100 * tot_us / total:.1f} | "
              f"{'-' if mb is None else f'{mb:.1f}'} | {'-' if gbps is None else f'{gbps:.0f}'} | "
              f"{'-' if pct is None else f'{pct:.1f}'} |")
    hbm = 0.0
    for n, (c, tot_us, avg_us) in tr.items():
        fs, wsz = f.get(n, {}).get("FETCH_SIZE"), w.get(n, {}).get("WRITE_SIZE")
        if fs is not None and wsz is not None:
            hbm += (2 * fs + wsz) * 1024 * c / steps
    print(f"\nkernel time per step: {total / steps / 1e3:.3f} ms over {steps:.0f} steady-state steps ({skip} warm-up cycles dropped); "
          f"HBM traffic per step (counter averages x launches): {hbm / 1e9:.2f} GB")
    # the three totals VERDICT round 5 (item 2) asks for, steady state, every kernel (also the rows below the 0.2 % cut):
    import re
    gemm = lambda n: bool(re.search(r"gconv_|wgrad_(s1|s2|kernel|im2col|c3|pipe)", n)) and "finalize" not in n and "reduce" not in n
    nl = sum(c for c, _, _ in tr.values()) / steps
    gl = sum(c for n, (c, _, _) in tr.items() if gemm(n)) / steps
    gms = sum(t for n, (_, t, _) in tr.items() if gemm(n)) / steps / 1e3
    short = sum(c for c, _, a in tr.values() if a < 10.0) / steps
    print(f"launches per step: {nl:.0f} (GEMM-class {gl:.0f} = {gms:.2f} ms; non-GEMM {nl - gl:.0f} = "
          f"{total / steps / 1e3 - gms:.2f} ms; {short:.0f} launches of kernels that average under 10 us)")


if __name__ == "__main__":
    main()
