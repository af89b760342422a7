#!/usr/bin/env python3
"""Check (and, if need be, split) rocprofv3 --pmc counter sets against what one pass can collect.

Round 2 lost a 5-minute GPU step to one over-full set: `TA_BUSY_avr TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum
TA_DATA_STALLED_BY_TC_CYCLES_sum` asks for THREE hardware counters of the TA block (TA_TA_BUSY, TA_ADDR_STALLED_BY_TC_CYCLES,
TA_DATA_STALLED_BY_TC_CYCLES; the derived names expand to them), the block has two counter registers per instance, and
rocprofiler-sdk aborts inside the traced program's set-up ("error code 38: Request exceeds the capabilities of the hardware to
collect"), after which the profiler sat in its signal handler until the step's own limit
(gpurun_out/pmc_spmv_relat9/pass7.err).  This tool expands every name of a set through `rocprofv3 --list-avail`'s
expressions into hardware counters, counts them per block, and prints passes that fit -- greedy, in the order given.

  pmc_sets.py AVAIL "set one" "set two" ...      -> one line per pass (space-separated names), '#' lines = what was split
  pmc_sets.py AVAIL --check "set"                -> exit 1 if the set does not fit one pass

Per-block counter registers (gfx9-family, as rocprofiler's own block tables have them; TA's 2 is the one error 38 above
confirms): anything not listed is taken as 4.
"""
import re
import sys

LIMIT = {"TA": 2, "TD": 2, "TCP": 4, "TCC": 4, "TCA": 4, "SQ": 8, "GRBM": 2, "GRBM_SE": 2, "SPI": 2, "CPC": 2, "CPF": 2, "GDS": 4,
         "SQC": 8, "SDMA": 2}


def parse_avail(path):
    block, expr = {}, {}
    name = None
    for ln in open(path, errors="replace"):
        mt = re.match(r"Counter_Name\s*:\s*(\S+)", ln)
        if mt:
            name = mt.group(1)
            continue
        mt = re.match(r"Block\s*:\s*(\S+)", ln)
        if mt and name:
            block[name] = mt.group(1)
            continue
        mt = re.match(r"Expression\s*:\s*(.+)", ln)
        if mt and name:
            expr[name] = mt.group(1).strip()
    return block, expr


def hardware(name, block, expr, seen=None):
    """the hardware counters (name, block) a --pmc name costs"""
    seen = seen or set()
    if name in seen:
        return set()
    seen.add(name)
    if name in block:
        return {(name, block[name])}
    if name not in expr:
        return set()        # a constant (CU_NUM, SE_NUM ...) or unknown: costs nothing here, rocprofv3 will say if it is wrong
    out = set()
    for tok in re.findall(r"[A-Za-z_][A-Za-z0-9_]*", expr[name]):
        if tok in ("reduce", "sum", "avr", "max", "min", "accumulate", "select", "HIGH_RES", "LOW_RES", "NONE"):
            continue
        out |= hardware(tok, block, expr, seen)
    return out


def fits(names, block, expr):
    per = {}
    for nm in names:
        for hw, blk in hardware(nm, block, expr):
            per.setdefault(blk, set()).add(hw)
    over = {b: sorted(v) for b, v in per.items() if len(v) > LIMIT.get(b, 4)}
    return not over, over


def split(names, block, expr):
    passes, cur = [], []
    for nm in names:
        ok, _ = fits(cur + [nm], block, expr)
        if ok or not cur:
            cur.append(nm)
        else:
            passes.append(cur)
            cur = [nm]
    if cur:
        passes.append(cur)
    return passes


def main():
    if len(sys.argv) < 3:
        sys.exit(__doc__)
    block, expr = parse_avail(sys.argv[1])
    args = sys.argv[2:]
    if args[0] == "--check":
        ok, over = fits(args[1].split(), block, expr)
        if not ok:
            print("does not fit one pass:", over)
            sys.exit(1)
        return
    for s in args:
        names = s.split()
        unknown = [n for n in names if n not in block and n not in expr]
        if unknown:
            print(f"# unknown on this device, dropped: {' '.join(unknown)}")
            names = [n for n in names if n not in unknown]
        ps = split(names, block, expr)
        if len(ps) > 1:
            print(f"# split into {len(ps)} passes (per-block counter registers): {s}")
        for q in ps:
            if not fits(q, block, expr)[0]:
                print(f"# dropped, needs more registers of one block than a pass has: {' '.join(q)}")
                continue
            print(" ".join(q))


if __name__ == "__main__":
    main()
