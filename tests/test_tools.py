"""tools/pmc_sets.py: rocprofv3 counter sets are checked against the per-block counter registers before a pass is started.

Round 2 lost a GPU step to a set that asked for three hardware counters of the TA block (two registers per instance):
rocprofiler-sdk aborts with error 38 inside the traced program.  The checker expands derived names through the device's own
`--list-avail` text, counts hardware counters per block and splits what does not fit.  CPU only: a small stand-in for the
list, in the format rocprofv3 prints.
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOOL = os.path.join(ROOT, "tools", "pmc_sets.py")

AVAIL = """GPU:0
Name:gfx950
Counter_Name        :	TA_TA_BUSY
Description         :	TA block is busy.
Block               :	TA
Dimensions          :	DIMENSION_INSTANCE[0:8]


Counter_Name        :	TA_BUSY_avr
Description         :	TA block is busy. Average over TA instances.
Expression          :	reduce(TA_TA_BUSY,avr)
Dimensions          :	DIMENSION_INSTANCE[0:0]


Counter_Name        :	TA_TA_BUSY_sum
Description         :	sum
Expression          :	reduce(TA_TA_BUSY,sum)


Counter_Name        :	TA_ADDR_STALLED_BY_TC_CYCLES
Description         :	x
Block               :	TA


Counter_Name        :	TA_ADDR_STALLED_BY_TC_CYCLES_sum
Description         :	x
Expression          :	reduce(TA_ADDR_STALLED_BY_TC_CYCLES,sum)


Counter_Name        :	TA_DATA_STALLED_BY_TC_CYCLES
Description         :	x
Block               :	TA


Counter_Name        :	TA_DATA_STALLED_BY_TC_CYCLES_sum
Description         :	x
Expression          :	reduce(TA_DATA_STALLED_BY_TC_CYCLES,sum)


Counter_Name        :	SQ_ACTIVE_INST_VALU
Description         :	x
Block               :	SQ


Counter_Name        :	GRBM_GUI_ACTIVE
Description         :	x
Block               :	GRBM


Counter_Name        :	VALUBusy
Description         :	x
Expression          :	100*reduce(SQ_ACTIVE_INST_VALU,sum)/CU_NUM/reduce(GRBM_GUI_ACTIVE,max)
"""


def run(tmp_path, *args):
    avail = tmp_path / "avail.txt"
    avail.write_text(AVAIL)
    return subprocess.run([sys.executable, TOOL, str(avail), *args], capture_output=True, text=True, timeout=60)


def test_the_set_that_aborted_the_profiler_is_split(tmp_path):
    bad = "TA_BUSY_avr TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum"
    r = run(tmp_path, "--check", bad)
    assert r.returncode == 1 and "TA" in r.stdout                       # three hardware counters of a block with two registers
    r = run(tmp_path, bad)
    passes = [ln for ln in r.stdout.splitlines() if ln and not ln.startswith("#")]
    assert passes == ["TA_BUSY_avr TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum", "TA_DATA_STALLED_BY_TC_CYCLES_sum"]
    assert any(ln.startswith("# split into 2 passes") for ln in r.stdout.splitlines())
    for p in passes:                                                    # every pass it prints fits
        assert run(tmp_path, "--check", p).returncode == 0


def test_derived_names_cost_their_hardware_counters_and_unknown_names_are_dropped(tmp_path):
    r = run(tmp_path, "VALUBusy MemUnitBusy TA_BUSY_avr")
    lines = r.stdout.splitlines()
    assert "# unknown on this device, dropped: MemUnitBusy" in lines
    assert "VALUBusy TA_BUSY_avr" in lines                              # SQ + GRBM + TA: one counter each, one pass
    assert run(tmp_path, "--check", "VALUBusy TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum").returncode == 0
