"""tools/gather_ceiling on the GPU: the program bench.py starts for the ceiling printed beside every product's rate.

Not a parity test (there is nothing to compare a rate with); what is checked is that the program runs on this box, prints the
one JSON line bench.py parses, and that its figures are rates of the right kind: positive, below the fabric's line-fill
ceiling by a wide margin of safety, and with the output-row stream never FASTER than the bare loop by more than noise."""
import os
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


@pytest.mark.parametrize("row_bytes", [64, 128])
def test_ceiling_program_prints_the_line_bench_reads(row_bytes):
    import subprocess
    import bench
    if not os.path.exists(os.path.join(ROOT, "tools", "gather_ceiling")):      # built by __graft_entry__.build(); same image here
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "block-lanczos-algorithm-parallelization_amd"), "../tools/gather_ceiling"])
    got = bench.live_gather_ceiling(row_bytes, 300e6, 19.4)
    assert "error" not in got, got
    assert got["row_bytes"] == row_bytes and got["gathers_per_output_row"] == 19
    keys = ("bare_8B_per_lane", "bare_16B_per_lane", "with_output_rows_8B_per_lane", "with_output_rows_16B_per_lane")
    assert all(1e9 < got[k] < 2e11 for k in keys), got
    assert max(got[keys[2]], got[keys[3]]) < 1.05 * max(got[keys[0]], got[keys[1]])
    assert "error" in bench.live_gather_ceiling(32, 300e6, 19.4)
