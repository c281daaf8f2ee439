"""tools/dispatch_timeline.py (round 6: the dispatch-by-dispatch account of the two-stream step) on a synthetic pair of
rocprofv3 kernel-trace CSVs with a known answer: three steps of four kernels, one of them on a second queue beside the second main-chain
kernel in trace B, rows shuffled (the tool must order by dispatch id). CPU only."""
import csv
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
COLS = ["Kind", "Agent_Id", "Queue_Id", "Stream_Id", "Thread_Id", "Dispatch_Id", "Kernel_Id", "Kernel_Name", "Correlation_Id", "Start_Timestamp",
        "End_Timestamp", "LDS_Block_Size", "Scratch_Size", "VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "Workgroup_Size_X", "Workgroup_Size_Y",
        "Workgroup_Size_Z", "Grid_Size_X", "Grid_Size_Y", "Grid_Size_Z"]


def _trace(path, two_streams, steps=8):
    rows, t, did = [], 1_000_000, 0
    for _ in range(steps):
        # kern_a 10 us, kern_b 20 us (40 us with company), side_w 30 us (one stream: after kern_b; two streams: beside kern_b), adamw 5 us
        plan = [("kern_a(int)", 1, 10_000, None), ("kern_b(int)", 1, 40_000 if two_streams else 20_000, None),
                ("side_w(int)", 3 if two_streams else 1, 30_000, "beside" if two_streams else None), ("adamw_kernel(float*)", 1, 5_000, None)]
        main_t = t
        b_start = None
        for name, q, dur, where in plan:
            did += 1
            if where == "beside":
                s = b_start + 2_000
            else:
                s = main_t + 1_000
                main_t = s + dur
            if name.startswith("kern_b"):
                b_start = s
            rows.append(dict(zip(COLS, ["KERNEL_DISPATCH", "Agent 2", q, 0, 1, did, 1, name, did, s, s + dur, 1024, 0, 64, 0, 32, 256, 1, 1, 256 * 256, 1, 1])))
        t = main_t
    rows = rows[::2] + rows[1::2]                     # not in launch order: the tool sorts by Dispatch_Id
    with open(path, "w", newline="") as f:
        w = csv.DictWriter(f, fieldnames=COLS, quoting=csv.QUOTE_NONNUMERIC)
        w.writeheader()
        w.writerows(rows)


def test_dispatch_timeline_on_a_synthetic_step(tmp_path):
    a, b, out, txt = (str(tmp_path / n) for n in ("a.csv", "b.csv", "out.json", "out.txt"))
    _trace(a, False)
    _trace(b, True)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "dispatch_timeline.py"), a, b, out, txt], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    d = json.load(open(out))
    s = d["summary"]
    assert s["dispatches"] == 4 and s["main_queue_kernels"] == 3 and s["side_queue_kernels"] == 1
    # one stream: 10 + 20 + 30 + 5 us of kernels + 4 gaps of 1 us = 69 us; as timed: 10 + 40 + 5 + 3 gaps = 58 us
    assert abs(s["wall_one_stream_ms"] - 0.069) < 1e-6 and abs(s["wall_as_timed_ms"] - 0.058) < 1e-6
    assert abs(s["main_kernels_sum_alone_ms"] - 0.035) < 1e-6 and abs(s["main_kernels_sum_in_step_ms"] - 0.055) < 1e-6
    assert abs(s["main_chain_stretch_ms"] - 0.020) < 1e-6 and abs(s["two_or_more_kernels_running_ms"] - 0.030) < 1e-6
    kb = [t for t in d["dispatches"] if t["kernel"].startswith("kern_b")][0]
    assert kb["beside_us"] == {"side_w": 30.0} and abs(kb["us_alone"] - 20.0) < 1e-6 and abs(kb["us_in_step"] - 40.0) < 1e-6
    assert "== every dispatch of the median step" in open(txt).read()
