"""Export what the judge needs from a rocprofv3 rocpd database (this image writes .db by default):
  python tools/rocpd_export.py stats  results.db out.csv   -> per-kernel calls / total / average (as --stats prints)
  python tools/rocpd_export.py pmc    results.db out.csv   -> one row per dispatch and counter (counter_collection.csv columns)"""
import csv, sqlite3, sys

mode, db, out = sys.argv[1:4]
c = sqlite3.connect(db)
with open(out, 'w', newline='') as f:
    w = csv.writer(f, quoting=csv.QUOTE_NONNUMERIC)
    if mode == 'stats':
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage"])
        for r in c.execute("select name, total_calls, total_duration * 1000.0, average * 1000.0, percentage from top_kernels order by total_duration desc"):
            w.writerow([r[0], r[1], round(r[2], 1), round(r[3], 1), round(r[4], 4)])
    else:
        w.writerow(["Dispatch_Id", "Grid_Size", "Kernel_Name", "Workgroup_Size", "LDS_Block_Size", "VGPR_Count", "Accum_VGPR_Count", "Counter_Name",
                    "Counter_Value", "Start_Timestamp", "End_Timestamp"])
        for r in c.execute("select dispatch_id, grid_size, kernel_name, workgroup_size, lds_block_size, vgpr_count, accum_vgpr_count, counter_name, "
                           "value, start, end from counters_collection order by dispatch_id"):
            w.writerow(list(r))
