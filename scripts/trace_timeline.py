"""Print the per-dispatch timeline of the last N dispatches in a rocprofv3 kernel trace CSV (queue, start offset us, duration us)."""
import csv, sys
f, n = sys.argv[1], int(sys.argv[2])
rows = [r for r in csv.DictReader(open(f)) if "lynse::" in r["Kernel_Name"] or "ccl" in r["Kernel_Name"].lower()]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
last = rows[-n:]
t0 = int(last[0]["Start_Timestamp"])
for r in last:
    print(r["Kernel_Name"].replace("void ", "").replace("lynse::", "")[:40].ljust(40), ("q" + r.get("Queue_Id", "?")).rjust(4), r.get("Grid_Size_X", "?").rjust(8),
          str(round((int(r["Start_Timestamp"]) - t0) / 1e3, 1)).rjust(8), str(round((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, 1)).rjust(8))
