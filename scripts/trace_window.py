"""Print a window of dispatches from a rocprofv3 kernel trace CSV: `n` dispatches starting `skip_back` before the end
(queue, start offset us, duration us, gap to the previous kernel's end)."""
import csv, sys
f, n, back = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
rows = [r for r in csv.DictReader(open(f)) if "lynse::" in r["Kernel_Name"] or "ccl" in r["Kernel_Name"].lower()]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
win = rows[-back:][:n] if back else rows[-n:]
t0 = int(win[0]["Start_Timestamp"])
prev_end = None
for r in win:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = "" if prev_end is None else str(round((s - prev_end) / 1e3, 1))
    print(r["Kernel_Name"].replace("void ", "").replace("lynse::", "")[:44].ljust(44), ("q" + r.get("Queue_Id", "?")).rjust(4),
          str(round((s - t0) / 1e3, 1)).rjust(9), str(round((e - s) / 1e3, 1)).rjust(8), gap.rjust(8))
    prev_end = e if prev_end is None else max(prev_end, e)
