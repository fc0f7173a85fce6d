"""Dev tool: largest idle gaps between consecutive kernels of a rocprofv3 kernel trace (csv), with the kernels around them."""
import csv
import glob
import sys

f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:60], r.get("Grid_Size", ""),
                r.get("Scratch_Size", r.get("Private_Segment_Size", ""))) for r in csv.DictReader(open(f))))
gaps = sorted(((rows[i + 1][0] - rows[i][1], i) for i in range(len(rows) - 1)), reverse=True)[:int(sys.argv[2]) if len(sys.argv) > 2 else 12]
for gap, i in gaps:
    print(f"gap {gap / 1e6:9.3f} ms after [{rows[i][2]} grid={rows[i][3]} scratch={rows[i][4]}] ({(rows[i][1] - rows[i][0]) / 1e3:.1f} us) before "
          f"[{rows[i + 1][2]} grid={rows[i + 1][3]} scratch={rows[i + 1][4]}] ({(rows[i + 1][1] - rows[i + 1][0]) / 1e3:.1f} us)")
