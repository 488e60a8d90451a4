import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# last 12 kernel dispatches
sel = rows[-14:]
t0 = int(sel[0]["Start_Timestamp"])
for r in sel:
    print(f'{(int(r["Start_Timestamp"]) - t0) / 1e3:10.1f} us  +{(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3:9.1f} us  {r["Kernel_Name"][:70]}')
