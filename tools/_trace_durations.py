import csv, sys, glob
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
pat = sys.argv[2]
for r in csv.DictReader(open(f)):
    nm = r["Kernel_Name"]
    if pat in nm:
        print(nm[:48], r.get("Grid_Size_X", r.get("Grid_Size", "")), int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
