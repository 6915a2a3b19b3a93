import csv,sys,glob
f=glob.glob(sys.argv[1]+"/**/*kernel_stats.csv",recursive=True)[0]
for r in csv.DictReader(open(f)):
    if "crypto" in r["Name"] and ("obs" in r["Name"] or "dyn" in r["Name"]): print(sys.argv[2], r["Name"][:40], r["Calls"], float(r["AverageNs"])/1e3)
