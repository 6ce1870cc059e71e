import csv,sys,glob,collections
for d in sys.argv[1:]:
    f=glob.glob(d+"/**/*kernel_stats.csv",recursive=True)
    print("==",d)
    for fn in f:
        rows=list(csv.DictReader(open(fn)))
        for r in rows[:12]:
            print("%-70s calls %5s avg %9.1f us  tot %8.2f ms"%(r["Name"][:70], r["Calls"], float(r["AverageNs"])/1e3, float(r["TotalDurationNs"])/1e6))
