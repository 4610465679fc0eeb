import csv,glob,sys
f=glob.glob(sys.argv[1]+"/**/*kernel_stats.csv",recursive=True)[0]
for r in csv.DictReader(open(f)):
    if float(r["Percentage"])>0.8: print(r["Name"][:90], r["Calls"], "avg",round(float(r["AverageNs"])/1000,1), "%",r["Percentage"], "min",round(float(r["MinNs"])/1e3,1), "max",round(float(r["MaxNs"])/1e3,1))
