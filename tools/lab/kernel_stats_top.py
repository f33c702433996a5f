import csv,sys,glob
f=sorted(glob.glob(sys.argv[1]+"/*/*kernel_stats.csv"))[-1]
rows=list(csv.DictReader(open(f)))
tot=sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:int(sys.argv[2]) if len(sys.argv)>2 else 14]:
    print(f'{r["Name"][:84]:84s} calls {r["Calls"]:>6s} avg {float(r["AverageNs"])/1e3:7.2f} us  {100*float(r["TotalDurationNs"])/tot:5.1f}%')
print("total ms", tot/1e6)
