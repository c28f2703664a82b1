import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
tot=sum(int(r["TotalDurationNs"]) for r in rows)
n=int(sys.argv[2]) if len(sys.argv)>2 else 1
print("total ms/step", tot/1e6/n)
for r in rows[:int(sys.argv[3]) if len(sys.argv)>3 else 28]:
    nm=r["Name"].replace("(anonymous namespace)::","").replace("void ","")
    print(f"  {nm[:84]:84s} {float(r['Percentage']):5.1f} %  calls/step {int(r['Calls'])/n:6.1f}  avg {float(r['AverageNs'])/1e3:7.1f} us")
