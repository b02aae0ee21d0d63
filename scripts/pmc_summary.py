"""Mean counter value per render kernel over the dispatches of a rocprofv3 --pmc output tree."""
import collections, csv, glob, json, sys
for d in sys.argv[1:]:
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "render_" not in k:
                continue
            k = k.split("render_")[1].split("(")[0][:40]
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        print(json.dumps({"dir": d.split("/")[-1], "kernel": k, "dispatches": len(next(iter(v.values()))), **{c: round(sum(x) / len(x)) for c, x in sorted(v.items())}}))
