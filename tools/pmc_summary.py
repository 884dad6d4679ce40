import sys, glob, csv, collections
root = sys.argv[1]
kern = sys.argv[2] if len(sys.argv) > 2 else "kpe_main"
agg = collections.defaultdict(list)
for f in glob.glob(root + "/*/runc/*counter_collection.csv"):
    for row in csv.DictReader(open(f)):
        if kern in row["Kernel_Name"]:
            agg[row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, v in sorted(agg.items()):
    print(f"{k:24s} n={len(v):3d} mean={sum(v)/len(v):.4g}")
