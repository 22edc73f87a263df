"""Aggregates rocprofv3 --pmc counter_collection CSVs per kernel (mean per dispatch)."""
import csv, glob, os, sys, collections
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob(os.path.join(out, "p*", "*counter_collection.csv"))):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "goi::" not in k:
            continue
        k = k.replace("void ", "").replace("goi::(anonymous namespace)::", "").split("(")[0]
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for extra in ("VGPR_Count", "LDS_Block_Size", "Accum_VGPR_Count", "SGPR_Count"):
            if extra in r:
                agg[k]["_" + extra] = [float(r[extra])]
for k, cs in agg.items():
    print("==", k)
    for c, v in sorted(cs.items()):
        print(f"   {c:32s} mean {sum(v)/len(v):16.1f}  n {len(v)}")
