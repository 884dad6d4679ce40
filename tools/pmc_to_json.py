"""Turn rocprofv3 --pmc passes (tools/pmc_kpe.sh) into profiles/<tag>_pmc_<kernel>.json with the
HBM traffic per launch corrected as MI355X_MICROARCH.md §HBM prescribes:
  FETCH_SIZE (KB) counts 64 B per 128-B request on gfx950 -> read bytes = 2 * FETCH_SIZE * 1024
  WRITE_SIZE (KB) is exact for streaming stores           -> write bytes = WRITE_SIZE * 1024
(the factor 2 was checked on kpe_main itself: it reads every frame byte once plus ~5 % halo)."""
import collections, csv, glob, json, sys

root, kern, out = sys.argv[1], sys.argv[2], sys.argv[3]
agg = collections.defaultdict(list)
for f in glob.glob(root + "/*/runc/*counter_collection.csv"):
    for row in csv.DictReader(open(f)):
        if kern in row["Kernel_Name"]:
            agg[row["Counter_Name"]].append(float(row["Counter_Value"]))
m = {k: sum(v) / len(v) for k, v in agg.items()}
res = {"kernel": kern.rstrip("("), "launches_sampled": {k: len(v) for k, v in agg.items()}, "counters_mean_per_launch": m}
if "FETCH_SIZE" in m and "WRITE_SIZE" in m:
    res["hbm_read_bytes_per_launch"] = 2 * m["FETCH_SIZE"] * 1024
    res["hbm_write_bytes_per_launch"] = m["WRITE_SIZE"] * 1024
    res["hbm_bytes_per_launch"] = res["hbm_read_bytes_per_launch"] + res["hbm_write_bytes_per_launch"]
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res)[:400])
