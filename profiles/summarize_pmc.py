#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection.csv files: mean counter value per launch of each hns:: kernel.
usage: summarize_pmc.py out.json dir1 dir2 ...   (each dir = a rocprofv3 -d output directory)"""
import collections, csv, glob, json, sys

def main():
    out_path, dirs = sys.argv[1], sys.argv[2:]
    out = {}
    for d in dirs:
        for f in glob.glob(f"{d}/**/*_counter_collection.csv", recursive=True):
            agg = collections.defaultdict(list)
            for r in csv.DictReader(open(f)):
                k = r["Kernel_Name"].split("(")[0].replace("void ", "")
                agg[(k, r["Counter_Name"])].append(float(r["Counter_Value"]))
            for (k, c), v in sorted(agg.items()):
                if k.startswith("hns::"):
                    out.setdefault(k, {})[c] = {"launches": len(v), "mean": sum(v) / len(v)}
    json.dump(out, open(out_path, "w"), indent=1, sort_keys=True)
    return out

if __name__ == "__main__":
    o = main()
    for k in o:
        print(k)
        for c, v in o[k].items():
            print(f"   {c:34s} {v['mean']:.5g}  (n={v['launches']})")
