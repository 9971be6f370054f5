"""Average PMC counter values per kernel from rocprofv3 --pmc csv output under gpurun_out/<dir>."""
import csv, glob, collections, sys
for d in sys.argv[1:]:
    f = glob.glob(f'gpurun_out/{d}/*/*counter_collection.csv')[0]
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        if 'qgemm' in r['Kernel_Name'] or 'paged' in r['Kernel_Name']:
            agg[(r['Kernel_Name'][:64], r['Grid_Size'])][r['Counter_Name']].append(float(r['Counter_Value']))
    for k, v in agg.items():
        print(k)
        for c, vals in sorted(v.items()):
            print(f"    {c:34s} {sum(vals)/len(vals):14.1f}  (n={len(vals)})")
