"""Summarise rocprofv3 counter_collection CSVs for the render kernel: python scripts/pmc_summary.py DIR..."""
import csv, glob, collections, sys
for d in sys.argv[1:]:
    for f in glob.glob(f'{d}/**/*counter_collection.csv', recursive=True):
        agg = collections.defaultdict(float); n = 0
        for r in csv.DictReader(open(f)):
            if 'render_k' in r['Kernel_Name']:
                agg[r['Counter_Name']] += float(r['Counter_Value']); last = r
        print(d, {k: f"{v:.4g}" for k, v in agg.items()})
        print("   ", {k: last[k] for k in ('VGPR_Count', 'SGPR_Count', 'LDS_Block_Size', 'Grid_Size', 'Workgroup_Size') if k in last})
