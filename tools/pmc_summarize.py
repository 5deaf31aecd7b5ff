"""Average the rocprofv3 --pmc counter CSVs per kernel name (helper of tools/pmc_conv.sh)."""
import csv
import glob
import os
import sys
from collections import defaultdict

acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
for d in sys.argv[1:]:
    for f in glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True):
        for row in csv.DictReader(open(f)):
            k = row['Kernel_Name'][:90] + ' grid=' + row.get('Grid_Size', '?')
            a = acc[k][row['Counter_Name']]
            a[0] += float(row['Counter_Value'])
            a[1] += 1
for k in sorted(acc):
    if 'pseg' not in k:
        continue
    print(k)
    for c, (v, n) in sorted(acc[k].items()):
        print('    %-28s %14.1f  (n=%d)' % (c, v / n, n))
