"""Digest of a kernel trace of tools/mid_rows_gemm_bench.py: per launch group (4 calls per mode and shape, in order) the median kernel time."""
import csv, glob, re, sys
rows = list(csv.DictReader(open(glob.glob(sys.argv[1] + '/*/*kernel_trace.csv')[0])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
keep = [r for r in rows if re.search(r'skinny_gemm_h3|gemm_h3_kernel', r['Kernel_Name'])]
i = 0
while i < len(keep):
    j = i
    key = (keep[i]['Kernel_Name'], keep[i]['Grid_Size_X'], keep[i]['Grid_Size_Y'])
    while j < len(keep) and (keep[j]['Kernel_Name'], keep[j]['Grid_Size_X'], keep[j]['Grid_Size_Y']) == key:
        j += 1
    d = sorted((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3 for r in keep[i:j])
    name = re.sub(r'void |tepose::|\(.*', '', key[0])
    print('%-48s grid %6s x %-3s n %2d  median %8.1f us  min %8.1f' % (name[:48], int(key[1]) // int(keep[i]['Workgroup_Size_X']), key[2], j - i, d[len(d) // 2], d[0]))
    i = j
