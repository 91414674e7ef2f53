"""print the kernel timeline of a few steps from a rocprofv3 kernel trace csv"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
def short(n):
    for k, v in (('step2ds', 'step2ds'), ('Euler5, 1', 'x-pass'), ('Euler5, 2', 'y-pass'), ('ncclDevKernel', 'nccl'), ('halo_pack', 'halo_pack'),
                 ('cfl_handover', 'cfl_handover'), ('fillBuffer', 'fill'), ('copyBuffer', 'copy'), ('frame_kernel', 'frame'),
                 ('unsplit_x', 'unsplit_x'), ('unsplit_y', 'unsplit_y'), ('sharp_kernel', 'sharp')):
        if k in n: return v
    return n[:40]
hand = [i for i, r in enumerate(rows) if 'cfl_handover' in r['Kernel_Name']]
k = int(sys.argv[2]) if len(sys.argv) > 2 else len(hand) // 2
a, b = hand[k], hand[k + 2]
t0 = int(rows[a]['End_Timestamp'])
for r in rows[a:b + 1]:
    print('%9.1f %9.1f  q%s s%s  %-14s grid %s' % ((int(r['Start_Timestamp']) - t0) / 1e3, (int(r['End_Timestamp']) - t0) / 1e3,
          r['Queue_Id'], r['Stream_Id'], short(r['Kernel_Name']), r['Grid_Size_X']))
