import csv, sys
rows=[r for r in csv.DictReader(open(sys.argv[1])) if 'wino' in r['Kernel_Name']]
cis=[32,64,128,256]; cos=[128,64,32]
i=0
for ci in cis:
    for co in cos:
        r=rows[i]; i+=1
        d=(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3
        px=2*256*480
        fl=2*px*9*ci*co
        print(f"ci={ci:4d} co={co:4d} {d:8.1f} us  direct-equiv {fl/d/1e6:7.1f} TF/s  (MFMA-bound: {fl/(157.3e12*2.25)*1e6:6.1f} us)")
